# kernel timeline of one of bench.py's extra configurations: bash scripts/gpu_cfg_timeline.sh c3 [n_last]
export TMPDIR=/tmp
ROOT=$(pwd)
C=${1:-c3}; N=${2:-40}
mkdir -p gpurun_out/tl
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/tl/$C -o u --output-format csv -- bash -c "cd $ROOT && python scripts/other_config.py $C" > $ROOT/gpurun_out/tl/$C.log 2>&1)
f=$(find gpurun_out/tl/$C -name "*kernel_trace.csv" | head -1)
python scripts/trace_timeline.py $f $N
find gpurun_out/tl -name "*kernel_trace.csv" -size +2M -delete
