# kernel timeline (last N dispatches) of a python script: bash scripts/gpu_script_timeline.sh scripts/f16_shard_ab.py 14
export TMPDIR=/tmp
ROOT=$(pwd)
S=$1; N=${2:-20}
mkdir -p gpurun_out/tl
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/tl/s -o u --output-format csv -- bash -c "cd $ROOT && python $S" > $ROOT/gpurun_out/tl/s.log 2>&1)
f=$(find gpurun_out/tl/s -name "*kernel_trace.csv" | head -1)
python scripts/trace_timeline.py $f $N
find gpurun_out/tl -name "*kernel_trace.csv" -size +2M -delete
