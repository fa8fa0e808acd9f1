# kernel timeline of one r6_latency configuration: bash scripts/gpu_r6_tl.sh c4 [n_last]
export TMPDIR=/tmp
ROOT=$(pwd); C=${1:-c4}; N=${2:-12}
mkdir -p gpurun_out/r6
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/r6/tl_$C -o u --output-format csv -- bash -c "cd $ROOT && python scripts/r6_latency.py $C" > $ROOT/gpurun_out/r6/tl_$C.log 2>&1)
f=$(find gpurun_out/r6/tl_$C -name "*kernel_trace.csv" | head -1)
python scripts/trace_timeline.py $f $N
find gpurun_out/r6/tl_$C -name "*.csv" -size +1M -delete
