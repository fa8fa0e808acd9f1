# kernel timeline of the 1.25M-row shard step (blocking and 3 in flight): rocprofv3 kernel trace, last dispatches printed
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/tl
for fl in 1 3; do
B="python bench.py --no-cpu-baseline --no-verify --no-configs --steps 40 --warmup 5 --rows 1250000 --in-flight $fl"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/tl/f$fl -o u --output-format csv -- bash -c "cd $ROOT && $B" > $ROOT/gpurun_out/tl/f$fl.log 2>&1)
f=$(find gpurun_out/tl/f$fl -name "*kernel_trace.csv" | head -1)
echo "== in flight $fl"; python scripts/trace_timeline.py $f 16
timeout 120 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'))"
done
find gpurun_out/tl -name "*kernel_trace.csv" -size +2M -delete
