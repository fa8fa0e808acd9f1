# round 6: latency decomposition + kernel timelines of the latency-shaped configurations
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/r6
python scripts/r6_latency.py c1 c3 c4 > gpurun_out/r6/lat1.jsonl 2> gpurun_out/r6/lat1.err
for C in c1 c3 c4; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/r6/tl_$C -o u --output-format csv -- bash -c "cd $ROOT && python scripts/r6_latency.py $C" > $ROOT/gpurun_out/r6/tl_$C.log 2>&1)
  f=$(find gpurun_out/r6/tl_$C -name "*kernel_trace.csv" | head -1)
  python scripts/trace_timeline.py $f 30 > gpurun_out/r6/tl_$C.txt
  find gpurun_out/r6/tl_$C -name "*.csv" -size +1M -delete
done
cat gpurun_out/r6/lat1.jsonl
