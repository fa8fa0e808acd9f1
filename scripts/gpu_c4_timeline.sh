# kernel timeline of the C4-share IVF batch (256 queries): rocprofv3 kernel trace, last dispatches printed
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/tl
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/tl/c4 -o u --output-format csv -- bash -c "cd $ROOT && python scripts/other_config.py c4_share" > $ROOT/gpurun_out/tl/c4.log 2>&1)
f=$(find gpurun_out/tl/c4 -name "*kernel_trace.csv" | head -1)
python scripts/trace_timeline.py $f 60 | tail -40
find gpurun_out/tl -name "*kernel_trace.csv" -size +2M -delete
