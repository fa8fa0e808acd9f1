#!/bin/bash
# runs build/qs_microbench on the GPU box with a power / clock sampler beside it; output under gpurun_out/qs/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/qs
( while true; do echo "T $(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Power|sclk|mclk|fclk" ; sleep 0.2; done ) > gpurun_out/qs/smi_${1:-run}.txt 2>&1 &
SMI=$!
timeout 600 build/qs_microbench ${QS_ARGS:-10000000 5 4.3} > gpurun_out/qs/microbench_${1:-run}.txt 2>&1
echo "exit $?" >> gpurun_out/qs/microbench_${1:-run}.txt
kill $SMI
tail -60 gpurun_out/qs/microbench_${1:-run}.txt
