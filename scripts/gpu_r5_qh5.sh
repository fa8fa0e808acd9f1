#!/bin/bash
# round 5: final state of the C3 work — parity (both k_scan_qh variants opt-in), rocprof kernel stats of C3 on the three tilings, the bench-line C3
mkdir -p gpurun_out/qh5
timeout 2400 python -m pytest tests/test_gpu_qh.py tests/test_gpu_baseline_configs.py -x -q -m gpu > gpurun_out/qh5/pytest.txt 2>&1; tail -4 gpurun_out/qh5/pytest.txt
LYNSE_HIP_QH_TEST=2 timeout 2400 python -m pytest tests/test_gpu_qh.py -x -q -m gpu > gpurun_out/qh5/pytest2.txt 2>&1; tail -3 gpurun_out/qh5/pytest2.txt
export TMPDIR=/tmp; ROOT=$(pwd)
for v in 0 1 2; do
  (cd /tmp && LYNSE_HIP_QH=$v timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/qh5/c3_qh$v -o u --output-format csv -- bash -c "cd $ROOT && python scripts/other_config.py c3" > $ROOT/gpurun_out/qh5/c3_qh$v.log 2>&1)
  f=$(find gpurun_out/qh5/c3_qh$v -name "*kernel_stats.csv" | head -1); echo "== QH=$v"; head -7 $f | cut -c1-230
  tail -1 gpurun_out/qh5/c3_qh$v.log | cut -c1-400
done
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
