#!/bin/bash
# round 5: the sample stage of the 64 / 128-column float plans on k_scan_qh<.., SMP> (LYNSE_HIP_QH_SAMPLE=0: k_scan_h16<.., EMIT = 2>) — parity, C3 A/B, kernel stats
mkdir -p gpurun_out/qh7
timeout 2400 python -m pytest tests/test_gpu_qh.py tests/test_gpu_baseline_configs.py tests/test_gpu_flat_parity.py -x -q -m gpu > gpurun_out/qh7/pytest.txt 2>&1; tail -4 gpurun_out/qh7/pytest.txt
(timeout 200 python scripts/stress_parity.py 150 71 2>&1 | tail -3) > gpurun_out/qh7/stress_parity.log; cat gpurun_out/qh7/stress_parity.log
c3() { python scripts/other_config.py c3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c3']; print('$1', {k: d.get(k) for k in ('ms','scan_us','oracle_parity','rescored_per_query','stages','fallback_queries','error')})"; }
for r in 1 2 3; do LYNSE_HIP_QH_SAMPLE=0 c3 "QH_SAMPLE=0"; LYNSE_HIP_QH_SAMPLE=1 c3 "QH_SAMPLE=1"; done
for g in 6 32; do LYNSE_HIP_SAMPLE_GROWTH=$g c3 "growth=$g"; done
for s0 in 131072 262144; do LYNSE_HIP_SAMPLE_ROWS_TO=$s0 c3 "sample=$s0"; LYNSE_HIP_SAMPLE_ROWS_TO=$s0 LYNSE_HIP_SAMPLE_GROWTH=32 c3 "sample=$s0 growth=32"; done
export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/qh7/c3 -o u --output-format csv -- bash -c "cd $ROOT && python scripts/other_config.py c3" > $ROOT/gpurun_out/qh7/c3.log 2>&1)
f=$(find gpurun_out/qh7/c3 -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
