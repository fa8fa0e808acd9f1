#!/bin/bash
# round 5: k_scan_qh (scan_qh.h) — parity, then C3 (1M x 128, squared L2, k = 100) against the 256 x 256 tile of k_scan_h16 on one box
mkdir -p gpurun_out/qh1
timeout 1200 python -m pytest tests/test_gpu_qh.py "tests/test_gpu_baseline_configs.py::test_c3_flat_l2_sift_like_1m_k100" -x -q -m gpu > gpurun_out/qh1/pytest.txt 2>&1; tail -15 gpurun_out/qh1/pytest.txt
c3() { python scripts/other_config.py c3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c3']; print('$1', {k: d.get(k) for k in ('ms','scan_us','oracle_parity','rescored_per_query','stages','fallback_queries','error')})"; }
for r in 1 2 3; do
  LYNSE_HIP_QH=0 c3 "QH=0"
  LYNSE_HIP_QH=1 c3 "QH=1"
done
for g in 2 3 4 6 8 32; do LYNSE_HIP_SAMPLE_GROWTH=$g c3 "QH=1 growth=$g"; done
for s0 in 98304 131072; do LYNSE_HIP_SAMPLE_ROWS_TO=$s0 c3 "QH=1 sample=$s0"; LYNSE_HIP_SAMPLE_ROWS_TO=$s0 LYNSE_HIP_SAMPLE_GROWTH=32 c3 "QH=1 sample=$s0 growth=32"; done
export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/qh1/c3 -o u --output-format csv -- bash -c "cd $ROOT && python scripts/other_config.py c3" > $ROOT/gpurun_out/qh1/c3.log 2>&1)
f=$(find gpurun_out/qh1/c3 -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
