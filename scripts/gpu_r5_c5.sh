#!/bin/bash
# round 5: C5 share (Hamming 12.5M x 1024 bit, 256 queries, k = 50) on k_scan_qh<.., QB = 2, F4> against k_scan_qs<.., F4>, one box
mkdir -p gpurun_out/c5
timeout 1500 python -m pytest "tests/test_gpu_baseline_configs.py" -x -q -m gpu -k "c5" > gpurun_out/c5/pytest.txt 2>&1; tail -4 gpurun_out/c5/pytest.txt
c5() { python scripts/other_config.py c5_share 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c5_share']['nq256']; print('$1', {k: d.get(k) for k in ('ms','scan_us','frac_of_hbm_peak','mfma_TOPs','oracle_parity','kernel')})"; }
for r in 1 2 3; do LYNSE_HIP_QH_F4=0 c5 "QH_F4=0"; LYNSE_HIP_QH_F4=1 c5 "QH_F4=1"; done
