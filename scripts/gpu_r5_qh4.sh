#!/bin/bash
# round 5: k_scan_qh variants (LYNSE_HIP_QH = 1: 32 queries per wave, two workgroups per CU; 2: 64 queries per wave) against k_scan_h16 on C3
mkdir -p gpurun_out/qh4
timeout 2400 python -m pytest tests/test_gpu_qh.py "tests/test_gpu_baseline_configs.py::test_c3_flat_l2_sift_like_1m_k100" -x -q -m gpu > gpurun_out/qh4/pytest.txt 2>&1; tail -6 gpurun_out/qh4/pytest.txt
LYNSE_HIP_QH=2 timeout 2400 python -m pytest tests/test_gpu_qh.py -x -q -m gpu > gpurun_out/qh4/pytest2.txt 2>&1; tail -3 gpurun_out/qh4/pytest2.txt
c3() { python scripts/other_config.py c3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c3']; print('$1', {k: d.get(k) for k in ('ms','scan_us','oracle_parity','rescored_per_query','stages','fallback_queries','error')})"; }
for r in 1 2 3; do
  LYNSE_HIP_QH=0 c3 "QH=0"
  LYNSE_HIP_QH=1 c3 "QH=1"
  LYNSE_HIP_QH=2 c3 "QH=2"
done
for g in 6 8 32; do LYNSE_HIP_QH=1 LYNSE_HIP_SAMPLE_GROWTH=$g c3 "QH=1 growth=$g"; done
echo "== qh=1 phases k=100"; LYNSE_HIP_QH=1 timeout 300 python scripts/qh_phase_timing.py 2>&1 | grep -v amdgpu.ids | head -12
echo "== qh=1 phases k=100 no emission"; QH_NOEMIT=1 LYNSE_HIP_QH=1 timeout 300 python scripts/qh_phase_timing.py 2>&1 | grep -v amdgpu.ids | head -12
