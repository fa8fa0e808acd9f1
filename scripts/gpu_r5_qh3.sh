#!/bin/bash
# round 5: the exactness rule of integer collections + k_scan_qh: parity, then C3 on both tilings and stage plans, one box
mkdir -p gpurun_out/qh3
timeout 2400 python -m pytest tests/test_gpu_qh.py tests/test_gpu_baseline_configs.py tests/test_gpu_flat_parity.py -x -q -m gpu > gpurun_out/qh3/pytest.txt 2>&1; tail -12 gpurun_out/qh3/pytest.txt
c3() { python scripts/other_config.py c3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c3']; print('$1', {k: d.get(k) for k in ('ms','scan_us','oracle_parity','rescored_per_query','stages','fallback_queries','error')})"; }
for r in 1 2; do
  LYNSE_HIP_NO_EXACT_INT=1 LYNSE_HIP_QH=0 c3 "margin QH=0"
  LYNSE_HIP_QH=0 c3 "exact QH=0"
  LYNSE_HIP_QH=1 c3 "exact QH=1"
done
for g in 3 6 8 32; do LYNSE_HIP_QH=0 LYNSE_HIP_SAMPLE_GROWTH=$g c3 "exact QH=0 growth=$g"; LYNSE_HIP_QH=1 LYNSE_HIP_SAMPLE_GROWTH=$g c3 "exact QH=1 growth=$g"; done
echo "== qh phases k=100 (exact)"; timeout 300 python scripts/qh_phase_timing.py 2>&1 | grep -v amdgpu.ids | head -4
