#!/bin/bash
# round 5: the LDS hand-over inside k_select_final (LYNSE_HIP_TAIL_LDS=0: through cand[q] in global memory) — parity, then A/B on one box
mkdir -p gpurun_out/tail1
timeout 2400 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_flat_parity.py tests/test_gpu_qh.py tests/test_gpu_inflight.py tests/test_gpu_i8c_hostile.py -x -q -m gpu > gpurun_out/tail1/pytest.txt 2>&1; tail -4 gpurun_out/tail1/pytest.txt
bash scripts/ab_env.sh LYNSE_HIP_TAIL_LDS 0 1
bash scripts/ab_env.sh LYNSE_HIP_TAIL_LDS 0 1 --rows 1250000 --in-flight 3 --steps 80
c3() { python scripts/other_config.py c3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c3']; print('$1', {k: d.get(k) for k in ('ms','scan_us','oracle_parity','rescored_per_query')})"; }
for r in 1 2 3; do LYNSE_HIP_TAIL_LDS=0 c3 "TAIL_LDS=0"; LYNSE_HIP_TAIL_LDS=1 c3 "TAIL_LDS=1"; done
