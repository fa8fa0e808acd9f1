#!/bin/bash
# round 5: k_scan_qh with the deferred emission as the default — parity (its tests, the baseline configs, the flat parity suite, the stress sweep), then C3 A/B
mkdir -p gpurun_out/qh6
timeout 2400 python -m pytest tests/test_gpu_qh.py tests/test_gpu_baseline_configs.py tests/test_gpu_flat_parity.py -x -q -m gpu > gpurun_out/qh6/pytest.txt 2>&1; tail -5 gpurun_out/qh6/pytest.txt
(timeout 300 python scripts/stress_parity.py 240 61 2>&1 | tail -4) > gpurun_out/qh6/stress_parity.log; cat gpurun_out/qh6/stress_parity.log
c3() { python scripts/other_config.py c3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c3']; print('$1', {k: d.get(k) for k in ('ms','scan_us','oracle_parity','rescored_per_query','stages','fallback_queries','error')})"; }
for r in 1 2 3; do LYNSE_HIP_QH=0 c3 "QH=0"; LYNSE_HIP_QH=1 c3 "QH=1"; done
for g in 6 32; do LYNSE_HIP_SAMPLE_GROWTH=$g c3 "QH=1 growth=$g"; done
for s0 in 98304 131072; do LYNSE_HIP_SAMPLE_ROWS_TO=$s0 c3 "QH=1 sample=$s0"; LYNSE_HIP_SAMPLE_ROWS_TO=$s0 LYNSE_HIP_SAMPLE_GROWTH=32 c3 "QH=1 sample=$s0 growth=32"; done
python scripts/c3_batch_sizes.py 2>&1 | grep -v amdgpu.ids
