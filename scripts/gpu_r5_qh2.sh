#!/bin/bash
mkdir -p gpurun_out/qh2
timeout 1200 python -m pytest tests/test_gpu_qh.py -x -q -m gpu > gpurun_out/qh2/pytest.txt 2>&1; tail -12 gpurun_out/qh2/pytest.txt
echo "== k=100"; timeout 300 python scripts/qh_phase_timing.py 2>&1 | grep -v amdgpu.ids
echo "== k=100 no emission"; QH_NOEMIT=1 timeout 300 python scripts/qh_phase_timing.py 2>&1 | grep -v amdgpu.ids
echo "== k=10"; QH_K=10 timeout 300 python scripts/qh_phase_timing.py 2>&1 | grep -v amdgpu.ids
