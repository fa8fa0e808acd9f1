#!/bin/bash
# round 5: the status-word fix (all-lists-empty flag of an IVF ticket through a communicator) + the sweep that found it, then C5 on k_scan_qh<.., F4>
mkdir -p gpurun_out/fix1
timeout 1500 python -m pytest tests/test_gpu_ivf_inflight.py tests/test_gpu_inflight.py tests/test_gpu_sharded_native.py -x -q -m gpu > gpurun_out/fix1/pytest.txt 2>&1; tail -4 gpurun_out/fix1/pytest.txt
(STRESS_COMM=1 timeout 240 python scripts/stress_ivf_inflight.py 150 56 2>&1 | grep -vE "^(RCCL|HIP version|ROCm version|Hostname|Librccl)" | tail -3) > gpurun_out/fix1/stress_ivf_inflight_comm.log; cat gpurun_out/fix1/stress_ivf_inflight_comm.log
(STRESS_COMM=1 timeout 200 python scripts/stress_inflight.py 60 54 2>&1 | grep -vE "^(RCCL|HIP version|ROCm version|Hostname|Librccl)" | tail -3) > gpurun_out/fix1/stress_inflight_comm.log; cat gpurun_out/fix1/stress_inflight_comm.log
bash scripts/gpu_r5_c5.sh
