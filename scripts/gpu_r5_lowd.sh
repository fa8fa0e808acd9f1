#!/bin/bash
# round 5: FLAT-IP at 64 / 128 columns on the float pass (k_scan_qh) — the whole GPU suite, the randomised sweeps that cover it, its speed
mkdir -p gpurun_out/lowd
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/lowd/pytest.txt 2>&1; tail -25 gpurun_out/lowd/pytest.txt | cut -c1-400
(timeout 260 python scripts/stress_parity.py 200 91 2>&1 | tail -3) > gpurun_out/lowd/stress_parity.log; cat gpurun_out/lowd/stress_parity.log
(timeout 200 python scripts/stress_inflight.py 100 92 2>&1 | tail -3) > gpurun_out/lowd/stress_inflight.log; cat gpurun_out/lowd/stress_inflight.log
(timeout 200 python scripts/stress_i8c_batches.py 100 93 2>&1 | tail -3) > gpurun_out/lowd/stress_i8c.log; cat gpurun_out/lowd/stress_i8c.log
python scripts/ip_lowd_ab.py 2>&1 | grep -v amdgpu.ids
