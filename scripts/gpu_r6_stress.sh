#!/bin/bash
# round 6, final binary: randomised parity sweeps (seeds on the command lines) + the bare `bench.py --gpus 8` self-launch under gloo
mkdir -p gpurun_out/r06s
(timeout 330 python scripts/stress_parity.py 240 61 2>&1 | tail -3) > gpurun_out/r06s/stress_parity.log
(timeout 240 python scripts/stress_ivf.py 150 62 2>&1 | tail -3) > gpurun_out/r06s/stress_ivf.log
(timeout 200 python scripts/stress_inflight.py 100 63 2>&1 | tail -3) > gpurun_out/r06s/stress_inflight.log
(STRESS_COMM=1 timeout 200 python scripts/stress_inflight.py 60 64 2>&1 | grep -vE "^(RCCL|HIP version|ROCm version|Hostname|Librccl)" | tail -3) > gpurun_out/r06s/stress_inflight_comm.log
(timeout 330 python scripts/stress_ivf_inflight.py 240 65 2>&1 | tail -3) > gpurun_out/r06s/stress_ivf_inflight.log
(STRESS_COMM=1 timeout 240 python scripts/stress_ivf_inflight.py 120 66 2>&1 | grep -vE "^(RCCL|HIP version|ROCm version|Hostname|Librccl)" | tail -3) > gpurun_out/r06s/stress_ivf_inflight_comm.log
(timeout 300 python scripts/stress_i8c_batches.py 180 67 2>&1 | tail -3) > gpurun_out/r06s/stress_i8c_batches.log
(timeout 300 python scripts/stress_ivf_large.py 180 68 2>&1 | tail -3) > gpurun_out/r06s/stress_ivf_large.log
(LYNSE_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 8 --rows 2000000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400) > gpurun_out/r06s/gloo8_self_launch.log
for f in gpurun_out/r06s/*.log; do echo "== $f"; cat $f; done
