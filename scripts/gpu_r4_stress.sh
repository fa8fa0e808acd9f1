#!/bin/bash
# round 4, final binary: randomised parity sweeps (seeds on the command lines) + the bare `bench.py --gpus 2` self-launch under gloo
mkdir -p gpurun_out/r04s
(timeout 330 python scripts/stress_parity.py 240 41 2>&1 | tail -3) > gpurun_out/r04s/stress_parity.log
(timeout 240 python scripts/stress_ivf.py 150 42 2>&1 | tail -3) > gpurun_out/r04s/stress_ivf.log
(timeout 200 python scripts/stress_inflight.py 100 43 2>&1 | tail -3) > gpurun_out/r04s/stress_inflight.log
(STRESS_COMM=1 timeout 200 python scripts/stress_inflight.py 60 44 2>&1 | grep -vE "^(RCCL|HIP version|ROCm version|Hostname|Librccl)" | tail -3) > gpurun_out/r04s/stress_inflight_comm.log
(timeout 330 python scripts/stress_ivf_inflight.py 240 45 2>&1 | tail -3) > gpurun_out/r04s/stress_ivf_inflight.log
(STRESS_COMM=1 timeout 240 python scripts/stress_ivf_inflight.py 120 46 2>&1 | grep -vE "^(RCCL|HIP version|ROCm version|Hostname|Librccl)" | tail -3) > gpurun_out/r04s/stress_ivf_inflight_comm.log
(timeout 300 python scripts/stress_i8c_batches.py 180 47 2>&1 | tail -3) > gpurun_out/r04s/stress_i8c_batches.log
(timeout 300 python scripts/stress_ivf_large.py 180 48 2>&1 | tail -3) > gpurun_out/r04s/stress_ivf_large.log
(LYNSE_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --rows 2000000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400) > gpurun_out/r04s/gloo2_self_launch.log
for f in gpurun_out/r04s/*.log; do echo "== $f"; cat $f; done
