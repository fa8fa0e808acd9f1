mkdir -p gpurun_out/r7
(timeout 300 python scripts/stress_parity.py 200 31 2>&1 | tail -3) > gpurun_out/r7/stress_parity.log
(timeout 200 python scripts/stress_ivf.py 120 32 2>&1 | tail -3) > gpurun_out/r7/stress_ivf.log
(timeout 200 python scripts/stress_inflight.py 90 33 2>&1 | tail -3) > gpurun_out/r7/stress_inflight.log
(STRESS_COMM=1 timeout 200 python scripts/stress_inflight.py 60 34 2>&1 | tail -3) > gpurun_out/r7/stress_inflight_comm.log
(LYNSE_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --rows 2000000 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/r7/gloo2.log
(timeout 300 python -m pytest tests/test_gpu_storage_formats.py tests/test_gpu_ivf_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed") > gpurun_out/r7/tests.log
(timeout 200 python scripts/ivf_c4_trace.py 2>&1 | grep "^nq") > gpurun_out/r7/ivf_c4.log
for f in gpurun_out/r7/*.log; do echo "== $f"; cat $f; done
