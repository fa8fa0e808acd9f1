#!/bin/bash
# --config c4 on one GPU: the 1-rank line at a reduced and at the full per-GPU size, and the 2-rank code path (gloo: two processes
# share the GPU — the training all-reduce and the exchange go through torch.distributed; RCCL needs one device per rank)
mkdir -p gpurun_out
python bench.py --config c4 --rows-per-gpu 500000 --steps 20 --warmup 3 > gpurun_out/c4_small.json 2> gpurun_out/c4_small.err; echo "rc=$?"; tail -c 2500 gpurun_out/c4_small.json; tail -5 gpurun_out/c4_small.err
LYNSE_BENCH_BACKEND=gloo python bench.py --config c4 --gpus 2 --rows-per-gpu 300000 --steps 10 --warmup 2 > gpurun_out/c4_gloo2.json 2> gpurun_out/c4_gloo2.err; echo "rc=$?"; tail -c 2500 gpurun_out/c4_gloo2.json; tail -5 gpurun_out/c4_gloo2.err
python bench.py --config c4 --steps 30 --warmup 3 > gpurun_out/c4_full.json 2> gpurun_out/c4_full.err; echo "rc=$?"; tail -c 2500 gpurun_out/c4_full.json; tail -5 gpurun_out/c4_full.err
