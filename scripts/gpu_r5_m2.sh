#!/bin/bash
# round 5, measurement 2: energy experiments of k_scan_qs (qs2 = one wave per SIMD x 64 queries; MFMA shape) + sample-stage phase stamps
mkdir -p gpurun_out/m2
N=10000000 timeout 600 python scripts/dbg_smp_stamps.py > gpurun_out/m2/smp_stamps_10m.txt 2>&1
N=1250000 timeout 600 python scripts/dbg_smp_stamps.py > gpurun_out/m2/smp_stamps_1p25m.txt 2>&1
grep -v amdgpu.ids gpurun_out/m2/smp_stamps_10m.txt | tail -14; grep -v amdgpu.ids gpurun_out/m2/smp_stamps_1p25m.txt | tail -7
QS_ARGS="10000000 5 4.3" bash scripts/qs_run.sh r5 > /dev/null 2>&1
cp gpurun_out/qs/microbench_r5.txt gpurun_out/m2/
grep -E "MISMATCH|mismatch|check pass" gpurun_out/qs/microbench_r5.txt | head; grep -A70 -- "---- 10000000" gpurun_out/qs/microbench_r5.txt | head -150
