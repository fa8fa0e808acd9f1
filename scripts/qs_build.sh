#!/bin/bash
# builds the stand-alone int8-scan A/B (scripts/qs_microbench.hip) into build/ (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -mllvm -pragma-unroll-threshold=262144 ${QS_DEFS:--DLYNSE_EXPERIMENTS} -o build/qs_microbench scripts/qs_microbench.hip
