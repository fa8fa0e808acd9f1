#!/bin/bash
# round 5: parity of the reworked sample stage / prologue + its stamps + bench (10M blocking, shard blocking / 3 in flight)
mkdir -p gpurun_out/t1
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_i8c_hostile.py tests/test_gpu_certificate.py tests/test_gpu_inflight.py -x -q -m gpu > gpurun_out/t1/pytest.txt 2>&1; tail -5 gpurun_out/t1/pytest.txt
N=10000000 timeout 600 python scripts/dbg_smp_stamps.py 2>&1 | grep -v amdgpu.ids | tail -7
N=1250000 timeout 600 python scripts/dbg_smp_stamps.py 2>&1 | grep -v amdgpu.ids | tail -7
B="python bench.py --no-cpu-baseline --no-verify --no-configs --steps 40 --warmup 5"
for i in 1 2; do timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('10M ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'), 'scan_us', d['roofline'].get('avg_launch_us'))"; done
for fl in 1 2 3; do timeout 300 $B --in-flight $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('10M in_flight $fl ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'))"; done
for fl in 1 3; do timeout 300 $B --rows 1250000 --steps 60 --in-flight $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1.25M in_flight $fl ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'))"; done
export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/t1/t10m -o u --output-format csv -- bash -c "cd $ROOT && $B --steps 20" > $ROOT/gpurun_out/t1/t10m.log 2>&1)
f=$(find gpurun_out/t1/t10m -name "*kernel_trace.csv" | head -1); echo "== 10M blocking (tail)"; python scripts/trace_window.py $f 9 0
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
