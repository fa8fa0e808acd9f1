#!/bin/bash
# round 5, measurement 1: where the fixed work of a step goes (10M rows and the 1.25M-row shard), spin-wait A/B
export TMPDIR=/tmp
ROOT=$(pwd)
O=gpurun_out/m1; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-verify --no-configs --steps 40 --warmup 5"
# 1. blocking headline, spin-wait on / off (alternating)
for i in 1 2; do
  for sp in 20000 0; do
    LYNSE_HIP_SPIN_US=$sp timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('10M spin_us $sp ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'), 'scan_us', d['roofline'].get('avg_launch_us'))"
  done
done > $O/spin_ab.txt 2>&1
for sp in 20000 0; do
  for fl in 1 3; do
  LYNSE_HIP_SPIN_US=$sp timeout 300 $B --rows 1250000 --steps 60 --in-flight $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1.25M spin_us $sp in_flight $fl ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'))"
  done
done >> $O/spin_ab.txt 2>&1
cat $O/spin_ab.txt
# 2. select stamps at 10M and 1.25M
N=10000000 timeout 600 python scripts/dbg_sel_stamps.py > $O/sel_stamps_10m.txt 2>&1
N=1250000 timeout 600 python scripts/dbg_sel_stamps.py > $O/sel_stamps_1p25m.txt 2>&1
tail -40 $O/sel_stamps_10m.txt
# 3. kernel timelines: 10M blocking; shard with 3 in flight (a window from the MIDDLE of the timed region)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $ROOT/$O/t10m -o u --output-format csv -- bash -c "cd $ROOT && $B --steps 20" > $ROOT/$O/t10m.log 2>&1)
f=$(find $O/t10m -name "*kernel_trace.csv" | head -1); echo "== 10M blocking (tail)"; python scripts/trace_window.py $f 16 0
for fl in 1 3; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $ROOT/$O/ts$fl -o u --output-format csv -- bash -c "cd $ROOT && $B --rows 1250000 --steps 60 --in-flight $fl" > $ROOT/$O/ts$fl.log 2>&1)
f=$(find $O/ts$fl -name "*kernel_trace.csv" | head -1); echo "== 1.25M in flight $fl (window inside the timed region)"; python scripts/trace_window.py $f 22 160
done
# 4. sample-stage size sweep at 10M (fixed cost vs bytes)
for s in 16384 32768 65536 131072 262144; do
  LYNSE_HIP_SAMPLE_ROWS_TO=$s timeout 300 $B --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample_rows $s ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'), d['roofline'].get('plan'))"
done > $O/sample_sweep.txt 2>&1
cat $O/sample_sweep.txt
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
