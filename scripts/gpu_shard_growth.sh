#!/bin/bash
# the 1.25M-row shard step against the growth factor between threshold stages (LYNSE_HIP_SAMPLE_GROWTH; 0 = default 32: one threshold stage)
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 80 --warmup 5 --rows 1250000"
for g in 0 4 6 8 12; do for fl in 3 1; do
  LYNSE_HIP_SAMPLE_GROWTH=$g $S --in-flight $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('growth $g in_flight $fl ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'), 'rescored/q', d.get('rescored_per_query'), 'stages', d['roofline']['plan']['stages'], 'scan us/step', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))"
done; done
