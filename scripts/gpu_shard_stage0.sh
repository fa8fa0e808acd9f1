#!/bin/bash
# the 1.25M-row shard step (and the 10M headline) against the size of the threshold-only sample stage (LYNSE_HIP_SAMPLE_ROWS_TO)
for rows in 1250000 10000000; do
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 60 --warmup 5 --rows $rows"
for s0 in 0 98304 131072; do for fl in 3 1; do
  LYNSE_HIP_SAMPLE_ROWS_TO=$s0 $S --in-flight $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows $rows sample rows $s0 in_flight $fl ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'), 'fallback', d.get('fallback_queries'), 'rescored/q', d.get('rescored_per_query'), 'stages', d['roofline']['plan']['stages'], 'pipeline us', d.get('pipeline_us_per_step'), 'scan us/step', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))"
done; done; done
