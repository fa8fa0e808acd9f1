#!/bin/bash
# the 1.25M-row shard step with and without the self-tightening single-launch scan (LYNSE_HIP_STS=1), blocking and 3 in flight
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 60 --warmup 5 --rows 1250000"
for sts in 0 1; do for fl in 1 3; do
  LYNSE_HIP_STS=$sts $S --in-flight $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sts $sts in_flight $fl ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'), 'fallback', d.get('fallback_queries'), 'rescored/q', d.get('rescored_per_query'), 'plan', d['roofline']['plan'], 'launch us', d['roofline']['avg_launch_us'], d.get('verify'))"
done; done
