#!/bin/bash
# A/B of one environment switch on the headline bench, alternating on the same box: scripts/ab_env.sh VAR A B [extra bench args]
V=$1; A=$2; B=$3; shift 3
for r in 1 2 3; do
  for x in $A $B; do
    env $V=$x timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$V=$x', 'ms_per_step', d['ms_per_step'], 'blocking', d['blocking_ms_per_batch'], 'frac', d['roofline']['frac'], 'rescored', d['rescored_per_query'], 'fallback', d['fallback_queries'])"
  done
done
