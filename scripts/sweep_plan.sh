#!/bin/bash
# scripts/sweep_plan.sh ROWS IN_FLIGHT — stage-plan sweep of bench.py (sample size x growth), one line per setting
ROWS=${1:-1250000}; D=${2:-3}
for S in 65536 131072 262144; do for G in 3 4 5 6 8 12 16 32; do
  LYNSE_HIP_SAMPLE_ROWS_TO=$S LYNSE_HIP_SAMPLE_GROWTH=$G python bench.py --rows $ROWS --in-flight $D --steps 60 --warmup 6 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('rows=$ROWS D=$D sample=$S growth=$G', r['ms_per_step'], r['roofline']['launches_per_step'], r['rescored_per_query'], r['fallback_queries'])"
done; done
