#!/bin/bash
# round 5: the 1.25M-row shard step (3 tickets in flight) and the 10M step (2 in flight) against LYNSE_HIP_SCAN_CUS (workgroups of the persistent scans)
B="python bench.py --no-cpu-baseline --no-verify --no-configs --warmup 5"
for r in 1 2; do for c in 256 252 248 240; do
  LYNSE_HIP_SCAN_CUS=$c timeout 300 $B --rows 1250000 --steps 80 --in-flight 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shard SCAN_CUS=$c ms_per_step', d['ms_per_step'])"
done; done
for c in 256 248 240; do
  LYNSE_HIP_SCAN_CUS=$c timeout 300 $B --steps 40 --in-flight 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('10M in flight 2 SCAN_CUS=$c ms_per_step', d['ms_per_step'])"
done
