#!/bin/bash
# C3 (FLAT-L2 1M x 128, k = 100, 256 queries) against the growth factor between the threshold stages (LYNSE_HIP_SAMPLE_GROWTH; 0 = default rule)
for g in 0 2 3 4 6 8; do
  LYNSE_HIP_SAMPLE_GROWTH=$g python scripts/other_config.py c3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c3']; print('growth $g', {k: d.get(k) for k in ('ms','scan_us','GBps','oracle_parity','rescored_per_query','stages','error')})"
done
