#!/bin/bash
# C3 (FLAT-L2 1M x 128, k = 100, 256 queries) against the size of the threshold-only sample stage (LYNSE_HIP_SAMPLE_ROWS_TO; 0 = default 65,536)
for s0 in 0 98304 131072 196608 250000; do
  LYNSE_HIP_SAMPLE_ROWS_TO=$s0 python scripts/other_config.py c3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['c3']; print('sample rows $s0', {k: d.get(k) for k in ('ms','scan_us','oracle_parity','rescored_per_query','stages','fallback_queries','error')})"
done
