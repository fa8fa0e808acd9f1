#!/bin/bash
# A/B of two builds of the library on the same box, alternating: scripts/ab_lib2.sh B.so rounds [bench args...]
# (A = lynsedb_amd/liblynse_hip.so); prints ms_per_step / blocking ms of each run
set -u
B=$1; R=$2; shift 2
cp lynsedb_amd/liblynse_hip.so /tmp/A.so; cp "$B" /tmp/B.so
for r in $(seq 1 $R); do
  for v in A B; do
    cp /tmp/$v.so lynsedb_amd/liblynse_hip.so
    timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 40 --warmup 5 "$@" 2>/dev/null | tail -1 > /tmp/line.json
    python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json'))
print(sys.argv[1], 'ms_per_step', d['ms_per_step'], 'blocking', d.get('blocking_ms_per_batch'), 'scan_us', d['roofline'].get('avg_launch_us'), 'frac', d['roofline']['frac'], 'in_flight', d['config']['batches_in_flight'])
PY
  done
done
cp /tmp/A.so lynsedb_amd/liblynse_hip.so
