#!/bin/bash
# A/B of two builds of the library on the same box: alternates lynsedb_amd/liblynse_hip.so (A) and lynsedb_amd/ab_*.so (B)
# over the headline bench (no CPU baseline, no extra configs), 3 rounds each, and prints ms_per_step / scan us of each run.
set -u
B=${1:-lynsedb_amd/ab_nozc.so}
mkdir -p gpurun_out
cp lynsedb_amd/liblynse_hip.so /tmp/A.so; cp "$B" /tmp/B.so
for r in 1 2 3; do
  for v in A B; do
    cp /tmp/$v.so lynsedb_amd/liblynse_hip.so
    timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 30 --warmup 5 2>/dev/null | tail -1 > /tmp/line.json
    python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json'))
print(sys.argv[1], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'blocking', d.get('blocking_ms_per_batch'))
PY
  done
done
cp /tmp/A.so lynsedb_amd/liblynse_hip.so
