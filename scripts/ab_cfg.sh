#!/bin/bash
# A/B of two builds of the library on one of the extra configurations (scripts/other_config.py <cfg>), alternating on the same box:
# scripts/ab_cfg.sh B.so rounds cfg   (A = lynsedb_amd/liblynse_hip.so)
set -u
B=$1; R=$2; CFG=$3
cp lynsedb_amd/liblynse_hip.so /tmp/A.so; cp "$B" /tmp/B.so
for r in $(seq 1 $R); do
  for v in A B; do
    cp /tmp/$v.so lynsedb_amd/liblynse_hip.so
    timeout 600 python scripts/other_config.py $CFG 2>/dev/null | tail -1 > /tmp/cfg.json
    python - "$v" "$CFG" <<'PY'
import json,sys
d=json.load(open('/tmp/cfg.json'))
c=d.get(sys.argv[2], d)
keep={k:v for k,v in c.items() if k in ("ms","ms_per_batch","scan_us","frac_of_hbm_peak","oracle_parity","GBps")}
print(sys.argv[1], keep if keep else str(c)[:400])
PY
  done
done
cp /tmp/A.so lynsedb_amd/liblynse_hip.so
