#!/bin/bash
# A/B of two builds of the library on the same box over the C3 configuration (SIFT-like 1M x 128, L2, 256 queries, k = 100):
# alternates lynsedb_amd/liblynse_hip.so (A) and lynsedb_amd/ab_*.so (B), 3 rounds each
set -u
B=${1:-lynsedb_amd/ab_nozc.so}
cp lynsedb_amd/liblynse_hip.so /tmp/A.so; cp "$B" /tmp/B.so
for r in 1 2 3; do
  for v in A B; do
    cp /tmp/$v.so lynsedb_amd/liblynse_hip.so
    echo "$v $(python scripts/other_config.py c3 2>/dev/null | grep -o '"ms": [0-9.]*, "queries_per_s": [0-9.]*, "scan_us": [0-9.]*')"
  done
done
cp /tmp/A.so lynsedb_amd/liblynse_hip.so
