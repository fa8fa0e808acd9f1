#!/bin/bash
# where the IVF training of --config c4 goes: rocprofv3 kernel stats of a 2M-row build (3 steps of search only)
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/tp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/tp/c4 -o u --output-format csv -- bash -c "cd $ROOT && python bench.py --config c4 --rows-per-gpu 2000000 --steps 3 --warmup 1" > $ROOT/gpurun_out/tp/c4.log 2>&1)
tail -c 600 gpurun_out/tp/c4.log | grep -o '"train_s": [0-9.]*'
f=$(find gpurun_out/tp/c4 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
find gpurun_out/tp -name "*kernel_trace.csv" -size +2M -delete
