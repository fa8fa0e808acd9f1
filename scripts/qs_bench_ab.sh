#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of bench.py for each scan variant: LYNSE_HIP_QS = 0 (256 x 256 tile), 1, 2, 3
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/${1:-qsab}
mkdir -p $OUT
for v in ${QS_LIST:-0 1 2 3}; do
  ( cd /tmp && LYNSE_HIP_QS=$v rocprofv3 --kernel-trace --stats -d $OUT/v$v -o v$v --output-format csv -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-verify ) > $OUT/v$v.log 2>&1
  f=$(find $OUT/v$v -name "*kernel_stats.csv" | head -1)
  echo "== LYNSE_HIP_QS=$v"; grep -o '"ms_per_step": [0-9.]*' $OUT/v$v.log | head -1
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "lynse::" not in n: continue
    if float(r["Percentage"]) < 0.3: continue
    print("  %-110s calls %4s avg %9.1f us  %5.1f%%" % (n.split("(")[0].replace("void lynse::","")[:110], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
