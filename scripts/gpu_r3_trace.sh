set -x
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/r4
B="python bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 5 --rows 1250000 --in-flight 1"
(cd /tmp && LYNSE_HIP_FUSED_SAMPLE=0 LYNSE_HIP_FUSED_TAIL=0 timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r4/unfused -o u --output-format csv -- bash -c "cd $ROOT && $B" > $ROOT/gpurun_out/r4/unfused.log 2>&1)
(cd /tmp && LYNSE_HIP_FUSED_SAMPLE=0 LYNSE_HIP_FUSED_TAIL=1 timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r4/ft -o u --output-format csv -- bash -c "cd $ROOT && $B" > $ROOT/gpurun_out/r4/ft.log 2>&1)
LYNSE_HIP_FUSED_SAMPLE=0 LYNSE_HIP_FUSED_TAIL=1 timeout 300 $B > gpurun_out/r4/b_ft.json 2>/dev/null
LYNSE_HIP_FUSED_SAMPLE=0 LYNSE_HIP_FUSED_TAIL=0 timeout 300 $B > gpurun_out/r4/b_unf.json 2>/dev/null
for d in unfused ft; do echo == $d; f=$(find gpurun_out/r4/$d -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print(r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
for f in gpurun_out/r4/*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['blocking_ms_per_batch'], d['pipeline_us_per_step'])"; done
# keep only the small csvs
find gpurun_out/r4 -name "*kernel_trace.csv" -size +2M -delete
