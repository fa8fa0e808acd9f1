#!/bin/bash
# round-5 evidence, third part: rocprofv3 stats + PMC of the lean headline command (blocking steps only) and its L2 / cosine variants
set -x
mkdir -p gpurun_out/r05; rm -rf gpurun_out/r05_c2 gpurun_out/r05_l2 gpurun_out/r05_cosine
bash scripts/prof.sh r05_c2 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 18 --warmup 3 > gpurun_out/r05/prof_c2.log 2>&1
python scripts/summarize_pmc.py gpurun_out/r05_c2 gpurun_out/r05/r05_c2_pmc k_scan_qs k_scan_h16 k_select k_select_final k_i8c_prep_queries > gpurun_out/r05/sum2.log 2>&1
python scripts/pmc_traffic.py gpurun_out/r05_c2 gpurun_out/r05/r05_pmc_traffic.json 10000000 768 0 >> gpurun_out/r05/sum2.log 2>&1
f=$(find gpurun_out/r05_c2/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/r05_c2_kernel_stats.csv
for M in l2 cosine; do
  bash scripts/prof.sh r05_$M python bench.py --metric $M --no-cpu-baseline --no-configs --no-verify --steps 12 --warmup 3 > gpurun_out/r05/prof_$M.log 2>&1
  python scripts/summarize_pmc.py gpurun_out/r05_$M gpurun_out/r05/r05_${M}_pmc k_scan >> gpurun_out/r05/sum2.log 2>&1
  f=$(find gpurun_out/r05_$M/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/r05_${M}_kernel_stats.csv
done
timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 18 --warmup 3 | tail -1 > gpurun_out/r05/bench_lean_no_profiler.json
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out -name "*counter_collection.csv" -size +12M -delete
python - <<'PY'
import csv, json
for tag in ("c2", "l2", "cosine"):
    print("==", tag)
    for r in list(csv.DictReader(open("gpurun_out/r05/r05_%s_kernel_stats.csv" % tag)))[:12]:
        if "lynse" in r["Name"]: print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1))
d=json.loads(open("gpurun_out/r05/bench_lean_no_profiler.json").read())
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_per_step"])
PY
