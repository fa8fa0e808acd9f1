#!/bin/bash
# round-5 evidence, second part (the default bench line is blocking on one GPU again): the full line, rocprofv3 stats + PMC passes of the
# headline command and of its L2 / cosine variants, PMC traffic
set -x
mkdir -p gpurun_out/r05
(time timeout 1500 python bench.py) > gpurun_out/r05/bench_full.json 2> gpurun_out/r05/bench_full.err
rm -rf gpurun_out/r05_c2 gpurun_out/r05_l2 gpurun_out/r05_cosine
bash scripts/prof.sh r05_c2 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 18 --warmup 3 > gpurun_out/r05/prof_c2.log 2>&1
python scripts/summarize_pmc.py gpurun_out/r05_c2 gpurun_out/r05/r05_c2_pmc k_scan_qs k_scan_h16 k_select k_select_final k_i8c_prep_queries > gpurun_out/r05/sum2.log 2>&1
python scripts/pmc_traffic.py gpurun_out/r05_c2 gpurun_out/r05/r05_pmc_traffic.json 10000000 768 0 >> gpurun_out/r05/sum2.log 2>&1
f=$(find gpurun_out/r05_c2/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/r05_c2_kernel_stats.csv
for M in l2 cosine; do
  bash scripts/prof.sh r05_$M python bench.py --metric $M --no-cpu-baseline --no-configs --no-verify --steps 12 --warmup 3 > gpurun_out/r05/prof_$M.log 2>&1
  python scripts/summarize_pmc.py gpurun_out/r05_$M gpurun_out/r05/r05_${M}_pmc k_scan >> gpurun_out/r05/sum2.log 2>&1
  f=$(find gpurun_out/r05_$M/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/r05_${M}_kernel_stats.csv
done
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out -name "*counter_collection.csv" -size +12M -delete
tail -3 gpurun_out/r05/bench_full.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05/bench_full.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("avg_launch_us"), d["blocking_ms_per_batch"], d["two_in_flight_ms_per_step"], d["config"]["batches_in_flight"])
print({k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d.get("same_shard_variants", {}).items()})
print(d.get("second_distribution"))
print({k: {kk: vv for kk, vv in v.items() if kk in ("ms","ms_per_batch","frac_of_hbm_peak","oracle_parity","nq256","nq1")} for k, v in d.get("configs", {}).items() if isinstance(v, dict)})
print(d.get("cpu_baseline"))
PY
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r05/r05_c2_kernel_stats.csv")))[:12]:
    if "lynse" in r["Name"]: print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
cat gpurun_out/r05/r05_pmc_traffic.json
