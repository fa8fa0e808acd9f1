# round-3 evidence: bench (full line), rocprofv3 stats + PMC passes of the headline command, shard-size runs
set -x
mkdir -p gpurun_out/r03
(time timeout 900 python bench.py) > gpurun_out/r03/bench_full.json 2> gpurun_out/r03/bench_full.err
bash scripts/prof.sh r03_c2 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 15 --warmup 3 > gpurun_out/r03/prof_c2.log 2>&1
python scripts/summarize_pmc.py gpurun_out/r03_c2 gpurun_out/r03/r03_c2_pmc k_scan_h16 k_select k_select_final k_i8c_prep_queries > gpurun_out/r03/sum.log 2>&1
python scripts/pmc_traffic.py gpurun_out/r03_c2 gpurun_out/r03/r03_pmc_traffic.json 10000000 768 0 >> gpurun_out/r03/sum.log 2>&1
f=$(find gpurun_out/r03_c2/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03/r03_c2_kernel_stats.csv
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 60 --warmup 5 --rows 1250000"
LYNSE_BENCH_FORCE_COMM=1 $S --in-flight 3 > gpurun_out/r03/shard_1p25m_in_flight_1rank_comm.json 2>/dev/null
$S --in-flight 3 > gpurun_out/r03/shard_1p25m_in_flight.json 2>/dev/null
$S --in-flight 1 > gpurun_out/r03/shard_1p25m_blocking.json 2>/dev/null
find gpurun_out/r03_c2 -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out/r03_c2 -name "*counter_collection.csv" -size +12M -delete
tail -3 gpurun_out/r03/bench_full.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if "ms_per_step" in d: print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("verify"))
    except Exception as e: print(f, "ERR", e)
PY
head -12 gpurun_out/r03/r03_c2_kernel_stats.csv
cat gpurun_out/r03/sum.log | tail -5
