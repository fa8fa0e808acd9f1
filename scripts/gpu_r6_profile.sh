#!/bin/bash
# round-6 evidence: the full default bench line, rocprofv3 stats + the three PMC passes of the lean headline command, its L2 / cosine variants and
# every other configuration on the bench line, --config c4 / c5, the 1.25M-row shard step (blocking / 3 in flight / 1-rank communicator) with its
# kernel timeline, the latency decomposition of the blocking calls.  Everything lands under gpurun_out/r06/; the judged copies go to profiles/r06_*.
set -x
O=gpurun_out/r06
mkdir -p $O
(time timeout 1500 python bench.py) > $O/bench_full.json 2> $O/bench_full.err
bash scripts/prof.sh r06_c2 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 18 --warmup 3 > $O/prof_c2.log 2>&1
python scripts/summarize_pmc.py gpurun_out/r06_c2 $O/r06_c2_pmc k_scan_qs k_scan_h16 k_select k_select_final k_i8c_prep_queries > $O/sum.log 2>&1
python scripts/pmc_traffic.py gpurun_out/r06_c2 $O/r06_pmc_traffic.json 10000000 768 0 >> $O/sum.log 2>&1
f=$(find gpurun_out/r06_c2/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_c2_kernel_stats.csv
timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 18 --warmup 3 | tail -1 > $O/bench_lean_no_profiler.json
for M in l2 cosine; do
  bash scripts/prof.sh r06_$M python bench.py --metric $M --no-cpu-baseline --no-configs --no-verify --steps 12 --warmup 3 > $O/prof_$M.log 2>&1
  python scripts/summarize_pmc.py gpurun_out/r06_$M $O/r06_${M}_pmc k_scan >> $O/sum.log 2>&1
  f=$(find gpurun_out/r06_$M/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_${M}_kernel_stats.csv
done
export LYNSE_BENCH_NO_INFLIGHT=1 LYNSE_BENCH_C4_ITERS=2 LYNSE_BENCH_C4_NO_SECOND=1
for C in c1 c3 c5_share c4_share; do
  bash scripts/prof.sh r06_$C python scripts/other_config.py $C > $O/prof_$C.log 2>&1
  python scripts/summarize_pmc.py gpurun_out/r06_$C $O/r06_${C}_pmc >> $O/sum.log 2>&1
  f=$(find gpurun_out/r06_$C/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_${C}_kernel_stats.csv
done
unset LYNSE_BENCH_NO_INFLIGHT LYNSE_BENCH_C4_ITERS LYNSE_BENCH_C4_NO_SECOND
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out -name "*counter_collection.csv" -size +12M -delete
(time timeout 900 python bench.py --config c4 --steps 30 --warmup 3) > $O/bench_c4.json 2> $O/bench_c4.err
(time timeout 900 python bench.py --config c5 --steps 30 --warmup 3) > $O/bench_c5.json 2> $O/bench_c5.err
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 80 --warmup 5 --rows 1250000"
LYNSE_BENCH_FORCE_COMM=1 $S --in-flight 3 > $O/shard_1p25m_in_flight_1rank_comm.json 2>/dev/null
$S --in-flight 3 > $O/shard_1p25m_in_flight.json 2>/dev/null
$S --in-flight 1 > $O/shard_1p25m_blocking.json 2>/dev/null
LYNSE_BENCH_FORCE_COMM=1 $S --in-flight 3 > $O/shard_1p25m_in_flight_1rank_comm_b.json 2>/dev/null
$S --in-flight 3 > $O/shard_1p25m_in_flight_b.json 2>/dev/null
bash scripts/gpu_shard_timeline.sh > $O/shard_timeline.txt 2>&1
python scripts/r6_latency.py c1 c3 c4 2>/dev/null | grep config > $O/latency.jsonl
for C in c1 c3 c4; do bash scripts/gpu_r6_tl.sh $C 16 > $O/timeline_$C.txt 2>&1; done
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
python - <<'PY'
import csv, json
for tag in ("c2", "l2", "cosine", "c1", "c3", "c5_share", "c4_share"):
    print("==", tag)
    try:
        for r in list(csv.DictReader(open("gpurun_out/r06/r06_%s_kernel_stats.csv" % tag)))[:10]:
            if "lynse" in r["Name"]: print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1))
    except Exception as e: print(e)
d=json.loads(open("gpurun_out/r06/bench_lean_no_profiler.json").read())
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_per_step"])
for f in ("shard_1p25m_in_flight", "shard_1p25m_in_flight_b", "shard_1p25m_blocking", "shard_1p25m_in_flight_1rank_comm", "shard_1p25m_in_flight_1rank_comm_b"):
    try: print(f, json.loads(open("gpurun_out/r06/%s.json" % f).read().strip().splitlines()[-1])["ms_per_step"])
    except Exception as e: print(f, e)
PY
