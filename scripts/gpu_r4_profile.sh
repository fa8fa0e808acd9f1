#!/bin/bash
# round-4 evidence: bench (full line), rocprofv3 stats + PMC passes of the headline command, of the other BASELINE configurations and of
# the L2 / cosine variants, shard-size runs.  Everything lands under gpurun_out/r04/; the judged copies go to profiles/r04_*.
set -x
mkdir -p gpurun_out/r04
(time timeout 1200 python bench.py) > gpurun_out/r04/bench_full.json 2> gpurun_out/r04/bench_full.err
bash scripts/prof.sh r04_c2 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 15 --warmup 3 > gpurun_out/r04/prof_c2.log 2>&1
python scripts/summarize_pmc.py gpurun_out/r04_c2 gpurun_out/r04/r04_c2_pmc k_scan_qs k_scan_h16 k_select k_select_final k_i8c_prep_queries > gpurun_out/r04/sum.log 2>&1
python scripts/pmc_traffic.py gpurun_out/r04_c2 gpurun_out/r04/r04_pmc_traffic.json 10000000 768 0 >> gpurun_out/r04/sum.log 2>&1
f=$(find gpurun_out/r04_c2/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r04/r04_c2_kernel_stats.csv
for C in c1 c3 c4_share c5_share; do
  bash scripts/prof.sh r04_$C python scripts/other_config.py $C > gpurun_out/r04/prof_$C.log 2>&1
  python scripts/summarize_pmc.py gpurun_out/r04_$C gpurun_out/r04/r04_${C}_pmc >> gpurun_out/r04/sum.log 2>&1
  f=$(find gpurun_out/r04_$C/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r04/r04_${C}_kernel_stats.csv
done
for M in l2 cosine; do
  bash scripts/prof.sh r04_$M python bench.py --metric $M --no-cpu-baseline --no-configs --no-verify --steps 10 --warmup 3 > gpurun_out/r04/prof_$M.log 2>&1
  python scripts/summarize_pmc.py gpurun_out/r04_$M gpurun_out/r04/r04_${M}_pmc k_scan >> gpurun_out/r04/sum.log 2>&1
  f=$(find gpurun_out/r04_$M/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r04/r04_${M}_kernel_stats.csv
done
# --config c4 (IVF tickets, one GPU = one eighth of the 50M collection): the line and a kernel trace of it
(time timeout 600 python bench.py --config c4 --steps 30 --warmup 3) > gpurun_out/r04/bench_c4.json 2> gpurun_out/r04/bench_c4.err
export TMPDIR=/tmp; R=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_c4bench -o c4 --output-format csv -- bash -c "cd $R && python bench.py --config c4 --steps 30 --warmup 3" > $R/gpurun_out/r04/prof_c4bench.log 2>&1)
f=$(find gpurun_out/r04_c4bench -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r04/r04_c4_tickets_kernel_stats.csv
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 60 --warmup 5 --rows 1250000"
LYNSE_BENCH_FORCE_COMM=1 $S --in-flight 3 > gpurun_out/r04/shard_1p25m_in_flight_1rank_comm.json 2>/dev/null
$S --in-flight 3 > gpurun_out/r04/shard_1p25m_in_flight.json 2>/dev/null
$S --in-flight 1 > gpurun_out/r04/shard_1p25m_blocking.json 2>/dev/null
# where the IVF training goes (2M-row build), the select stamps in the assignment's shape, the phases of the single-query search
bash scripts/gpu_c4_train_prof.sh > gpurun_out/r04/c4_train_prof.log 2>&1
cp gpurun_out/tp/c4/u_kernel_stats.csv gpurun_out/r04/r04_c4_train_kernel_stats.csv
python scripts/dbg_assign_stamps.py > gpurun_out/r04/assign_stamps.log 2>&1
python scripts/c1_phases.py 2> gpurun_out/r04/c1_phases.log > /dev/null
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out -name "*counter_collection.csv" -size +12M -delete
tail -3 gpurun_out/r04/bench_full.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if "ms_per_step" in d: print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("verify"))
    except Exception as e: print(f, "ERR", e)
PY
head -12 gpurun_out/r04/r04_c2_kernel_stats.csv
tail -5 gpurun_out/r04/sum.log
