#!/bin/bash
# round-5 evidence: the full default bench line, rocprofv3 stats + PMC passes of the headline command and of the other BASELINE configurations,
# the L2 / cosine variants, --config c4 / c5, the 1.25M-row shard step (blocking / 3 in flight / 1-rank communicator) with its kernel timeline,
# select + sample-stage stamps.  Everything lands under gpurun_out/r05/; the judged copies go to profiles/r05_*.
set -x
mkdir -p gpurun_out/r05
(time timeout 1500 python bench.py) > gpurun_out/r05/bench_full.json 2> gpurun_out/r05/bench_full.err
bash scripts/prof.sh r05_c2 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 18 --warmup 3 > gpurun_out/r05/prof_c2.log 2>&1
python scripts/summarize_pmc.py gpurun_out/r05_c2 gpurun_out/r05/r05_c2_pmc k_scan_qs k_scan_h16 k_select k_select_final k_i8c_prep_queries > gpurun_out/r05/sum.log 2>&1
python scripts/pmc_traffic.py gpurun_out/r05_c2 gpurun_out/r05/r05_pmc_traffic.json 10000000 768 0 >> gpurun_out/r05/sum.log 2>&1
f=$(find gpurun_out/r05_c2/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/r05_c2_kernel_stats.csv
for C in c1 c3 c4_share c5_share; do
  bash scripts/prof.sh r05_$C python scripts/other_config.py $C > gpurun_out/r05/prof_$C.log 2>&1
  python scripts/summarize_pmc.py gpurun_out/r05_$C gpurun_out/r05/r05_${C}_pmc >> gpurun_out/r05/sum.log 2>&1
  f=$(find gpurun_out/r05_$C/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/r05_${C}_kernel_stats.csv
done
for M in l2 cosine; do
  bash scripts/prof.sh r05_$M python bench.py --metric $M --no-cpu-baseline --no-configs --no-verify --steps 12 --warmup 3 > gpurun_out/r05/prof_$M.log 2>&1
  python scripts/summarize_pmc.py gpurun_out/r05_$M gpurun_out/r05/r05_${M}_pmc k_scan >> gpurun_out/r05/sum.log 2>&1
  f=$(find gpurun_out/r05_$M/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/r05_${M}_kernel_stats.csv
done
(time timeout 900 python bench.py --config c4 --steps 30 --warmup 3) > gpurun_out/r05/bench_c4.json 2> gpurun_out/r05/bench_c4.err
(time timeout 900 python bench.py --config c5 --steps 30 --warmup 3) > gpurun_out/r05/bench_c5.json 2> gpurun_out/r05/bench_c5.err
export TMPDIR=/tmp; R=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_c4train -o c4 --output-format csv -- bash -c "cd $R && python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline --no-second-dataset" > $R/gpurun_out/r05/prof_c4train.log 2>&1)
f=$(find gpurun_out/r05_c4train -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05/r05_c4_train_kernel_stats.csv
S="timeout 300 python bench.py --no-cpu-baseline --no-configs --no-verify --steps 80 --warmup 5 --rows 1250000"
LYNSE_BENCH_FORCE_COMM=1 $S --in-flight 3 > gpurun_out/r05/shard_1p25m_in_flight_1rank_comm.json 2>/dev/null
$S --in-flight 3 > gpurun_out/r05/shard_1p25m_in_flight.json 2>/dev/null
$S --in-flight 1 > gpurun_out/r05/shard_1p25m_blocking.json 2>/dev/null
LYNSE_BENCH_FORCE_COMM=1 $S --in-flight 3 > gpurun_out/r05/shard_1p25m_in_flight_1rank_comm_b.json 2>/dev/null
bash scripts/gpu_shard_timeline.sh > gpurun_out/r05/shard_timeline.txt 2>&1
N=10000000 timeout 600 python scripts/dbg_sel_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/sel_stamps_10m.txt
N=10000000 timeout 600 python scripts/dbg_smp_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/smp_stamps_10m.txt
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out -name "*counter_collection.csv" -size +12M -delete
tail -3 gpurun_out/r05/bench_full.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if "ms_per_step" in d: print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("avg_launch_us"), d.get("verify"))
    except Exception as e: print(f, "ERR", e)
PY
head -12 gpurun_out/r05/r05_c2_kernel_stats.csv
tail -5 gpurun_out/r05/sum.log
