#!/bin/bash
# round 5, last binary: rocprofv3 stats + the three PMC passes of C3 (scripts/other_config.py c3) with k_scan_qh as the default
mkdir -p gpurun_out/r05b
LYNSE_BENCH_NO_INFLIGHT=1 bash scripts/prof.sh r05b_c3 python scripts/other_config.py c3 > gpurun_out/r05b/prof_c3.log 2>&1
python scripts/summarize_pmc.py gpurun_out/r05b_c3 gpurun_out/r05b/r05_c3_pmc k_scan_qh k_scan_h16 k_select k_select_final k_prep_queries > gpurun_out/r05b/sum.log 2>&1
f=$(find gpurun_out/r05b_c3/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05b/r05_c3_kernel_stats.csv
find gpurun_out -name "*kernel_trace.csv" -size +3M -delete
find gpurun_out -name "*counter_collection.csv" -size +12M -delete
head -6 gpurun_out/r05b/r05_c3_kernel_stats.csv | cut -c1-180; head -30 gpurun_out/r05b/r05_c3_pmc.md
