#!/bin/bash
# scripts/prof.sh TAG CMD...  — rocprofv3 evidence for one command, written under gpurun_out/TAG/ on the GPU box:
#   stats/   --kernel-trace --stats            (per-kernel time)
#   pmc_a/   SQ issue / wait / MFMA-busy counters     (own pass: never combined with the trace domains gpurun refuses)
#   pmc_b/   LDS + instruction-mix counters
#   pmc_c/   FETCH_SIZE (HBM bytes; gfx950 reports 1/2 of a wide coalesced stream — MI355X_MICROARCH.md)
# Summaries are distilled into profiles/ by scripts/summarize_pmc.py.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
run() {  # name, rocprof args...
    local name=$1; shift
    ( cd /tmp && rocprofv3 "$@" -d "$OUT/$name" -o "$name" --output-format csv -- "${CMD[@]}" ) > "$OUT/$name.log" 2>&1
    echo "$name rc=$?" >> "$OUT/status.txt"
}
CMD=("$@")
# the command's relative paths resolve against the repo root
CMD=(bash -c "cd $ROOT && $(printf '%q ' "$@")")
run stats --kernel-trace --stats
run pmc_a --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
run pmc_b --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
run pmc_c --kernel-trace --pmc FETCH_SIZE
cat "$OUT/status.txt"
