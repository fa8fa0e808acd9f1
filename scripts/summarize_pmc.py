#!/usr/bin/env python3
"""Distil the rocprofv3 passes of scripts/prof.sh into one markdown + json summary under profiles/.

    python scripts/summarize_pmc.py gpurun_out/TAG profiles/r02_TAG [kernel-substring ...]

Per lynse:: kernel (aggregated over its dispatches, largest-grid dispatches listed separately for the scan kernels):
calls, average duration (kernel trace), the SQ counters of pass a / b as per-dispatch averages, derived ratios
(MFMA-busy share of the CU-cycles, issue-stall and wait shares of wave-cycles) and FETCH_SIZE corrected for gfx950
(x2, MI355X_MICROARCH.md HBM section).
"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

csv.field_size_limit(1 << 30)
src, dst = Path(sys.argv[1]), Path(sys.argv[2])
filt = sys.argv[3:]


def short(name):
    m = re.search(r"lynse::(\w+)(<[^>]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")


def find(pass_name, suffix):
    d = src / pass_name
    hits = sorted(d.rglob(f"*{suffix}")) if d.exists() else []
    return hits[0] if hits else None


kern = defaultdict(lambda: {"calls": 0, "dur_ns": 0.0, "counters": defaultdict(float), "ndisp": defaultdict(int)})
stats_trace = find("stats", "kernel_trace.csv")
if stats_trace:
    for r in csv.DictReader(open(stats_trace)):
        s = short(r["Kernel_Name"])
        if not s or (filt and not any(f in s for f in filt)):
            continue
        k = kern[s]
        k["calls"] += 1
        k["dur_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        k["grid"] = max(k.get("grid", 0), int(r["Grid_Size_X"]))
        k["vgpr"] = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
        k["lds"] = int(r["LDS_Block_Size"])
for p in ("pmc_a", "pmc_b", "pmc_c"):
    f = find(p, "counter_collection.csv")
    if not f:
        continue
    seen = set()
    for r in csv.DictReader(open(f)):
        s = short(r["Kernel_Name"])
        if not s or s not in kern:
            continue
        c = r["Counter_Name"]
        kern[s]["counters"][c] += float(r["Counter_Value"])
        key = (s, c, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            kern[s]["ndisp"][c] += 1
            kern[s].setdefault("pmc_dur_ns_" + p, 0.0)
            if c in ("SQ_WAVE_CYCLES", "SQ_INSTS_MFMA", "FETCH_SIZE"):
                kern[s]["pmc_dur_ns_" + p] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

out = {}
lines = [f"# rocprofv3 counters — {src.name}", "",
         "Passes: `--kernel-trace --stats`, `--pmc` SQ issue/wait/MFMA (a), LDS + mix (b), FETCH_SIZE (c); each its own run "
         "(scripts/prof.sh).  Counter values are per-dispatch averages; SQ_* cycle counters are in quad-cycles summed over "
         "waves / CUs as rocprofv3 reports them, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES in cycles.", ""]
for s, k in sorted(kern.items(), key=lambda kv: -kv[1]["dur_ns"]):
    calls = max(k["calls"], 1)
    avg_us = k["dur_ns"] / calls / 1e3
    c = {n: v / max(k["ndisp"][n], 1) for n, v in k["counters"].items()}
    d = {"calls": k["calls"], "avg_us": round(avg_us, 2), "total_ms": round(k["dur_ns"] / 1e6, 3), "grid": k.get("grid"),
         "vgpr+agpr": k.get("vgpr"), "lds_bytes": k.get("lds"), "counters_avg_per_dispatch": {n: round(v, 1) for n, v in sorted(c.items())}}
    der = {}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if n in c:
                der[n + "/SQ_WAVE_CYCLES"] = round(c[n] / wc, 4)
    if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # SQ_BUSY_CYCLES is summed over the SQs that were busy (one per XCD-SE group); the MFMA-busy counter over SIMDs.
        der["SQ_VALU_MFMA_BUSY_CYCLES/SQ_BUSY_CYCLES"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], 4)
    if "SQ_INSTS_MFMA" in c and c.get("GRBM_GUI_ACTIVE"):
        der["mfma_insts_per_gui_cycle"] = round(c["SQ_INSTS_MFMA"] / c["GRBM_GUI_ACTIVE"], 4)
    if "SQ_INSTS_MFMA" in c and "SQ_INSTS_VALU" in c:
        der["valu_insts_per_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / max(c["SQ_INSTS_MFMA"], 1), 3)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "FETCH_SIZE" in c:
        der["hbm_bytes_corrected_per_dispatch"] = int(c["FETCH_SIZE"] * 1024 * 2)
    if c.get("GRBM_GUI_ACTIVE") and k.get("pmc_dur_ns_pmc_b"):
        nd = max(k["ndisp"].get("GRBM_GUI_ACTIVE", 1), 1)
        der["effective_clock_GHz"] = round(c["GRBM_GUI_ACTIVE"] / (k["pmc_dur_ns_pmc_b"] / nd), 3)
    d["derived"] = der
    out[s] = d
    lines += [f"## `{s}`", "", f"- calls {k['calls']}, avg {avg_us:.1f} us, total {k['dur_ns']/1e6:.2f} ms, grid {k.get('grid')}, "
              f"VGPR+AGPR {k.get('vgpr')}, LDS {k.get('lds')} B", ""]
    if c:
        lines += ["| counter | avg / dispatch |", "|---|---|"] + [f"| {n} | {v:,.0f} |" for n, v in sorted(c.items())] + [""]
    if der:
        lines += ["| derived | value |", "|---|---|"] + [f"| {n} | {v} |" for n, v in der.items()] + [""]
dst.parent.mkdir(parents=True, exist_ok=True)
Path(str(dst) + ".md").write_text("\n".join(lines) + "\n")
Path(str(dst) + ".json").write_text(json.dumps(out, indent=1) + "\n")
print(f"wrote {dst}.md / .json ({len(out)} kernels)")
