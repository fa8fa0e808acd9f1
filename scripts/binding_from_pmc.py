#!/usr/bin/env python3
"""profiles/rNN_binding.json from the committed rocprofv3 --pmc summaries (profiles/rNN_<config>_pmc.json, scripts/summarize_pmc.py):
what LIMITS the dominant kernel of every configuration on the bench line (name the binding roofline).

    python scripts/binding_from_pmc.py r06          # reads profiles/r06_*_pmc.json (falls back to r05_* where a config was not re-profiled)

Per kernel, per dispatch (MI355X: 256 CUs x 4 SIMDs, 8 XCDs; MI355X_MICROARCH.md):
  cycles            = GRBM_GUI_ACTIVE / 8 XCDs                      (the counter sums the XCDs)
  effective clock   = cycles / average duration                     (what the 1400 W socket cap leaves under this instruction mix)
  mfma_busy_frac    = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles
  valu_busy_frac    = SQ_INSTS_VALU x 4 / 1024 / cycles             (a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles)
  hbm_frac          = FETCH_SIZE [KiB] x 1024 x 2 (gfx950 correction) / duration / 8 TB/s
The binding resource is the largest of the three; when none reaches 0.4 the kernel is latency-bound (dependent memory round trips, launch ramp,
barriers) and says so."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"

# bench-line key -> (pmc file stem, substring that names the dominant kernel of that shape)
WHAT = {
    "c2": ("c2", "k_scan_qs<6, 2, 6, 3"), "l2": ("l2", "k_scan_qs<6, 2, 6, 3"), "cosine": ("cosine", "k_scan_qs<6, 2, 6, 3"),
    "c1": ("c1", "k_small_search"), "c3": ("c3", "k_scan_qh<2, 1, 3, 2, 1, 0>"),
    "c4_share_nq256": ("c4_share", "3, 3, 2, true")   # the TILED work-list scan over the probed slabs (the <2,4,4,2> launches of that profile are the k-means assignment)
    , "c4_share_nq1": ("c4_share", "k_small_search"),
    "c5_share_nq256": ("c5_share", "k_scan_qs<4, 2, 4, 3"), "c5_share_nq1": ("c5_share", "k_scan_binary_rows"),
}
out = {}
for key, (stem, sub) in WHAT.items():
    src = next((f for f in (ROOT / "profiles" / f"{tag}_{stem}_pmc.json", ROOT / "profiles" / f"r05_{stem}_pmc.json") if f.exists()), None)
    if src is None:
        continue
    d = json.loads(src.read_text())
    cands = [(k, v) for k, v in d.items() if sub in k]
    if not cands:
        continue
    name, v = max(cands, key=lambda kv: kv[1]["total_ms"])
    c = v["counters_avg_per_dispatch"]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    us = v["avg_us"]
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cycles
    valu = c.get("SQ_INSTS_VALU", 0.0) * 4.0 / 1024.0 / cycles
    hbm = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / (us * 1e-6) / 8.0e12
    fr = {"matrix pipe (MFMA busy cycles)": mfma, "VALU (epilogue / selection arithmetic)": valu, "HBM": hbm}
    top = max(fr, key=fr.get)
    out[key] = {
        "kernel": name, "avg_us_under_profiler": us,
        # (GRBM_GUI_ACTIVE also counts the dispatch ramp around a kernel: for launches of a few tens of microseconds cycles / duration exceeds the
        # 2.4 GHz the part can clock — no clock is reported there)
        "effective_clock_GHz": (round(cycles / us / 1e3, 3) if cycles / us / 1e3 <= 2.45 else None),
        "mfma_busy_frac": round(mfma, 3), "valu_busy_frac": round(valu, 3), "hbm_frac_of_8TBps_from_FETCH_SIZE": round(hbm, 3),
        "valu_insts_per_mfma_inst": (round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2) if c.get("SQ_INSTS_MFMA") else None),
        "binding": top if fr[top] >= 0.4 else "latency: no unit is 40 % busy (dependent memory round trips, launch ramp, barriers between short phases)",
        "pmc_file": src.name,
    }
dst = ROOT / "profiles" / f"{tag}_binding.json"
dst.write_text(json.dumps(out, indent=1) + "\n")
for k, v in out.items():
    print(k.ljust(16), v["kernel"][:44].ljust(44), "clk %s" % v["effective_clock_GHz"], "mfma %.2f valu %.2f hbm %.2f" % (v["mfma_busy_frac"], v["valu_busy_frac"], v["hbm_frac_of_8TBps_from_FETCH_SIZE"]), "->", v["binding"][:40])
