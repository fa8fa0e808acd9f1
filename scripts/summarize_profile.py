#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output into profiles/rNN_summary.md (run in the repo root).

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py ...
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex k_scan --output-format csv ...   (separate pass)
"""
import csv
import json
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = Path("profiles")
out = [f"# rocprofv3 summary {tag}", ""]
stats = P / f"{tag}_bench_kernel_stats.csv"
if stats.exists():
    out += ["## kernel stats (`rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2`)", "",
            "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats)):
        name = r["Name"]
        if not ("lynse::" in name):
            continue
        short = name.split("(")[0].replace("void ", "")
        out.append(f"| `{short}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |")
    out.append("")
benchj = P / f"{tag}_bench_under_rocprof.json"
if benchj.exists():
    d = json.loads(benchj.read_text())
    rf = d["roofline"]
    out += ["## bench line of the same run", "",
            f"- {d['metric']}: **{d['value']} {d['unit']}**, {d['ms_per_step']} ms/step",
            f"- roofline (HIP events inside bench.py): {rf['achieved']} GB/s = {rf['frac']*100:.1f}% of {rf['peak']} GB/s; "
            f"avg launch {rf['avg_launch_us']} us over {rf['launches']} launches",
            f"- verify: {d.get('verify')}", ""]
pmc = P / f"{tag}_pmc_fetch_counter_collection.csv"
if pmc.exists():
    h16 = "h16" in tag  # k_scan_h16 streams the 2-byte shadow rows
    out += ["## HBM traffic of the scan kernel (separate `--pmc FETCH_SIZE` pass, bench.py --rows 2000000)", "",
            "FETCH_SIZE is in KiB and, on gfx950, reports 1/2 of the bytes of a wide coalesced stream "
            "(MI355X_MICROARCH.md §HBM) — corrected = FETCH_SIZE x 1024 x 2."
            + (" k_scan_h16 reads the f16 shadow rows: kernel bytes = rows x 768 x 2, algorithmic (SURVEY 8d) = rows x 768 x 4." if h16 else ""), "",
            "| dispatch | stage rows | FETCH_SIZE KiB | corrected GB | kernel GB | algorithmic GB | hbm / kernel | hbm / algorithmic |", "|---|---|---|---|---|---|---|---|"]
    # stage plan of bench.py --rows 2000000: f32 scan = contiguous stages; h16 = sampled plan (one 256-row tile per CU, then everything)
    plan = [65536, 2000000] if h16 else [4096, 32768 - 4096, 262144 - 32768, 2000000 - 262144]
    rows = [r for r in csv.DictReader(open(pmc)) if r["Counter_Name"] == "FETCH_SIZE"]
    last = None
    for i, r in enumerate(rows):
        n = plan[i % len(plan)]
        alg = n * 768 * 4 / 1e9
        ker = alg / 2 if h16 else alg
        corr = float(r["Counter_Value"]) * 1024 * 2 / 1e9
        out.append(f"| {r['Dispatch_Id']} | {n} | {float(r['Counter_Value']):.0f} | {corr:.4f} | {ker:.4f} | {alg:.4f} | {corr/ker:.3f} | {corr/alg:.3f} |")
        if i % len(plan) == len(plan) - 1:
            last = (n, alg, corr)
    out.append("")
    if last:
        tj = P / "r01_pmc_traffic.json"
        d = json.loads(tj.read_text()) if tj.exists() else {}
        kname = "k_scan_h16" if h16 else "k_scan_glds"
        d[kname] = {"kernel": kname, "source": f"profiles/{pmc.name} (rocprofv3 --pmc FETCH_SIZE --kernel-trace, separate pass, bench.py --rows 2000000)",
                    "correction": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950 reports 1/2 of a wide coalesced stream, MI355X_MICROARCH.md HBM section)",
                    "large_stage_rows": last[0], "algorithmic_bytes": int(last[1] * 1e9), "hbm_bytes_corrected": int(last[2] * 1e9),
                    "ratio_hbm_over_algorithmic": round(last[2] / last[1], 4)}
        tj.write_text(json.dumps(d, indent=2))
(P / f"{tag}_summary.md").write_text("\n".join(out))
print("\n".join(out))
