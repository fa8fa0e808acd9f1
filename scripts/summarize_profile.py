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
    out += ["## HBM traffic of the scan kernel (separate `--pmc FETCH_SIZE` pass, bench.py --rows 2000000)", "",
            "FETCH_SIZE is in KiB and, on gfx950, reports 1/2 of the bytes of a wide coalesced stream "
            "(MI355X_MICROARCH.md §HBM) — corrected = FETCH_SIZE x 1024 x 2.", "",
            "| dispatch | stage rows (grid blocks x tiles) | FETCH_SIZE KiB | corrected GB | algorithmic GB | ratio |", "|---|---|---|---|---|---|"]
    plan = [4096, 32768 - 4096, 262144 - 32768, 2000000 - 262144]
    rows = [r for r in csv.DictReader(open(pmc)) if r["Counter_Name"] == "FETCH_SIZE"]
    for i, r in enumerate(rows):
        n = plan[i % 4]
        alg = n * 768 * 4 / 1e9
        corr = float(r["Counter_Value"]) * 1024 * 2 / 1e9
        out.append(f"| {r['Dispatch_Id']} | {n} | {float(r['Counter_Value']):.0f} | {corr:.4f} | {alg:.4f} | {corr/alg:.3f} |")
    out.append("")
(P / f"{tag}_summary.md").write_text("\n".join(out))
print("\n".join(out))
