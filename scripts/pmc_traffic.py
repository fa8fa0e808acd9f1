#!/usr/bin/env python3
"""HBM traffic of the scan kernels from the FETCH_SIZE pass of scripts/prof.sh -> profiles/r02_pmc_traffic.json.

    python scripts/pmc_traffic.py gpurun_out/TAG profiles/r02_pmc_traffic.json ROWS DIM STEPS_IN_RUN

FETCH_SIZE is reported in KiB and, on gfx950, tallies a wide coalesced stream at half its bytes
(MI355X_MICROARCH.md, HBM section): corrected bytes = FETCH_SIZE x 1024 x 2.  The ratio is taken over ALL launches of the
threshold-stage scan kernel in the run against the bytes those launches have to stream (rows x padded row bytes of the
coarse-pass copy: 1 B / element for the certified int8 pass, 2 B for the f16 shadow)."""
import csv
import json
import re
import sys
from pathlib import Path

csv.field_size_limit(1 << 30)
src, dst, rows, dim, steps = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
f = sorted((src / "pmc_c").rglob("*counter_collection.csv"))[0]
tot = {}
kname = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    mq = re.search(r"lynse::k_scan_qs<([^>]*)>", r["Kernel_Name"])   # the query-stationary tiling: threshold stages of the int8 pass
    m = re.search(r"lynse::k_scan_h16<([^>]*)>", r["Kernel_Name"])
    if mq:
        qa = [x.strip() for x in mq.group(1).split(",")]
        if len(qa) > 10 and qa[10] == "1":
            continue   # SMP: the sample stage on the same tiling (its 65,536 rows are streamed again by the threshold stages)
        key = ("i8c", "0")
        kname[key] = "k_scan_qs<%s>" % mq.group(1)
    elif m:
        a = [x.strip() for x in m.group(1).split(",")]
        i8q, emit = (a[12] if len(a) > 12 else "0"), (a[13] if len(a) > 13 else "-1")
        key = ("i8c" if i8q == "2" else "f16", emit)
    else:
        continue
    t = tot.setdefault(key, {"fetch_kib": 0.0, "dispatches": set()})
    t["fetch_kib"] += float(r["Counter_Value"])
    t["dispatches"].add(r["Dispatch_Id"])
out = {}
for (kind, emit), t in tot.items():
    if emit != "0":
        continue  # threshold stages: together they stream every row of the shard exactly once per step
    elem = 1 if kind == "i8c" else 2
    pad = 16 if kind == "i8c" else 8
    if steps <= 0:   # 0 = derive from the trace: every step of the headline plan launches TWO threshold stages (two-level + DENSE)
        steps = len(t["dispatches"]) // 2
    stream = rows * (-(-dim // pad) * pad) * elem * steps
    hbm = t["fetch_kib"] * 1024 * 2
    out[kind] = {"kernel": kname.get((kind, emit), "k_scan_h16<%s, EMIT=0>" % kind), "source": str(f.relative_to(src.parent)) if src.parent in f.parents else str(f),
                 "launches": len(t["dispatches"]), "steps": steps, "rows": rows, "dim": dim,
                 "kernel_stream_bytes": stream, "hbm_bytes_corrected": int(hbm), "ratio_hbm_over_kernel_bytes": round(hbm / stream, 4),
                 "correction": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950, MI355X_MICROARCH.md HBM section)"}
dst.write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out))
