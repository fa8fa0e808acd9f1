#!/usr/bin/env python3
"""Decomposition experiment (EXPERIMENTS build): time of the scan launches of one 10M x 768 x 256 IP batch with compile-time
pieces of the hot kernel removed (LYNSE_HIP_DEBUG_FLAGS = DBG << 8).  Results are wrong by construction; only time matters."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
code = r'''
import sys, json, torch
sys.path.insert(0, %r)
import lynsedb_amd as L
dev = torch.device("cuda", 0)
N, D = 10_000_000, 768
idx = L.FlatIndex(None, D, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for b in range(0, N, 500_000):
    idx.write_device(torch.rand((500_000, D), generator=g, device=dev))
idx.finalize()
q = torch.rand((256, D), generator=g, device=dev)
rows = torch.zeros((256, 10), dtype=torch.int64, device=dev); d = torch.zeros((256, 10), device=dev); c = torch.zeros(256, dtype=torch.int32, device=dev)
for _ in range(3): idx.search_device(q, 10, "ip", rows, d, c)
idx.profile_enable(True); idx.profile_get(reset=True)
for _ in range(5): idx.search_device(q, 10, "ip", rows, d, c)
p = idx.profile_get(reset=True)
print(json.dumps({"scan_us_per_step": round(p["scan_us"] / 5, 1), "launches": p["scan_launches"] // 5}))
''' % str(ROOT)
names = {0: "everything", 3: "DMA only (no MFMA, no LDS reads)", 1: "no MFMA (DMA + LDS reads)", 2: "no LDS reads (DMA + MFMA)",
         12: "MFMA + LDS reads only (no DMA)", 13: "LDS reads only", 14: "MFMA only", 4: "no query-image DMA", 8: "no row DMA",
         7: "row DMA only", 11: "query-image DMA only"}
for dbg, name in names.items():
    env = dict(os.environ, LYNSE_HIP_DEBUG_FLAGS=str((dbg << 8) | 2))  # | 2: no emission (garbage scores must not flood the candidate buffers)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    print(dbg, name, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
