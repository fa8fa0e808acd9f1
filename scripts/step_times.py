#!/usr/bin/env python3
"""Per-step wall times of the blocking headline search (10M x 768 x 256) after a short idle: shows clock / power settling."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
dev = torch.device("cuda", 0)
n, dim, nq = 10_000_000, 768, 256
idx = L.FlatIndex(None, dim, 0); idx.reserve(n)
g = torch.Generator(device=dev); g.manual_seed(n)
for b in range(0, n, 500_000):
    idx.write_device(torch.rand((500_000, dim), generator=g, device=dev))
idx.finalize()
dq = torch.rand((nq, dim), generator=g, device=dev)
rows = torch.zeros((nq, 10), dtype=torch.int64, device=dev); d = torch.zeros((nq, 10), dtype=torch.float32, device=dev); c = torch.zeros(nq, dtype=torch.int32, device=dev)
for _ in range(3): idx.search_device(dq, 10, "ip", rows, d, c)
torch.cuda.synchronize()
for rep in range(2):
    time.sleep(0.05)
    ts = []
    for _ in range(40):
        t = time.perf_counter(); idx.search_device(dq, 10, "ip", rows, d, c); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("rep", rep, " ".join("%.2f" % x for x in ts))
