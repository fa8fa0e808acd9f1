#!/usr/bin/env python3
"""Looks for one-off stalls in a long run of batches in flight (1.25M x 768, 256 queries, 3 in flight): prints every step whose
completion came more than 1 ms after the previous one."""
import sys, time, gc
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
dev = torch.device("cuda", 0)
n, dim, nq, K = 1_250_000, 768, 256, 10
idx = L.FlatIndex(None, dim, 0); idx.reserve(n)
g = torch.Generator(device=dev); g.manual_seed(n)
for b in range(0, n, 250_000):
    idx.write_device(torch.rand((250_000, dim), generator=g, device=dev))
idx.finalize(); idx.prepare("ip", nq)
dq = torch.rand((nq, dim), generator=g, device=dev)
outs = [(torch.zeros((nq, K), dtype=torch.int64, device=dev), torch.zeros((nq, K), dtype=torch.float32, device=dev), torch.zeros(nq, dtype=torch.int32, device=dev)) for _ in range(3)]
if len(sys.argv) > 1 and sys.argv[1] == "nogc": gc.disable()
if len(sys.argv) > 2: idx.profile_enable(int(sys.argv[2]))
pending, last, slow = [], time.perf_counter(), []
t0 = last
for i in range(600):
    o = outs[i % 3]
    pending.append(idx.search_submit(dq, K, "ip", o[0], o[1], o[2]))
    if len(pending) >= 3:
        pending.pop(0).wait()
        now = time.perf_counter()
        if now - last > 1e-3: slow.append((i, round((now - last) * 1e3, 2)))
        last = now
for t in pending: t.wait()
print("total ms", round((time.perf_counter() - t0) * 1e3, 1), "slow steps", slow)
