"""FLAT-IP-SQ8 (two-pass) against the exact search on 10M x 768, 256 queries (host API)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L
dev = torch.device("cuda", 0)
n, dim = 10_000_000, 768
idx = L.FlatIndex(None, dim, 0); idx.reserve(n)
g = torch.Generator(device=dev); g.manual_seed(1)
for b in range(0, n, 500_000):
    idx.write_device(torch.rand((500_000, dim), generator=g, device=dev))
idx.finalize()
q = torch.rand((256, dim), generator=g, device=dev).cpu().numpy()
for name, fn in (("exact", lambda: idx.search_batch_arrays(q, 10, "ip")), ("sq8 two-pass", lambda: idx.search_sq8_batch_arrays(q, 10, "ip"))):
    for _ in range(3): fn()
    ts = []
    for _ in range(8):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print(name, "median ms", round(float(np.median(ts)) * 1e3, 3))
