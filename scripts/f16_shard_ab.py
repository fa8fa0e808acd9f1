#!/usr/bin/env python3
"""F16 shard 10M x 768 (one f16 copy of the rows), 256 / 32 / 1 queries, IP: median ms per call; LYNSE_HIP_COARSE=f16 for the A/B."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
dev = torch.device("cuda", 0)
n, dim = 10_000_000, 768
idx = L.FlatIndex(None, dim, 0, dtype="f16"); idx.reserve(n)
g = torch.Generator(device=dev); g.manual_seed(n)
for b in range(0, n, 500_000):
    idx.write_device(torch.rand((500_000, dim), generator=g, device=dev))
idx.finalize()
q_all = torch.rand((256, dim), generator=g, device=dev)
print("hbm GB", round(idx.hbm_bytes() / 1e9, 2))
for nq in (256, 32, 1):
    dq = q_all[:nq].contiguous()
    rows = torch.zeros((nq, 10), dtype=torch.int64, device=dev); d = torch.zeros((nq, 10), dtype=torch.float32, device=dev); c = torch.zeros(nq, dtype=torch.int32, device=dev)
    fn = lambda: idx.search_device(dq, 10, "ip", rows, d, c)  # noqa: E731
    for _ in range(12): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(10):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    idx.profile_enable(True); idx.profile_get(reset=True); fn(); p = idx.profile_get(reset=True); idx.profile_enable(False)
    print("f16 shard nq", nq, "median_ms", round(float(np.median(ts)) * 1e3, 3), "scan_us", round(p["scan_us"], 1), "int8", bool(int(p["last_plan"]) & 4))
print("hbm GB after", round(idx.hbm_bytes() / 1e9, 2))
