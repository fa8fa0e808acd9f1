#!/usr/bin/env python3
"""Mid-size batches (40 / 64 / 100 / 128 queries) over 10M x 768: median ms per call; run with LYNSE_HIP_MID_TILINGS=0 / 1."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
dev = torch.device("cuda", 0)
n, dim = 10_000_000, 768
idx = L.FlatIndex(None, dim, 0); idx.reserve(n)
g = torch.Generator(device=dev); g.manual_seed(n)
for b in range(0, n, 500_000):
    idx.write_device(torch.rand((500_000, dim), generator=g, device=dev))
idx.finalize()
q_all = torch.rand((256, dim), generator=g, device=dev)
for metric in ("ip", "l2"):
    for nq in (40, 64, 100, 128, 256):
        dq = q_all[:nq].contiguous()
        rows = torch.zeros((nq, 10), dtype=torch.int64, device=dev); d = torch.zeros((nq, 10), dtype=torch.float32, device=dev); c = torch.zeros(nq, dtype=torch.int32, device=dev)
        fn = lambda: idx.search_device(dq, 10, metric, rows, d, c)  # noqa: E731
        for _ in range(3): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(10):
            t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        idx.profile_enable(True); idx.profile_get(reset=True); fn(); p = idx.profile_get(reset=True); idx.profile_enable(False)
        print(metric, "nq", nq, "median_ms", round(float(np.median(ts)) * 1e3, 3), "scan_us", round(p["scan_us"], 1), "q/s", round(nq / float(np.median(ts))), "pool/q", round(p["pool_entries"] / nq, 1), "fallback", p["fallback_queries"])
