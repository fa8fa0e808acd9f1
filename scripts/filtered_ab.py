#!/usr/bin/env python3
"""A/B helper: masked-scan filtered search, 10M x 768, 256 queries, 50 % and 10 % subsets; prints median ms and the
scan profile (run with different LYNSE_HIP_* knobs to compare stage plans)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402

dev = torch.device("cuda", 0)
import os
n, dim, nq = 10_000_000, 768, int(os.environ.get("FILTERED_AB_NQ", "256"))
idx = L.FlatIndex(None, dim, 0)
idx.reserve(n)
g = torch.Generator(device=dev); g.manual_seed(n)
for b in range(0, n, 500_000):
    idx.write_device(torch.rand((500_000, dim), generator=g, device=dev))
idx.finalize()
rng = np.random.default_rng(42)
qs = np.ascontiguousarray(rng.random((nq, dim), dtype=np.float32))
for frac in (0.5, 0.1):
    subset = np.sort(rng.choice(n, int(n * frac), replace=False)).astype(np.uint64)
    words = np.zeros((n + 63) // 64, np.uint64)
    np.bitwise_or.at(words, (subset // 64).astype(np.int64), np.uint64(1) << (subset % np.uint64(64)))
    fn = lambda: idx.search_filtered_bitset_batch_arrays(qs, 10, "ip", words)  # noqa: E731
    for _ in range(3): fn()
    idx.profile_enable(True); idx.profile_get(reset=True)
    ts = []
    for _ in range(8):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    p = idx.profile_get(reset=True); idx.profile_enable(False)
    s = max(p["searches"], 1)
    print("sel", frac, "median_ms", round(float(np.median(ts)) * 1e3, 3), "scan_us", round(p["scan_us"] / s, 1), "launches", p["scan_launches"] / s,
          "rows", p["scan_rows"] / s, "pipeline_us", round(p["total_us"] / s, 1), "fallback", p["fallback_queries"], "pool/q", round(p["pool_entries"] / s / nq, 1))
