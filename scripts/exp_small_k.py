#!/usr/bin/env python3
"""Fused single-launch search: kernel time vs k and shard size (separates the scan from the k-proportional merge)."""
import json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L
dev = torch.device("cuda", 0)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for n in (1000, 25000, 100000):
    rng = np.random.default_rng(1)
    data = rng.random((n, dim), dtype=np.float32)
    idx = L.FlatIndex(None, dim, 0); idx.write(data); idx.finalize()
    q = torch.as_tensor(data[:1] + 0.01, device=dev)
    for k in (1, 10, 32, 64):
        rows = torch.zeros((1, k), dtype=torch.int64, device=dev); d = torch.zeros((1, k), device=dev); c = torch.zeros(1, dtype=torch.int32, device=dev)
        for _ in range(20): idx.search_device(q, k, "ip", rows, d, c)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); idx.search_device(q, k, "ip", rows, d, c); ts.append(time.perf_counter() - t0)
        idx.profile_enable(True); idx.profile_get(reset=True)
        for _ in range(50): idx.search_device(q, k, "ip", rows, d, c)
        p = idx.profile_get(reset=True); idx.profile_enable(False)
        print(json.dumps({"n": n, "dim": dim, "k": k, "wall_us": round(sorted(ts)[100] * 1e6, 1), "kernel_us": round(p["scan_us"] / 50, 1)}), flush=True)
