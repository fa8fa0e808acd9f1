#!/usr/bin/env python3
"""C3-shaped timing experiment (FLAT-L2 SIFT-like n x 128, batch 256, k): prints one JSON line; env knobs select the plan."""
import json, os, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lynsedb_amd as L
from lynsedb_amd.datasets import sift_like
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
metric = sys.argv[3] if len(sys.argv) > 3 else "l2"
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 128
dev = torch.device("cuda", 0)
data = sift_like(n, dim, 42); qs = sift_like(256, dim, 43)
idx = L.FlatIndex(None, dim, 0); idx.write(data)
dq = torch.as_tensor(qs, device=dev)
rows = torch.zeros((256, k), dtype=torch.int64, device=dev); d = torch.zeros((256, k), dtype=torch.float32, device=dev); c = torch.zeros(256, dtype=torch.int32, device=dev)
fn = lambda: idx.search_device(dq, k, metric, rows, d, c)
for _ in range(3): fn()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
idx.profile_enable(True); idx.profile_get(reset=True)
for _ in range(5): fn()
p = idx.profile_get(reset=True)
env = {k_: v for k_, v in os.environ.items() if k_.startswith("LYNSE_HIP")}
print(json.dumps({"n": n, "k": k, "metric": metric, "env": env, "median_ms": round(sorted(ts)[5] * 1e3, 3), "scan_us": round(p["scan_us"] / 5, 1),
                  "launches": p["scan_launches"] // 5, "pipeline_us": round(p["total_us"] / 5, 1), "pool": p["pool_entries"] // 5 // 256, "fallback": p["fallback_queries"]}))
