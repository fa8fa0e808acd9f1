#!/usr/bin/env python3
"""IVF-IP timing experiment (a slice of config 4): n x 768 clustered unit rows generated on the device, nlist lists,
nprobe 32, k 10; device-resident build and search.  Prints one JSON line per batch size.  Under
`rocprofv3 --kernel-trace --stats` the kernel table shows where a search spends its device time.

usage: exp_ivf.py [n] [nlist] [iters] [calls]"""
import json, os, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lynsedb_amd as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dim = 768
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
centers = torch.randn((K, dim), generator=g, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
rows = torch.empty((n, dim), device=dev)
for b in range(0, n, 250_000):
    e = min(n, b + 250_000)
    blk = centers[torch.arange(b, e, device=dev) % K] + 0.03 * torch.randn((e - b, dim), generator=g, device=dev)
    rows[b:e] = blk / blk.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
ivf = L.IvfFlatIndex.build_device(rows, dim, K, iters, "ip", l2_partitions=False)
torch.cuda.synchronize()
build_s = time.perf_counter() - t0
print(json.dumps({"n": n, "nlist": K, "kmeans_iters": iters, "build_s": round(build_s, 3)}), flush=True)
pick = torch.randint(0, n, (256,), generator=g, device=dev)
qs_all = (rows[pick] + 0.01 * torch.randn((256, dim), generator=g, device=dev)).contiguous()
del rows
for nq in (1, 8, 64, 256):
    dq = qs_all[:nq].contiguous()
    r = torch.zeros((nq, 10), dtype=torch.int64, device=dev); d = torch.zeros((nq, 10), dtype=torch.float32, device=dev)
    c = torch.zeros(nq, dtype=torch.int32, device=dev)
    fn = lambda: ivf.search_device(dq, 10, 32, r, d, c)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(json.dumps({"nq": nq, "median_ms": round(ts[len(ts) // 2] * 1e3, 3), "best_ms": round(ts[0] * 1e3, 3),
                      "qps": round(nq / ts[len(ts) // 2], 1), "top1_is_source": int((r[:, 0] == pick[:nq]).sum().item())}), flush=True)
