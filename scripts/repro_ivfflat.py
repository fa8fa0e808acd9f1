#!/usr/bin/env python3
"""Reproduces one stress_ivf.py configuration many times with fresh seeds and prints the first mismatches in detail."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402
orc = O.get()
n, dim, nlist, nprobe, nq, k = [int(x) for x in sys.argv[1:7]]
metric = {"ip": O.IP, "l2": O.L2, "cosine": O.COS}[sys.argv[7]]
NAME = {O.IP: "ip", O.L2: "l2", O.COS: "cosine"}
bad = 0
for seed in range(int(sys.argv[8])):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((max(nlist // 2, 2), dim)).astype(np.float32)
    data = (centers[rng.integers(0, centers.shape[0], n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    cen, asg = orc.kmeans_train(data, nlist, 8, O.L2)
    off, orig = orc.ivf_flat_layout(asg, cen.shape[0])
    slab = data[orig.astype(np.int64)]
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric], ivfflat_routing=True)
    rd = orc.ivf_routing_dims(cen)
    g = idx.search_batch_arrays(queries, k, nprobe)
    for qi in range(nq):
        e_ids, e_d = orc.ivf_flat_search(queries[qi], slab, cen, off, orig, nprobe, k, metric, routing_dims=rd)
        c = int(g[2][qi])
        if c != len(e_ids) or not np.array_equal(g[0][qi, :c].astype(np.uint64), np.asarray(e_ids, np.uint64)) or not np.array_equal(g[1][qi, :c].view(np.uint32), e_d.view(np.uint32)):
            bad += 1
            if bad <= 5:
                print("seed", seed, "q", qi, "got", g[0][qi, :c], g[1][qi, :c], "count", c, "expected", e_ids, e_d)
print("mismatching (seed, query) pairs:", bad)
