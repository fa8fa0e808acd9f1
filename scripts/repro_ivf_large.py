#!/usr/bin/env python3
"""Repeats one stress_ivf_large.py configuration over fresh seeds and prints mismatches in detail:
repro_ivf_large.py n dim nlist metric kind nq nprobe k seeds"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402
orc = O.get()
n, dim, nlist = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mname, kind = sys.argv[4], sys.argv[5]
nq, nprobe, k, seeds = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])
metric = {"ip": O.IP, "l2": O.L2, "cosine": O.COS}[mname]
bad = 0
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    if kind == "uniform":
        data = rng.random((n, dim), dtype=np.float32)
    elif kind == "gaussian":
        data = rng.standard_normal((n, dim)).astype(np.float32)
    else:
        c = rng.standard_normal((max(nlist // 2, 2), dim)).astype(np.float32)
        data = (c[rng.integers(0, c.shape[0], n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    built = L.IvfFlatIndex.build(None, data, dim, nlist, 2, mname, l2_partitions=False)
    cen, asg, _, _ = built.export()
    del built
    idx = L.IvfFlatIndex.load(data, cen, asg, mname)
    off, rows = orc.lists_from_assignments(asg, cen.shape[0])
    for rep in range(8):
        queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
        g = idx.search_batch_arrays(queries, k, nprobe)
        g2 = idx.search_batch_arrays(queries, k, nprobe)
        same = np.array_equal(g[0], g2[0]) and np.array_equal(g[1].view(np.uint32), g2[1].view(np.uint32))
        for qi in range(nq):
            e_ids, e_d, probes = orc.ivf_search(queries[qi], data, cen, off, rows, nprobe, k, metric)
            c = int(g[2][qi])
            if c != len(e_ids) or not np.array_equal(g[0][qi, :c].astype(np.uint64), np.asarray(e_ids, np.uint64)) or not np.array_equal(g[1][qi, :c].view(np.uint32), e_d.view(np.uint32)):
                bad += 1
                if bad <= 4:
                    diff = np.nonzero(g[0][qi, :c].astype(np.uint64) != np.asarray(e_ids, np.uint64)[:c])[0] if c == len(e_ids) else []
                    print("seed", seed, "rep", rep, "q", qi, "count", c, len(e_ids), "repeatable", same, "first diffs at", diff[:6],
                          "got", g[0][qi, :c][diff[:3]] if len(diff) else "", g[1][qi, :c][diff[:3]] if len(diff) else "",
                          "exp", np.asarray(e_ids)[diff[:3]] if len(diff) else "", e_d[diff[:3]] if len(diff) else "", "probes", probes)
    del idx
print("mismatching queries:", bad)
