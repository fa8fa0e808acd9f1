#!/usr/bin/env python3
"""Randomised IVF parity sweep at sizes that reach the int8 staged path and the device-side grouping (slabs of 70K..320K rows,
whole 128-column slabs or not, 1..256 queries, IP / L2 / cosine): the device k-means makes SOME partition, both sides search it.
Usage: stress_ivf_large.py [seconds] [seed]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402
orc = O.get()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NAME = {O.IP: "ip", O.L2: "l2", O.COS: "cosine"}
t0, cases, bad, on_int8 = time.time(), 0, [], 0
while time.time() - t0 < budget:
    n = int(rng.choice([70_000, 150_000, 270_000, 320_000]))
    dim = int(rng.choice([128, 200, 256, 384]))
    nlist = int(rng.choice([16, 100, 256, 1000]))
    metric = int(rng.choice([O.IP, O.IP, O.L2, O.COS]))
    kind = str(rng.choice(["uniform", "gaussian", "clustered"]))
    if kind == "uniform":
        data = rng.random((n, dim), dtype=np.float32)
    elif kind == "gaussian":
        data = rng.standard_normal((n, dim)).astype(np.float32)
    else:
        c = rng.standard_normal((max(nlist // 2, 2), dim)).astype(np.float32)
        data = (c[rng.integers(0, c.shape[0], n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    built = L.IvfFlatIndex.build(None, data, dim, nlist, 2, NAME[metric], l2_partitions=False)
    cen, asg, _, _ = built.export()
    del built
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
    off, rows = orc.lists_from_assignments(asg, cen.shape[0])
    idx.profile_enable(True)
    for nq in (int(rng.integers(1, 5)), int(rng.integers(5, 33)), int(rng.integers(33, 257))):
        nprobe = int(rng.choice([1, 4, 16, 40]))
        k = int(rng.choice([1, 10, 40]))
        queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
        g = idx.search_batch_arrays(queries, k, nprobe)
        on_int8 += 1 if int(idx.profile_get()["last_plan"]) & 64 else 0
        ok = True
        for qi in sorted({0, nq - 1}):
            e_ids, e_d, _ = orc.ivf_search(queries[qi], data, cen, off, rows, nprobe, k, metric)
            c = int(g[2][qi])
            ok = ok and c == len(e_ids) and np.array_equal(g[0][qi, :c].astype(np.uint64), np.asarray(e_ids, np.uint64)) and np.array_equal(g[1][qi, :c].view(np.uint32), e_d.view(np.uint32))
        cases += 1
        if not ok:
            bad.append((n, dim, nlist, NAME[metric], kind, nq, nprobe, k))
    del idx
print("cases", cases, "started on int8", on_int8, "mismatches", len(bad))
for b in bad[:20]:
    print("  ", b)
