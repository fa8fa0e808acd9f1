#!/usr/bin/env python3
"""Randomised parity sweep of the certified int8 pass across batch sizes (the <= 32-query, 64-query, 128-query and 256-query
tilings) on shards large enough to select it: random n in [256K, 500K], dim, metric, data distribution, k; three queries of every
batch against the oracle, the whole batch against a second run (determinism).  Usage: stress_i8c_batches.py [seconds] [seed]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402
orc = O.get()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
M = {"ip": O.IP, "l2": O.L2, "cosine": O.COS}
t0, cases, bad, on_int8 = time.time(), 0, [], 0
while time.time() - t0 < budget:
    n = int(rng.integers(262_144, 500_000))
    dim = int(rng.choice([64, 128, 200, 256, 384, 512, 768]))
    metric = str(rng.choice(["ip", "ip", "l2", "cosine"]))
    kind = str(rng.choice(["uniform", "gaussian", "clustered", "lognormal"]))
    if kind == "uniform":
        data = rng.random((n, dim), dtype=np.float32)
    elif kind == "gaussian":
        data = rng.standard_normal((n, dim)).astype(np.float32)
    elif kind == "clustered":
        c = rng.standard_normal((64, dim)).astype(np.float32)
        data = (c[rng.integers(0, 64, n)] + 0.2 * rng.standard_normal((n, dim))).astype(np.float32)
    else:
        data = (rng.standard_normal((n, dim)) * np.exp(0.7 * rng.standard_normal((n, 1)))).astype(np.float32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    idx.profile_enable(True)
    for nq in (int(rng.integers(1, 33)), int(rng.integers(33, 65)), int(rng.integers(65, 129)), int(rng.integers(129, 257))):
        k = int(rng.choice([1, 10, 50]))
        queries = (data[rng.integers(0, n, nq)] * (1.0 + 0.02 * rng.standard_normal((nq, 1))) + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
        if rng.random() < 0.3:
            queries[0] = -queries[0]
        idx.profile_get(reset=True)
        r, d, c = idx.search_batch_arrays(queries, k, metric)
        p = idx.profile_get(reset=True)
        on_int8 += 1 if int(p["last_plan"]) & 64 else 0
        r2, d2, c2 = idx.search_batch_arrays(queries, k, metric)
        ok = np.array_equal(r, r2) and np.array_equal(d.view(np.uint32), d2.view(np.uint32))
        for qi in sorted({0, nq // 2, nq - 1}):
            e_ids, e_d = orc.canonical_topk(queries[qi], data, k, M[metric])
            ok = ok and int(c[qi]) == len(e_ids) and np.array_equal(r[qi, :len(e_ids)].astype(np.uint64), e_ids.astype(np.uint64)) \
                and np.array_equal(d[qi, :len(e_ids)].view(np.uint32), e_d.view(np.uint32))
        cases += 1
        if not ok:
            bad.append((n, dim, metric, kind, nq, k, int(p["last_plan"]) & 0xff, int(p["fallback_queries"])))
    del idx
print("cases", cases, "started on int8", on_int8, "mismatches", len(bad))
for b in bad[:20]:
    print("  ", b)
