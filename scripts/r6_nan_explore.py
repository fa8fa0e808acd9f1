#!/usr/bin/env python3
"""What the FLAT search returns for NaN / +-inf rows and queries (round 6; tests/test_gpu_flat_parity.py pins it afterwards)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import lynsedb_amd as L
import oracle as O

orc = O.get()
rng = np.random.default_rng(3)


def model(q, data, k, metric):
    d = np.asarray(orc.all_distances(q, data, metric), np.float32)
    asc = metric != O.IP
    key = np.where(np.isnan(d), np.inf, d if asc else -d)
    order = np.lexsort((np.arange(len(d)), key))          # (score best-first, NaN as the worst value, row ascending)
    return order[:k].astype(np.uint64), d[order[:k]]


for n, dim, nq, k in ((64, 96, 1, 64), (64, 96, 8, 64), (5000, 96, 3, 20), (5000, 96, 40, 20), (100000, 128, 2, 10), (300000, 128, 64, 10), (300000, 256, 64, 10)):
    data = rng.standard_normal((n, dim)).astype(np.float32)
    sp = rng.choice(n, 12, replace=False)
    data[sp[0:4], 3] = np.nan
    data[sp[4:6], 5] = np.inf
    data[sp[6:8], 5] = -np.inf
    data[sp[8], 1] = np.inf; data[sp[8], 2] = -np.inf
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    queries[:, 1] = np.abs(queries[:, 1]); queries[:, 2] = np.abs(queries[:, 2])
    if nq > 1:
        queries[-1, 7] = np.nan
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    for name, m in (("ip", O.IP), ("l2", O.L2), ("cosine", O.COS)):
        try:
            rows, dists, counts = idx.search_batch_arrays(queries, k, name)
        except Exception as e:  # noqa: BLE001
            print(n, dim, nq, k, name, "ERROR", repr(e))
            continue
        bad = 0
        for qi in range(nq):
            e_r, e_d = model(queries[qi], data, k, m)
            c = int(counts[qi])
            same_r = c == len(e_r) and np.array_equal(rows[qi, :c].astype(np.uint64), e_r)
            same_d = c == len(e_r) and np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32))
            if not (same_r and same_d):
                bad += 1
                if bad <= 2:
                    print("  q", qi, "count", c, "\n   got ", rows[qi, :min(c, 14)].astype(np.int64), dists[qi, :min(c, 14)], "\n   want", e_r[:14].astype(np.int64), e_d[:14])
        print(n, dim, nq, k, name, "mismatching queries:", bad, "of", nq, "coarse", idx.coarse_state())
