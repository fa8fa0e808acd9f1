#!/usr/bin/env python3
"""Randomised parity sweep (GPU): FLAT exact / filtered / SQ8 / f16-dtype / binary searches with random shapes against the
oracle.  Usage: python scripts/stress_parity.py [seconds] [seed] -> prints the number of cases and any mismatch."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402

orc = O.get()
# first argument: seconds, or "c<N>" = exactly N cases (machine-independent case list for a given seed)
_a1 = sys.argv[1] if len(sys.argv) > 1 else "60"
max_cases = int(_a1[1:]) if _a1.startswith("c") else None
budget = float("inf") if max_cases is not None else float(_a1)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NAME = {O.IP: "ip", O.L2: "l2", O.COS: "cosine", O.HAMMING: "hamming", O.JACCARD: "jaccard", O.DICE: "dice"}
t0, cases, bad = time.time(), 0, []
while time.time() - t0 < budget and (max_cases is None or cases < max_cases):
    n = int(rng.choice([1, 7, 300, 4097, 20000, 70001, 200000, 600000, 1100000]))
    dim = int(rng.choice([1, 3, 8, 17, 64, 100, 128, 200, 384]))
    nq = int(rng.choice([1, 2, 31, 33, 70, 256, 300]))
    k = int(rng.choice([1, 5, 10, 64, 300]))
    mode = str(rng.choice(["exact", "filtered", "sq8", "f16", "binary"]))
    metric = int(rng.choice([O.IP, O.L2, O.COS])) if mode != "binary" else int(rng.choice([O.HAMMING, O.JACCARD, O.DICE]))
    if n * dim > 40_000_000:
        continue
    kind = rng.integers(0, 4)
    os.environ["LYNSE_HIP_QH"] = str(int(rng.integers(0, 3)))   # (read per call) the threshold stages of 64 / 128-column float batches: k_scan_h16 / k_scan_qh variants
    if mode == "binary":
        data = (rng.random((n, dim)) < 0.4).astype(np.float32)
    elif kind == 0:
        data = rng.standard_normal((n, dim)).astype(np.float32)
    elif kind == 1:
        data = rng.integers(0, 3, (n, dim)).astype(np.float32)  # heavy exact ties
    elif kind == 3:   # integer collections (the exactness rule of k_prep_queries when the queries are integers too): signed or not, up to ~300
        hi = int(rng.choice([2, 16, 100, 256, 300]))
        data = rng.integers(-hi if rng.random() < 0.5 else 0, hi + 1, (n, dim)).astype(np.float32)
    else:
        data = (rng.random((n, dim)) * rng.choice([1e-3, 1.0, 300.0])).astype(np.float32)
    queries = data[rng.integers(0, n, nq)] + (0.1 * rng.standard_normal((nq, dim)).astype(np.float32) if mode != "binary" else 0)
    if mode != "binary" and kind in (1, 3) and rng.random() < 0.6:   # integer queries: rows of the shard, half of them with integer noise
        queries = data[rng.integers(0, n, nq)] + rng.integers(-2, 3, (nq, dim)).astype(np.float32) * (rng.random((nq, 1)) < 0.5)
    queries = np.ascontiguousarray(queries, np.float32)
    try:
        if mode == "f16":
            idx = L.FlatIndex(None, dim, 0, dtype="f16")
            idx.write(data)
            dec = orc.round_f16(data)
            rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
            check = lambda qi: orc.canonical_topk_f16(queries[qi], dec, k, metric)  # noqa: E731
        else:
            idx = L.FlatIndex(None, dim, 0)
            idx.write(data)
            if mode == "exact":
                rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
                check = lambda qi: orc.canonical_topk(queries[qi], data, k, metric)  # noqa: E731
            elif mode == "binary":
                rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
                words = orc.pack_binary(data)
                check = lambda qi: orc.canonical_topk_packed(orc.pack_binary(queries[qi].reshape(1, -1))[0], words, k, metric)  # noqa: E731
            elif mode == "filtered":
                m = int(rng.integers(1, n + 1))
                subset = np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
                rows, dists, counts = idx.search_filtered_batch_arrays(queries, k, NAME[metric], subset)
                check = lambda qi: orc.canonical_topk_filtered(queries[qi], data, k, metric, subset)  # noqa: E731
            else:
                if (20 * k if metric != O.COS else 100 * k) > 4096 and n > 16384:
                    continue
                mins, scales, codes = orc.sq8_fit(data)
                rows, dists, counts = idx.search_sq8_batch_arrays(queries, k, NAME[metric])
                check = lambda qi: orc.sq8_search(queries[qi], data, mins, scales, codes, k, metric)  # noqa: E731
        for qi in sorted(set([0, nq - 1, int(rng.integers(0, nq))])):
            e_ids, e_d = check(qi)
            c = int(counts[qi])
            if c != len(e_ids) or not np.array_equal(rows[qi, :c].astype(np.uint32), e_ids) or not np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32)):
                bad.append((mode, NAME[metric], n, dim, nq, k, int(kind), qi))
                break
    except Exception as e:  # noqa: BLE001
        if "not supported" not in str(e):
            bad.append((mode, NAME[metric], n, dim, nq, k, int(kind), "EXC " + str(e)[:80]))
    cases += 1
print("cases", cases, "mismatches", len(bad))
for b in bad[:20]:
    print("  ", b)
