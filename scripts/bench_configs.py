#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE configs on ONE GPU (per-GPU slices of the 8-GPU configs).

    python scripts/bench_configs.py [c1] [c3] [c4] [c5] [small]   -> one JSON line per config

Every number is the HIP path through the C ABI with inputs resident in HBM, timed with
torch.cuda.synchronize() brackets; kernel time of the dominant scan kernel comes from the library's
HIP-event profile.  Parity of each config is checked against the oracle on a few queries.
"""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
orc = O.get()
HBM = 8000.0


def timeit(fn, warm=3, reps=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def prof(idx, fn, reps=10):
    idx.profile_enable(True)
    idx.profile_get(reset=True)
    for _ in range(reps):
        fn()
    p = idx.profile_get(reset=True)
    idx.profile_enable(False)
    gbps = p["scan_bytes"] / (p["scan_us"] * 1e-6) / 1e9 if p["scan_us"] else 0.0
    return {"scan_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / HBM, 4), "scan_us_per_call": round(p["scan_us"] / reps, 1),
            "pipeline_us_per_call": round(p["total_us"] / max(p["searches"], 1), 1), "fallback_queries": int(p["fallback_queries"])}


def c1():
    """flat_search_bench.py protocol: FLAT-IP 100k x 128, single query, k=10 (config 1) on the GPU."""
    rng = np.random.default_rng(42)
    q = rng.random(128, dtype=np.float32)
    data = rng.random((100_000, 128), dtype=np.float32)
    data[0] = q
    idx = L.FlatIndex(None, 128, 0)
    idx.write(data)
    dq = torch.as_tensor(q.reshape(1, -1), device=dev)
    rows = torch.zeros((1, 10), dtype=torch.int64, device=dev)
    d = torch.zeros((1, 10), dtype=torch.float32, device=dev)
    c = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lambda: idx.search_device(dq, 10, "ip", rows, d, c)  # noqa: E731
    med, best = timeit(fn, 20, 30)
    e_ids, e_d = orc.canonical_topk(q, data, 10, O.IP)
    ok = np.array_equal(rows.cpu().numpy()[0].astype(np.uint32), e_ids) and np.array_equal(d.cpu().numpy()[0], e_d)
    host_med, _ = timeit(lambda: idx.search(q, 10, "ip"), 20, 30)
    return {"config": "C1 FLAT-IP 100000x128 f32, single query, k=10", "median_ms": round(med * 1e3, 4),
            "best_ms": round(best * 1e3, 4), "host_api_median_ms": round(host_med * 1e3, 4), "parity": bool(ok),
            "top1": int(rows[0, 0]), **prof(idx, fn)}


def c3():
    """FLAT-L2 on a SIFT-like 1M x 128 set (ints 0..218 as f32), k=100, batch of 256 queries."""
    from lynsedb_amd.datasets import sift_like

    data = sift_like(1_000_000, 128, 42)
    qs = sift_like(256, 128, 43)
    idx = L.FlatIndex(None, 128, 0)
    idx.write(data)
    dq = torch.as_tensor(qs, device=dev)
    rows = torch.zeros((256, 100), dtype=torch.int64, device=dev)
    d = torch.zeros((256, 100), dtype=torch.float32, device=dev)
    c = torch.zeros(256, dtype=torch.int32, device=dev)
    fn = lambda: idx.search_device(dq, 100, "l2", rows, d, c)  # noqa: E731
    med, best = timeit(fn, 3, 10)
    hit = 0
    r = rows.cpu().numpy()
    for i in range(4):
        e_ids, e_d = orc.canonical_topk(qs[i], data, 100, O.L2)
        hit += int(np.array_equal(r[i].astype(np.uint32), e_ids))
    return {"config": "C3 FLAT-L2 SIFT-like 1000000x128, batch 256, k=100", "median_ms": round(med * 1e3, 3),
            "qps": round(256 / med, 1), "exact_id_parity_4_queries": hit == 4, **prof(idx, fn, 5)}


def c5():
    """Packed Hamming, one GPU's slice of config 5: 12.5M x 1024-bit fingerprints, k=50, single query and batch 32."""
    from lynsedb_amd.datasets import packed_bernoulli

    n = 12_500_000
    words = packed_bernoulli(n, 1024, 0.5, 42)
    idx = L.FlatIndex(None, 1024, 0)
    idx.write_packed(words)
    out = {}
    for nq in (1, 2, 4, 8, 32, 256):
        qw = words[np.arange(nq) * 1000 + 7].copy()
        qw[:, 0] ^= np.uint64(0xFFFF)
        dq = torch.as_tensor(qw.view(np.int64), device=dev)
        rows = torch.zeros((nq, 50), dtype=torch.int64, device=dev)
        d = torch.zeros((nq, 50), dtype=torch.float32, device=dev)
        c = torch.zeros(nq, dtype=torch.int32, device=dev)
        fn = lambda: idx.search_packed_device(dq, 50, "hamming", rows, d, c)  # noqa: E731
        med, best = timeit(fn, 3, 10)
        e_ids, e_d = orc.canonical_topk_packed(qw[0], words, 50, O.HAMMING)
        ok = np.array_equal(rows.cpu().numpy()[0].astype(np.uint32), e_ids) and np.array_equal(d.cpu().numpy()[0], e_d)
        out[f"nq{nq}"] = {"median_ms": round(med * 1e3, 3), "qps": round(nq / med, 1), "parity": bool(ok), **prof(idx, fn, 5)}
    return {"config": "C5 slice: Hamming 12500000x1024-bit, k=50", **out}


def c4(n=2_000_000, K=1024):
    """IVF-Flat IP, a reduced slice of config 4: 2M x 768 clustered unit vectors, nlist=1024, nprobe=32, k=10
    (c4full: one GPU's 6.25M-row share of the 50M x 768, nlist=4096 configuration)."""
    dim = 768
    rng = np.random.default_rng(7)
    centers = rng.standard_normal((K, dim)).astype(np.float32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    data = np.empty((n, dim), np.float32)
    for b in range(0, n, 200_000):
        idxs = np.arange(b, min(n, b + 200_000))
        blk = centers[idxs % K] + 0.03 * rng.standard_normal((idxs.size, dim)).astype(np.float32)
        data[b:b + idxs.size] = blk / np.linalg.norm(blk, axis=1, keepdims=True)
    t0 = time.time()
    ivf = L.IvfFlatIndex.build(None, data, dim, K, 5, "ip", l2_partitions=False)
    build_s = time.time() - t0
    qs = data[rng.integers(0, n, 64)] + 0.01 * rng.standard_normal((64, dim)).astype(np.float32)
    qs = np.ascontiguousarray(qs, dtype=np.float32)
    med, best = timeit(lambda: ivf.search_batch_arrays(qs, 10, 32), 2, 5)
    rows, d, c = ivf.search_batch_arrays(qs, 10, 32)
    flat = L.FlatIndex(None, dim, 0)
    flat.write(data)
    frows, fd, fc = flat.search_batch_arrays(qs, 10, "ip")
    rec = np.mean([len(set(rows[i].tolist()) & set(frows[i].tolist())) / 10 for i in range(64)])
    out = {"config": "C4 slice: IVF-IP %dx768 nlist=%d nprobe=32 k=10, batch 64 (host API)" % (n, K), "build_s": round(build_s, 1),
           "median_ms": round(med * 1e3, 2), "qps": round(64 / med, 1), "recall_at_10_vs_flat": round(float(rec), 4)}
    for nq in (1, 256):
        qq = np.ascontiguousarray(np.tile(qs, (4, 1))[:nq])
        m2, _ = timeit(lambda: ivf.search_batch_arrays(qq, 10, 32), 2, 5)
        f2, _ = timeit(lambda: flat.search_batch_arrays(qq, 10, "ip"), 2, 5)
        out["nq%d" % nq] = {"ivf_ms": round(m2 * 1e3, 3), "ivf_qps": round(nq / m2, 1), "flat_ms": round(f2 * 1e3, 3)}
    return out


def c4full():
    return c4(6_250_000, 4096)


def small():
    """FLAT-IP 10M x 768 with small batches (the <=32-query kernel): single query and 16 queries."""
    n, dim = 4_000_000, 768
    idx = L.FlatIndex(None, dim, 0)
    idx.reserve(n)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for b in range(0, n, 500_000):
        idx.write_device(torch.rand((500_000, dim), generator=g, device=dev))
    idx.finalize()
    out = {}
    for nq in (1, 16, 32):
        dq = torch.rand((nq, dim), generator=g, device=dev)
        rows = torch.zeros((nq, 10), dtype=torch.int64, device=dev)
        d = torch.zeros((nq, 10), dtype=torch.float32, device=dev)
        c = torch.zeros(nq, dtype=torch.int32, device=dev)
        fn = lambda: idx.search_device(dq, 10, "ip", rows, d, c)  # noqa: E731
        med, best = timeit(fn, 3, 10)
        out[f"nq{nq}"] = {"median_ms": round(med * 1e3, 3), "qps": round(nq / med, 1), **prof(idx, fn, 5)}
    return {"config": "FLAT-IP 4000000x768 small batches (k=10)", **out}


def sq8():
    """FLAT-IP-SQ8 / FLAT-L2-SQ8 (two-pass mode) on 10M x 768 uniform rows, 256 queries, k=10: time and recall@10 against the
    exact search of the same index (the mode is approximate by design)."""
    n, dim, nq = 10_000_000, 768, 256
    idx = L.FlatIndex(None, dim, 0)
    idx.reserve(n)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    for b in range(0, n, 500_000):
        idx.write_device(torch.rand((500_000, dim), generator=g, device=dev))
    idx.finalize()
    qs = np.ascontiguousarray(idx.read_rows(12345, nq) + 0.03 * np.random.default_rng(1).standard_normal((nq, dim)).astype(np.float32))
    out = {}
    t0 = time.perf_counter()
    idx.sq8_params()
    torch.cuda.synchronize()
    out["build_s"] = round(time.perf_counter() - t0, 3)
    for name in ("ip", "l2"):
        med, best = timeit(lambda: idx.search_sq8_batch_arrays(qs, 10, name), 2, 6)
        ex, _ = timeit(lambda: idx.search_batch_arrays(qs, 10, name), 2, 6)
        r1 = idx.search_sq8_batch_arrays(qs, 10, name)[0]
        r2 = idx.search_batch_arrays(qs, 10, name)[0]
        rec = np.mean([len(set(r1[i].tolist()) & set(r2[i].tolist())) / 10 for i in range(nq)])
        out[name] = {"sq8_ms": round(med * 1e3, 3), "sq8_qps": round(nq / med, 1), "exact_ms": round(ex * 1e3, 3), "recall_at_10_vs_exact": round(float(rec), 4)}
    return {"config": "FLAT-*-SQ8 two-pass, 10000000x768, 256 queries, k=10 (host API)", **out}


def filtered():
    """Filtered FLAT-IP (docs/comparisons/vector_database_benchmarks.md:64-66, :99-101 quote 0.178 ms at 100k and 2.16 ms at 1M
    on the reference's CPU path): single query, k=10, subsets of 10 % and 50 % of the rows, through the host API
    (subset ids travel host -> device on every call); plus 10M x 768 with 256 queries."""
    out = {}
    rng = np.random.default_rng(42)
    for n, dim, nq in ((100_000, 128, 1), (1_000_000, 128, 1), (10_000_000, 768, 256)):
        idx = L.FlatIndex(None, dim, 0)
        idx.reserve(n)
        g = torch.Generator(device=dev)
        g.manual_seed(n)
        for b in range(0, n, 500_000):
            idx.write_device(torch.rand((min(500_000, n - b), dim), generator=g, device=dev))
        idx.finalize()
        qs = np.ascontiguousarray(rng.random((nq, dim), dtype=np.float32))
        for frac in (0.001, 0.1, 0.5):
            m = int(n * frac)
            subset = np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
            words = np.zeros((n + 63) // 64, np.uint64)
            np.bitwise_or.at(words, (subset // 64).astype(np.int64), np.uint64(1) << (subset % np.uint64(64)))
            res = {}
            for api, fn in (("ids", lambda: idx.search_filtered_batch_arrays(qs, 10, "ip", subset)),
                            ("bitset", lambda: idx.search_filtered_bitset_batch_arrays(qs, 10, "ip", words))):
                med, best = timeit(fn, 3, 10)
                rows, d, c = fn()
                ok = bool(np.isin(rows[0, :int(c[0])], subset).all() and int(c[0]) == 10)
                res[api] = {"median_ms": round(med * 1e3, 3), "qps": round(nq / med, 1), "subset_ok": ok}
            out[f"{n}x{dim}_nq{nq}_sel{frac}"] = res
        del idx
    return {"config": "filtered FLAT-IP, k=10 (host API, subset ids uploaded per call)", **out}


if __name__ == "__main__":
    todo = sys.argv[1:] or ["c1", "c3", "c5", "small", "c4"]
    for name in todo:
        print(json.dumps(globals()[name]()), flush=True)
