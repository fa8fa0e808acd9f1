"""Concurrent readers on ONE index (`VectorIndex: Send + Sync`, `Arc<RwLock<Collection>>` with `inner.read()` on the search
path: src/index/mod.rs:78, src/python/mod.rs:950, :1187).  Unfiltered searches take the handle's lock shared and run on
separate search contexts (workspace + stream); append / finalize stay exclusive.  Results must not depend on concurrency."""
import os
import threading
import time

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
f32 = np.float32


def test_eight_threads_search_one_index_concurrently(oracle):
    os.environ["LYNSE_HIP_CONTEXTS"] = "8"     # read when a handle is created
    import lynsedb_amd as L

    n, dim, k, per_thread, n_threads = 100_000, 128, 10, 400, 8
    rng = np.random.default_rng(3)
    data = rng.random((n, dim), dtype=f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    queries = (data[rng.integers(0, n, n_threads * per_thread)] + 0.01 * rng.standard_normal((n_threads * per_thread, dim))).astype(f32)
    # reference answers, one thread
    t0 = time.perf_counter()
    ref = [idx.search(queries[i], k, "ip") for i in range(per_thread)]
    t_single = (time.perf_counter() - t0) / per_thread
    for i in (0, 7, per_thread - 1):
        e_ids, e_d = oracle.canonical_topk(queries[i], data, k, O.IP)
        assert np.array_equal(ref[i][0], e_ids) and np.array_equal(ref[i][1].view(np.uint32), e_d.view(np.uint32))
    out = [None] * n_threads
    errors = []

    def worker(t):
        try:
            res = []
            for i in range(per_thread):
                q = queries[t * per_thread + i]
                metric = ("ip", "l2", "cosine")[i % 3] if t % 2 else "ip"   # mixed metrics on half of the threads
                res.append((metric, idx.search(q, k, metric)))
            out[t] = res
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    t_conc = (time.perf_counter() - t0) / (n_threads * per_thread)
    assert not errors, errors
    # thread 0 ran the same queries as the single-threaded pass: identical answers
    for i in range(per_thread):
        assert np.array_equal(out[0][i][1][0], ref[i][0]) and np.array_equal(out[0][i][1][1].view(np.uint32), ref[i][1].view(np.uint32))
    # spot checks of the other threads against the oracle
    for t in (1, 5, 7):
        for i in (0, 1, 2, per_thread - 1):
            metric, (ids, d) = out[t][i]
            m = {"ip": O.IP, "l2": O.L2, "cosine": O.COS}[metric]
            e_ids, e_d = oracle.canonical_topk(queries[t * per_thread + i], data, k, m)
            assert np.array_equal(ids, e_ids) and np.array_equal(d.view(np.uint32), e_d.view(np.uint32)), (t, i, metric)
    speedup = t_single / t_conc
    print(f"single thread {t_single * 1e6:.1f} us/query, 8 threads {t_conc * 1e6:.1f} us/query: x{speedup:.2f}")
    assert speedup > 0.8, speedup   # concurrency must not cost throughput; the gain is bounded by host-side launch + GIL time per call (reported above)
    # writers stay exclusive and searches see the appended rows afterwards
    extra = (queries[:5] * 1.0).astype(f32)
    idx.write(extra)
    idx.finalize()
    ids, d = idx.search(queries[0], 1, "l2")
    assert ids[0] == n and d[0] == 0.0


def test_filtered_searches_run_concurrently_with_each_other_and_with_unfiltered_ones(oracle, monkeypatch):
    """Subset-filtered searches on the masked-scan strategy keep their bitmask / id staging in their own search context and take
    the shared lock (the reference's filtered search is a reader like any other, `inner.read()`, src/python/mod.rs:1187); the
    gathered-rows strategy stays exclusive.  Every thread has its OWN subset; answers must equal the single-threaded ones and
    the oracle's."""
    os.environ["LYNSE_HIP_CONTEXTS"] = "8"
    import lynsedb_amd as L

    n, dim, k, per_thread, n_threads = 120_000, 64, 10, 60, 6
    rng = np.random.default_rng(31)
    data = rng.standard_normal((n, dim)).astype(f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    queries = (data[rng.integers(0, n, n_threads * per_thread)] + 0.05 * rng.standard_normal((n_threads * per_thread, dim))).astype(f32)
    subsets = [np.sort(rng.choice(n, n // (2 + t), replace=False)).astype(np.uint64) for t in range(n_threads)]
    out = [None] * n_threads
    errors = []

    def worker(t):
        try:
            res = []
            for i in range(per_thread):
                q = queries[t * per_thread + i]
                if t == n_threads - 1:
                    res.append(idx.search(q, k, "l2"))                      # an unfiltered reader among them
                elif i % 7 == 3:
                    res.append(idx.search_filtered(q, k, "l2", subsets[t][:200]))   # few ids: the gathered-rows strategy (exclusive)
                else:
                    res.append(idx.search_filtered(q, k, "l2", subsets[t]))         # masked scan (shared)
            out[t] = res
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(n_threads):
        for i in (0, 3, 10, per_thread - 1):
            q = queries[t * per_thread + i]
            ids, d = out[t][i]
            if t == n_threads - 1:
                e_ids, e_d = oracle.canonical_topk(q, data, k, O.L2)
            else:
                sub = subsets[t][:200] if i % 7 == 3 else subsets[t]
                e_ids, e_d = oracle.canonical_topk_filtered(q, data, k, O.L2, sub)
            assert np.array_equal(ids, e_ids) and np.array_equal(d.view(np.uint32), e_d.view(np.uint32)), (t, i)
    # a batch on the masked int8 scan from two threads at once (>= 64K rows, 33..256 queries)
    big = (data[rng.integers(0, n, 2 * 64)] + 0.05 * rng.standard_normal((2 * 64, dim))).astype(f32)
    monkeypatch.setenv("LYNSE_HIP_FILTER_STRATEGY", "2")
    idx.search_filtered_batch_arrays(big[:64], k, "ip", subsets[0])   # (builds the SQ8 codes: exclusive, once)
    got = [None, None]

    def batch_worker(t):
        try:
            got[t] = idx.search_filtered_batch_arrays(big[t * 64:(t + 1) * 64], k, "ip", subsets[t])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=batch_worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(2):
        for qi in (0, 33, 63):
            e_ids, e_d = oracle.canonical_topk_filtered(big[t * 64 + qi], data, k, O.IP, subsets[t])
            assert np.array_equal(got[t][0][qi].astype(np.uint64), e_ids.astype(np.uint64)) and np.array_equal(got[t][1][qi].view(np.uint32), e_d.view(np.uint32))
