"""world_size-2 / 4 / 8 `gloo` tests of the multi-GPU exchange step on CPU (no GPU needed).

Covers the N>1 path of lynsedb_amd/sharded.py: the row partition rule (global row g on rank g % G),
the fixed-size per-rank result block, the all-gather and the canonical (distance, row) k-way merge
(lynse_hip_merge_topk, host side of the C ABI).  The per-shard scan itself needs a GPU, so each rank
produces its shard's top-k with the CPU oracle (allowed in tests) and the merged result must equal
the oracle's answer on the unsharded collection — bit-exact ids and distances, ties included.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, metric, ret):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle as O
    from lynsedb_amd.sharded import ShardedFlat, shard_of_row

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = O.get()
        rng = np.random.default_rng(1234)  # same global data on every rank
        n, dim, nq, k = 3001, 24, 5, 12
        data = rng.integers(0, 4, size=(n, dim)).astype(np.float32)  # small ints -> many exact ties
        queries = rng.integers(0, 4, size=(nq, dim)).astype(np.float32)
        mine = np.arange(rank, n, world)
        assert all(shard_of_row(int(g), world) == rank for g in mine[:10])
        local = data[mine]
        rows = np.full((nq, k), np.iinfo(np.uint64).max, np.uint64)
        dists = np.zeros((nq, k), np.float32)
        counts = np.zeros(nq, np.uint32)
        for q in range(nq):
            i, d = orc.canonical_topk(queries[q], local, k, metric, O.IPFORM_SINGLE)
            rows[q, :len(i)] = i.astype(np.uint64) * world + rank  # local row -> global row (row map)
            dists[q, :len(i)] = d
            counts[q] = len(i)
        m_rows, m_dists, m_counts = ShardedFlat.allgather_merge_host(dist, world, rows, dists, counts, k, metric)
        ok = True
        for q in range(nq):
            e_i, e_d = orc.canonical_topk(queries[q], data, k, metric, O.IPFORM_SINGLE)
            ok &= int(m_counts[q]) == len(e_i)
            ok &= np.array_equal(m_rows[q, :len(e_i)], e_i.astype(np.uint64))
            ok &= np.array_equal(m_dists[q, :len(e_i)], e_d)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,metric", [(2, 0), (2, 1), (2, 3), (4, 0), (4, 3), (8, 1), (8, 3)])  # IP (descending), L2, Hamming (heavy ties)
def test_allgather_merge(world, metric):
    import torch.multiprocessing as mp

    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, metric, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _ivf_worker(rank, world, port, ret):
    """Row-sharded IVF (BASELINE config 4): each rank answers from ITS rows of the probed lists (CPU oracle stands in for
    the per-shard GPU scan), ShardedIvf.search does the exchange + merge; the result must equal the unsharded index."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle as O
    from lynsedb_amd.sharded import ShardedIvf

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = O.get()
        rng = np.random.default_rng(99)
        n, dim, nlist, nprobe, nq, k = 2000, 16, 12, 3, 6, 8
        data = rng.standard_normal((n, dim)).astype(np.float32)
        queries = rng.standard_normal((nq, dim)).astype(np.float32)
        cen, asg = orc.kmeans_train(data, nlist, 10, O.L2)
        off, rows = orc.lists_from_assignments(asg, cen.shape[0])
        mine = np.arange(rank, n, world)
        l_off, l_rows = orc.lists_from_assignments(asg[mine], cen.shape[0])

        class _Shard(ShardedIvf):  # the GPU scan of the local shard, restated with the oracle
            def search_local(self, q, k_, nprobe_):
                r = np.full((q.shape[0], k_), np.iinfo(np.uint64).max, np.uint64)
                d = np.zeros((q.shape[0], k_), np.float32)
                c = np.zeros(q.shape[0], np.uint32)
                for i in range(q.shape[0]):
                    # probed lists come from the GLOBAL centroids; an empty local part contributes nothing (no fallback)
                    _, _, probed = orc.ivf_search(q[i], data, cen, off, rows, nprobe_, k_, O.L2)
                    cand = np.concatenate([l_rows[int(l_off[p]):int(l_off[p + 1])] for p in probed]).astype(np.int64)
                    if cand.size:
                        ids, dd = orc.canonical_topk(q[i], data[mine][cand], k_, O.L2, O.IPFORM_SINGLE)
                        g = mine[cand[ids.astype(np.int64)]]
                        order = np.lexsort((g, dd))
                        r[i, :len(ids)], d[i, :len(ids)], c[i] = g[order], dd[order], len(ids)
                return r, d, c

        s = _Shard(dim, rank=rank, world=world, group=dist)
        s.metric = "l2"
        m_rows, m_d, m_c = s.search(queries, k, nprobe)
        ok = True
        for i in range(nq):
            e_ids, e_d, _ = orc.ivf_search(queries[i], data, cen, off, rows, nprobe, k, O.L2)
            ok &= int(m_c[i]) == len(e_ids) and np.array_equal(m_rows[i, :len(e_ids)], e_ids) and np.array_equal(m_d[i, :len(e_ids)], e_d)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_ivf_exchange(world):
    import torch.multiprocessing as mp

    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ivf_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _ordered_values(rank, n=1000):
    """f32 values whose sum depends on the order of the additions (magnitudes spread over 2^24 and both signs)."""
    rng = np.random.default_rng(1000 + rank)
    return (rng.standard_normal(n) * np.exp2(rng.integers(-12, 13, n))).astype(np.float32)


def _reduce_worker(rank, world, port, ret):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C

    import torch.distributed as dist

    import oracle as O
    from lynsedb_amd.sharded import ShardedIvf

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = ShardedIvf(8, rank=rank, world=world, device=0, group=dist)
        fn = sh._host_reduce(None, 0)                      # the lynse_hip_reduce_fn the all-reduced k-means calls (lynse_hip_ivf_kmeans_sharded)
        f = (np.arange(1000, dtype=np.float32) * (rank + 1) * 0.37).astype(np.float32)
        u = (np.arange(77, dtype=np.uint32) + 5 * rank).astype(np.uint32)
        o = _ordered_values(rank)
        assert fn(None, f.ctypes.data_as(C.c_void_p), f.size, 0) == 0
        assert fn(None, u.ctypes.data_as(C.c_void_p), u.size, 1) == 0
        assert fn(None, o.ctypes.data_as(C.c_void_p), o.size, 2) == 0      # dtype 2: f32 in RANK order
        # the sharded Lloyd loop restated over the launcher's reduction (every rank holds the rows g % world == rank; the device
        # half of lynse_hip_ivf_kmeans_sharded is tests/test_gpu_sharded_kmeans.py): centroids + assignments must be the oracle's
        # lo_kmeans_train_sharded(world) bit for bit — the rank-ordered sum is what makes that true beyond two ranks
        orc = O.get()
        rng = np.random.default_rng(5)
        nlist, dim, n, iters = 12, 24, 3000, 20
        centers = (rng.standard_normal((nlist, dim)) * 4).astype(np.float32)
        data = (centers[rng.integers(0, nlist, n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
        km = {}
        for metric in (O.L2, O.IP):
            cen, _ = orc.kmeans_train(data, nlist, 0, metric)      # max_iter 0: the FastRng sample + farthest-first init (kmeans.rs:141-196)
            cen = cen.copy()
            local = np.ascontiguousarray(data[rank::world])
            cur = np.full(local.shape[0], 0xFFFFFFFF, np.uint32)
            for _ in range(iters):
                nxt = orc.kmeans_assign(local, cen, metric)
                words = np.zeros(nlist + 1, np.uint32)
                words[:nlist] = np.bincount(nxt, minlength=nlist).astype(np.uint32)
                words[nlist] = 1 if not np.array_equal(nxt, cur) else 0
                cur = nxt
                sums = np.zeros((nlist, dim), np.float32)
                np.add.at(sums, cur.astype(np.int64), local)       # unbuffered: one f32 add per member row, ascending row order (kmeans.rs:273-286)
                flat = sums.reshape(-1)
                assert fn(None, flat.ctypes.data_as(C.c_void_p), flat.size, 2) == 0
                assert fn(None, words.ctypes.data_as(C.c_void_p), words.size, 1) == 0
                counts = words[:nlist]
                max_c = nlist - 1 - int(np.argmax(counts[::-1]))  # max_by_key keeps the LAST maximum (kmeans.rs:105-110)
                for c in range(nlist):
                    if counts[c] > 0:
                        cen[c] = sums[c] * (np.float32(1.0) / np.float32(counts[c]))
                    elif counts[max_c] > 1:
                        cen[c] = cen[max_c] * (np.float32(1.0) + np.float32(1e-4) * np.arange(dim, dtype=np.float32))
                if words[nlist] == 0:
                    break
            km[int(metric)] = (cen.copy(), orc.kmeans_assign(local, cen, metric))
        ret[rank] = (f.copy(), u.copy(), o.copy(), km)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_kmeans_reduction_and_the_restatement_it_is_tested_against(world):
    """Host side of the row-sharded IVF training (lynse_hip_ivf_kmeans_sharded; the device part: tests/test_gpu_sharded_kmeans.py):
    (1) the reduction callback of the launcher sums f32 / u32 host buffers over the gloo ranks in place, and its RANK-ORDERED form
    (dtype 2) returns ((p0 + p1) + p2) + ... bit for bit on values whose sum depends on the order; (2) the sharded Lloyd loop run
    over that callback by `world` gloo processes ends with the centroids and assignments of the oracle's restatement
    (lo_kmeans_train_sharded: per-rank sequential sums added in rank order) — bit for bit at 2, 4 and 8 ranks; (3) that
    restatement is kmeans_train itself for one rank and stays within rounding of it otherwise."""
    import multiprocessing as mp

    import oracle as O

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        ps = [ctx.Process(target=_reduce_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(300)
            assert p.exitcode == 0
        got = [ret[r] for r in range(world)]
    want_f = np.zeros(1000, np.float32)
    want_o = _ordered_values(0)
    for r in range(world):
        want_f = want_f + (np.arange(1000, dtype=np.float32) * (r + 1) * 0.37).astype(np.float32)
        if r:
            want_o = want_o + _ordered_values(r)
    want_u = np.arange(77, dtype=np.uint32) * world + 5 * sum(range(world))
    rev = _ordered_values(world - 1)
    for r in range(world - 2, -1, -1):
        rev = rev + _ordered_values(r)
    if world > 2:
        assert not np.array_equal(rev.view(np.uint32), want_o.view(np.uint32))   # (the order matters on these values)
    for f, u, o, _ in got:
        assert np.allclose(f, want_f, rtol=1e-6)
        assert np.array_equal(u, want_u)
        assert np.array_equal(o.view(np.uint32), want_o.view(np.uint32))
    orc = O.get()
    rng = np.random.default_rng(5)
    centers = (rng.standard_normal((12, 24)) * 4).astype(np.float32)
    data = (centers[rng.integers(0, 12, 3000)] + 0.3 * rng.standard_normal((3000, 24))).astype(np.float32)
    for metric in (O.L2, O.IP):
        c_w, a_w = orc.kmeans_train_sharded(data, 12, 20, metric, world)
        for r in range(world):
            cen, asg = got[r][3][int(metric)]
            assert np.array_equal(cen.view(np.uint32), c_w.view(np.uint32)), (world, r, metric)
            assert np.array_equal(asg, a_w[r::world]), (world, r, metric)
    for metric in (O.L2, O.IP, O.COS):
        c_one, a_one = orc.kmeans_train(data, 12, 20, metric)
        c_w1, a_w1 = orc.kmeans_train_sharded(data, 12, 20, metric, 1)
        assert np.array_equal(c_one.view(np.uint32), c_w1.view(np.uint32)) and np.array_equal(a_one, a_w1)
        c_w, a_w = orc.kmeans_train_sharded(data, 12, 20, metric, world)
        assert np.allclose(c_one, c_w, rtol=1e-5, atol=1e-5) and np.array_equal(a_one, a_w), metric
