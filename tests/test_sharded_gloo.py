"""world_size-2 `gloo` test of the multi-GPU exchange step on CPU (no GPU needed).

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


@pytest.mark.parametrize("metric", [0, 1, 3])  # IP (descending), L2, Hamming (heavy ties)
def test_allgather_merge_world2(metric):
    import torch.multiprocessing as mp

    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, metric, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


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


def test_sharded_ivf_exchange_world2():
    import torch.multiprocessing as mp

    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ivf_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _reduce_worker(rank, world, port, ret):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C

    import torch.distributed as dist

    from lynsedb_amd.sharded import ShardedIvf

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = ShardedIvf(8, rank=rank, world=world, device=0, group=dist)
        fn = sh._host_reduce(None, 0)                      # the lynse_hip_reduce_fn the all-reduced k-means calls (lynse_hip_ivf_kmeans_sharded)
        f = (np.arange(1000, dtype=np.float32) * (rank + 1) * 0.37).astype(np.float32)
        u = (np.arange(77, dtype=np.uint32) + 5 * rank).astype(np.uint32)
        assert fn(None, f.ctypes.data_as(C.c_void_p), f.size, 0) == 0
        assert fn(None, u.ctypes.data_as(C.c_void_p), u.size, 1) == 0
        ret[rank] = (f.copy(), u.copy())
    finally:
        dist.destroy_process_group()


def test_sharded_kmeans_reduction_world2_and_the_restatement_it_is_tested_against():
    """Host side of the row-sharded IVF training (lynse_hip_ivf_kmeans_sharded; the device part: tests/test_gpu_sharded_kmeans.py):
    (1) the reduction callback of the launcher sums f32 / u32 host buffers over the gloo ranks in place; (2) the oracle's
    restatement of the sharded algorithm (lo_kmeans_train_sharded: per-rank sequential sums added in rank order) is kmeans_train
    itself for one rank — bit for bit — and stays within rounding of it for two (same lists on well-separated data)."""
    import multiprocessing as mp

    import oracle as O

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        ps = [ctx.Process(target=_reduce_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(120)
            assert p.exitcode == 0
        f0, u0 = ret[0]
        f1, u1 = ret[1]
    want_f = (np.arange(1000, dtype=np.float32) * 0.37).astype(np.float32) + (np.arange(1000, dtype=np.float32) * 2 * 0.37).astype(np.float32)
    assert np.array_equal(f0.view(np.uint32), want_f.view(np.uint32)) and np.array_equal(f1.view(np.uint32), want_f.view(np.uint32))
    assert np.array_equal(u0, np.arange(77, dtype=np.uint32) * 2 + 5) and np.array_equal(u1, u0)
    orc = O.get()
    rng = np.random.default_rng(5)
    centers = (rng.standard_normal((12, 24)) * 4).astype(np.float32)
    data = (centers[rng.integers(0, 12, 5000)] + 0.3 * rng.standard_normal((5000, 24))).astype(np.float32)
    for metric in (O.L2, O.IP, O.COS):
        c_one, a_one = orc.kmeans_train(data, 12, 20, metric)
        c_w1, a_w1 = orc.kmeans_train_sharded(data, 12, 20, metric, 1)
        assert np.array_equal(c_one.view(np.uint32), c_w1.view(np.uint32)) and np.array_equal(a_one, a_w1)
        c_w2, a_w2 = orc.kmeans_train_sharded(data, 12, 20, metric, 2)
        assert np.allclose(c_one, c_w2, rtol=1e-5, atol=1e-5) and np.array_equal(a_one, a_w2), metric
