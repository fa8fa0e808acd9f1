"""Row-sharded IVF training on the device (`lynse_hip_ivf_kmeans_sharded`, SURVEY 8e / VERDICT r3 "missing" 2): every rank holds the
rows g % world == rank, k-means runs over the whole collection with ONE all-reduce of the centroid sums + counts per Lloyd
iteration, every rank ends with the same centroids and the assignments of its rows.

Checked against the oracle's restatement of that algorithm (`lo_kmeans_train_sharded`: kmeans_train on the union — kmeans.rs:74-139 —
with the sums formed per rank, sequentially, and added in rank order):
  * two / three / four / eight ranks as THREADS of this process, the reduction a callback (`lynse_hip_reduce_fn`) that adds the host
    buffers in rank order — centroid bits and assignments equal the restatement's for ip / l2 / cosine, host rows and device rows;
  * the communicator's own device-side reduction (all-gather + rank-ordered add kernel + integer all-reduce) on a 1-rank communicator;
  * one rank: identical to the single-index device k-means (`IvfFlatIndex.build`) and to kmeans_train itself;
  * two and four gloo PROCESSES sharing the GPU: `ShardedIvf.train` through torch.distributed — same bits again, and the row-sharded index
    built from the result answers like the oracle's IVFIndex over the union.
"""
import os
import socket
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
f32 = np.float32
ROOT = Path(__file__).resolve().parent.parent
NAME = {O.IP: "ip", O.L2: "l2", O.COS: "cosine"}


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1
    return L_


def clustered(rng, n, dim, nc=24, spread=0.3):
    centers = (rng.standard_normal((nc, dim)) * 3).astype(f32)
    return (centers[rng.integers(0, nc, n)] + spread * rng.standard_normal((n, dim))).astype(f32)


class RankOrderSum:
    """In-process stand-in for the reduction: every rank deposits its buffer, all leave with the sum taken in RANK order,
    ((p0 + p1) + p2) + ... — what `lynse_hip_reduce_fn` promises for dtype 2 (and harmless for the integer words)."""

    def __init__(self, world=2):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def reducer(self, rank):
        def reduce(arr):
            self.slots[rank] = arr.copy()
            self.bar.wait()
            total = self.slots[0].copy()
            for r in range(1, self.world):
                total = total + self.slots[r]
            self.bar.wait()
            arr[:] = total
        return reduce


TwoRankSum = RankOrderSum


@pytest.mark.parametrize("metric", [O.L2, O.IP, O.COS])
@pytest.mark.parametrize("on_device", [False, True])
def test_two_ranks_in_one_process_equal_the_restatement(L, oracle, metric, on_device):
    _ranks_in_one_process(oracle, 2, metric, on_device)


@pytest.mark.parametrize("world,metric,on_device", [(4, O.L2, True), (4, O.IP, False), (8, O.L2, False), (8, O.COS, True), (3, O.IP, True)])
def test_more_ranks_in_one_process_equal_the_restatement(L, oracle, world, metric, on_device):
    """From three ranks on the ORDER of the reduction matters: the per-rank sums are added in rank order (VERDICT r5 'missing' 3)."""
    _ranks_in_one_process(oracle, world, metric, on_device)


def _ranks_in_one_process(oracle, world, metric, on_device):
    import torch

    from lynsedb_amd.sharded import ShardedIvf

    rng = np.random.default_rng(900 + metric)
    n, dim, nlist, iters = 20_001, 40, 96, 12          # an odd row count: the ranks hold unequal numbers of rows
    data = clustered(rng, n, dim)
    want_c, want_a = oracle.kmeans_train_sharded(data, nlist, iters, metric, world)
    hub = RankOrderSum(world)
    out = [None] * world
    errs = []

    def run(rank):
        try:
            sh = ShardedIvf(dim, rank=rank, world=world, device=0)
            local = np.ascontiguousarray(data[rank::world])
            rows = torch.from_numpy(local).to("cuda:0") if on_device else local
            out[rank] = sh.train(rows, n, nlist, iters, NAME[metric], reduce=hub.reducer(rank))
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            hub.bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs
    for rank in range(world):
        cen, asg = out[rank]
        assert cen.shape == want_c.shape and np.array_equal(cen.view(np.uint32), want_c.view(np.uint32)), (rank, np.abs(cen - want_c).max())
        assert np.array_equal(asg, want_a[rank::world]), rank
    # training on the union differs in the rounding of the sums only: a handful of rows on a cell border change sides
    one_c, one_a = oracle.kmeans_train(data, nlist, iters, metric)
    assert np.mean(np.concatenate([out[r][1] for r in range(world)]) == np.concatenate([one_a[r::world] for r in range(world)])) > 0.99


def test_the_communicators_device_reduction_runs_on_one_rank(L, oracle, monkeypatch):
    """The communicator's form of a Lloyd iteration's reduction (ncclAllGather of the per-rank sums -> k_sum_rank_order -> integer
    ncclAllReduce of counts + stop word, all on device buffers, comm_host.inc) on the one GPU there is: LYNSE_HIP_KM_FORCE_COLLECTIVE=1
    sends a 1-rank training through it — same centroids and assignments as kmeans_train, bit for bit."""
    import torch

    from lynsedb_amd.sharded import NativeComm, ShardedIvf

    rng = np.random.default_rng(78)
    n, dim, nlist, iters = 12_000, 48, 64, 10
    data = clustered(rng, n, dim)
    want_c, want_a = oracle.kmeans_train(data, nlist, iters, O.IP)
    monkeypatch.setenv("LYNSE_HIP_KM_FORCE_COLLECTIVE", "1")
    sh = ShardedIvf(dim, rank=0, world=1, device=0)
    sh.comm = NativeComm(None, 0, 1, 0)
    for rows in (data, torch.from_numpy(data).to("cuda:0")):
        cen, asg = sh.train(rows, n, nlist, iters, "ip")
        assert np.array_equal(cen.view(np.uint32), want_c.view(np.uint32)) and np.array_equal(asg, want_a)


def test_one_rank_is_the_single_index_training(L, oracle):
    from lynsedb_amd.sharded import ShardedIvf

    rng = np.random.default_rng(77)
    n, dim, nlist, iters = 9000, 32, 50, 10
    data = clustered(rng, n, dim)
    sh = ShardedIvf(dim, rank=0, world=1, device=0)
    cen, asg = sh.train(data, n, nlist, iters, "l2")
    want_c, want_a = oracle.kmeans_train(data, nlist, iters, O.L2)
    assert np.array_equal(cen.view(np.uint32), want_c.view(np.uint32)) and np.array_equal(asg, want_a)
    idx = L.IvfFlatIndex.build(None, data, dim, nlist, iters, "l2", l2_partitions=False)
    c2, a2, _, _ = idx.export()
    assert np.array_equal(c2.view(np.uint32), cen.view(np.uint32)) and np.array_equal(a2, asg)


def test_build_sharded_device_in_one_call(L, oracle):
    """`lynse_hip_ivf_build_sharded_device`: one rank = the single-index device build; two ranks (threads + callback reduction) = the
    restatement's centroids, and their merged answers are IVFIndex::search over the union under those centroids."""
    import torch

    from lynsedb_amd.sharded import ShardedIvf

    rng = np.random.default_rng(5)
    n, dim, nlist, iters, nprobe, k = 16_000, 32, 48, 6, 5, 10
    data = clustered(rng, n, dim)
    queries = (data[rng.integers(0, n, 20)] + 0.05 * rng.standard_normal((20, dim))).astype(f32)
    one = ShardedIvf(dim, rank=0, world=1, device=0)
    one.build_device(torch.from_numpy(data).to("cuda:0"), n, nlist, iters, "l2")
    c1, a1, _, _ = one.index.export()
    want_c, want_a = oracle.kmeans_train(data, nlist, iters, O.L2)
    assert np.array_equal(c1.view(np.uint32), want_c.view(np.uint32)) and np.array_equal(a1, want_a)
    hub = TwoRankSum()
    shards, errs = [None, None], []

    def run(rank):
        try:
            sh = ShardedIvf(dim, rank=rank, world=2, device=0)
            sh.build_device(torch.from_numpy(np.ascontiguousarray(data[rank::2])).to("cuda:0"), n, nlist, iters, "l2", reduce=hub.reducer(rank))
            shards[rank] = sh
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            hub.bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs, errs
    want_c, want_a = oracle.kmeans_train_sharded(data, nlist, iters, O.L2, 2)
    for rank in range(2):
        c, a, _, _ = shards[rank].index.export()
        assert np.array_equal(c.view(np.uint32), want_c.view(np.uint32)) and np.array_equal(a, want_a[rank::2]), rank
    offsets, list_rows = oracle.lists_from_assignments(want_a, want_c.shape[0])
    parts = [shards[r].search_local(queries, k, nprobe) for r in range(2)]
    for qi in range(queries.shape[0]):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, want_c, offsets, list_rows, nprobe, k, O.L2)
        ids = np.concatenate([parts[r][0][qi, :int(parts[r][2][qi])] for r in range(2)])
        ds = np.concatenate([parts[r][1][qi, :int(parts[r][2][qi])] for r in range(2)])
        m_ids, m_d = oracle.merge_results(ids.astype(np.uint64), ds, k, O.L2)
        assert np.array_equal(np.asarray(m_ids, np.uint64), e_ids.astype(np.uint64)) and np.array_equal(np.asarray(m_d, f32).view(np.uint32), e_d.view(np.uint32)), qi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, ret):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from lynsedb_amd.sharded import ShardedIvf

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(4242)           # the same global data on every rank
        n, dim, nlist, nprobe, k = 30_000, 48, 64, 6, 10
        data = clustered(rng, n, dim)
        queries = data[rng.integers(0, n, 24)] + 0.05 * rng.standard_normal((24, dim)).astype(np.float32)
        sh = ShardedIvf(dim, rank=rank, world=world, device=0, group=dist)
        local = np.ascontiguousarray(data[rank::world])
        cen, asg = sh.train(local, n, nlist, 8, "ip")          # torch.distributed (gloo) sums the host buffers
        sh.load_local(local, cen, asg, "ip")
        rows, dists, counts = sh.search(queries.astype(np.float32), k, nprobe)
        ret[rank] = (cen, asg, rows, dists, counts)
    finally:
        dist.destroy_process_group()
        del torch


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_processes_train_and_search_like_the_union(L, oracle, world):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        ps = [ctx.Process(target=_gloo_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(900)
            assert p.exitcode == 0
        got = [ret[r] for r in range(world)]
    rng = np.random.default_rng(4242)
    n, dim, nlist, nprobe, k = 30_000, 48, 64, 6, 10
    data = clustered(rng, n, dim)
    queries = (data[rng.integers(0, n, 24)] + 0.05 * rng.standard_normal((24, dim)).astype(f32)).astype(f32)
    want_c, want_a = oracle.kmeans_train_sharded(data, nlist, 8, O.IP, world)
    for rank in range(world):
        cen, asg, rows, dists, counts = got[rank]
        assert np.array_equal(cen.view(np.uint32), want_c.view(np.uint32)) and np.array_equal(asg, want_a[rank::world]), rank
    # the row-sharded index answers like IVFIndex::search over the union under these centroids / lists (ivf.rs:181-348)
    offsets, list_rows = oracle.lists_from_assignments(want_a, want_c.shape[0])
    rows, dists, counts = got[0][2], got[0][3], got[0][4]
    for rank in range(1, world):
        assert np.array_equal(got[rank][2], rows) and np.array_equal(got[rank][3].view(np.uint32), dists.view(np.uint32))
    for qi in range(queries.shape[0]):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, want_c, offsets, list_rows, nprobe, k, O.IP)
        c = int(counts[qi])
        assert c == len(e_ids) and np.array_equal(rows[qi, :c].astype(np.uint64), e_ids.astype(np.uint64)), qi
        assert np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32)), qi
