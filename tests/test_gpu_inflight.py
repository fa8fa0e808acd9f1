"""Searches IN FLIGHT (lynse_hip_flat_search_submit_* / lynse_hip_flat_search_wait, include/lynse_hip.h).

The reference answers concurrent readers (Arc<RwLock<Collection>>, src/python/mod.rs:950, :1187); the submit / wait pair
keeps several batches enqueued on the device.  The results must be the blocking entry points' results bit for bit — and
therefore the oracle's — including when a batch overflows its candidate buffers and is re-run inside wait()."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
IP, L2, COS, HAMMING = 0, 1, 2, 3


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd

    if lynsedb_amd._lib.device_count() < 1:
        pytest.skip("no HIP device")
    return lynsedb_amd


def _tensors(torch, nq, k, dev):
    return (torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev),
            torch.zeros(nq, dtype=torch.int32, device=dev))


def _host(t):
    r, d, c = t
    return r.cpu().numpy().view(np.uint64), d.cpu().numpy(), c.cpu().numpy().view(np.uint32)


def _assert_oracle(oracle, q, data, k, metric, rows, dists, count, tag):
    e_ids, e_d = oracle.canonical_topk(q, data, k, metric)
    assert int(count) == len(e_ids), tag
    assert np.array_equal(rows[:len(e_ids)].astype(np.uint32), e_ids), tag
    assert np.array_equal(dists[:len(e_ids)].view(np.uint32), e_d.view(np.uint32)), tag


@pytest.mark.parametrize("name,metric,nq,k", [("ip", IP, 256, 10), ("ip", IP, 40, 10), ("l2", L2, 100, 25), ("cosine", COS, 33, 7)])
def test_batches_in_flight_equal_blocking_and_oracle(L, oracle, name, metric, nq, k):
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    n, dim = 150_000, 64          # >= 64K rows: IP batches of 33..256 queries take the certified int8 coarse pass
    data = rng.random((n, dim), dtype=f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    batches = [(data[rng.integers(0, n, nq)] + 0.02 * rng.standard_normal((nq, dim))).astype(f32) for _ in range(7)]
    dq = [torch.as_tensor(b, device=dev) for b in batches]
    blocking = []
    for q in dq:
        o = _tensors(torch, nq, k, dev)
        idx.search_device(q, k, name, *o)
        blocking.append(_host(o))
    idx.profile_enable(True)
    idx.profile_get(reset=True)
    outs = [_tensors(torch, nq, k, dev) for _ in batches]
    pending = []
    for i, q in enumerate(dq):                       # three batches in flight
        pending.append(idx.search_submit(q, k, name, *outs[i]))
        if len(pending) == 3:
            pending.pop(0).wait()
    for t in pending:
        t.wait()
    p = idx.profile_get(reset=True)
    assert p["searches"] == len(batches) and p["fallback_queries"] == 0 and p["scan_launches"] >= len(batches)
    for i in range(len(batches)):
        r, d, c = _host(outs[i])
        br, bd, bc = blocking[i]
        assert np.array_equal(c, bc) and np.array_equal(r, br) and np.array_equal(d.view(np.uint32), bd.view(np.uint32)), i
        for qi in (0, nq // 2, nq - 1):
            _assert_oracle(oracle, batches[i][qi], data, k, metric, r[qi], d[qi], c[qi], (name, i, qi))


def test_packed_hamming_in_flight(L, oracle):
    import torch

    from lynsedb_amd.datasets import packed_bernoulli

    dev = torch.device("cuda", 0)
    n, bits, nq, k = 200_000, 256, 16, 50
    words = packed_bernoulli(n, bits, 0.5, 9)
    idx = L.FlatIndex(None, bits)
    idx.write_packed(words)
    qws = []
    for s in range(4):
        qw = words[np.arange(nq) * 977 + s].copy()
        qw[:, 0] ^= np.uint64(0xFF0F)
        qws.append(qw)
    outs = [_tensors(torch, nq, k, dev) for _ in qws]
    tickets = [idx.search_submit(torch.as_tensor(qw.view(np.int64), device=dev), k, "hamming", *outs[i]) for i, qw in enumerate(qws)]
    for t in tickets:
        t.wait()
    for i, qw in enumerate(qws):
        r, d, c = _host(outs[i])
        for qi in (0, nq - 1):
            e_ids, e_d = oracle.canonical_topk_packed(qw[qi], words, k, HAMMING)
            assert int(c[qi]) == k and np.array_equal(r[qi].astype(np.uint32), e_ids) and np.array_equal(d[qi], e_d), (i, qi)


def test_float_queries_of_a_binary_metric_are_answered_by_the_blocking_path(L, oracle):
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    n, bits, nq, k = 50_000, 256, 40, 10
    data = (rng.random((n, bits)) < 0.5).astype(f32)
    idx = L.FlatIndex(None, bits)
    idx.write(data)
    q = data[rng.integers(0, n, nq)].copy()
    q[:, :7] = 1.0 - q[:, :7]
    o = _tensors(torch, nq, k, dev)
    idx.search_submit(torch.as_tensor(q, device=dev), k, "hamming", *o).wait()   # packed by pack_binary_query inside the blocking call
    r, d, c = _host(o)
    for qi in (0, nq - 1):
        _assert_oracle(oracle, q[qi], data, k, HAMMING, r[qi], d[qi], c[qi], qi)


def test_shapes_that_cannot_be_pipelined_are_answered_inside_submit(L, oracle):
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(6)
    n, dim, k = 40_000, 32, 10
    data = rng.random((n, dim), dtype=f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    for nq in (2, 300):     # the fused few-query search; more than one 256-query pass
        q = rng.random((nq, dim), dtype=f32)
        o = _tensors(torch, nq, k, dev)
        t = idx.search_submit(torch.as_tensor(q, device=dev), k, "ip", *o)   # (also builds the derived data: first search)
        t.wait()
        r, d, c = _host(o)
        for qi in (0, nq - 1):
            _assert_oracle(oracle, q[qi], data, k, IP, r[qi], d[qi], c[qi], (nq, qi))


def test_more_tickets_than_contexts_is_an_error_not_a_deadlock(L):
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    n, dim, nq, k = 80_000, 32, 64, 5
    data = rng.random((n, dim), dtype=f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    q = torch.as_tensor(rng.random((nq, dim), dtype=f32), device=dev)
    o = [_tensors(torch, nq, k, dev) for _ in range(9)]
    idx.search_device(q, k, "l2", *o[0])           # derived data up to date
    tickets = []
    with pytest.raises(Exception, match="in flight"):
        for i in range(9):                          # at most 8 contexts per handle (default 4)
            tickets.append(idx.search_submit(q, k, "l2", *o[i]))
    assert 1 <= len(tickets) <= 8
    # a search that needs the writer lock while tickets are outstanding is refused, not deadlocked
    with pytest.raises(Exception, match="outstanding"):
        idx.search_submit(torch.as_tensor(rng.random((300, dim), dtype=f32), device=dev), k, "l2", *_tensors(torch, 300, k, dev))
    for t in tickets:
        t.wait()
    ref = _host(o[0])
    for i in range(1, len(tickets)):
        r, d, c = _host(o[i])
        assert np.array_equal(r, ref[0]) and np.array_equal(d, ref[1]) and np.array_equal(c, ref[2])
    idx.search_submit(q, k, "l2", *o[0]).wait()    # contexts are free again


def test_overflowed_batch_is_rerun_inside_wait(L, oracle):
    """Scores increase with the row index (test_float_parity_adversarial_monotone): the submitted plan overflows its
    candidate buffers; wait() must climb the plan levels and return the exact answer."""
    import torch

    dev = torch.device("cuda", 0)
    n, dim, nq, k = 60_000, 8, 40, 10
    data = np.zeros((n, dim), f32)
    data[:, 0] = np.arange(n, dtype=f32) / f32(n)
    data[:, 1] = f32(0.5)
    q = np.zeros((nq, dim), f32)
    q[:, 0] = 1.0
    q[:, 1] = np.linspace(0.0, 0.5, nq, dtype=f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    idx.set_fused_search(False)
    dq = torch.as_tensor(q, device=dev)
    o0 = _tensors(torch, nq, k, dev)
    idx.search_device(dq, k, "ip", *o0)
    idx.profile_enable(True)
    idx.profile_get(reset=True)
    outs = [_tensors(torch, nq, k, dev) for _ in range(3)]
    tickets = [idx.search_submit(dq, k, "ip", *o) for o in outs]
    for t in tickets:
        t.wait()
    p = idx.profile_get(reset=True)
    assert p["fallback_queries"] > 0, p
    for o in outs:
        r, d, c = _host(o)
        for qi in (0, 17, nq - 1):
            _assert_oracle(oracle, q[qi], data, k, IP, r[qi], d[qi], c[qi], qi)


def test_sharded_submit_with_a_one_rank_communicator(L, oracle):
    """The exchange half of a ticket (status word in the result block, merge kernel, pinned status) on the one GPU a test
    box has: a 1-rank RCCL communicator, global rows through the row map."""
    import torch

    from lynsedb_amd.sharded import NativeComm, ShardedFlat

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    n, dim, nq, k = 120_000, 64, 128, 10
    data = rng.random((n, dim), dtype=f32)
    sh = ShardedFlat(dim, rank=0, world=1, device=0, group=None)
    sh.index.write(data)
    sh.index.finalize()
    sh.comm = NativeComm(None, 0, 1, 0)
    batches = [(data[rng.integers(0, n, nq)] + 0.01).astype(f32) for _ in range(4)]
    outs = [sh.alloc_outputs(nq, k) for _ in batches]
    tickets = []
    for b, o in zip(batches, outs):
        tickets.append(sh.index.search_submit(torch.as_tensor(b, device=dev), k, IP, o.rows, o.dists, o.counts, comm=sh.comm.handle))
    for t in tickets:
        t.wait()
    # the blocking sharded entry point still works next to it (its own block pair)
    o2 = sh.alloc_outputs(nq, k)
    L._lib.check(L._lib.lib.lynse_hip_flat_search_sharded_f32_device(
        sh.index.handle, sh.comm.handle, C.c_void_p(torch.as_tensor(batches[0], device=dev).data_ptr()), nq, k, IP,
        C.c_void_p(o2.rows.data_ptr()), C.c_void_p(o2.dists.data_ptr()), C.c_void_p(o2.counts.data_ptr())))
    torch.cuda.synchronize()
    assert torch.equal(o2.rows, outs[0].rows) and torch.equal(o2.dists, outs[0].dists) and torch.equal(o2.counts, outs[0].counts)
    for b, o in zip(batches, outs):
        r, d, c = o.rows.cpu().numpy().view(np.uint64), o.dists.cpu().numpy(), o.counts.cpu().numpy()
        for qi in (0, 63, nq - 1):
            _assert_oracle(oracle, b[qi], data, k, IP, r[qi], d[qi], c[qi], qi)


def test_a_ticket_that_needs_the_lazy_f16_shadow_while_int8_tickets_are_outstanding(L, oracle):
    """the f16 shadow is a lazy copy — a shard that has only answered int8 batches does not hold it.  A batch that needs it
    (<= 32 queries on a shard under 256K rows; L2 below 256 dimensions) submitted while int8 tickets are outstanding used to fail in
    submit ("cannot be pipelined"), and on a sharded collection that rank's peers stalled in the all-gather.  Now submit builds the
    copy under the handle's exclusive lock; both kinds of ticket are in flight together and equal the oracle."""
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(17)
    n, dim, k = 100_000, 64, 10
    data = rng.random((n, dim), dtype=f32)
    for with_comm in (False, True):
        idx = L.FlatIndex(None, dim)
        idx.write(data)
        idx.finalize()
        comm = None
        if with_comm:
            from lynsedb_amd.sharded import NativeComm
            try:
                comm = NativeComm(None, 0, 1, 0)      # a 1-rank communicator: the sharded code path without a second GPU
            except Exception as e:  # noqa: BLE001
                pytest.skip("RCCL unavailable: %r" % (e,))
        big = (data[rng.integers(0, n, 64)] + 0.02 * rng.standard_normal((64, dim))).astype(f32)     # 64 queries, IP: certified int8 pass
        small = (data[rng.integers(0, n, 8)] + 0.02 * rng.standard_normal((8, dim))).astype(f32)     # 8 queries: the f16 shadow (shard < 256K rows)
        l2q = (data[rng.integers(0, n, 40)] + 0.02 * rng.standard_normal((40, dim))).astype(f32)     # L2 at 64 dimensions: the f16 shadow
        dbig, dsmall, dl2 = (torch.as_tensor(x, device=dev) for x in (big, small, l2q))
        o1, o2, o3, o4 = _tensors(torch, 64, k, dev), _tensors(torch, 8, k, dev), _tensors(torch, 40, k, dev), _tensors(torch, 64, k, dev)
        kw = {"comm": comm.handle} if comm is not None else {}
        t1 = idx.search_submit(dbig, k, "ip", *o1, **kw)
        t2 = idx.search_submit(dsmall, k, "ip", *o2, **kw)     # needs the shadow while t1 is outstanding
        t3 = idx.search_submit(dl2, k, "l2", *o3, **kw)
        t4 = idx.search_submit(dbig, k, "ip", *o4, **kw)
        for t in (t1, t2, t3, t4):
            t.wait()
        for (o, q, metric, name) in ((o1, big, IP, "big"), (o2, small, IP, "small"), (o3, l2q, L2, "l2"), (o4, big, IP, "big again")):
            r, d, c = _host(o)
            for qi in (0, len(q) - 1):
                _assert_oracle(oracle, q[qi], data, k, metric, r[qi], d[qi], c[qi], (with_comm, name, qi))
        if comm is not None:
            comm.close()


def test_a_local_failure_of_a_sharded_submit_rides_the_exchange(L, oracle, monkeypatch):
    """once a sharded batch's shape is pipelined, a LOCAL failure of the preparation (a lazy build out of memory, the writer
    lock refused behind outstanding tickets, ...) must not end `submit` in front of the all-gather — the peers would wait out the collective
    timeout.  LYNSE_HIP_DEBUG_FAIL_SUBMIT=1 simulates one on a 1-rank communicator: submit still returns a ticket (the exchange is enqueued with an
    empty block + the failure bit), `wait` reports the local error, the context is given back and the next batch on the same handle is answered."""
    import torch

    from lynsedb_amd.sharded import NativeComm

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(23)
    n, dim, nq, k = 90_000, 64, 64, 10
    data = rng.random((n, dim), dtype=f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    comm = NativeComm(None, 0, 1, 0)
    q = (data[rng.integers(0, n, nq)] + 0.01).astype(f32)
    dq = torch.as_tensor(q, device=dev)
    o_bad, o_good = _tensors(torch, nq, k, dev), _tensors(torch, nq, k, dev)
    monkeypatch.setenv("LYNSE_HIP_DEBUG_FAIL_SUBMIT", "1")
    t = idx.search_submit(dq, k, "ip", *o_bad, comm=comm.handle)          # no exception here: the failure travels with the block
    with pytest.raises(MemoryError, match="simulated local failure"):
        t.wait()
    monkeypatch.delenv("LYNSE_HIP_DEBUG_FAIL_SUBMIT")
    for _ in range(9):                                                      # every context was given back: more batches than contexts in a row
        idx.search_submit(dq, k, "ip", *o_good, comm=comm.handle).wait()
    r, d, c = _host(o_good)
    for qi in (0, nq - 1):
        _assert_oracle(oracle, q[qi], data, k, IP, r[qi], d[qi], c[qi], qi)
    idx.write(data[:10])                                                    # ... and the writer guard is open again (no ticket left counted)
    comm.close()
