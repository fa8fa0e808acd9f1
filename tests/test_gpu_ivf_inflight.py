"""IVF searches IN FLIGHT (lynse_hip_ivf_search_submit_f32_device / lynse_hip_ivf_search_wait, include/lynse_hip.h; VERDICT r3 "missing" 3).

IVFIndex::search (ivf.rs:181-348) under the reference's concurrent readers (src/python/mod.rs:950, :1187): several batches enqueued on
the device — centroid ranking, device-side grouping, list scans, rescoring, and for a row-sharded index the exchange — with no host
synchronisation between their steps.  The results must be the blocking entry points' bit for bit, and therefore the oracle's,
including when something the blocking path would have noticed on the host happens in flight: the all-lists-empty fallback of
ivf.rs:258-265, a candidate overflow of the int8 pass."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
f32 = np.float32
IP, L2, COS = O.IP, O.L2, O.COS
NAME = {IP: "ip", L2: "l2", COS: "cosine"}


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


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])


def _index(L, oracle, rng, n, dim, nlist, metric, iters=2, nc=None):
    centers = rng.standard_normal((nc or max(nlist // 2, 4), dim)).astype(f32)
    data = (centers[rng.integers(0, centers.shape[0], n)] + 0.3 * rng.standard_normal((n, dim))).astype(f32)
    built = L.IvfFlatIndex.build(None, data, dim, nlist, iters, NAME[metric], l2_partitions=False)
    cen, asg, _, _ = built.export()
    del built
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    return data, cen, asg, off, rows, idx


def _assert_oracle(oracle, q, data, cen, off, rows, nprobe, k, metric, got, qi, tag=""):
    e_ids, e_d, _ = oracle.ivf_search(q, data, cen, off, rows, nprobe, k, metric)
    c = int(got[2][qi])
    assert c == len(e_ids), (tag, qi, c, len(e_ids))
    assert np.array_equal(got[0][qi, :c], e_ids.astype(np.uint64)), (tag, qi, got[0][qi, :c], e_ids)
    assert np.array_equal(got[1][qi, :c].view(np.uint32), e_d.view(np.uint32)), (tag, qi)


@pytest.mark.parametrize("metric,n,dim,nlist,nprobe,nq,k", [
    (IP, 120_000, 128, 256, 8, 256, 10),      # >= 64K rows, whole 128-column slabs, > 32 queries: the certified int8 pass in flight
    (IP, 50_000, 48, 64, 6, 40, 10),          # the f16 shadow
    (L2, 150_000, 256, 512, 12, 130, 10),     # augmented-L2 int8 codes
    (COS, 60_000, 64, 100, 5, 33, 7),
    (L2, 40_000, 32, 64, 4, 3, 5),            # a few queries (the blocking path answers these with its fused two-launch search)
])
def test_ivf_batches_in_flight_equal_blocking_and_oracle(L, oracle, metric, n, dim, nlist, nprobe, nq, k):
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n + dim + nq)
    data, cen, asg, off, rows, idx = _index(L, oracle, rng, n, dim, nlist, metric)
    batches = [(data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32) for _ in range(6)]
    dq = [torch.as_tensor(b, device=dev) for b in batches]
    blocking = []
    for q in dq:
        o = _tensors(torch, nq, k, dev)
        idx.search_device(q, k, nprobe, *o)
        blocking.append(_host(o))
    outs = [_tensors(torch, nq, k, dev) for _ in batches]
    # three contexts serve tickets (LYNSE_HIP_CONTEXTS = 4, context 0 stays with the blocking searches): three in flight at a time
    for base in (0, 3):
        tickets = [idx.search_submit(dq[i], k, nprobe, *outs[i]) for i in range(base, base + 3)]
        # a blocking search next to the tickets (context 0)
        o = _tensors(torch, nq, k, dev)
        idx.search_device(dq[base], k, nprobe, *o)
        assert _same(_host(o), blocking[base])
        for t in reversed(tickets):            # any order
            t.wait()
    assert idx.ticket_stats() == {"in_flight": 6, "inside_submit": 0, "redone_in_wait": 0}
    for i in range(6):
        got = _host(outs[i])
        assert _same(got, blocking[i]), i
        for qi in sorted({0, nq // 2, nq - 1}):
            _assert_oracle(oracle, batches[i][qi], data, cen, off, rows, nprobe, k, metric, got, qi, f"batch {i}")


def test_ivf_all_lists_empty_fallback_is_answered_inside_wait(L, oracle):
    """A query that probes only EMPTY lists: IVFIndex scans every list then (ivf.rs:258-265).  In flight the grouping kernel only flags
    it; wait() must notice and answer the batch with the blocking ladder."""
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31)
    n, dim, nlist, nq, k, nprobe = 30_000, 64, 40, 64, 10, 2
    data, cen, asg, off, rows, _ = _index(L, oracle, rng, n, dim, nlist, L2, nc=10)
    far = (200.0 + np.arange(3 * dim, dtype=f32)).reshape(3, dim)      # three centroids that own no row
    cen = np.concatenate([cen, far])
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    idx = L.IvfFlatIndex.load(data, cen, asg, "l2")
    qs = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    clean = qs.copy()
    qs[41] = far[1]
    outs = [_tensors(torch, nq, k, dev) for _ in range(3)]
    dqs = [torch.as_tensor(x, device=dev) for x in (clean, qs, clean)]
    tickets = [idx.search_submit(dq, k, nprobe, *o) for dq, o in zip(dqs, outs)]
    for t in tickets:
        t.wait()
    assert idx.ticket_stats() == {"in_flight": 3, "inside_submit": 0, "redone_in_wait": 1}
    got = [_host(o) for o in outs]
    assert _same(got[0], got[2])
    assert int(got[1][2][41]) == k        # every list was scanned for it
    for qi in (0, 40, 41, 42, nq - 1):
        _assert_oracle(oracle, qs[qi], data, cen, off, rows, nprobe, k, L2, got[1], qi, "with the far query")
        _assert_oracle(oracle, clean[qi], data, cen, off, rows, nprobe, k, L2, got[0], qi, "clean")


def test_ivf_int8_overflow_in_flight_is_rerun_inside_wait(L, oracle):
    """test_ivf_int8_pass_overflow_goes_back_to_the_f16_shadow, in flight: one row collapses an SQ8 scale, the int8 margin lets every
    probed row through, the candidate pool overflows — the ticket's status says so and wait() answers exactly."""
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5150)
    n, dim, nlist, nprobe, nq, k = 100_000, 128, 32, 8, 64, 10
    data = rng.standard_normal((n, dim)).astype(f32)
    data[4321, 9] = 1.0e4
    queries = rng.standard_normal((nq, dim)).astype(f32)
    cen, asg = oracle.kmeans_train(data[:4000], nlist, 3, IP)
    asg = np.concatenate([asg, rng.integers(0, cen.shape[0], n - 4000).astype(asg.dtype)])
    idx = L.IvfFlatIndex.load(data, cen, asg, "ip")
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    dq = torch.as_tensor(queries, device=dev)
    outs = [_tensors(torch, nq, k, dev) for _ in range(2)]
    tickets = [idx.search_submit(dq, k, nprobe, *o) for o in outs]
    for t in tickets:
        t.wait()
    assert idx.ticket_stats() == {"in_flight": 2, "inside_submit": 0, "redone_in_wait": 2}
    got = [_host(o) for o in outs]
    assert _same(got[0], got[1])
    for qi in (0, 33, 63):
        _assert_oracle(oracle, queries[qi], data, cen, off, rows, nprobe, k, IP, got[0], qi)


def test_ivf_shapes_the_grouping_does_not_take_and_the_rules(L, oracle):
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    n, dim, nlist, nq, k = 20_000, 32, 16, 50, 5
    data, cen, asg, off, rows, idx = _index(L, oracle, rng, n, dim, nlist, IP)
    q = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    dq = torch.as_tensor(q, device=dev)
    # nprobe >= nlist: every list — answered by the blocking ladder inside submit, the ticket is complete
    o = _tensors(torch, nq, k, dev)
    idx.search_submit(dq, k, nlist, *o).wait()
    assert idx.ticket_stats() == {"in_flight": 0, "inside_submit": 1, "redone_in_wait": 0}
    ob = _tensors(torch, nq, k, dev)
    idx.search_device(dq, k, nlist, *ob)
    assert _same(_host(o), _host(ob))
    _assert_oracle(oracle, q[7], data, cen, off, rows, nlist, k, IP, _host(o), 7)
    # more tickets than contexts 1..3: an error, not a deadlock; insert / delete are refused while tickets are outstanding
    import os

    n_ctx = max(1, min(8, int(os.environ.get("LYNSE_HIP_CONTEXTS", "4"))))   # (context 0 stays with the blocking searches)
    outs = [_tensors(torch, nq, k, dev) for _ in range(9)]
    tickets = []
    with pytest.raises(Exception, match="in flight"):
        for i in range(9):
            tickets.append(idx.search_submit(dq, k, 3, *outs[i]))
    assert len(tickets) == n_ctx - 1
    with pytest.raises(Exception, match="in flight"):
        idx.insert(data[:10])
    with pytest.raises(Exception, match="in flight"):
        idx.delete([1, 2])
    for t in tickets:
        t.wait()
    idx.insert(data[:10])                               # free again
    assert len(idx) == n + 10
    t = idx.search_submit(dq, k, 3, *outs[8])           # a new store behind the handle: derived data is rebuilt by the first submit
    t.wait()
    ob = _tensors(torch, nq, k, dev)
    idx.search_device(dq, k, 3, *ob)
    assert _same(_host(outs[8]), _host(ob))
    # more than 256 queries, k = 0
    with pytest.raises(Exception):
        idx.search_submit(torch.as_tensor(np.zeros((300, dim), f32), device=dev), k, 3, *_tensors(torch, 300, k, dev))
    # binary indexes keep to the blocking calls
    bits = (rng.random((4000, 64)) < 0.5).astype(f32)
    bidx = L.IvfFlatIndex.build(None, bits, 64, 8, 3, "hamming")
    with pytest.raises(Exception, match="float IVF"):
        bidx.search_submit(torch.as_tensor(bits[:4], device=dev), 3, 2, *_tensors(torch, 4, 3, dev))


def test_sharded_ivf_tickets_with_a_one_rank_communicator(L, oracle):
    """The exchange half of an IVF ticket (status word inside the result block, all-gather slot, merge kernel, pinned status) on the
    one GPU a test box has: a 1-rank RCCL communicator, global rows through the row map (stride 3, offset 1)."""
    import torch

    from lynsedb_amd.sharded import NativeComm, ShardedIvf

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(77)
    n, dim, nlist, nprobe, nq, k = 90_000, 128, 128, 6, 128, 10
    data, cen, asg, off, rows, _ = _index(L, oracle, rng, n, dim, nlist, IP)
    sh = ShardedIvf(dim, rank=0, world=1, device=0, group=None)
    sh.load_local(data, cen, asg, "ip")
    L._lib.check(L._lib.lib.lynse_hip_ivf_set_row_map(sh.index._h, 3, 1))
    sh.comm = NativeComm(None, 0, 1, 0)
    from lynsedb_amd.sharded import ShardOutputs

    batches = [(data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32) for _ in range(3)]
    dqs = [torch.as_tensor(b, device=dev) for b in batches]
    outs = [ShardOutputs(nq, k, 1, dev) for _ in batches]
    tickets = [sh.search_submit(dq, k, nprobe, o) for dq, o in zip(dqs, outs)]
    for t in tickets:
        t.wait()
    assert sh.index.ticket_stats() == {"in_flight": 3, "inside_submit": 0, "redone_in_wait": 0}
    o2 = ShardOutputs(nq, k, 1, dev)
    sh.search_device(dqs[0], k, nprobe, o2)            # the blocking sharded entry point next to it (its own block pair)
    torch.cuda.synchronize()
    assert torch.equal(o2.rows, outs[0].rows) and torch.equal(o2.dists, outs[0].dists) and torch.equal(o2.counts, outs[0].counts)
    for b, o in zip(batches, outs):
        r, d, c = o.rows.cpu().numpy().view(np.uint64), o.dists.cpu().numpy(), o.counts.cpu().numpy()
        for qi in (0, 63, nq - 1):
            e_ids, e_d, _ = oracle.ivf_search(b[qi], data, cen, off, rows, nprobe, k, IP)
            assert int(c[qi]) == len(e_ids) and np.array_equal(r[qi, :len(e_ids)], e_ids.astype(np.uint64) * 3 + 1), qi
            assert np.array_equal(d[qi, :len(e_ids)].view(np.uint32), e_d.view(np.uint32)), qi


def test_sharded_ivf_ticket_with_an_all_lists_empty_query_is_answered_again_not_failed(L, oracle):
    """The status word of a result block carries TWO kinds of flags (csrc/lynse_hip.hip, STATUS_*): the low byte asks for the batch to be
    answered again — bit 0 a candidate overflow, bit 1 an IVF query whose probed lists are all empty (ivf.rs:258-265) — and only bits above
    it mean that a rank failed.  Round 5 first read bit 1 as a failure: every such ticket through a communicator raised (found by
    scripts/stress_ivf_inflight.py with STRESS_COMM=1, not by a test — this is that test)."""
    import torch

    from lynsedb_amd.sharded import NativeComm, ShardedIvf, ShardOutputs

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(32)
    n, dim, nlist, nq, k, nprobe = 30_000, 64, 40, 64, 10, 2
    data, cen, asg, off, rows, _ = _index(L, oracle, rng, n, dim, nlist, L2, nc=10)
    far = (200.0 + np.arange(3 * dim, dtype=f32)).reshape(3, dim)      # three centroids that own no row
    cen = np.concatenate([cen, far])
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    sh = ShardedIvf(dim, rank=0, world=1, device=0, group=None)
    sh.load_local(data, cen, asg, "l2")
    sh.comm = NativeComm(None, 0, 1, 0)
    qs = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    clean = qs.copy()
    qs[41] = far[1]
    dqs = [torch.as_tensor(x, device=dev) for x in (clean, qs, clean)]
    outs = [ShardOutputs(nq, k, 1, dev) for _ in dqs]
    tickets = [sh.search_submit(dq, k, nprobe, o) for dq, o in zip(dqs, outs)]
    for t in tickets:
        t.wait()                      # (raised LynseHipError "another rank ... failed" before the fix)
    assert sh.index.ticket_stats() == {"in_flight": 3, "inside_submit": 0, "redone_in_wait": 1}
    torch.cuda.synchronize()
    assert torch.equal(outs[0].rows, outs[2].rows) and torch.equal(outs[0].dists, outs[2].dists)
    c1 = outs[1].counts.cpu().numpy()
    assert int(c1[41]) == k          # every list was scanned for it
    r, d = outs[1].rows.cpu().numpy().view(np.uint64), outs[1].dists.cpu().numpy()
    for qi in (0, 41, nq - 1):
        e_ids, e_d, _ = oracle.ivf_search(qs[qi], data, cen, off, rows, nprobe, k, L2)
        assert int(c1[qi]) == len(e_ids) and np.array_equal(r[qi, :len(e_ids)], e_ids.astype(np.uint64)), qi
        assert np.array_equal(d[qi, :len(e_ids)].view(np.uint32), e_d.view(np.uint32)), qi


def test_ivf_tickets_from_three_threads(L, oracle):
    """Three submitting threads, one ticket each at a time (contexts 1..3), next to a fourth thread running blocking searches: every
    answer equals the blocking call's."""
    import threading

    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99)
    n, dim, nlist, nprobe, nq, k = 80_000, 128, 128, 6, 96, 10
    data, cen, asg, off, rows, idx = _index(L, oracle, rng, n, dim, nlist, IP)
    batches = [(data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32) for _ in range(12)]
    dq = [torch.as_tensor(b, device=dev) for b in batches]
    want = []
    for q in dq:
        o = _tensors(torch, nq, k, dev)
        idx.search_device(q, k, nprobe, *o)
        want.append(_host(o))
    got = [None] * len(batches)
    errs = []

    def submitter(tid):
        try:
            torch.cuda.set_device(0)
            for rep in range(3):
                for i in range(tid, len(batches), 3):
                    o = _tensors(torch, nq, k, dev)
                    torch.cuda.synchronize()
                    idx.search_submit(dq[i], k, nprobe, *o).wait()
                    got[i] = _host(o)
        except Exception as e:  # noqa: BLE001
            errs.append((tid, repr(e)))

    def blocker():
        try:
            torch.cuda.set_device(0)
            for rep in range(20):
                o = _tensors(torch, nq, k, dev)
                idx.search_device(dq[rep % len(dq)], k, nprobe, *o)
                assert _same(_host(o), want[rep % len(dq)])
        except Exception as e:  # noqa: BLE001
            errs.append(("blocking", repr(e)))

    ts = [threading.Thread(target=submitter, args=(t,)) for t in range(3)] + [threading.Thread(target=blocker)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs, errs
    for i in range(len(batches)):
        assert _same(got[i], want[i]), i
    st = idx.ticket_stats()
    assert st["in_flight"] == 36 and st["inside_submit"] == 0, st


def test_a_local_failure_of_a_sharded_ivf_submit_rides_the_exchange(L, oracle, monkeypatch):
    """The IVF twin of tests/test_gpu_inflight.py::test_a_local_failure_of_a_sharded_submit_rides_the_exchange: a local failure
    of the preparation of a sharded IVF submit (LYNSE_HIP_DEBUG_FAIL_SUBMIT=1 simulates one) does not end the call in front of the all-gather: the
    ticket's exchange runs with an empty block + the failure bit, `wait` reports the error, the next tickets are answered like the oracle."""
    import torch

    from lynsedb_amd.sharded import NativeComm, ShardedIvf, ShardOutputs

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(33)
    n, dim, nlist, nq, k, nprobe = 30_000, 64, 40, 64, 10, 4
    data, cen, asg, off, rows, _ = _index(L, oracle, rng, n, dim, nlist, L2, nc=10)
    sh = ShardedIvf(dim, rank=0, world=1, device=0, group=None)
    sh.load_local(data, cen, asg, "l2")
    sh.comm = NativeComm(None, 0, 1, 0)
    qs = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    dq = torch.as_tensor(qs, device=dev)
    sh.search_submit(dq, k, nprobe, ShardOutputs(nq, k, 1, dev)).wait()      # (derived data built; a first clean ticket)
    monkeypatch.setenv("LYNSE_HIP_DEBUG_FAIL_SUBMIT", "1")
    bad = sh.search_submit(dq, k, nprobe, ShardOutputs(nq, k, 1, dev))       # no exception here
    with pytest.raises(MemoryError, match="simulated local failure"):
        bad.wait()
    monkeypatch.delenv("LYNSE_HIP_DEBUG_FAIL_SUBMIT")
    out = ShardOutputs(nq, k, 1, dev)
    for _ in range(9):                                                       # more tickets than contexts, one after the other: every context came back
        sh.search_submit(dq, k, nprobe, out).wait()
    torch.cuda.synchronize()
    r, d, c = out.rows.cpu().numpy().view(np.uint64), out.dists.cpu().numpy(), out.counts.cpu().numpy()
    for qi in (0, nq - 1):
        e_ids, e_d, _ = oracle.ivf_search(qs[qi], data, cen, off, rows, nprobe, k, L2)
        assert int(c[qi]) == len(e_ids) and np.array_equal(r[qi, :len(e_ids)], e_ids.astype(np.uint64)), qi
        assert np.array_equal(d[qi, :len(e_ids)].view(np.uint32), e_d.view(np.uint32)), qi
