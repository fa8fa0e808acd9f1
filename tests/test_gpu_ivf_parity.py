"""GPU parity tests for IVF-Flat: HIP path (C ABI) vs the CPU oracle's restatement of IVFIndex
(src/index/ivf.rs) and IvfFlatMmap (src/storage/ivf_flat_mmap.rs).

Parity definition: the same centroids + assignments in -> the same probes, candidates
and results out, bit-exact ids and distances.  The device k-means is additionally compared with the
oracle's k-means where the reference itself is deterministic (n < 8192, sequential sums)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
IP, L2, COS = O.IP, O.L2, O.COS
NAME = {IP: "ip", L2: "l2", COS: "cosine"}
f32 = np.float32


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1
    return L_


def oracle_ivf(oracle, data, nlist, metric, iters=20, train_metric=None):
    cen, asg = oracle.kmeans_train(data, nlist, iters, metric if train_metric is None else train_metric)
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    return cen, asg, off, rows


def check_ivfindex(L, oracle, data, queries, nlist, nprobe, k, metric):
    cen, asg, off, rows = oracle_ivf(oracle, data, nlist, metric)
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
    assert len(idx) == data.shape[0] and idx.n_partitions == cen.shape[0]
    g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
    for qi in range(queries.shape[0]):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows, nprobe, k, metric)
        c = int(g_c[qi])
        assert c == len(e_ids), (qi, c, len(e_ids))
        assert np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), (qi, g_d[qi, :c], e_d)
        assert np.array_equal(g_rows[qi, :c], e_ids), (qi, g_rows[qi, :c], e_ids)


@pytest.mark.parametrize("metric", [IP, L2, COS])
@pytest.mark.parametrize("n,dim,nlist,nprobe,nq,k", [
    (800, 32, 32, 2, 4, 10), (800, 32, 32, 32, 3, 10), (3000, 20, 16, 4, 40, 5), (6000, 100, 64, 10, 9, 10),
    (5000, 48, 300, 7, 70, 10), (400, 8, 8, 1, 2, 25),
])
def test_ivfindex_parity(L, oracle, metric, n, dim, nlist, nprobe, nq, k):
    rng = np.random.default_rng(n + dim + nlist)
    centers = rng.standard_normal((max(nlist // 2, 2), dim)).astype(f32)
    data = (centers[rng.integers(0, centers.shape[0], n)] + 0.3 * rng.standard_normal((n, dim))).astype(f32)
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    check_ivfindex(L, oracle, data, queries, nlist, nprobe, k, metric)


def test_ivf_reference_kats(L, oracle):
    # ivf.rs:578-638 — n=800, D=32 generator, nlist=32: nprobe=32 => exact; recall monotone in nprobe
    n, dim = 800, 32
    i = np.arange(n)[:, None]
    j = np.arange(dim)[None, :]
    data = (((i * 131 + j * 17 + 1) % 997).astype(f32) / f32(997.0) + f32(0.01)).astype(f32)
    cen, asg, off, rows = oracle_ivf(oracle, data, 32, IP)
    idx = L.IvfFlatIndex.load(data, cen, asg, "ip")
    q = data[0]
    flat_ids, _ = oracle.canonical_topk(q, data, 10, IP, O.IPFORM_SINGLE)
    hi, _ = idx.search(q, 10, 32, "ip")
    lo, _ = idx.search(q, 10, 2, "ip")
    assert set(hi.tolist()) == set(flat_ids.tolist())
    assert len(set(lo.tolist()) & set(flat_ids.tolist())) <= 10
    # ivf.rs:546-575 (unfiltered part): two well separated clusters, nprobe=1 from the near one
    d2 = np.array([[0, 0], [0.1, 0], [10, 10], [10.1, 10]], f32)
    cen, asg, off, rows = oracle_ivf(oracle, d2, 2, L2)
    idx = L.IvfFlatIndex.load(d2, cen, asg, "l2")
    ids, _ = idx.search(np.zeros(2, f32), 2, 1, "l2")
    assert sorted(ids.tolist()) == [0, 1]
    ids0, d0 = idx.search(np.zeros(2, f32), 2, 0, "l2")  # nprobe 0 == 1 (ivf.rs:192-196)
    assert np.array_equal(ids0, ids)


@pytest.mark.parametrize("n,dim,nlist,nprobe", [(12, 4, 3, 1), (1000, 8, 10, 10), (4000, 64, 64, 5), (6000, 96, 128, 12), (4, 2, 2, 2)])
def test_ivfflat_parity_and_heuristic_routing(L, oracle, n, dim, nlist, nprobe):
    """IvfFlatMmap semantics: L2-trained partitions, IP search, 16-dim shortlist routing when
    D >= 64 and nlist >= 64 (ivf_flat_mmap.rs:381-421)."""
    rng = np.random.default_rng(n * 7 + dim)
    data = rng.standard_normal((n, dim)).astype(f32)
    cen, asg = oracle.kmeans_train(data, nlist, 10, L2)
    offsets, orig = oracle.ivf_flat_layout(asg, cen.shape[0])
    slab = data[orig]
    rd = oracle.ivf_routing_dims(cen)
    idx = L.IvfFlatIndex.load(data, cen, asg, "ip", ivfflat_routing=True)
    c2, a2, o2, orig2 = idx.export()
    assert np.array_equal(o2, offsets) and np.array_equal(orig2, orig) and np.array_equal(a2, asg)
    for qi in range(6):
        q = (data[rng.integers(0, n)] + 0.1 * rng.standard_normal(dim)).astype(f32)
        e_ids, e_d = oracle.ivf_flat_search(q, slab, cen, offsets, orig, nprobe, 5, IP, rd)
        g_ids, g_d = idx.search(q, 5, nprobe, "ip")
        assert np.array_equal(g_d.view(np.uint32), e_d.view(np.uint32)), (qi, g_d, e_d)
        assert np.array_equal(g_ids, e_ids), (qi, g_ids, e_ids)


@pytest.mark.parametrize("metric,l2p", [(IP, False), (L2, False), (COS, False), (IP, True)])
def test_device_kmeans_matches_oracle(L, oracle, metric, l2p):
    """lynse_hip_ivf_build vs kmeans::train_for_metric (kmeans.rs:74-139) restated by the oracle:
    identical FastRng(42) init, assignments and (sequential-order) centroid sums -> bit-identical."""
    rng = np.random.default_rng(17)
    n, dim, nlist = 3000, 24, 40
    data = rng.random((n, dim), dtype=f32) + (rng.integers(0, 5, size=(n, 1)) * 0.7).astype(f32)
    idx = L.IvfFlatIndex.build(None, data, dim, nlist, 20, NAME[metric], l2_partitions=l2p)
    cen, asg, off, orig = idx.export()
    e_cen, e_asg = oracle.kmeans_train(data, nlist, 20, L2 if l2p else metric)
    assert cen.shape == e_cen.shape
    assert np.array_equal(asg, e_asg)
    assert np.array_equal(cen.view(np.uint32), e_cen.view(np.uint32))
    q = data[5]
    e_off, e_rows = oracle.lists_from_assignments(e_asg, e_cen.shape[0])
    if not l2p:
        e_ids, e_d, _ = oracle.ivf_search(q, data, e_cen, e_off, e_rows, 6, 10, metric)
        g_ids, g_d = idx.search(q, 10, 6, NAME[metric])
        assert np.array_equal(g_ids.astype(np.uint64), e_ids) and np.array_equal(g_d, e_d)


@pytest.mark.parametrize("n,dim,nlist,iters,metric,device_rows", [
    (50_000, 64, 256, 4, IP, False),      # assignment in wide passes (8192 rows per launch, several query chunks per launch)
    (50_000, 64, 256, 3, L2, True),       # ... and the device-resident twin (rows never leave HBM)
    (20_000, 32, 5000, 3, L2, False),     # more than 4096 centroids: the centroid store is past the single / batch-8 boundary of the
    (20_000, 32, 5000, 2, IP, False),     # IP kernels (flat_mmap.rs:4852-4854) and past one 4096-row plan stage
    (9_000, 40, 4097, 2, COS, False),     # nlist just past the boundary, rows ~ 2 per list: empty-cluster reseeding in play
])
def test_device_kmeans_matches_oracle_at_scale(L, oracle, n, dim, nlist, iters, metric, device_rows):
    """The device k-means (csrc/ivf_host.inc: FastRng(42) + farthest-point init, assignment = FLAT k=1 search against the
    centroid matrix in wide passes, sequential-order centroid sums) against the oracle's kmeans::train_for_metric
    (kmeans.rs:74-139, :237-315) where the wide-pass assignment and > 4096 centroids are actually reached (VERDICT r2 weak #2:
    the n = 3000 case above never leaves one launch): centroid bits and assignments identical."""
    rng = np.random.default_rng(1000 + n + nlist)
    centers = rng.standard_normal((max(nlist // 3, 8), dim)).astype(f32)
    data = (centers[rng.integers(0, len(centers), n)] + 0.35 * rng.standard_normal((n, dim))).astype(f32)
    if device_rows:
        import torch

        idx = L.IvfFlatIndex.build_device(torch.as_tensor(data, device="cuda:0"), dim, nlist, iters, NAME[metric], l2_partitions=False)
    else:
        idx = L.IvfFlatIndex.build(None, data, dim, nlist, iters, NAME[metric], l2_partitions=False)
    cen, asg, off, orig = idx.export()
    e_cen, e_asg = oracle.kmeans_train(data, nlist, iters, metric)
    assert cen.shape == e_cen.shape, (cen.shape, e_cen.shape)
    assert np.array_equal(asg, e_asg), int((asg != e_asg).sum())
    assert np.array_equal(cen.view(np.uint32), e_cen.view(np.uint32)), int((cen.view(np.uint32) != e_cen.view(np.uint32)).sum())
    # and a search over the index built from them equals the oracle's IVFIndex::search on its own centroids / lists
    e_off, e_rows = oracle.lists_from_assignments(e_asg, e_cen.shape[0])
    for qi in (0, n // 2):
        q = (data[qi] + 0.05 * rng.standard_normal(dim)).astype(f32)
        e_ids, e_d, _ = oracle.ivf_search(q, data, e_cen, e_off, e_rows, 8, 10, metric)
        g_ids, g_d = idx.search(q, 10, 8, NAME[metric])
        assert np.array_equal(g_ids.astype(np.uint64), e_ids) and np.array_equal(g_d.view(np.uint32), e_d.view(np.uint32)), (qi, g_ids, e_ids)


@pytest.mark.parametrize("metric", [IP, L2, COS])
def test_kmeans_assignment_with_the_arg_best_epilogue_equals_the_search_path(L, oracle, metric, monkeypatch):
    """kmeans::assign_metric (kmeans.rs:237-264) through the lane-max scan + k_assign_pick against the same
    build with every row going through the exact top-1 search (LYNSE_HIP_ASSIGN_FAST=0) and against the oracle — on data made of
    NEAR-TIES: 40 clusters under 200 lists (sibling centroids of a cluster score within the f16 margin of each other for most rows: the
    in-margin candidates are rescored exactly inside the pick kernel), duplicated rows, and integer-valued rows whose inner products
    tie exactly between candidates (the first-strictly-smaller rule = lowest centroid id must win)."""
    rng = np.random.default_rng(7700 + metric)
    n, dim, nlist, iters = 30_000, 96, 200, 4
    centers = rng.standard_normal((40, dim)).astype(f32)
    data = (centers[rng.integers(0, 40, n)] + 0.25 * rng.standard_normal((n, dim))).astype(f32)
    data[5000:7000] = data[4999]                                   # 2000 copies of one row
    data[9000:12000] = rng.integers(-2, 3, size=(3000, dim)).astype(f32)   # small integers: exact ties between candidates
    built = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("LYNSE_HIP_ASSIGN_FAST", fast)
        idx = L.IvfFlatIndex.build(None, data, dim, nlist, iters, NAME[metric], l2_partitions=False)
        built[fast] = idx.export()[:2]
        del idx
    assert np.array_equal(built["1"][1], built["0"][1]), int((built["1"][1] != built["0"][1]).sum())
    assert np.array_equal(built["1"][0].view(np.uint32), built["0"][0].view(np.uint32))
    e_cen, e_asg = oracle.kmeans_train(data, nlist, iters, metric)
    assert np.array_equal(built["1"][1], e_asg), int((built["1"][1] != e_asg).sum())
    assert np.array_equal(built["1"][0].view(np.uint32), e_cen.view(np.uint32))


@pytest.mark.parametrize("metric", [IP, L2, COS])
def test_ivfindex_insert_and_delete_keep_the_centroids(L, oracle, metric):
    """IVFIndex::insert / ::delete (ivf.rs:350-441): build -> add rows -> search -> delete rows -> search.  New rows are
    assigned to the EXISTING centroids (kmeans::assign_metric's loop, restated by oracle.kmeans_assign) and appended; a
    delete renumbers the remaining rows in order and reassigns all of them; the centroids never move.  Assignments and
    search results equal the oracle's IVFIndex::search over the same centroids and the lists those rules give."""
    rng = np.random.default_rng(4100 + metric)
    n0, n1, dim, nlist, nprobe, k = 6000, 1500, 48, 40, 5, 10
    centers = rng.standard_normal((12, dim)).astype(f32)
    mk = lambda m: (centers[rng.integers(0, 12, m)] + 0.3 * rng.standard_normal((m, dim))).astype(f32)  # noqa: E731
    base, extra = mk(n0), mk(n1)
    idx = L.IvfFlatIndex.build(None, base, dim, nlist, 6, NAME[metric], l2_partitions=False)
    cen0, asg0, _, _ = idx.export()
    e_new = oracle.kmeans_assign(extra, cen0, metric)
    assert np.array_equal(idx.assign(extra), e_new)
    idx.insert(extra)
    cen1, asg1, off1, orig1 = idx.export()
    assert len(idx) == n0 + n1 and np.array_equal(cen1.view(np.uint32), cen0.view(np.uint32))
    assert np.array_equal(asg1, np.concatenate([asg0, e_new]))
    data = np.concatenate([base, extra])
    e_off, e_rows = oracle.lists_from_assignments(asg1, cen0.shape[0])
    assert np.array_equal(off1, e_off) and np.array_equal(orig1, e_rows)      # new rows sit at the END of their lists
    queries = (data[rng.integers(0, n0 + n1, 12)] + 0.05 * rng.standard_normal((12, dim))).astype(f32)
    g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
    for qi in range(len(queries)):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen0, e_off, e_rows, nprobe, k, metric)
        c = int(g_c[qi])
        assert c == len(e_ids) and np.array_equal(g_rows[qi, :c], e_ids) and np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), qi
    # delete: every third of the first 3000 rows + a few of the inserted ones + ids that do not exist
    gone = np.concatenate([np.arange(0, 3000, 3), n0 + np.arange(0, 200, 7), [10**9]]).astype(np.uint64)
    idx.delete(gone)
    keep = np.setdiff1d(np.arange(n0 + n1), gone[gone < n0 + n1].astype(np.int64))
    left = data[keep]
    cen2, asg2, off2, orig2 = idx.export()
    assert len(idx) == len(keep) and np.array_equal(cen2.view(np.uint32), cen0.view(np.uint32))
    e_asg2 = oracle.kmeans_assign(left, cen0, metric)
    assert np.array_equal(asg2, e_asg2)
    e_off2, e_rows2 = oracle.lists_from_assignments(e_asg2, cen0.shape[0])
    g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
    for qi in range(len(queries)):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], left, cen0, e_off2, e_rows2, nprobe, k, metric)
        c = int(g_c[qi])
        assert c == len(e_ids) and np.array_equal(g_rows[qi, :c], e_ids) and np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), qi


def test_ivf_binary_insert_goes_through_the_quantizer(L, oracle):
    """IVF-HAMMING-BINARY: inserted rows are encoded with the thresholds FITTED AT BUILD (quantizer.encode -> decode,
    ivf.rs:398-404), routed with L2 on the codes, and searched by popcount like the rest."""
    rng = np.random.default_rng(77)
    n0, n1, dim, nlist, nprobe, k = 4000, 900, 96, 24, 6, 10
    base = (rng.standard_normal((n0, dim)) * 2 + 1).astype(f32)
    extra = (rng.standard_normal((n1, dim)) * 2 + 1).astype(f32)
    idx = L.IvfFlatIndex.build(None, base, dim, nlist, 8, "hamming", l2_partitions=False)
    thr, ab = idx.thresholds()
    cen, asg0, _, _ = idx.export()
    enc_new = oracle.binary_quantize(extra, thr)
    e_new = oracle.kmeans_assign(enc_new, cen, O.L2)
    idx.insert(extra)
    _, asg1, off1, orig1 = idx.export()
    assert np.array_equal(asg1, np.concatenate([asg0, e_new]))
    enc = np.concatenate([oracle.binary_quantize(base, thr), enc_new])
    packed = oracle.pack_binary(enc)
    e_off, e_rows = oracle.lists_from_assignments(asg1, cen.shape[0])
    queries = np.concatenate([base[:5], extra[:5]]).astype(f32)
    g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
    for qi in range(len(queries)):
        eq = oracle.binary_quantize(queries[qi], thr)[0]
        e_ids, e_d, _ = oracle.ivf_search(eq, enc, cen, e_off, e_rows, nprobe, k, O.HAMMING, packed=packed)
        c = int(g_c[qi])
        assert c == len(e_ids) and np.array_equal(g_rows[qi, :c], e_ids) and np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), qi


def test_ivf_build_recall_and_errors(L, oracle):
    rng = np.random.default_rng(3)
    n, dim = 20000, 64
    centers = rng.standard_normal((50, dim)).astype(f32)
    data = (centers[np.arange(n) % 50] + 0.2 * rng.standard_normal((n, dim))).astype(f32)
    idx = L.IvfFlatIndex.build(None, data, dim, 64, 10, "l2")
    qs = (data[:20] + 0.01).astype(f32)
    rows, d, c = idx.search_batch_arrays(qs, 10, 8)
    hit = 0
    for i in range(20):
        e_ids, _ = oracle.canonical_topk(qs[i], data, 10, L2)
        hit += len(set(rows[i, :c[i]].tolist()) & set(e_ids.tolist()))
    assert hit / 200 >= 0.85  # reference gate floor for IVF (benchmarks/gate_index_modes.py:271)
    rows_all, d_all, c_all = idx.search_batch_arrays(qs[:3], 10, 64)  # all partitions -> exact
    for i in range(3):
        e_ids, e_d = oracle.canonical_topk(qs[i], data, 10, L2)
        assert np.array_equal(rows_all[i].astype(np.uint32), e_ids) and np.array_equal(d_all[i], e_d)
    with pytest.raises(IOError):
        L.IvfFlatIndex.build(None, data[:5], dim, 8)  # fewer vectors than partitions
    with pytest.raises(ValueError):
        idx.search(np.zeros(dim + 1, f32), 5, 4, "l2")
    with pytest.raises(ValueError, match="Unknown metric"):
        idx.search(np.zeros(dim, f32), 5, 4, "nope")


# ------------------------------------------------------------------------- IVF-*-BINARY (ivf.rs:77-127)
BIN_NAME = {O.HAMMING: "hamming", O.JACCARD: "jaccard", O.DICE: "dice", O.TANIMOTO: "tanimoto"}


def check_ivf_binary(L, oracle, data, queries, nlist, nprobe, k, metric, build_on_device):
    ab, thr = oracle.binary_fit(data)
    enc = oracle.binary_quantize(data, thr)
    packed = oracle.pack_binary(enc)
    if build_on_device:
        idx = L.IvfFlatIndex.build(None, data, data.shape[1], nlist, 20, BIN_NAME[metric], l2_partitions=False)
        g_thr, g_ab = idx.thresholds()
        assert g_ab == ab
        assert np.array_equal(g_thr.view(np.uint32), thr.view(np.uint32)), (g_thr, thr)
        cen, asg, off, orig = idx.export()
        rows = orig
        if data.shape[0] < 8192:  # deterministic k-means (sequential sums): bit-identical to the oracle's on the codes
            e_cen, e_asg = oracle.kmeans_train(enc, nlist, 20, O.L2)
            assert np.array_equal(e_asg, asg)
            assert np.array_equal(e_cen.view(np.uint32), cen.view(np.uint32))
    else:
        cen, asg = oracle.kmeans_train(enc, nlist, 20, O.L2)
        off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
        idx = L.IvfFlatIndex.load(data, cen, asg, BIN_NAME[metric], thresholds=thr)
    g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
    for qi in range(queries.shape[0]):
        eq = oracle.binary_quantize(queries[qi], thr)[0]
        e_ids, e_d, _ = oracle.ivf_search(eq, enc, cen, off, rows, nprobe, k, metric, packed=packed)
        c = int(g_c[qi])
        assert c == len(e_ids), (qi, c, len(e_ids))
        assert np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), (qi, g_d[qi, :c], e_d)
        assert np.array_equal(g_rows[qi, :c], e_ids), (qi, g_rows[qi, :c], e_ids)


@pytest.mark.parametrize("metric", [O.HAMMING, O.JACCARD, O.DICE])
@pytest.mark.parametrize("n,dim,nlist,nprobe,nq,k,kind", [
    (2000, 64, 16, 4, 8, 10, "float"), (3000, 100, 32, 32, 5, 10, "float"), (5000, 256, 64, 8, 40, 20, "bits"),
    (1500, 33, 8, 2, 3, 50, "float"), (6000, 1024, 32, 6, 33, 10, "bits"), (700, 130, 200, 9, 4, 10, "skew"),
])
def test_ivf_binary_parity_loaded(L, oracle, metric, n, dim, nlist, nprobe, nq, k, kind):
    rng = np.random.default_rng(n * 3 + dim + nlist)
    if kind == "bits":
        data = (rng.random((n, dim)) < 0.3).astype(f32)
        queries = data[rng.integers(0, n, nq)].copy()
        flip = rng.random(queries.shape) < 0.05
        queries = np.where(flip, 1 - queries, queries).astype(f32)
    elif kind == "skew":  # many constant / two-valued columns: median == min or max -> midrange fallback
        data = rng.integers(0, 3, (n, dim)).astype(f32)
        data[:, ::4] = 7.0
        data[:, 1::4] = (rng.random((n, dim))[:, 1::4] < 0.9).astype(f32) * 2
        queries = data[rng.integers(0, n, nq)].copy()
    else:
        data = rng.standard_normal((n, dim)).astype(f32) * 3 + rng.standard_normal(dim).astype(f32)
        queries = (data[rng.integers(0, n, nq)] + 0.5 * rng.standard_normal((nq, dim))).astype(f32)
    check_ivf_binary(L, oracle, data, queries, nlist, nprobe, k, metric, build_on_device=False)


@pytest.mark.parametrize("metric,n,dim,nlist,kind", [(O.HAMMING, 3000, 96, 24, "float"), (O.JACCARD, 2500, 128, 16, "bits"),
                                                      (O.HAMMING, 20000, 64, 32, "float")])
def test_ivf_binary_build_on_device(L, oracle, metric, n, dim, nlist, kind):
    rng = np.random.default_rng(n + dim)
    if kind == "bits":
        data = (rng.random((n, dim)) < 0.4).astype(f32)
    else:
        data = (rng.standard_normal((n, dim)) * 2 + 1).astype(f32)
    queries = data[rng.integers(0, n, 12)].copy()
    check_ivf_binary(L, oracle, data, queries, nlist, 5, 10, metric, build_on_device=True)


def test_ivf_hamming_binary_reference_kat(L, oracle):
    # ivf.rs:641-679: n=256, dim=32, bit (i*17 + j*3) % 2 == 0, IVF(16 lists), nprobe=16 -> the flat Hamming distances
    n, dim = 256, 32
    i = np.arange(n)[:, None]
    j = np.arange(dim)[None, :]
    vectors = (((i * 17 + j * 3) % 2) == 0).astype(f32)
    idx = L.IvfFlatIndex.build(None, vectors, dim, 16, 20, "hamming", l2_partitions=False)
    thr, ab = idx.thresholds()
    assert ab and np.all(thr == 0.5)
    q = vectors[0]
    exact = np.sort(np.array([oracle.compute_distance(q, vectors[r], O.HAMMING) for r in range(n)], f32), kind="stable")[:10]
    ids, d = idx.search(q, 10, 16, "hamming")
    assert np.array_equal(d, exact)
    # ties are ordered by original row id (canonical order): the 10 smallest ids among the zero-distance rows
    zero = np.nonzero(np.array([oracle.compute_distance(q, vectors[r], O.HAMMING) for r in range(n)]) == 0)[0]
    assert np.array_equal(ids, zero[:10].astype(np.uint32))


def test_ivf_binary_many_ties_and_large_lists(L, oracle):
    # few distinct codes -> huge tie groups at the k-th distance; exercises the non-strict cut + overflow plan
    rng = np.random.default_rng(5)
    protos = (rng.random((12, 64)) < 0.5).astype(f32)
    data = protos[rng.integers(0, 12, 30000)]
    queries = protos[:6].copy()
    queries[:, :3] = 1 - queries[:, :3]
    check_ivf_binary(L, oracle, data, queries, 8, 8, 25, O.HAMMING, build_on_device=False)


# ---------------------------------------------------------------- SearchParams.subset (ivf.rs:251-265)
@pytest.mark.parametrize("metric", [IP, L2, COS, O.HAMMING])
@pytest.mark.parametrize("n,dim,nlist,nprobe,nq,k,frac", [
    (3000, 24, 16, 3, 10, 10, 0.3), (6000, 64, 64, 8, 33, 10, 0.05), (5000, 32, 32, 2, 6, 20, 0.002), (2000, 16, 8, 8, 4, 5, 0.9),
])
def test_ivf_filtered_parity(L, oracle, metric, n, dim, nlist, nprobe, nq, k, frac):
    rng = np.random.default_rng(n + dim + nlist + int(frac * 1000))
    binary = metric == O.HAMMING
    if binary:
        data = (rng.random((n, dim)) < 0.35).astype(f32)
        queries = data[rng.integers(0, n, nq)].copy()
        thr = np.full(dim, 0.5, f32)
        enc = data
        packed = oracle.pack_binary(enc)
        cen, asg = oracle.kmeans_train(enc, nlist, 20, O.L2)
        off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
        idx = L.IvfFlatIndex.load(data, cen, asg, "hamming", thresholds=thr)
    else:
        centers = rng.standard_normal((max(nlist // 2, 2), dim)).astype(f32)
        data = (centers[rng.integers(0, centers.shape[0], n)] + 0.3 * rng.standard_normal((n, dim))).astype(f32)
        queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
        cen, asg, off, rows = oracle_ivf(oracle, data, nlist, metric)
        idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
        packed = None
    m = max(1, int(n * frac))
    subset = np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
    g_rows, g_d, g_c = idx.search_filtered_batch_arrays(queries, k, nprobe, subset)
    fell_back = 0
    for qi in range(nq):
        e_ids, e_d = oracle.ivf_search_filtered(queries[qi], data, cen, off, rows, nprobe, k, metric, subset, packed=packed)
        c = int(g_c[qi])
        assert c == len(e_ids), (qi, c, len(e_ids))
        assert np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), (qi, g_d[qi, :c], e_d)
        assert np.array_equal(g_rows[qi, :c], e_ids), (qi, g_rows[qi, :c], e_ids)
        assert np.isin(g_rows[qi, :c], subset).all()
    # empty subset / all-invalid subset -> no results
    _, _, c0 = idx.search_filtered_batch_arrays(queries[:2], k, nprobe, [])
    assert c0.tolist() == [0, 0]
    _, _, c1 = idx.search_filtered_batch_arrays(queries[:2], k, nprobe, [n + 5])
    assert c1.tolist() == [0, 0]


@pytest.mark.parametrize("metric", [IP, L2])
@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_ivf_equals_single_index(L, oracle, metric, world):
    # BASELINE config 4 in miniature: the shards of lynsedb_amd.sharded.ShardedIvf (all in this process, one GPU)
    # searched separately + the k-way merge of the exchange step == the unsharded index == the oracle
    from lynsedb_amd.sharded import ShardedIvf

    rng = np.random.default_rng(world * 7 + metric)
    n, dim, nlist, nprobe, nq, k = 6000, 32, 48, 5, 12, 10
    centers = rng.standard_normal((20, dim)).astype(f32)
    data = (centers[rng.integers(0, 20, n)] + 0.3 * rng.standard_normal((n, dim))).astype(f32)
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    cen, asg, off, rows = oracle_ivf(oracle, data, nlist, metric)
    # device-side assignment (kmeans::assign_metric) agrees with the oracle's
    assert np.array_equal(ShardedIvf.assign(data, cen, NAME[metric]), oracle.kmeans_assign(data, cen, metric))
    shards = []
    for r in range(world):
        s = ShardedIvf(dim, rank=r, world=world, device=0)
        s.load_global(data, cen, asg, NAME[metric])
        shards.append(s)
    parts = [s.search_local(queries, k, nprobe) for s in shards]
    for qi in range(nq):
        cand_r = np.full((world, k), np.iinfo(np.uint64).max, np.uint64)
        cand_d = np.zeros((world, k), f32)
        cnt = np.zeros(world, np.uint32)
        for r, (pr, pd, pc) in enumerate(parts):
            c = int(pc[qi])
            cand_r[r, :c], cand_d[r, :c], cnt[r] = pr[qi, :c], pd[qi, :c], c
        ids, d = L.merge_topk(cand_r, cand_d, cnt, k, metric)
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows, nprobe, k, metric)
        assert np.array_equal(ids, e_ids) and np.array_equal(d.view(np.uint32), e_d.view(np.uint32))


def test_device_resident_ivf_build_load_search(L, oracle):
    """The `_device` twins (rows, queries and outputs in HBM): same centroids / assignments / results as the host-array
    entry points (IvfFlatMmap::build reads an mmapped store, ivf_flat_mmap.rs:56-159; here device memory)."""
    import torch

    rng = np.random.default_rng(5)
    n, dim, nlist, nprobe, k, nq = 20_000, 48, 32, 6, 10, 40
    centers = rng.standard_normal((16, dim)).astype(f32)
    data = (centers[rng.integers(0, 16, n)] + 0.3 * rng.standard_normal((n, dim))).astype(f32)
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    dev = torch.device("cuda", 0)
    d_rows = torch.as_tensor(data, device=dev)
    d_q = torch.as_tensor(queries, device=dev)
    for metric in (IP, L2):
        host = L.IvfFlatIndex.build(None, data, dim, nlist, 5, NAME[metric], l2_partitions=False)
        devi = L.IvfFlatIndex.build_device(d_rows, dim, nlist, 5, NAME[metric], l2_partitions=False)
        h_cen, h_asg, h_off, h_orig = host.export()
        d_cen, d_asg, d_off, d_orig = devi.export()
        assert np.array_equal(h_asg, d_asg) and np.array_equal(h_cen.view(np.uint32), d_cen.view(np.uint32))
        assert np.array_equal(h_off, d_off) and np.array_equal(h_orig, d_orig)
        loaded = L.IvfFlatIndex.load_device(d_rows, d_cen, d_asg, NAME[metric])
        off, rows_l = oracle.lists_from_assignments(d_asg, d_cen.shape[0])
        r = torch.zeros((nq, k), dtype=torch.int64, device=dev)
        d = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        c = torch.zeros(nq, dtype=torch.int32, device=dev)
        for idx in (devi, loaded):
            idx.search_device(d_q, k, nprobe, r, d, c)
            torch.cuda.synchronize()
            g_rows, g_d, g_c = r.cpu().numpy().view(np.uint64), d.cpu().numpy(), c.cpu().numpy()
            for qi in (0, 7, 33, nq - 1):
                e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, d_cen, off, rows_l, nprobe, k, metric)
                cnt = int(g_c[qi])
                assert cnt == len(e_ids)
                assert np.array_equal(g_d[qi, :cnt].view(np.uint32), e_d.view(np.uint32)) and np.array_equal(g_rows[qi, :cnt], e_ids)
    with pytest.raises(NotImplementedError):
        L.IvfFlatIndex.build_device(d_rows, dim, nlist, 5, "hamming", l2_partitions=False)


@pytest.mark.parametrize("metric", [IP, L2, COS])
def test_ivf_few_queries_fused_path_equals_staged_and_oracle(L, oracle, metric):
    """One to four queries take the two-launch fused path (centroid ranking + probed-list scan in k_small_search, no host
    round trip); more queries and `set_fused_search(False)` take the staged path.  Both must return the oracle's rows and
    distance bits — including k larger than the probed lists hold, duplicate rows (ties broken by the original row id),
    lists left empty by the clustering, and the all-probed-lists-empty fallback of IVFIndex (ivf.rs:258-265)."""
    rng = np.random.default_rng(90 + metric)
    n, dim, nlist = 9000, 72, 96
    centers = rng.standard_normal((40, dim)).astype(f32)
    data = (centers[rng.integers(0, 40, n)] + 0.25 * rng.standard_normal((n, dim))).astype(f32)
    data[100:140] = data[7]                                   # 40 copies of one row: ties on (distance) -> original row order
    cen, asg, off, rows = oracle_ivf(oracle, data, nlist, metric)
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
    queries = (data[rng.integers(0, n, 4)] + 0.05 * rng.standard_normal((4, dim))).astype(f32)
    queries[1] = data[7]
    for nq in (1, 2, 4):
        for nprobe, k in ((1, 10), (6, 10), (32, 64), (64, 50), (3, 1)):
            got = {}
            for fused in (True, False):
                idx.set_fused_search(fused)
                got[fused] = idx.search_batch_arrays(queries[:nq], k, nprobe)
            idx.set_fused_search(True)
            for qi in range(nq):
                e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows, nprobe, k, metric)
                for fused in (True, False):
                    g_rows, g_d, g_c = got[fused]
                    c = int(g_c[qi])
                    assert c == len(e_ids), (fused, nq, nprobe, k, qi, c, len(e_ids))
                    assert np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), (fused, nq, nprobe, k, qi)
                    assert np.array_equal(g_rows[qi, :c], e_ids), (fused, nq, nprobe, k, qi, g_rows[qi, :c], e_ids)
    # every probed list empty: centroids far from the data own no row; a query next to one probes only empty lists.
    # IVFIndex then searches every row (the fused path flags it and the staged path answers).
    cen2 = np.concatenate([cen, (100.0 + np.arange(4 * dim, dtype=f32)).reshape(4, dim)])
    idx2 = L.IvfFlatIndex.load(data, cen2, asg, NAME[metric])
    off2, rows2 = oracle.lists_from_assignments(asg, cen2.shape[0])
    q_far = cen2[-1:].copy() if metric != COS else cen2[-2:-1].copy()
    g_rows, g_d, g_c = idx2.search_batch_arrays(q_far, 5, 1)
    e_ids, e_d, _ = oracle.ivf_search(q_far[0], data, cen2, off2, rows2, 1, 5, metric)
    probes = oracle.ivf_search(q_far[0], data, cen2, off2, rows2, 1, 5, metric)[2]
    c = int(g_c[0])
    assert c == len(e_ids) and np.array_equal(g_rows[0, :c], e_ids) and np.array_equal(g_d[0, :c].view(np.uint32), e_d.view(np.uint32)), (probes, g_rows[0], e_ids)
    # device-resident twin through the fused path
    import torch
    dev = torch.device("cuda", 0)
    dq = torch.as_tensor(queries[:3], device=dev)
    r = torch.zeros((3, 10), dtype=torch.int64, device=dev); d = torch.zeros((3, 10), dtype=torch.float32, device=dev)
    cc = torch.zeros(3, dtype=torch.int32, device=dev)
    idx.search_device(dq, 10, 6, r, d, cc)
    for qi in range(3):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows, 6, 10, metric)
        assert np.array_equal(r[qi].cpu().numpy().astype(np.uint64)[:len(e_ids)], e_ids.astype(np.uint64))
        assert np.array_equal(d[qi].cpu().numpy()[:len(e_ids)].view(np.uint32), e_d.view(np.uint32))


def test_sharded_ivf_entry_point_with_a_one_rank_communicator(L, oracle):
    """lynse_hip_ivf_search_sharded_f32_device (local part of the probed lists -> RCCL exchange -> device merge behind the
    C-ABI) on the one GPU a test box has: a 1-rank RCCL communicator, global rows through the row map, results equal to
    the oracle's IVFIndex::search over the same centroids / lists — and to the torch.distributed-free device path."""
    import torch

    from lynsedb_amd.sharded import NativeComm, ShardedIvf, ShardOutputs

    rng = np.random.default_rng(91)
    n, dim, nlist, nprobe, nq, k = 30_000, 64, 64, 6, 40, 10
    centers = rng.standard_normal((20, dim)).astype(f32)
    data = (centers[rng.integers(0, 20, n)] + 0.3 * rng.standard_normal((n, dim))).astype(f32)
    cen, asg, off, rows = oracle_ivf(oracle, data, nlist, IP, iters=5)
    sh = ShardedIvf(dim, rank=0, world=1, device=0, group=None)
    sh.load_local(data, cen, asg, "ip")
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    dq = torch.as_tensor(queries, device="cuda:0")
    plain = ShardOutputs(nq, k, 1, dq.device)
    sh.search_device(dq, k, nprobe, plain)
    torch.cuda.synchronize()
    sh.comm = NativeComm(None, 0, 1, 0)
    out = ShardOutputs(nq, k, 1, dq.device)
    sh.search_device(dq, k, nprobe, out)
    torch.cuda.synchronize()
    assert torch.equal(out.rows, plain.rows) and torch.equal(out.dists, plain.dists) and torch.equal(out.counts, plain.counts)
    r, d, c = out.rows.cpu().numpy().view(np.uint64), out.dists.cpu().numpy(), out.counts.cpu().numpy()
    for qi in range(nq):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows, nprobe, k, IP)
        cc = int(c[qi])
        assert cc == len(e_ids) and np.array_equal(r[qi, :cc], e_ids) and np.array_equal(d[qi, :cc].view(np.uint32), e_d.view(np.uint32)), qi


@pytest.mark.parametrize("metric", [IP, L2])
def test_ivf_any_k_up_to_max_top_k(L, oracle, metric):
    """IVFIndex::search takes any k (ivf.rs:304-310; the server caps at MAX_TOP_K = 10,000, src/server/mod.rs:46): beyond the
    candidate capacity of the staged pipeline (k > 4096) every row of the probed lists is scored exactly and the host keeps
    the k best keys — ids and distance bits equal the oracle's, also when the probed lists hold fewer than k rows and for
    the all-lists-empty fallback."""
    rng = np.random.default_rng(300 + metric)
    n, dim, nlist, nprobe = 60_000, 32, 16, 8
    centers = rng.standard_normal((16, dim)).astype(f32)
    data = (centers[rng.integers(0, 16, n)] + 0.4 * rng.standard_normal((n, dim))).astype(f32)
    cen, asg, off, rows = oracle_ivf(oracle, data, nlist, metric, iters=4)
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
    queries = (data[rng.integers(0, n, 3)] + 0.05 * rng.standard_normal((3, dim))).astype(f32)
    for k, npr in ((10_000, nprobe), (5_000, 1), (4_097, 3)):
        g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, npr)
        for qi in range(len(queries)):
            e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows, npr, k, metric)
            c = int(g_c[qi])
            assert c == len(e_ids), (k, npr, qi, c, len(e_ids))
            assert np.array_equal(g_rows[qi, :c], e_ids) and np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), (k, npr, qi)


def test_ivf_any_k_with_a_subset_and_wide_binary_rows(L, oracle):
    """The exact-scoring IVF path (k beyond the staged pipeline's capacity, or packed rows wider than 4096 bits) with
    SearchParams.subset (ivf.rs:251-265) and for the IVF-HAMMING-BINARY mode (quantizer -> L2 routing on the codes -> popcount)."""
    rng = np.random.default_rng(808)
    # float index, k = 6000 within a 40 % subset
    n, dim, nlist, nprobe, k = 40_000, 24, 12, 6, 6_000
    data = rng.standard_normal((n, dim)).astype(f32)
    cen, asg, off, rows = oracle_ivf(oracle, data, nlist, L2, iters=3)
    idx = L.IvfFlatIndex.load(data, cen, asg, "l2")
    subset = np.sort(rng.choice(n, n * 2 // 5, replace=False)).astype(np.uint64)
    q = rng.standard_normal((2, dim)).astype(f32)
    g_rows, g_d, g_c = idx.search_filtered_batch_arrays(q, k, nprobe, subset)
    for qi in range(2):
        e_ids, e_d = oracle.ivf_search_filtered(q[qi], data, cen, off, rows, nprobe, k, L2, subset)
        c = int(g_c[qi])
        assert c == len(e_ids) and np.array_equal(g_rows[qi, :c], e_ids) and np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), qi
    # binary index over 4160-bit codes (wider than the tiled popcount kernel's 4096) and k = 5000 on a narrow one
    for n, dim, nlist, nprobe, k in ((3_000, 4160, 8, 3, 10), (30_000, 64, 8, 6, 5_000)):
        data = (rng.standard_normal((n, dim)) + 0.3).astype(f32)
        ab, thr = oracle.binary_fit(data)
        enc = oracle.binary_quantize(data, thr)
        packed = oracle.pack_binary(enc)
        cen, asg = oracle.kmeans_train(enc, nlist, 4, O.L2)
        off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
        idx = L.IvfFlatIndex.load(data, cen, asg, "hamming", thresholds=thr)
        queries = data[rng.integers(0, n, 3)].copy()
        g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
        for qi in range(3):
            eq = oracle.binary_quantize(queries[qi], thr)[0]
            e_ids, e_d, _ = oracle.ivf_search(eq, enc, cen, off, rows, nprobe, k, O.HAMMING, packed=packed)
            c = int(g_c[qi])
            assert c == len(e_ids) and np.array_equal(g_rows[qi, :c], e_ids) and np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), (dim, qi)


# ---- the staged path on the certified int8 coarse pass (k_scan_h16<.., TILED, I8C>): IP batches of more than 32 queries over
# an f32 slab of >= 64K rows and whole 128-column slabs scan the SQ8 codes of the probed lists; survivors are rescored exactly
# from the f32 slab as always (IVFIndex::search, ivf.rs:181-348)
@pytest.mark.parametrize("n,dim,nlist,nprobe,nq,k,kind,metric", [
    (120_000, 128, 256, 8, 100, 10, "clustered", IP), (90_000, 256, 64, 64, 40, 25, "gaussian", IP),
    (200_000, 384, 1024, 16, 256, 10, "uniform", IP), (70_000, 128, 300, 5, 33, 100, "clustered", IP),
    (150_000, 256, 512, 12, 130, 10, "uniform", L2), (100_000, 300, 128, 9, 64, 10, "clustered", L2),
    (150_000, 384, 512, 12, 130, 10, "gaussian", COS), (80_000, 256, 100, 100, 48, 20, "clustered", COS),
    (300_000, 128, 256, 8, 12, 10, "clustered", IP), (280_000, 256, 128, 6, 30, 5, "gaussian", L2),   # >= 256K rows: batches of 5..32 queries too
])
def test_ivf_staged_path_on_the_certified_int8_pass(L, oracle, n, dim, nlist, nprobe, nq, k, kind, metric):
    rng = np.random.default_rng(n + dim + nlist + nq)
    if kind == "clustered":
        centers = rng.standard_normal((nlist // 2, dim)).astype(f32)
        data = (centers[rng.integers(0, centers.shape[0], n)] + 0.3 * rng.standard_normal((n, dim))).astype(f32)
    elif kind == "gaussian":
        data = rng.standard_normal((n, dim)).astype(f32)
    else:
        data = rng.random((n, dim), dtype=f32)
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    # (the partitions only have to be SOME centroids + assignments: a short device k-means, exported and fed to both sides)
    built = L.IvfFlatIndex.build(None, data, dim, nlist, 2, NAME[metric], l2_partitions=False)
    cen, asg, _, _ = built.export()
    del built
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    idx.profile_enable(True)
    g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
    p = idx.profile_get()
    assert int(p["last_plan"]) & 64, ("the staged IVF search did not start on the int8 pass", p["last_plan"])
    assert int(p["last_plan"]) & 4, ("the int8 pass overflowed on benign data", p["last_plan"])
    for qi in sorted({0, 1, min(31, nq - 1), min(32, nq - 1), nq // 2, nq - 1}):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows, nprobe, k, metric)
        c = int(g_c[qi])
        assert c == len(e_ids), (qi, c, len(e_ids))
        assert np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), (qi, g_d[qi, :c], e_d)
        assert np.array_equal(g_rows[qi, :c], e_ids), (qi, g_rows[qi, :c], e_ids)
    # 32 queries or fewer stay on the f16 shadow below 256K rows (from there on they take the int8 pass as well): the same answers
    m = min(20, nq)
    r2, d2, c2 = idx.search_batch_arrays(queries[:m], k, nprobe)
    assert bool(int(idx.profile_get()["last_plan"]) & 64) == (n >= 262_144)
    assert np.array_equal(r2, g_rows[:m]) and np.array_equal(d2.view(np.uint32), g_d[:m].view(np.uint32))


def test_ivf_int8_pass_overflow_goes_back_to_the_f16_shadow(L, oracle):
    """One row with 1e4 in one dimension collapses that dimension's SQ8 scale: the int8 margin lets every probed row through,
    the candidate pool overflows, the chunk is answered by the f16 shadow (a strike), results stay exact."""
    rng = np.random.default_rng(5150)
    n, dim, nlist, nprobe, nq, k = 100_000, 128, 32, 8, 64, 10
    data = rng.standard_normal((n, dim)).astype(f32)
    data[4321, 9] = 1.0e4
    queries = rng.standard_normal((nq, dim)).astype(f32)
    cen, asg = oracle.kmeans_train(data[:4000], nlist, 3, IP)
    asg = np.concatenate([asg, rng.integers(0, cen.shape[0], n - 4000).astype(asg.dtype)])
    idx = L.IvfFlatIndex.load(data, cen, asg, "ip")
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    idx.profile_enable(True)
    g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
    plan = int(idx.profile_get()["last_plan"])
    assert plan & 64
    for qi in (0, 33, 63):
        e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows, nprobe, k, IP)
        assert np.array_equal(g_rows[qi], e_ids) and np.array_equal(g_d[qi].view(np.uint32), e_d.view(np.uint32))


@pytest.mark.parametrize("metric", [IP, L2, COS])
def test_ivf_device_side_grouping_equals_the_host_path(L, oracle, metric, monkeypatch):
    """k_ivf_group / k_ivf_emit_tiles / k_ivf_gather_groups against the host loop they replace (LYNSE_HIP_IVF_DEVICE_GROUPING=0)
    and the oracle: many queries per list (groups of 32 split), lists left empty, a query that probes ONLY empty lists in the
    middle of the batch (IVFIndex then scans every list, ivf.rs:258-265: the kernel flags it, the host path answers), lists
    longer than the second position window."""
    rng = np.random.default_rng(700 + metric)
    n, dim, nlist, nq, k = 60_000, 64, 48, 150, 10
    centers = rng.standard_normal((12, dim)).astype(f32)
    data = (centers[rng.integers(0, 12, n)] + 0.25 * rng.standard_normal((n, dim))).astype(f32)
    cen, asg, off, rows = oracle_ivf(oracle, data, nlist, metric, iters=3)
    far = (200.0 + np.arange(3 * dim, dtype=f32)).reshape(3, dim)      # three centroids that own no row
    cen = np.concatenate([cen, far])
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    for nprobe, with_far in ((6, False), (1, False), (20, False), (2, True)):
        qs = queries.copy()
        if with_far:
            qs[77] = far[2] if metric != COS else far[1]
        got = {}
        for dg in ("1", "0"):
            monkeypatch.setenv("LYNSE_HIP_IVF_DEVICE_GROUPING", dg)
            got[dg] = idx.search_batch_arrays(qs, k, nprobe)
        assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][2], got["0"][2])
        assert np.array_equal(got["1"][1].view(np.uint32), got["0"][1].view(np.uint32))
        for qi in (0, 31, 32, 77, 149):
            e_ids, e_d, _ = oracle.ivf_search(qs[qi], data, cen, off, rows, nprobe, k, metric)
            c = int(got["1"][2][qi])
            assert c == len(e_ids) and np.array_equal(got["1"][0][qi, :c], e_ids), (nprobe, with_far, qi, got["1"][0][qi, :c], e_ids)
            assert np.array_equal(got["1"][1][qi, :c].view(np.uint32), e_d.view(np.uint32))
    # more pairs than 8192 (256 queries x 40 probes): the pair ranks live in global memory, the per-list arrays in LDS
    big = (data[rng.integers(0, n, 256)] + 0.05 * rng.standard_normal((256, dim))).astype(f32)
    got = {}
    for dg in ("1", "0"):
        monkeypatch.setenv("LYNSE_HIP_IVF_DEVICE_GROUPING", dg)
        got[dg] = idx.search_batch_arrays(big, k, 40)
    assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1].view(np.uint32), got["0"][1].view(np.uint32))
    for qi in (0, 100, 255):
        e_ids, e_d, _ = oracle.ivf_search(big[qi], data, cen, off, rows, 40, k, metric)
        assert np.array_equal(got["1"][0][qi, :len(e_ids)], e_ids) and np.array_equal(got["1"][1][qi, :len(e_ids)].view(np.uint32), e_d.view(np.uint32))


def test_all_lists_fallback_with_many_lists_and_padding_of_short_results(L, oracle):
    """Found by scripts/stress_ivf_inflight.py (round 4).  (1) A query whose probed lists are all empty scans EVERY list (ivf.rs:258-265):
    with more than 64 lists the first position window emitted a0 x nlist keys per query — a0 was sized for nprobe lists — and every plan
    overflowed ('candidate overflow on the exhaustive IVF plan').  (2) Results shorter than k are padded the same way by the fused
    few-query search and by the staged pipeline (row ~0, the worst distance of the metric)."""
    rng = np.random.default_rng(4045)
    n, dim, nlist, k = 40_000, 64, 300, 10
    data = rng.random((n, dim), dtype=f32)
    cen, asg, off, rows = oracle_ivf(oracle, data, nlist, L2, iters=2)
    far = (150.0 + np.arange(2 * dim, dtype=f32)).reshape(2, dim)      # two centroids that own no row
    cen = np.concatenate([cen, far])
    off, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    idx = L.IvfFlatIndex.load(data, cen, asg, "l2")
    for nq in (1, 33, 256):
        qs = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
        qs[nq // 2] = far[1]                                            # nprobe = 1: only an empty list is probed
        g_rows, g_d, g_c = idx.search_batch_arrays(qs, k, 1)
        for qi in sorted({0, nq // 2, nq - 1}):
            e_ids, e_d, _ = oracle.ivf_search(qs[qi], data, cen, off, rows, 1, k, L2)
            c = int(g_c[qi])
            assert c == len(e_ids) and np.array_equal(g_rows[qi, :c], e_ids) and np.array_equal(g_d[qi, :c].view(np.uint32), e_d.view(np.uint32)), (nq, qi)
        assert int(g_c[nq // 2]) == k                                   # every row was a candidate
    # short results: 300 lists over 700 rows, nprobe 1 -> a handful of rows per query, k = 50
    small = data[:700]
    cen2, asg2, off2, rows2 = oracle_ivf(oracle, small, 300, L2, iters=2)
    idx2 = L.IvfFlatIndex.load(small, cen2, asg2, "l2")
    q4 = (small[rng.integers(0, 700, 4)] + 0.001).astype(f32)
    fused = idx2.search_batch_arrays(q4, 50, 1)
    idx2.set_fused_search(False)
    staged = idx2.search_batch_arrays(q4, 50, 1)
    assert np.array_equal(fused[2], staged[2]) and int(fused[2].min()) < 50    # (a query that probes an empty list scans every row: 50 results)
    assert np.array_equal(fused[0], staged[0]) and np.array_equal(fused[1].view(np.uint32), staged[1].view(np.uint32))
    for qi in range(4):
        c = int(fused[2][qi])
        assert np.all(fused[0][qi, c:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isposinf(fused[1][qi, c:]))
