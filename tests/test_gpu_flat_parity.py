"""GPU parity tests: the HIP path (through the C ABI of liblynse_hip.so) vs the CPU oracle.

Bar (task ③): ids/ranks bit-exact under the canonical (distance, row) order; float distances are
ALSO compared bit-exactly because the final rescoring pass reproduces the reference's accumulation
order (tolerance stated per test: 0 ulp unless noted; north_star allows 1e-5 relative).
Run on the GPU box with `pytest -m gpu`.
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

IP, L2, COS, HAM, JAC, DICE, TANI = O.IP, O.L2, O.COS, O.HAMMING, O.JACCARD, O.DICE, O.TANIMOTO
NAME = {IP: "ip", L2: "l2", COS: "cosine", HAM: "hamming", JAC: "jaccard", DICE: "dice", TANI: "tanimoto"}
f32 = np.float32


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1, "no HIP device: GPU tests need the MI355X box"
    return L_


def make_index(L, data):
    idx = L.FlatIndex(None, data.shape[1])
    if data.shape[0]:
        idx.write(data)
    return idx


def check_batch(L, oracle, idx, data, queries, k, metric, ip_form=O.IPFORM_AUTO, exact_dist=True):
    if queries.shape[0] <= 4 and k <= 64:  # small batches take the fused single-launch search: check the staged pipeline as well
        idx.set_fused_search(False)
        try:
            _check_batch(L, oracle, idx, data, queries, k, metric, ip_form, exact_dist)
        finally:
            idx.set_fused_search(True)
    _check_batch(L, oracle, idx, data, queries, k, metric, ip_form, exact_dist)


def _check_batch(L, oracle, idx, data, queries, k, metric, ip_form=O.IPFORM_AUTO, exact_dist=True):
    rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
    for qi in range(queries.shape[0]):
        e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, metric, ip_form)
        c = int(counts[qi])
        assert c == len(e_ids), (qi, c, len(e_ids))
        got_ids = rows[qi, :c].astype(np.uint32)
        got_d = dists[qi, :c]
        if exact_dist:
            assert np.array_equal(got_d.view(np.uint32), e_d.view(np.uint32)), \
                (NAME[metric], qi, got_d[:5], e_d[:5], got_ids[:5], e_ids[:5])
        else:
            assert np.allclose(got_d, e_d, rtol=1e-5, atol=1e-6)
        assert np.array_equal(got_ids, e_ids), (NAME[metric], qi, got_ids[:10], e_ids[:10], got_d[:10], e_d[:10])


# ------------------------------------------------------------------ reference KATs through the GPU

def test_kat_flat_mmap_write_search(L):  # flat_mmap.rs:6022-6054
    data = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0.5, 0.5, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], f32)
    idx = make_index(L, data)
    assert len(idx) == 5 and idx.dim == 4
    ids, d = idx.search(np.array([1, 0, 0, 0], f32), 2, "ip")
    assert len(ids) == 2 and ids[0] == 0 and abs(d[0] - 1.0) < 1e-6
    ids, _ = idx.search(np.zeros(4, f32), 1, "l2")
    assert ids[0] == 2
    assert ids.dtype == np.uint32 and d.dtype == np.float32


def test_kat_top_k_functions(L):  # distance/mod.rs:502-527, :571-621; test_backend.py:25-92
    ids, d = L.py_top_k_search(np.array([1, 0, 0, 0], f32), np.array([[1, 0, 0, 0], [.5, .5, 0, 0], [0, 1, 0, 0]], f32), "IP", 2)
    assert list(ids) == [0, 1] and abs(d[0] - 1.0) < 1e-6
    ids, _ = L.py_top_k_search(np.zeros(3, f32), np.array([[1, 0, 0], [.1, 0, 0], [2, 0, 0]], f32), "L2", 2)
    assert ids[0] == 1
    ids, d = L.py_top_k_search(np.array([0, 0], f32), np.array([[2, 0], [1, 0]], f32), "l2", 10)  # clamp
    assert list(ids) == [1, 0] and d[0] <= d[1]
    ids, d = L.py_top_k_search(np.array([1, 2], f32), np.zeros((0, 2), f32), "l2", 5)  # empty
    assert len(ids) == 0 and len(d) == 0
    ids, d = L.py_top_k_search(np.array([1, 2], f32), np.array([[1, 2], [3, 4]], f32), "l2", 0)  # k == 0
    assert len(ids) == 0
    q = np.array([1, 0, 1, 0], f32)
    c = np.array([[1, 0, 1, 0], [1, 1, 1, 0], [0, 1, 0, 1]], f32)
    ids, d = L.py_top_k_search(q, c, "hamming", 3)
    assert list(ids) == [0, 1, 2] and list(d) == [0.0, 1.0, 4.0]
    ids, d = L.py_top_k_search(q, c, "jaccard", 3)
    assert list(ids) == [0, 1, 2] and abs(d[1] - 1 / 3) < 1e-6 and abs(d[2] - 1) < 1e-6
    for a, b, m, exp, tol in [([1, 0, 0], [0, 1, 0], "IP", 0.0, 1e-5), ([0, 0], [3, 4], "L2", 25.0, 1e-4),
                              ([1, 0], [0, 1], "cosine", 1.0, 1e-5), ([1, 1, 0], [1, 0, 1], "dice", 0.5, 1e-5),
                              ([1, 1, 0], [1, 0, 1], "tanimoto", 2 / 3, 1e-5), ([3, 4], [3, 4], "IP", 25.0, 1e-4)]:
        assert abs(L.py_compute_distance(np.array(a, f32), np.array(b, f32), m) - exp) < tol
    with pytest.raises(ValueError, match="Unknown metric"):
        L.py_compute_distance(np.zeros(2, f32), np.zeros(2, f32), "bogus")
    with pytest.raises(ValueError):
        L.py_compute_distance(np.zeros(2, f32), np.zeros(3, f32), "ip")


def test_kat_packed_binary_dim130(L, oracle):  # flat_mmap.rs:6386-6421 (3 words, tail bits)
    dim = 130
    rows = np.zeros((3, dim), f32)
    for i in (0, 1, 64, 129):
        rows[0, i] = rows[1, i] = 1.0
    rows[1, 5] = 1.0
    for i in (2, 3, 65):
        rows[2, i] = 1.0
    idx = make_index(L, rows)
    assert np.array_equal(idx.read_packed(0, 3), oracle.pack_binary(rows))  # ballot pack == pack_binary_f32
    for m in ("hamming", "jaccard", "tanimoto", "dice"):
        ids, d = idx.search(rows[0], 3, m)
        r_ids, r_d = oracle.top_k_search(rows[0], rows, 3, oracle.metric_from_str(m))
        assert np.array_equal(ids, r_ids) and np.all(np.abs(d - r_d) < 1e-6)


def test_kat_eye_and_backend_fixtures(L, oracle):  # test_backend.py:107-190
    eye = np.eye(16, dtype=f32)
    idx = make_index(L, eye)
    for i in range(16):
        ids, d = idx.search(eye[i], 1, "ip")
        assert ids[0] == i and abs(d[0] - 1) < 1e-5
    ids, d = idx.search(eye[3], 1, "l2")
    assert ids[0] == 3 and abs(d[0]) < 1e-5
    np.random.seed(7)
    vecs = np.random.rand(200, 16).astype(f32)
    np.random.seed(1)
    q = np.random.rand(16).astype(f32)
    idx = make_index(L, vecs)
    assert idx.search(q, 1, "ip")[0][0] == int(np.argmax(vecs @ q))
    assert idx.search(q, 1, "l2")[0][0] == int(np.argmin(((vecs - q) ** 2).sum(1)))
    ids, d = idx.search(q, 300, "ip")
    assert len(ids) == 200
    check_batch(L, oracle, idx, vecs, q.reshape(1, -1), 20, IP)
    check_batch(L, oracle, idx, vecs, q.reshape(1, -1), 20, L2)
    check_batch(L, oracle, idx, vecs, q.reshape(1, -1), 20, COS)


# ------------------------------------------------------------------ randomized differential parity

@pytest.mark.parametrize("metric", [IP, L2, COS])
@pytest.mark.parametrize("n,dim,nq,k", [
    (1, 4, 1, 1), (5, 3, 2, 10), (100, 16, 7, 10), (1000, 130, 3, 5), (4095, 32, 4, 10), (4097, 32, 33, 10),
    (20000, 128, 40, 10), (50000, 96, 256, 10), (30000, 64, 300, 7), (9000, 24, 5, 100), (70000, 768, 12, 10),
])
def test_float_parity_uniform(L, oracle, metric, n, dim, nq, k):
    rng = np.random.default_rng(n * 31 + dim)
    data = rng.random((n, dim), dtype=f32)
    queries = rng.random((nq, dim), dtype=f32)
    if n > 10:
        queries[0] = data[n // 2]  # exact self match
    idx = make_index(L, data)
    sel = list(range(min(nq, 6))) + ([nq - 1] if nq > 6 else [])
    rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
    for qi in sel:
        e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, metric)
        c = int(counts[qi])
        assert c == len(e_ids)
        assert np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32)), (qi, dists[qi, :c], e_d)
        assert np.array_equal(rows[qi, :c].astype(np.uint32), e_ids), (qi, rows[qi, :c], e_ids)


@pytest.mark.parametrize("metric", [IP, L2, COS, HAM, JAC])
@pytest.mark.parametrize("nq", [3, 40, 300])
def test_top1_takes_the_minimum_key_path(L, oracle, metric, nq):
    """k = 1 (the k-means assignment's shape): select_body finds the smallest key by a block-wide minimum instead of the radix
    passes.  Uniform rows, rows duplicated beyond k (ties go to the lowest row) and a self match; 0 ulp on the distance."""
    rng = np.random.default_rng(1000 + nq)
    n, dim = 30000, 64
    if metric in (HAM, JAC):
        data = (rng.random((n, dim)) < 0.4).astype(f32)
        queries = (rng.random((nq, dim)) < 0.4).astype(f32)
    else:
        data = rng.random((n, dim), dtype=f32)
        queries = rng.random((nq, dim), dtype=f32)
    data[20000:20050] = data[77]          # 51 copies of one row: the tie of the best score must resolve to row 77
    if metric == IP:                      # (the largest row wins every IP query: copies of it, the lowest row of the tie)
        data[123] = f32(0.999)
        data[25000:25010] = data[123]
    queries[0] = data[77]
    queries[1] = data[n - 1]
    idx = make_index(L, data)
    sel = np.arange(nq) if nq <= 40 else np.concatenate([np.arange(8), rng.choice(nq, 12, replace=False)])
    if nq <= 40:   # (3 queries: the fused single-launch search and the staged pipeline both)
        check_batch(L, oracle, idx, data, queries, 1, metric)
    rows, dists, counts = idx.search_batch_arrays(queries, 1, NAME[metric])
    assert int(rows[0, 0]) == (123 if metric == IP else 77)
    for qi in sel:
        e_ids, e_d = oracle.canonical_topk(queries[qi], data, 1, metric)
        assert int(counts[qi]) == 1
        assert np.array_equal(dists[qi, :1].view(np.uint32), e_d.view(np.uint32)), (qi, dists[qi, :1], e_d)
        assert int(rows[qi, 0]) == int(e_ids[0]), (qi, rows[qi, :1], e_ids)


@pytest.mark.parametrize("k", [1, 10, 33, 64])
@pytest.mark.parametrize("order", ["random", "best_first", "best_last", "one_block"])
def test_fused_search_merge_of_the_workgroup_lists(L, oracle, k, order):
    """k_small_search's last-workgroup merge (threshold from the list heads, candidates ranked by counting; more than 512
    candidates: sorted).  Rows stored in score order put whole lists under the threshold (k = 64: thousands of candidates);
    `one_block` keeps every good row inside 128 consecutive rows (one workgroup's share: fewer good lists than k)."""
    rng = np.random.default_rng(k * 7 + len(order))
    n, dim = 70000, 16
    data = rng.random((n, dim), dtype=f32)
    q = rng.random((4, dim), dtype=f32)
    s = data @ q[0]
    if order == "best_first":
        data = data[np.argsort(-s, kind="stable")]
    elif order == "best_last":
        data = data[np.argsort(s, kind="stable")]
    elif order == "one_block":
        data *= f32(0.01)
        data[4096:4224] = rng.random((128, dim), dtype=f32)
    idx = make_index(L, data)
    for nq in (1, 4):
        rows, dists, counts = idx.search_batch_arrays(q[:nq], k, "ip")
        for qi in range(nq):
            e_ids, e_d = oracle.canonical_topk(q[qi], data, k, IP)
            assert int(counts[qi]) == k
            assert np.array_equal(rows[qi, :k].astype(np.uint32), e_ids), (order, k, qi, rows[qi, :k], e_ids)
            assert np.array_equal(dists[qi, :k].view(np.uint32), e_d.view(np.uint32))
    small = make_index(L, data[:40])          # one workgroup, fewer rows than k: every key is a candidate, short result padded
    rows, dists, counts = small.search_batch_arrays(q[:2], k, "l2")
    for qi in range(2):
        e_ids, e_d = oracle.canonical_topk(q[qi], data[:40], k, L2)
        c = int(counts[qi])
        assert c == min(k, 40) and np.array_equal(rows[qi, :c].astype(np.uint32), e_ids)
        assert np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32))


@pytest.mark.parametrize("metric", [IP, L2, COS])
@pytest.mark.parametrize("k,nq", [(1, 40), (5, 40), (5, 256), (10, 130)])
def test_decisive_best_rows_keep_their_threshold_rescoring(L, oracle, metric, k, nq, monkeypatch):
    """Every query has a family of exactly k near-copies in the shard and nothing else within the coarse margin: the survivors of
    the last select are the rows it rescored for the threshold, and k_select_final hands their exact scores on
    (SelectArgs::hand_exact) instead of rescoring them again.  Same answer as the oracle, and as LYNSE_HIP_HAND_EXACT=0."""
    rng = np.random.default_rng(300 + k + nq)
    n, dim = 80000, 96
    data = (rng.random((n, dim), dtype=f32) * f32(0.2)).astype(f32)
    queries = (rng.random((nq, dim), dtype=f32) + f32(0.5)).astype(f32)
    slots = rng.choice(n, size=(nq, k), replace=False)
    for qi in range(nq):
        for j in range(k):
            data[slots[qi, j]] = queries[qi] * f32(1.0 + 0.002 * j) + rng.standard_normal(dim).astype(f32) * f32(1e-3)
    idx = make_index(L, data)
    rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
    monkeypatch.setenv("LYNSE_HIP_HAND_EXACT", "0")
    rows0, dists0, counts0 = idx.search_batch_arrays(queries, k, NAME[metric])
    monkeypatch.delenv("LYNSE_HIP_HAND_EXACT")
    assert np.array_equal(rows, rows0) and np.array_equal(dists.view(np.uint32), dists0.view(np.uint32)) and np.array_equal(counts, counts0)
    for qi in list(range(6)) + [nq - 1]:
        e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, metric)
        assert int(counts[qi]) == k
        assert np.array_equal(rows[qi, :k].astype(np.uint32), e_ids), (qi, rows[qi, :k], e_ids)
        assert np.array_equal(dists[qi, :k].view(np.uint32), e_d.view(np.uint32))
        if metric != IP:
            assert set(int(r) for r in rows[qi, :k]) == set(int(r) for r in slots[qi])


@pytest.mark.parametrize("metric", [IP, L2, COS])
def test_float_parity_gaussian_mixed_scale(L, oracle, metric):
    """Signed data with very different row norms: stresses the f16 scale + certified margin."""
    rng = np.random.default_rng(99)
    n, dim = 40000, 100
    data = rng.standard_normal((n, dim)).astype(f32) * (10.0 ** rng.uniform(-2, 2, size=(n, 1))).astype(f32)
    queries = rng.standard_normal((9, dim)).astype(f32) * f32(3.0)
    idx = make_index(L, data)
    check_batch(L, oracle, idx, data, queries, 10, metric)


@pytest.mark.parametrize("metric", [IP, L2, COS])
def test_float_parity_massive_ties(L, oracle, metric):
    """Duplicated rows -> exact score ties far beyond k: order must still be (distance, row asc)."""
    rng = np.random.default_rng(5)
    base = rng.random((7, 48), dtype=f32)
    data = np.repeat(base, 3000, axis=0)  # 21000 rows, 7 distinct
    data = data[rng.permutation(data.shape[0])]
    queries = np.vstack([base[2], rng.random(48, dtype=f32)])
    idx = make_index(L, data)
    check_batch(L, oracle, idx, data, queries, 25, metric)


def test_float_parity_adversarial_monotone(L, oracle):
    """Scores increase with the row index: every later row beats the running threshold, so the
    candidate buffers overflow and the exhaustive safe plan must take over."""
    n, dim = 60000, 8
    data = np.zeros((n, dim), f32)
    data[:, 0] = np.arange(n, dtype=f32) / f32(n)
    data[:, 1] = f32(0.5)
    q = np.zeros((2, dim), f32)
    q[:, 0] = 1.0
    q[1, 1] = 0.25
    idx = make_index(L, data)
    idx.profile_enable(True)
    check_batch(L, oracle, idx, data, q, 10, IP)   # (the staged pipeline is one of the two passes of check_batch)
    assert idx.profile_get()["fallback_queries"] > 0


def test_zero_vectors_and_zero_query(L, oracle):
    rng = np.random.default_rng(3)
    data = rng.random((5000, 20), dtype=f32)
    data[::7] = 0.0
    queries = np.vstack([np.zeros(20, f32), rng.random(20, dtype=f32)])
    idx = make_index(L, data)
    for m in (IP, L2, COS):
        check_batch(L, oracle, idx, data, queries, 12, m)


def test_append_after_search_and_small_plan(L, oracle):
    rng = np.random.default_rng(8)
    data = rng.random((30000, 40), dtype=f32)
    idx = L.FlatIndex(None, 40)
    idx.set_plan(stage0_rows=256, growth=2, cap=1024)  # many stages
    idx.write(data[:12000])
    q = rng.random((3, 40), dtype=f32)
    check_batch(L, oracle, idx, data[:12000], q, 10, L2)
    idx.write(data[12000:] * f32(50.0))  # new rows change the collection scale
    full = np.vstack([data[:12000], data[12000:] * f32(50.0)])
    for m in (IP, L2, COS):
        check_batch(L, oracle, idx, full, q, 10, m)


def test_ip_forms(L, oracle):
    rng = np.random.default_rng(21)
    data = rng.random((3000, 37), dtype=f32)
    q = rng.random((2, 37), dtype=f32)
    idx = make_index(L, data)
    idx.set_ip_form(O.IPFORM_BATCH8)
    check_batch(L, oracle, idx, data, q, 10, IP, ip_form=O.IPFORM_BATCH8)
    idx.set_ip_form(O.IPFORM_SINGLE)
    check_batch(L, oracle, idx, data, q, 10, IP, ip_form=O.IPFORM_SINGLE)
    # reference policy for n < 4096 is the single-row kernel (flat_mmap.rs:4852): identical to lo_flat_search
    idx.set_ip_form(O.IPFORM_AUTO)
    ids, d = idx.search(q[0], 10, "ip")
    r_ids, r_d = oracle.flat_search(q[0], data, 10, IP)
    assert np.array_equal(ids, r_ids) and np.array_equal(d.view(np.uint32), r_d.view(np.uint32))


# ------------------------------------------------------------------ packed binary (integer path)

@pytest.mark.parametrize("metric", [HAM, JAC, DICE, TANI])
@pytest.mark.parametrize("n,bits,nq,k,p", [(50, 64, 3, 10, 0.5), (4000, 130, 5, 50, 0.5), (100000, 1024, 4, 50, 0.5),
                                           (60000, 1024, 2, 50, 0.05), (30000, 2048, 33, 10, 0.3), (20000, 100, 260, 5, 0.5),
                                           # wider than 4096 bits (no width limit in BinaryData, flat_mmap.rs:145-160)
                                           (3000, 8192, 5, 20, 0.5), (2500, 5000, 3, 10, 0.3)])
def test_binary_parity_packed(L, oracle, metric, n, bits, nq, k, p):
    rng = np.random.default_rng(n + bits)
    W = (bits + 63) // 64
    dense = rng.random((n, bits)) < p
    rows = np.zeros((n, W), np.uint64)
    for w in range(W):
        chunk = dense[:, w * 64:(w + 1) * 64]
        rows[:, w] = (chunk.astype(np.uint64) << np.arange(chunk.shape[1], dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
    queries = rows[rng.integers(0, n, size=nq)].copy()
    queries[-1] ^= np.uint64(0x5)
    idx = L.FlatIndex(None, bits)
    idx.write_packed(rows)
    r, d, c = idx.search_packed_arrays(queries, k, NAME[metric])
    for qi in list(range(min(nq, 4))) + [nq - 1]:
        e_ids, e_d = oracle.canonical_topk_packed(queries[qi], rows, k, metric)
        cc = int(c[qi])
        assert cc == len(e_ids)
        assert np.array_equal(d[qi, :cc].view(np.uint32), e_d.view(np.uint32)), (qi, d[qi, :cc], e_d)
        assert np.array_equal(r[qi, :cc].astype(np.uint32), e_ids), (qi, r[qi, :cc], e_ids)


def test_binary_from_f32_rows_all_identical(L, oracle):
    """f32 {0,1} rows packed lazily on the device (ensure_binary); all rows identical -> every
    distance ties, ids must be 0..k-1."""
    n, dim = 30000, 96
    row = (np.arange(dim) % 3 == 0).astype(f32)
    data = np.tile(row, (n, 1))
    idx = make_index(L, data)
    for m in ("hamming", "jaccard", "dice"):
        ids, d = idx.search(row, 20, m)
        assert list(ids) == list(range(20)) and np.all(d == 0.0)
    q = row.copy()
    q[:5] = 1.0 - q[:5]
    ids, d = idx.search(q, 20, "hamming")
    assert list(ids) == list(range(20)) and np.all(d == 5.0)
    e_ids, e_d = oracle.canonical_topk(q, data, 20, HAM)
    assert np.array_equal(ids, e_ids) and np.array_equal(d, e_d)


# ------------------------------------------------------------------ boundary behaviour

def test_boundary_errors_and_edges(L):
    idx = L.FlatIndex(None, 8)
    ids, d = idx.search(np.zeros(8, f32), 5, "ip")  # empty store -> empty result, not an error
    assert len(ids) == 0 and len(d) == 0
    data = np.random.default_rng(0).random((10, 8), dtype=f32)
    idx.write(data)
    ids, d = idx.search(data[3], 0, "ip")  # k == 0
    assert len(ids) == 0
    ids, d = idx.search(data[3], 50, "l2")  # k > N clamps
    assert len(ids) == 10 and ids[0] == 3
    with pytest.raises(ValueError, match="Unknown metric"):
        idx.search(data[0], 5, "nope")
    with pytest.raises(ValueError, match="dimension mismatch"):
        idx.search(np.zeros(7, f32), 5, "ip")
    with pytest.raises(ValueError):
        idx.write(np.zeros((2, 9), f32))
    res = idx.batch_search(data[:4], 3, "cosine")
    assert len(res) == 4 and all(len(r[0]) == 3 for r in res) and [int(r[0][0]) for r in res] == [0, 1, 2, 3]


def test_collection_surface_like_flat_search_bench(L, oracle, tmp_path):
    """The `_core` call sequence of benchmarks/flat_search_bench.py:43-96 at a reduced row count."""
    rows, dim, k = 20000, 128, 10
    rng = np.random.default_rng(42)
    query = rng.random(dim, dtype=f32)
    mgr = L.DatabaseManager(str(tmp_path))
    mgr.create_database("bench_db")
    mgr.require_collection("bench_db", "bench_vectors", dim)
    coll = mgr.get_collection("bench_db", "bench_vectors", dim)
    allv = []
    for start in range(0, rows, 5000):
        v = rng.random((5000, dim), dtype=f32)
        if start == 0:
            v[0] = query
        coll.add_items(v, list(range(start, start + 5000)), None)
        allv.append(v)
    coll.commit()
    coll.build_index("FLAT-IP", None)
    assert coll.shape() == (rows, dim)
    res = coll.search(query, k, None, 10)
    assert len(res) == k and res.ids()[0] == 0 and res.ids().dtype == np.int64 and res.index_mode() == "FLAT-IP"
    data = np.vstack(allv)
    e_ids, e_d = oracle.canonical_topk(query, data, k, IP)
    assert np.array_equal(res.ids(), e_ids.astype(np.int64)) and np.array_equal(res.distances(), e_d)
    coll.build_index("FLAT-HAMMING-BINARY", None)
    res = coll.search((query > 0.5).astype(f32), 5)
    e_ids, e_d = oracle.canonical_topk((query > 0.5).astype(f32), data, 5, HAM)
    assert np.array_equal(res.ids(), e_ids.astype(np.int64)) and np.array_equal(res.distances(), e_d)


# ------------------------------------------------------------------ multi-GPU exchange pieces on one GPU

@pytest.mark.parametrize("metric", [IP, L2, HAM])
@pytest.mark.parametrize("world,nq,k", [(1, 3, 10), (2, 7, 10), (8, 256, 10), (8, 5, 100), (4, 33, 1)])
def test_device_merge_matches_host_merge(L, oracle, metric, world, nq, k):
    """k_merge (device k-way merge after the RCCL all-gather) == lynse_hip_merge_topk (host) == the
    oracle's VectorStore::merge_results order, with ties and short blocks."""
    import ctypes as C

    import torch

    from lynsedb_amd.sharded import ShardedFlat, block_layout

    rng = np.random.default_rng(world * 100 + nq + k)
    ro, do, co, total = block_layout(nq, k)
    blocks, per_rank = [], []
    for r in range(world):
        rows = (rng.permutation(nq * k).reshape(nq, k).astype(np.uint64) * world + r)
        d = rng.integers(0, 5, size=(nq, k)).astype(f32)  # heavy ties
        d.sort(axis=1)
        if metric == IP:
            d = d[:, ::-1].copy()
        c = rng.integers(0, k + 1, size=nq).astype(np.uint32)
        per_rank.append((rows, d, c))
        blocks.append(ShardedFlat.pack_block(rows, d, c))
    gathered = torch.as_tensor(np.concatenate(blocks), device="cuda")
    out_r = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    out_d = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    out_c = torch.zeros(nq, dtype=torch.int32, device="cuda")
    L._lib.check(L._lib.lib.lynse_hip_merge_topk_device(
        C.c_void_p(gathered.data_ptr()), total, ro, do, co, world, nq, k, metric, C.c_void_p(out_r.data_ptr()),
        C.c_void_p(out_d.data_ptr()), C.c_void_p(out_c.data_ptr()), None))
    torch.cuda.synchronize()
    g_r, g_d, g_c = out_r.cpu().numpy().view(np.uint64), out_d.cpu().numpy(), out_c.cpu().numpy().view(np.uint32)
    for q in range(nq):
        ids = np.stack([per_rank[r][0][q] for r in range(world)])
        ds = np.stack([per_rank[r][1][q] for r in range(world)])
        cs = np.array([per_rank[r][2][q] for r in range(world)], np.uint32)
        h_i, h_d = L.merge_topk(ids, ds, cs, k, metric)
        flat_i = np.concatenate([ids[r, :cs[r]] for r in range(world)])
        flat_d = np.concatenate([ds[r, :cs[r]] for r in range(world)])
        o_i, o_d = oracle.merge_results(flat_i, flat_d, k, metric) if flat_i.size else (np.zeros(0, np.uint64), np.zeros(0, f32))
        assert int(g_c[q]) == len(h_i) == len(o_i)
        assert np.array_equal(g_r[q, :len(h_i)], h_i) and np.array_equal(h_i, o_i)
        assert np.array_equal(g_d[q, :len(h_i)], h_d) and np.array_equal(h_d, o_d)


def test_sharded_flat_world1_and_row_map(L, oracle):
    """ShardedFlat on one rank (device-resident queries/outputs) + the shard row map g = l*stride+offset."""
    import torch

    from lynsedb_amd.sharded import ShardedFlat

    rng = np.random.default_rng(12)
    n, dim, k = 9000, 40, 10
    data = rng.random((n, dim), dtype=f32)
    sh = ShardedFlat(dim, rank=0, world=1, device=0, group=None)
    sh.add_global_rows(data)
    q = rng.random((6, dim), dtype=f32)
    rows, dists, counts = sh.search(q, k, IP)
    for i in range(6):
        e_i, e_d = oracle.canonical_topk(q[i], data, k, IP)
        assert np.array_equal(rows[i].astype(np.uint32), e_i) and np.array_equal(dists[i], e_d)
    # emulate rank 1 of 3: local rows are global rows 1, 4, 7, ...; returned ids must be global
    part = L.FlatIndex(None, dim, 0)
    part.set_row_map(3, 1)
    part.write(np.ascontiguousarray(data[1::3]))
    r, d, c = part.search_batch_arrays(q[:2], k, "l2")
    for i in range(2):
        e_i, e_d = oracle.canonical_topk(q[i], np.ascontiguousarray(data[1::3]), k, L2)
        assert np.array_equal(r[i], e_i.astype(np.uint64) * 3 + 1) and np.array_equal(d[i], e_d)


# ------------------------------------------------------------------ filtered search
@pytest.mark.parametrize("strategy", ["auto", "direct", "mask"])
@pytest.mark.parametrize("metric", [O.IP, O.L2, O.COS])
@pytest.mark.parametrize("n,dim,m,k,nq", [
    (200, 4, 1, 1, 1), (5000, 32, 50, 10, 3), (5000, 32, 2500, 10, 5), (20000, 64, 10000, 25, 40),
    (70000, 48, 60000, 10, 9), (70000, 48, 7, 10, 2), (150000, 24, 100000, 100, 4), (3000, 17, 3000, 8, 33),
    (60000, 1100, 20000, 10, 2),
])
def test_filtered_search_parity(L, oracle, metric, n, dim, m, k, nq, strategy, monkeypatch):
    # "direct" = exact scores of the listed rows (<= 50,000 ids), "mask" = bitmask applied in the scan epilogue
    if strategy != "auto":
        monkeypatch.setenv("LYNSE_HIP_FILTER_STRATEGY", "1" if strategy == "direct" else "2")
    rng = np.random.default_rng(n + dim + m)
    data = rng.standard_normal((n, dim)).astype(f32)
    queries = (data[rng.integers(0, n, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(f32)
    subset = np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
    idx = L.FlatIndex(None, dim, 0)
    idx.write(data)
    rows, dists, counts = idx.search_filtered_batch_arrays(queries, k, NAME[metric], subset)
    for qi in range(nq):
        e_ids, e_d = oracle.canonical_topk_filtered(queries[qi], data, k, metric, subset)
        c = int(counts[qi])
        assert c == len(e_ids) == min(k, m)
        assert np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32)), (qi, dists[qi, :c], e_d)
        assert np.array_equal(rows[qi, :c].astype(np.uint32), e_ids)
        # continuous random data has no exact ties: the reference's own policy gives the same answer
        r_ids, r_d = oracle.flat_search_filtered(queries[qi], data, k, metric, subset)
        assert np.array_equal(r_ids, e_ids) and np.array_equal(r_d.view(np.uint32), e_d.view(np.uint32))


def test_filtered_search_reference_kat_and_edges(L, oracle):
    # vector_store.rs:1309-1329: 200 rows of dim 4 (0..399 written twice), L2, subset [100] -> row 100
    data = np.concatenate([np.arange(400, dtype=f32)] * 2).reshape(200, 4)
    idx = L.FlatIndex(None, 4, 0)
    idx.write(data)
    q = np.array([0, 1, 2, 3], f32)
    ids, d = idx.search_filtered(q, 1, "l2", [100])
    assert ids.tolist() == [100] and d.tolist() == [0.0]
    # empty subset -> empty; ids >= n are skipped; duplicates count once; k is clamped to the subset length
    ids, d = idx.search_filtered(q, 5, "l2", [])
    assert ids.size == 0
    ids, d = idx.search_filtered(q, 5, "l2", [7, 7, 1000, 3, 7])
    e_ids, e_d = oracle.canonical_topk_filtered(q, data, 5, O.L2, [7, 7, 1000, 3, 7])
    assert np.array_equal(ids, e_ids) and np.array_equal(d, e_d) and sorted(ids.tolist()) == [3, 7]
    # massive ties inside the subset: canonical (distance, row) order
    data2 = np.tile(np.array([[1, 0, 0, 0], [0, 1, 0, 0]], f32), (20000, 1))
    idx2 = L.FlatIndex(None, 4, 0)
    idx2.write(data2)
    subset = np.arange(1, 40000, 2, dtype=np.uint64)  # all the [0,1,0,0] rows
    ids, d = idx2.search_filtered(np.array([0, 1, 0, 0], f32), 10, "ip", subset)
    assert ids.tolist() == list(range(1, 21, 2)) and np.all(d == 1.0)


@pytest.mark.parametrize("metric", [O.HAMMING, O.JACCARD, O.DICE])
@pytest.mark.parametrize("n,dim,m,k,nq", [(6000, 128, 3000, 10, 4), (50000, 256, 200, 5, 2), (40000, 100, 30000, 50, 35)])
def test_filtered_binary_search_parity(L, oracle, metric, n, dim, m, k, nq):
    rng = np.random.default_rng(n + dim + m + metric)
    data = (rng.random((n, dim)) < 0.4).astype(f32)
    queries = data[rng.integers(0, n, nq)].copy()
    flip = rng.random(queries.shape) < 0.1
    queries = np.where(flip, 1 - queries, queries).astype(f32)
    subset = np.sort(rng.choice(n, m, replace=False)).astype(np.uint64)
    idx = L.FlatIndex(None, dim, 0)
    idx.write(data)
    rows, dists, counts = idx.search_filtered_batch_arrays(queries, k, NAME[metric], subset)
    words = oracle.pack_binary(data)
    for qi in range(nq):
        qw = oracle.pack_binary(queries[qi].reshape(1, -1))[0]
        e_ids, e_d = oracle.canonical_topk_filtered(None, None, k, metric, subset, packed_query=qw, packed_rows=words)
        c = int(counts[qi])
        assert c == len(e_ids)
        assert np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32))
        assert np.array_equal(rows[qi, :c].astype(np.uint32), e_ids)
        # the reference's strict subset-order admission keeps the earliest rows among ties == canonical for a sorted subset
        r_ids, r_d = oracle.packed_search_filtered(qw, words, k, metric, subset)
        assert np.array_equal(r_d, e_d) and np.array_equal(r_ids, e_ids)


@pytest.mark.parametrize("strategy", ["auto", "direct", "mask"])
def test_filtered_search_bitset_words(L, oracle, strategy, monkeypatch):
    # SearchParams.subset is a BitSet (u64 words, bit r of word r // 64): same answer as the id-list entry point;
    # "direct" expands the words to row ids on the device, "mask" uses them as the scan mask
    if strategy != "auto":
        monkeypatch.setenv("LYNSE_HIP_FILTER_STRATEGY", "1" if strategy == "direct" else "2")
    rng = np.random.default_rng(11)
    n, dim = 30000, 40
    data = rng.standard_normal((n, dim)).astype(f32)
    queries = rng.standard_normal((6, dim)).astype(f32)
    member = rng.random(n) < 0.3
    words = np.zeros((n + 63) // 64 + 2, np.uint64)  # two spare words with bits beyond len set: must be ignored
    ids = np.nonzero(member)[0].astype(np.uint64)
    np.bitwise_or.at(words, (ids // 64).astype(np.int64), np.uint64(1) << (ids % np.uint64(64)))
    words[-1] = np.uint64(0xFFFF)
    idx = L.FlatIndex(None, dim, 0)
    idx.write(data)
    rows, dists, counts = idx.search_filtered_bitset_batch_arrays(queries, 12, "l2", words)
    for qi in range(queries.shape[0]):
        e_ids, e_d = oracle.canonical_topk_filtered(queries[qi], data, 12, O.L2, ids)
        assert int(counts[qi]) == 12
        assert np.array_equal(rows[qi].astype(np.uint32), e_ids) and np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32))


@pytest.mark.parametrize("metric", [O.IP, O.L2])
def test_insertion_ordered_shard_needs_no_fallback(L, oracle, metric):
    # rows sorted so that every later row is a better match than all earlier ones: the contiguous stage plan would
    # overflow at every stage; the sampled first stage gives a representative threshold and no query falls back
    rng = np.random.default_rng(77)
    n, dim, nq, k = 300_000, 16, 40, 10
    data = rng.standard_normal((n, dim)).astype(f32)
    q0 = rng.standard_normal(dim).astype(f32)
    s = data @ q0 if metric == O.IP else -((data - q0) ** 2).sum(1)
    data = np.ascontiguousarray(data[np.argsort(s, kind="stable")])
    queries = (q0 + 0.01 * rng.standard_normal((nq, dim))).astype(f32)
    idx = L.FlatIndex(None, dim, 0)
    idx.write(data)
    idx.profile_enable(True)
    idx.profile_get(reset=True)
    rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
    prof = idx.profile_get(reset=True)
    assert prof["fallback_queries"] == 0
    for qi in (0, 7, 39):
        e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, metric)
        assert np.array_equal(rows[qi].astype(np.uint32), e_ids)
        assert np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32))


@pytest.mark.parametrize("metric", [IP, L2, HAM])
def test_large_k_up_to_the_server_cap(L, oracle, metric):
    """k beyond the candidate capacity of one pass (k > cap / 4 over more than cap rows): the reference accepts any k
    (k.min(n), flat_mmap.rs:836; the server caps requests at MAX_TOP_K = 10,000, src/server/mod.rs:46)."""
    rng = np.random.default_rng(21 + metric)
    n, dim = 50_000, 24
    data = rng.standard_normal((n, dim)).astype(f32) if metric != HAM else (rng.random((n, 64)) < 0.5).astype(f32)
    queries = data[rng.integers(0, n, 3)] + (0.05 if metric != HAM else 0.0)
    queries = np.ascontiguousarray(queries, f32)
    idx = make_index(L, data)
    for k in (10_000, 5000):
        rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
        for qi in range(queries.shape[0]):
            e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, metric)
            assert int(counts[qi]) == k == len(e_ids)
            assert np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32)), (metric, k, qi)
            assert np.array_equal(rows[qi].astype(np.uint32), e_ids), (metric, k, qi)
    # device API + k > n
    import torch

    dev = torch.device("cuda", 0)
    k = 60_000
    dq = torch.as_tensor(queries[:1], device=dev)
    r = torch.zeros((1, k), dtype=torch.int64, device=dev)
    d = torch.zeros((1, k), dtype=torch.float32, device=dev)
    c = torch.zeros(1, dtype=torch.int32, device=dev)
    idx.search_device(dq, k, NAME[metric], r, d, c)
    torch.cuda.synchronize()
    assert int(c[0]) == n
    e_ids, e_d = oracle.canonical_topk(queries[0], data, n, metric)
    assert np.array_equal(r[0, :n].cpu().numpy().astype(np.uint32), e_ids) and np.array_equal(d[0, :n].cpu().numpy().view(np.uint32), e_d.view(np.uint32))


@pytest.mark.parametrize("metric", [IP, L2])
def test_large_k_with_a_subset_filter(L, oracle, metric):
    """k beyond one pass's candidate capacity together with a subset (search_filtered: k.min(subset.len()),
    flat_mmap.rs:498-501): row-range views, each searched with the subset ids that fall into it."""
    rng = np.random.default_rng(33 + metric)
    n, dim = 45_000, 16
    data = rng.standard_normal((n, dim)).astype(f32)
    queries = np.ascontiguousarray(data[rng.integers(0, n, 2)] + 0.05, f32)
    idx = make_index(L, data)
    subset = np.sort(rng.choice(n, 30_000, replace=False)).astype(np.uint64)
    for k in (6000, 40_000):          # > cap / 4 = 4096; the second is clamped to the subset length
        rows, dists, counts = idx.search_filtered_batch_arrays(queries, k, NAME[metric], subset)
        for qi in range(queries.shape[0]):
            e_ids, e_d = oracle.canonical_topk_filtered(queries[qi], data, k, metric, subset)
            c = int(counts[qi])
            assert c == len(e_ids) == min(k, subset.size)
            assert np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32)), (metric, k, qi)
            assert np.array_equal(rows[qi, :c].astype(np.uint32), e_ids), (metric, k, qi)
    # BitSet words, a subset confined to two row ranges, duplicates / out-of-range ids in the list form
    words = np.zeros((n + 63) // 64, np.uint64)
    some = np.concatenate([np.arange(100, 9000), np.arange(20_000, 26_000)]).astype(np.uint64)
    np.bitwise_or.at(words, (some // 64).astype(np.int64), np.uint64(1) << (some % 64))
    rows, dists, counts = idx.search_filtered_bitset_batch_arrays(queries, 5000, NAME[metric], words)
    rows2, dists2, counts2 = idx.search_filtered_batch_arrays(queries, 5000, NAME[metric], np.concatenate([some, some[:50], [n + 5]]).astype(np.uint64))
    for qi in range(queries.shape[0]):
        e_ids, e_d = oracle.canonical_topk_filtered(queries[qi], data, 5000, metric, some)
        for r, d, c in ((rows, dists, counts), (rows2, dists2, counts2)):
            assert int(c[qi]) == 5000
            assert np.array_equal(r[qi].astype(np.uint32), e_ids) and np.array_equal(d[qi].view(np.uint32), e_d.view(np.uint32))


def test_sharded_search_entry_point_with_a_one_rank_communicator(L, oracle):
    """The C-ABI exchange path (lynse_hip_comm_* + lynse_hip_flat_search_sharded_f32_device) on the one GPU a test box has:
    RCCL is loaded at run time, a 1-rank communicator is created from its own unique id, the self-check sees 1 rank, and
    the sharded entry point returns the answers of the plain search (global rows through the row map)."""
    import torch

    from lynsedb_amd.sharded import NativeComm, ShardedFlat

    rng = np.random.default_rng(17)
    n, dim, nq, k = 30_000, 64, 40, 10
    data = rng.standard_normal((n, dim)).astype(f32)
    queries = (data[rng.integers(0, n, nq)] + 0.05).astype(f32)
    sh = ShardedFlat(dim, rank=0, world=1, device=0, group=None)
    sh.index.write(data)
    sh.index.finalize()
    sh.comm = NativeComm(None, 0, 1, 0)
    assert sh.comm.ranks_seen() == 1
    dev = torch.device("cuda", 0)
    out = sh.alloc_outputs(nq, k)
    dq = torch.as_tensor(queries, device=dev)
    for name, metric in (("ip", IP), ("l2", L2)):
        m = L.metric_from_str(name)
        import ctypes as C
        L._lib.check(L._lib.lib.lynse_hip_flat_search_sharded_f32_device(
            sh.index.handle, sh.comm.handle, C.c_void_p(dq.data_ptr()), nq, k, m, C.c_void_p(out.rows.data_ptr()),
            C.c_void_p(out.dists.data_ptr()), C.c_void_p(out.counts.data_ptr())))
        torch.cuda.synchronize()
        rows = out.rows.cpu().numpy().view(np.uint64)
        dists = out.dists.cpu().numpy()
        for qi in (0, 7, nq - 1):
            e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, metric)
            assert np.array_equal(rows[qi].astype(np.uint32), e_ids) and np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32))


@pytest.mark.parametrize("n,bits,nq,k,p_one", [(200_000, 1024, 256, 50, 0.5), (150_000, 200, 130, 10, 0.3), (70_000, 64, 100, 64, 0.5),
                                              (300_000, 1152, 97, 5, 0.1), (100_000, 4096, 72, 20, 0.5)])
def test_batched_hamming_on_the_matrix_pipe_equals_the_popcount_kernels(L, oracle, n, bits, nq, k, p_one):
    """Hamming batches of >= 72 queries run as an exact +-1 GEMM in FP4 on the MFMA (popcount(x ^ q) = (D - dot) / 2, strict
    cut, no rescoring; packed_binary_search, flat_mmap.rs:1345-1409).  Same ids and distances as the oracle and as the
    popcount kernels a small batch takes — widths that are no multiple of 256 (zero nibbles in the pad columns), narrow rows
    with huge tie groups at the k-th distance, sparse rows."""
    from lynsedb_amd.datasets import packed_bernoulli

    words = packed_bernoulli(n, bits, p_one, 100 + bits)
    if bits % 64:
        words[:, -1] &= np.uint64((1 << (bits % 64)) - 1)
    idx = L.FlatIndex(None, bits)
    idx.write_packed(words)
    idx.finalize()
    rng = np.random.default_rng(bits)
    qw = words[rng.integers(0, n, nq)].copy()
    qw[:, 0] ^= np.uint64(0x5A5A)
    rows, dists, counts = idx.search_packed_arrays(qw, k, "hamming")          # MFMA path
    r_small = [idx.search_packed_arrays(qw[i:i + 32], k, "hamming") for i in range(0, nq, 32)]   # <= 32 queries: popcount kernels
    rs = np.concatenate([r[0] for r in r_small]); ds = np.concatenate([r[1] for r in r_small]); cs = np.concatenate([r[2] for r in r_small])
    assert np.array_equal(rows, rs) and np.array_equal(dists.view(np.uint32), ds.view(np.uint32)) and np.array_equal(counts, cs)
    for qi in sorted({0, nq // 2, nq - 1}):
        e_ids, e_d = oracle.canonical_topk_packed(qw[qi], words, k, O.HAMMING)
        c = int(counts[qi])
        assert c == len(e_ids) and np.array_equal(rows[qi, :c].astype(np.uint32), e_ids) and np.array_equal(dists[qi, :c], e_d), qi
    assert idx.coarse_state()["bpm_rows"] == n                                 # the +-1 copy was built: the MFMA path ran


def _nan_rule_model(oracle, q, data, k, metric):
    """The order include/lynse_hip.h pins for non-finite scores: the reference kernels' own arithmetic per (query, row) — the oracle's
    single-pair kernels propagate NaN / inf like simd.rs does —, a NaN score replaced by the WORST value of the metric, then the
    canonical (score best-first, row ascending) order."""
    d = np.asarray(oracle.all_distances(q, data, metric), f32)
    asc = metric != IP
    d = np.where(np.isnan(d), f32(np.inf if asc else -np.inf), d)
    order = np.lexsort((np.arange(len(d)), d if asc else -d))
    return order[:k].astype(np.uint64), d[order[:k]]


@pytest.mark.parametrize("n,dim,nq,k", [(64, 96, 1, 64), (64, 96, 8, 64), (64, 96, 40, 64), (5000, 96, 3, 20), (5000, 96, 40, 20),
                                         (100_000, 128, 2, 10), (300_000, 128, 64, 10), (300_000, 256, 64, 10)])
def test_nan_and_infinite_rows_and_queries(L, oracle, n, dim, nq, k):
    """NaN / +-inf elements in rows and in queries through lynse_hip_flat_search_f32 (VERDICT r5 'weak' 3): the reference leaves the order
    of NaN distances to partial_cmp(..).unwrap_or(Equal) (flat_mmap.rs:2141-2149) — here ONE order is pinned, on every path the shapes
    below reach (the fused few-query search, the <= 32-query tiling, the 33..256-query tilings with the int8 pass switched off by the
    non-finite rows, sampled and contiguous plans): NaN = the worst value of the metric, ties by row; a query that scores NaN against
    every row gets rows 0 .. k - 1."""
    rng = np.random.default_rng(3 + n + nq)
    data = rng.standard_normal((n, dim)).astype(f32)
    sp = rng.choice(n, 12, replace=False)
    data[sp[0:4], 3] = np.nan
    data[sp[4:6], 5] = np.inf
    data[sp[6:8], 5] = -np.inf
    data[sp[8], 1] = np.inf
    data[sp[8], 2] = -np.inf                                # (IP / L2 of this row: inf - inf)
    queries = rng.standard_normal((nq, dim)).astype(f32)
    queries[:, 1] = np.abs(queries[:, 1])
    queries[:, 2] = np.abs(queries[:, 2])
    if nq > 1:
        queries[-1, 7] = np.nan                             # every score NaN
    if nq > 2:
        queries[-2, 9] = np.inf                             # IP: +-inf / NaN by row; L2: +inf everywhere; cosine: NaN everywhere
    idx = make_index(L, data)
    idx.finalize()
    for metric in (IP, L2, COS):
        for fused in ((True, False) if (nq <= 4 and k <= 64) else (True,)):
            idx.set_fused_search(fused)
            try:
                rows, dists, counts = idx.search_batch_arrays(queries, k, NAME[metric])
            finally:
                idx.set_fused_search(True)
            for qi in range(nq):
                e_r, e_d = _nan_rule_model(oracle, queries[qi], data, k, metric)
                c = int(counts[qi])
                assert c == len(e_r), (NAME[metric], fused, qi, c)
                assert np.array_equal(rows[qi, :c].astype(np.uint64), e_r), (NAME[metric], fused, qi, rows[qi, :8], e_r[:8])
                assert np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32)), (NAME[metric], fused, qi, dists[qi, :8], e_d[:8])
    if nq > 1:      # the all-NaN query of the pinned rule, spelled out
        rows, dists, counts = idx.search_batch_arrays(queries, k, "cosine")
        assert int(counts[-1]) == min(k, n) and np.array_equal(rows[-1, :min(k, n)], np.arange(min(k, n), dtype=rows.dtype)) and np.all(np.isposinf(dists[-1, :min(k, n)]))
