"""GPU: a collection directory / IvfFlat file pair in the reference's on-disk formats is opened into HBM and searched;
results equal the oracle's on the same bytes."""
import numpy as np
import pytest

import oracle as O
from lynsedb_amd import storage as S

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1
    return L_


def test_open_segmented_collection_and_search(L, oracle, tmp_path):
    # vector_store.rs:1309-1329: rows 0..399 written twice (segment target 1024 B) -> 200 rows in two segments
    data = np.arange(400, dtype=f32).reshape(100, 4)
    S.write_flat_collection(tmp_path, [data, data], ids=np.arange(1000, 1200, dtype=np.uint64), segment_target_bytes=1024)
    idx, id_map, m = S.open_flat_collection(tmp_path, 4)
    assert len(idx) == 200 and len(m.segments) == 2
    q = np.array([0, 1, 2, 3], f32)
    ids, d = idx.search(q, 1, "l2")
    assert ids.tolist() == [0]  # canonical tie-break: rows 0 and 100 are identical, the lower row wins
    ids, d = idx.search_filtered(q, 1, "l2", [100])
    assert ids.tolist() == [100] and S.rows_to_user_ids(ids, id_map).tolist() == [1100]
    # a larger random collection split over several segments
    rng = np.random.default_rng(3)
    big = rng.standard_normal((5000, 24)).astype(f32)
    S.write_flat_collection(tmp_path / "big", [big[:1800], big[1800:1900], big[1900:]], segment_target_bytes=100_000)
    idx2, id_map2, m2 = S.open_flat_collection(tmp_path / "big", 24)
    assert len(m2.segments) >= 2 and len(idx2) == 5000 and id_map2.size == 0
    qs = rng.standard_normal((5, 24)).astype(f32)
    rows, dists, counts = idx2.search_batch_arrays(qs, 10, "ip")
    for i in range(5):
        e_ids, e_d = oracle.canonical_topk(qs[i], big, 10, O.IP)
        assert np.array_equal(rows[i].astype(np.uint32), e_ids) and np.array_equal(dists[i].view(np.uint32), e_d.view(np.uint32))
    assert S.rows_to_user_ids(rows[0], id_map2).tolist() == rows[0].tolist()


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_ivf_flat_files_roundtrip(L, oracle, tmp_path, metric):
    # IvfFlatMmap::build writes <data> (slab order) + <data>.ivf_meta.bin; reopen (ivf_flat_mmap.rs:752-773) and search
    rng = np.random.default_rng(9)
    n, dim, nlist = 3000, 16, 24
    centers = rng.standard_normal((12, dim)).astype(f32)
    data = (centers[rng.integers(0, 12, n)] + 0.2 * rng.standard_normal((n, dim))).astype(f32)
    built = L.IvfFlatIndex.build(None, data, dim, nlist, 10, metric, l2_partitions=True)
    meta = S.save_ivf_flat(tmp_path / "vecs.bin", built, data)
    assert (tmp_path / "vecs.ivf_meta.bin").exists()
    assert meta.partition_offsets[-1] == n and sorted(meta.original_ids.tolist()) == list(range(n))
    slab = np.fromfile(tmp_path / "vecs.bin", "<f4").reshape(n, dim)
    assert np.array_equal(slab, data[meta.original_ids.astype(np.int64)])  # rows physically grouped by partition
    reopened = S.open_ivf_flat(tmp_path / "vecs.bin", metric)
    cen2, asg2, off2, orig2 = reopened.export()
    assert np.array_equal(off2, meta.partition_offsets) and np.array_equal(orig2, meta.original_ids)
    qs = (data[rng.integers(0, n, 6)] + 0.05 * rng.standard_normal((6, dim))).astype(f32)
    a = built.search_batch_arrays(qs, 10, 4)
    b = reopened.search_batch_arrays(qs, 10, 4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    mid = O.IP if metric == "ip" else O.L2
    rd = oracle.ivf_routing_dims(meta.centroids)
    for i in range(6):
        e_ids, e_d = oracle.ivf_flat_search(qs[i], slab, meta.centroids, meta.partition_offsets, meta.original_ids, 4, 10, mid, routing_dims=rd)
        c = int(b[2][i])
        assert np.array_equal(b[0][i, :c].astype(np.uint64), np.asarray(e_ids, np.uint64)) and np.array_equal(b[1][i, :c].view(np.uint32), e_d.view(np.uint32))


def test_flushed_plus_pending_rows_and_tombstones(L, oracle):
    # Collection::search tail (src/engine.rs:4797-4822): flushed rows from the store, pending rows scored with
    # top_k_search, merge_row_results, then filter_tombstoned_limit — equals a brute-force answer over the live rows
    from lynsedb_amd import shard_node as N

    rng = np.random.default_rng(21)
    dim, n_flushed, n_pending, k = 20, 4000, 300, 10
    allrows = rng.standard_normal((n_flushed + n_pending, dim)).astype(f32)
    idx = L.FlatIndex(None, dim, 0)
    idx.write(allrows[:n_flushed])
    pending_rows = np.arange(n_flushed, n_flushed + n_pending, dtype=np.uint64)
    tomb = rng.choice(n_flushed + n_pending, 400, replace=False).astype(np.uint64)
    for qi in range(4):
        q = rng.standard_normal(dim).astype(f32)
        search_k = k + tomb.size  # the engine over-fetches by the tombstone count
        f_ids, f_d = idx.search(q, search_k, "l2")
        p_idx, p_d = L.py_top_k_search(q, allrows[n_flushed:], "l2", search_k)
        m_ids, m_d = N.merge_row_results(f_ids, f_d, pending_rows[p_idx.astype(np.int64)], p_d, search_k, "l2")
        ids, d = N.filter_tombstoned_limit(m_ids, m_d, tomb, k)
        live = np.setdiff1d(np.arange(n_flushed + n_pending), tomb.astype(np.int64))
        e_ids, e_d = oracle.canonical_topk(q, allrows[live], k, O.L2, O.IPFORM_SINGLE)
        assert np.array_equal(ids, live[e_ids.astype(np.int64)].astype(np.uint64))
        assert np.allclose(d, e_d, rtol=1e-6, atol=0)


# ------------------------------------------------------------------ VectorDtype::F16 storage
@pytest.mark.parametrize("metric,name", [(O.IP, "ip"), (O.L2, "l2"), (O.COS, "cosine")])
@pytest.mark.parametrize("n,dim,nq,k", [(3000, 24, 5, 10), (40000, 100, 33, 10), (90000, 33, 3, 25)])
def test_f16_storage_search_parity(L, oracle, metric, name, n, dim, nq, k):
    # rows live as f16 (rounded like encode_f32_slice_as_le_bytes); distances follow simd::*_f16 (sequential f32 sums)
    rng = np.random.default_rng(n + dim)
    data = (rng.standard_normal((n, dim)) * 3).astype(f32)
    data[5] = 0  # a zero row: cosine distance 1.0 by the `== 0` rule of cosine_distance_f16
    queries = (data[rng.integers(0, n, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(f32)
    decoded = oracle.round_f16(data)
    idx = L.FlatIndex(None, dim, 0, dtype="f16")
    idx.write(data[: n // 2])                                          # f32 in: rounded on the device
    idx.write_f16_bits(data[n // 2:].astype(np.float16).view(np.uint16))  # the bytes of an F16 segment file
    assert np.array_equal(idx.read_rows(0, n), decoded)
    rows, dists, counts = idx.search_batch_arrays(queries, k, name)
    for qi in range(nq):
        e_ids, e_d = oracle.canonical_topk_f16(queries[qi], decoded, k, metric)
        assert np.array_equal(rows[qi].astype(np.uint32), e_ids), (qi, rows[qi], e_ids)
        assert np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32)), (qi, dists[qi], e_d)
    # filtered search on the f16 store (search_filtered_f16, flat_mmap.rs:5329-5437) uses the same kernels
    subset = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.uint64)
    frows, fd, fc = idx.search_filtered_batch_arrays(queries[:2], k, name, subset)
    for qi in range(2):
        sub_ids, sub_d = oracle.canonical_topk_f16(queries[qi], decoded[subset.astype(np.int64)], k, metric)
        assert np.array_equal(frows[qi], subset[sub_ids.astype(np.int64)]) and np.array_equal(fd[qi].view(np.uint32), sub_d.view(np.uint32))


def test_open_f16_collection(L, oracle, tmp_path):
    rng = np.random.default_rng(5)
    data = rng.standard_normal((2500, 12)).astype(f32)
    S.write_flat_collection(tmp_path, [data[:1000], data[1000:]], segment_target_bytes=20_000, dtype="f16")
    m = S.load_manifest(tmp_path, 12, "f16")
    assert sum(s.rows for s in m.segments) == 2500 and len(m.segments) == 2
    idx, id_map, _ = S.open_flat_collection(tmp_path, 12, "float16")
    assert idx.dtype == "f16" and len(idx) == 2500
    q = rng.standard_normal(12).astype(f32)
    ids, d = idx.search(q, 7, "l2")
    e_ids, e_d = oracle.canonical_topk_f16(q, oracle.round_f16(data), 7, O.L2)
    assert np.array_equal(ids, e_ids) and np.array_equal(d.view(np.uint32), e_d.view(np.uint32))
    with pytest.raises(ValueError):
        L.FlatIndex(None, 4, 0, dtype="int8")


@pytest.mark.parametrize("metric,name,k", [(O.IP, "ip", 300), (O.L2, "l2", 250), (O.COS, "cosine", 60)])
def test_sq8_more_candidates_than_one_pass_holds(L, oracle, metric, name, k):
    """n_cand = 20 k (cosine 100 k, flat_mmap.rs:5883-5893) beyond cap / 4 over more than cap rows: pass 1 per row range, host
    merge in (code score, row) order, pass 2 as the subset-filtered exact search."""
    rng = np.random.default_rng(77 + metric)
    n, dim, nq = 40_000, 24, 3
    data = rng.standard_normal((n, dim)).astype(f32)
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    idx = L.FlatIndex(None, dim, 0)
    idx.write(data)
    mins, scales, codes = oracle.sq8_fit(data)
    rows, dists, counts = idx.search_sq8_batch_arrays(queries, k, name)
    for qi in range(nq):
        e_ids, e_d = oracle.sq8_search(queries[qi], data, mins, scales, codes, k, metric)
        assert int(counts[qi]) == k == len(e_ids)
        assert np.array_equal(rows[qi].astype(np.uint32), e_ids), (qi, rows[qi][:8], e_ids[:8])
        assert np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32))


# ------------------------------------------------------------------ FLAT-*-SQ8 two-pass mode
@pytest.mark.parametrize("metric,name", [(O.IP, "ip"), (O.L2, "l2"), (O.COS, "cosine")])
@pytest.mark.parametrize("n,dim,nq,k,kind", [(3000, 32, 6, 10, "normal"), (40000, 96, 33, 5, "normal"), (70000, 40, 4, 10, "positive"),
                                             (150, 16, 3, 10, "normal"), (25000, 130, 5, 3, "const")])
def test_sq8_two_pass_parity(L, oracle, metric, name, n, dim, nq, k, kind):
    rng = np.random.default_rng(n + dim + metric)
    if kind == "positive":
        data = rng.random((n, dim)).astype(f32)
    else:
        data = rng.standard_normal((n, dim)).astype(f32)
    if kind == "const":
        data[:, 3] = 1.5      # constant dimension: scale 0, code 0
        data[:, 7] *= 1e-20   # range below 1e-30
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(f32)
    queries[0] *= 3  # a query outside the rows' range: codes clamp to 0 / 255
    idx = L.FlatIndex(None, dim, 0)
    idx.write(data)
    mins, scales, codes = oracle.sq8_fit(data)
    g_mins, g_scales = idx.sq8_params()
    assert np.array_equal(g_mins.view(np.uint32), mins.view(np.uint32)) and np.array_equal(g_scales.view(np.uint32), scales.view(np.uint32))
    rows, dists, counts = idx.search_sq8_batch_arrays(queries, k, name)
    for qi in range(nq):
        e_ids, e_d = oracle.sq8_search(queries[qi], data, mins, scales, codes, k, metric)
        c = int(counts[qi])
        assert c == len(e_ids)
        assert np.array_equal(rows[qi, :c].astype(np.uint32), e_ids), (qi, rows[qi, :c], e_ids)
        assert np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32))
    # appending rows changes the collection-wide min / max: the codes are rebuilt (ensure_sq8 after a write)
    extra = (rng.standard_normal((500, dim)) * 4).astype(f32)
    idx.write(extra)
    both = np.concatenate([data, extra])
    mins2, scales2, codes2 = oracle.sq8_fit(both)
    rows, dists, counts = idx.search_sq8_batch_arrays(queries[:2], k, name)
    for qi in range(2):
        e_ids, e_d = oracle.sq8_search(queries[qi], both, mins2, scales2, codes2, k, metric)
        assert np.array_equal(rows[qi, :len(e_ids)].astype(np.uint32), e_ids) and np.array_equal(dists[qi, :len(e_ids)].view(np.uint32), e_d.view(np.uint32))


def test_hand_assembled_reference_layout_opens_and_searches(L, oracle, tmp_path, golden_dir):
    """The hand-assembled directory of tests/golden/reference_layout_fixture.json (see tests/test_storage_formats.py) opened
    into an HBM shard: searches equal the oracle's over the rows the fixture's closed form defines; ids go through id_map.bin."""
    from test_storage_formats import _materialise_reference_layout

    fx = _materialise_reference_layout(tmp_path, golden_dir)
    dim, n = fx["dim"], fx["rows"]
    idx, id_map, m = S.open_flat_collection(tmp_path, dim)
    assert len(idx) == n and len(m.segments) == 3
    want = np.array([[((r * 37 + d * 11) % 101 - 50) * 0.125 for d in range(dim)] for r in range(n)], f32)
    rng = np.random.default_rng(8)
    for name, metric in (("ip", O.IP), ("l2", O.L2), ("cosine", O.COS)):
        for _ in range(3):
            q = rng.standard_normal(dim).astype(f32)
            ids, d = idx.search(q, 7, name)
            e_ids, e_d = oracle.canonical_topk(q, want, 7, metric)
            assert np.array_equal(ids, e_ids) and np.array_equal(d.view(np.uint32), e_d.view(np.uint32))
    ids, _ = idx.search(want[59], 2, "l2")
    assert ids[0] == 59 and S.rows_to_user_ids(ids[:1], id_map).tolist() == [10_000_000_177]
    ids, _ = idx.search(want[92], 1, "l2")
    assert S.rows_to_user_ids(ids, id_map).tolist() == [92]              # past the end of the map: the row itself


def test_f16_shard_holds_one_copy_of_its_rows(L, oracle):
    """VectorDtype::F16 (src/storage/dtype.rs, flat_mmap.rs:187-221): the shard keeps the f16 bits and nothing else of the rows —
    no f32 decode, and (values within the unscaled f16 range) the coarse-pass shadow IS that buffer: 2 B per element + the
    per-row norms, against 4 + 2 B for an f32 shard.  Rows read back decode exactly; appends after a search keep working."""
    rng = np.random.default_rng(12)
    n, dim = 50_000, 96
    data = oracle.round_f16(rng.standard_normal((n, dim)).astype(f32)).reshape(n, dim)
    idx = L.FlatIndex(None, dim, dtype="f16")
    idx.write(data[:30_000])
    idx.finalize()
    q = rng.standard_normal((5, dim)).astype(f32)
    idx.search_batch_arrays(q, 10, "ip")
    idx.write(data[30_000:])
    idx.finalize()
    assert idx.hbm_bytes() <= n * dim * 2 * 1.6 + (1 << 20), idx.hbm_bytes()     # one f16 copy (+ growth slack, norms)
    f32_idx = L.FlatIndex(None, dim)
    f32_idx.write(data)
    f32_idx.finalize()
    assert f32_idx.hbm_bytes() <= n * dim * 4 * 1.3 + (1 << 20), f32_idx.hbm_bytes()   # (round 4: the f16 shadow of an f32 shard is a LAZY copy ...)
    f32_idx.search_batch_arrays(q, 10, "ip")                                              # ... built by the first search that scans it
    assert f32_idx.hbm_bytes() >= n * dim * 6
    assert np.array_equal(idx.read_rows(0, n).view(np.uint32), data.view(np.uint32))
    for name, metric in (("ip", O.IP), ("l2", O.L2), ("cosine", O.COS)):
        rows, dists, counts = idx.search_batch_arrays(q, 10, name)
        for i in range(5):
            e_ids, e_d = oracle.canonical_topk_f16(q[i], data, 10, metric)
            assert np.array_equal(rows[i].astype(np.uint32), e_ids) and np.array_equal(dists[i].view(np.uint32), e_d.view(np.uint32)), (name, i)
    # values beyond 2^15: the shadow is a scaled copy of its own, the results do not change
    big = oracle.round_f16((rng.standard_normal((20_000, dim)) * 9000.0).astype(f32)).reshape(20_000, dim)
    idx2 = L.FlatIndex(None, dim, dtype="f16")
    idx2.write(big)
    idx2.finalize()
    rows, dists, counts = idx2.search_batch_arrays(q, 10, "l2")
    for i in range(5):
        e_ids, e_d = oracle.canonical_topk_f16(q[i], big, 10, O.L2)
        assert np.array_equal(rows[i].astype(np.uint32), e_ids) and np.array_equal(dists[i].view(np.uint32), e_d.view(np.uint32)), i


@pytest.mark.parametrize("name,metric", [("ip", O.IP), ("l2", O.L2), ("cosine", O.COS)])
def test_f16_shard_on_the_certified_int8_pass(L, oracle, name, metric):
    """F16 shards take the certified int8 coarse pass like f32 ones: the SQ8 codes are built from the exactly decoded halves, the
    survivors are rescored with the f16 kernels' sequential f32 sums (simd.rs:805-846) — 1 B per element streamed instead of 2."""
    rng = np.random.default_rng(77)
    n, dim, k = 300_000, 256, 10
    data = oracle.round_f16((rng.standard_normal((n, dim)) * 2).astype(f32)).reshape(n, dim)
    queries = (data[rng.integers(0, n, 200)] + 0.1 * rng.standard_normal((200, dim))).astype(f32)
    idx = L.FlatIndex(None, dim, dtype="f16")
    for b in range(0, n, 100_000):
        idx.write(data[b:b + 100_000])
    idx.finalize()
    idx.profile_enable(True)
    for nq in (200, 40, 1):
        idx.profile_get(reset=True)
        rows, dists, counts = idx.search_batch_arrays(queries[:nq], k, name)
        p = idx.profile_get(reset=True)
        assert int(p["last_plan"]) & 64 and int(p["last_plan"]) & 4 and p["fallback_queries"] == 0, (name, nq, p)
        for qi in sorted({0, nq // 2, nq - 1}):
            e_ids, e_d = oracle.canonical_topk_f16(queries[qi], data, k, metric)
            assert np.array_equal(rows[qi].astype(np.uint32), e_ids), (name, nq, qi, rows[qi], e_ids)
            assert np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32)), (name, nq, qi)
