"""GPU tests of the engine glue inside `Collection`: `search_with_precomputed_filter`
(src/engine.rs:4718-4833) = k inflated by the tombstones, index / flat / subset-filtered search of the flushed rows,
`pending_search` of the un-flushed rows, `merge_row_results`, row -> user id, `filter_tombstoned_limit`; index-mode
routing (`FLAT-*-SQ8`, flat_mmap.rs:891-905; `IVF-{HAMMING,JACCARD}-BINARY`, src/index/mod.rs:376-385); and the
search-time metric of `IvfFlatIndex.search` (ivf_flat_mmap.rs:225-305).  Expected values come from the CPU oracle."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
f32 = np.float32
NAME = {O.IP: "ip", O.L2: "l2", O.COS: "cosine", O.HAMMING: "hamming", O.JACCARD: "jaccard", O.DICE: "dice"}


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1
    return L_


def expected_live_topk(oracle, q, data, user_ids, k, metric, dead, subset_rows=None):
    """Exact top-k over the rows that are alive (user id not tombstoned) and inside the subset, canonical
    (distance in metric order, row) order — what k + |tombstones| inflation followed by filter_tombstoned_limit yields."""
    rows = np.arange(data.shape[0])
    keep = ~np.isin(user_ids, np.fromiter(dead, np.int64, len(dead)))
    if subset_rows is not None:
        keep &= np.isin(rows, subset_rows)
    rows = rows[keep]
    if rows.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, f32)
    d = oracle.all_distances(q, data, metric, ip_form=O.IPFORM_SINGLE if subset_rows is not None else O.IPFORM_AUTO)[rows]
    order = np.lexsort((rows, -d if metric == O.IP else d))[:k]
    return user_ids[rows[order]], d[order]


@pytest.mark.parametrize("mode,metric", [("FLAT-IP", O.IP), ("FLAT-L2", O.L2), ("FLAT-COS", O.COS)])
def test_collection_pending_tombstones_and_subset(L, oracle, mode, metric):
    rng = np.random.default_rng(11 + metric)
    n_flushed, n_pending, dim, k = 12_000, 300, 48, 10
    data = rng.standard_normal((n_flushed + n_pending, dim)).astype(f32)
    user_ids = (np.arange(n_flushed + n_pending, dtype=np.int64) * 7 + 3)  # user ids differ from row indices
    coll = L.Collection("c", dim)
    coll.add_items(data[:n_flushed], user_ids[:n_flushed].tolist())        # >= PENDING_INGEST_FLUSH_ROWS: flushed at once
    assert coll.pending_len() == 0
    coll.build_index(mode, None)
    coll.add_items(data[n_flushed:], user_ids[n_flushed:].tolist())        # stays in the pending buffer
    assert coll.pending_len() == n_pending and coll.shape() == (n_flushed + n_pending, dim)
    queries = (data[rng.integers(0, data.shape[0], 9)] + 0.01 * rng.standard_normal((9, dim))).astype(f32)
    queries[0] = data[n_flushed + 5]                                        # best hit lives in the pending rows

    # (1) no tombstones: flushed + pending merge
    for qi, res in enumerate(coll.batch_search(queries, k)):
        e_ids, e_d = expected_live_topk(oracle, queries[qi], data, user_ids, k, metric, set())
        assert np.array_equal(res.ids(), e_ids), (mode, qi, res.ids(), e_ids)
        np.testing.assert_allclose(res.distances(), e_d, rtol=1e-5, atol=1e-6)
    assert coll.search(queries[0], 1).ids()[0] == user_ids[n_flushed + 5]

    # (2) tombstones covering the current best hits of every query, flushed and pending alike
    dead = set()
    for qi in range(queries.shape[0]):
        dead.update(int(i) for i in coll.search(queries[qi], 4).ids())
    coll.delete_items(dead)
    assert coll.list_deleted_ids() == sorted(dead)
    for qi, res in enumerate(coll.batch_search(queries, k)):
        e_ids, e_d = expected_live_topk(oracle, queries[qi], data, user_ids, k, metric, dead)
        assert len(res) == k and not (set(res.ids().tolist()) & dead)
        assert np.array_equal(res.ids(), e_ids), (mode, qi, res.ids(), e_ids)
        np.testing.assert_allclose(res.distances(), e_d, rtol=1e-5, atol=1e-6)

    # (3) the same with a precomputed row filter (BitSet and row list), tombstones still in force
    subset_rows = np.sort(rng.choice(n_flushed + n_pending, 4000, replace=False)).astype(np.uint64)
    bits = L.BitSet.from_rows(subset_rows, n_flushed + n_pending)
    assert bits.count() == subset_rows.size and np.array_equal(bits.to_vec(), subset_rows)
    for sub in (bits, subset_rows):
        for qi, res in enumerate(coll.batch_search(queries, k, subset=sub)):
            e_ids, e_d = expected_live_topk(oracle, queries[qi], data, user_ids, k, metric, dead, subset_rows)
            assert np.array_equal(res.ids(), e_ids), (mode, qi, res.ids(), e_ids)
            np.testing.assert_allclose(res.distances(), e_d, rtol=1e-5, atol=1e-6)
    assert len(coll.search(queries[1], k, subset=np.zeros(0, np.uint64))) == 0       # empty subset -> empty result

    # (4) restore + commit: everything flushed, same answers as a plain exact search
    coll.restore_items(dead)
    coll.commit()
    assert coll.pending_len() == 0 and coll.list_deleted_ids() == []
    for qi, res in enumerate(coll.batch_search(queries, k)):
        e_ids, e_d = expected_live_topk(oracle, queries[qi], data, user_ids, k, metric, set())
        assert np.array_equal(res.ids(), e_ids)
    with pytest.raises(NotImplementedError):
        coll.search(queries[0], k, where_expr="x > 1")


@pytest.mark.parametrize("mode,metric", [("FLAT-IP-SQ8", O.IP), ("FLAT-L2-SQ8", O.L2), ("FLAT-COS-SQ8", O.COS)])
def test_collection_sq8_mode_routes_to_the_two_pass_search(L, oracle, mode, metric):
    rng = np.random.default_rng(5)
    n, dim, k = 20_000, 64, 10
    data = rng.random((n, dim), dtype=f32)
    coll = L.Collection("c", dim)
    coll.add_items(data, list(range(n)))
    coll.commit()
    coll.build_index(mode, None)
    queries = (data[rng.integers(0, n, 6)] + 0.02 * rng.standard_normal((6, dim))).astype(f32)
    mins, scales, codes = oracle.sq8_fit(data)
    res = coll.batch_search(queries, k)
    flat = L.FlatIndex(None, dim)
    flat.write(data)
    r2, d2, c2 = flat.search_sq8_batch_arrays(queries, k, NAME[metric])
    for qi in range(queries.shape[0]):
        e_ids, e_d = oracle.sq8_search(queries[qi], data, mins, scales, codes, k, metric)
        assert res[qi].index_mode() == mode
        assert np.array_equal(res[qi].ids(), r2[qi, :int(c2[qi])].astype(np.int64))
        assert np.array_equal(res[qi].distances().view(np.uint32), d2[qi, :int(c2[qi])].view(np.uint32))
        # the oracle's two-pass result: same distances; ids equal wherever the distances are distinct
        assert np.array_equal(res[qi].distances().view(np.uint32), e_d.view(np.uint32)), (qi, res[qi].distances(), e_d)
        assert np.array_equal(res[qi].ids(), e_ids.astype(np.int64)) or len(set(e_d.tolist())) < k


@pytest.mark.parametrize("mode,metric", [("IVF-HAMMING-BINARY", O.HAMMING), ("IVF-JACCARD-BINARY", O.JACCARD)])
def test_collection_binary_ivf_mode_builds_the_binary_index(L, oracle, mode, metric):
    rng = np.random.default_rng(77)
    n, dim, nlist, nprobe, k = 3000, 96, 16, 4, 10
    data = (rng.standard_normal((n, dim)) * 2 + 1).astype(f32)
    coll = L.Collection("c", dim)
    coll.add_items(data, list(range(100, 100 + n)))
    coll.commit()
    coll.build_index(mode, {"n_clusters": nlist, "nprobe": nprobe})
    queries = data[rng.integers(0, n, 7)].copy()
    # the index inside the collection is the binary-quantised IVF: thresholds / centroids / assignments equal the oracle's
    ab, thr = oracle.binary_fit(data)
    g_thr, g_ab = coll._ivf.thresholds()
    assert g_ab == ab and np.array_equal(g_thr.view(np.uint32), thr.view(np.uint32))
    enc = oracle.binary_quantize(data, thr)
    packed = oracle.pack_binary(enc)
    cen, asg, off, orig = coll._ivf.export()
    e_cen, e_asg = oracle.kmeans_train(enc, nlist, 20, O.L2)
    assert np.array_equal(e_asg, asg) and np.array_equal(e_cen.view(np.uint32), cen.view(np.uint32))
    off_o, rows_o = oracle.lists_from_assignments(asg, cen.shape[0])
    res = coll.batch_search(queries, k)            # nprobe=None -> the build default
    for qi in range(queries.shape[0]):
        eq = oracle.binary_quantize(queries[qi], thr)[0]
        e_ids, e_d, _ = oracle.ivf_search(eq, enc, cen, off_o, rows_o, nprobe, k, metric, packed=packed)
        assert res[qi].index_mode() == mode
        assert np.array_equal(res[qi].distances().view(np.uint32), e_d.view(np.uint32)), (qi, res[qi].distances(), e_d)
        assert np.array_equal(res[qi].ids(), e_ids.astype(np.int64) + 100)
    if metric == O.HAMMING:
        d0 = res[0].distances()
        assert np.all(d0 == np.round(d0)) and d0[0] == 0.0   # Hamming counts, self-match first — not L2 distances


def test_ivfflat_search_uses_the_metric_of_the_call(L, oracle):
    """PyIvfFlatIndex.search(query, k, nprobe, metric): L2 partitions at build, metric per search."""
    rng = np.random.default_rng(3)
    n, dim, nlist, nprobe, k = 4000, 40, 24, 5, 10
    centers = rng.standard_normal((12, dim)).astype(f32)
    data = (centers[rng.integers(0, 12, n)] + 0.4 * rng.standard_normal((n, dim))).astype(f32)
    cen, asg = oracle.kmeans_train(data, nlist, 20, O.L2)
    idx = L.IvfFlatIndex.load(data, cen, asg, "ip", ivfflat_routing=True)   # built "for ip" ...
    off, orig = oracle.ivf_flat_layout(asg, cen.shape[0])
    slab = data[orig]
    rd = oracle.ivf_routing_dims(cen)
    queries = (data[rng.integers(0, n, 5)] + 0.05 * rng.standard_normal((5, dim))).astype(f32)
    for metric in (O.L2, O.COS, O.IP):                                       # ... searched with each metric
        for qi in range(queries.shape[0]):
            g_ids, g_d = idx.search(queries[qi], k, nprobe, NAME[metric])
            e_ids, e_d = oracle.ivf_flat_search(queries[qi], slab, cen, off, orig, nprobe, k, metric, routing_dims=rd)[:2]
            assert np.array_equal(g_d.view(np.uint32), e_d.view(np.uint32)), (metric, qi, g_d, e_d)
            assert np.array_equal(g_ids, e_ids.astype(np.uint32)), (metric, qi, g_ids, e_ids)
        asc = metric != O.IP
        assert np.all(np.diff(g_d) >= 0) if asc else np.all(np.diff(g_d) <= 0)
    with pytest.raises(ValueError):
        idx.search(queries[0], k, nprobe, "hamming")   # a float index cannot serve a binary metric
    with pytest.raises(NotImplementedError):
        idx.search(queries[0], k, nprobe, "manhattan")  # valid in the reference, outside this path
    with pytest.raises(ValueError, match="Unknown metric"):
        idx.search(queries[0], k, nprobe, "bogus")


def test_search_profile_carries_the_reference_fields(L, oracle):
    """`Collection.search_profile` (src/python/mod.rs:1240-1271, QueryProfile engine.rs:6906-6919; the reference's own test
    engine.rs:9145-9158): the result of `search` + the profile dict with the reference's field names, index_path "flat_mmap" /
    "flat_mmap_filtered" / "ann_index", and this build's `device` block from the library's HIP-event profile."""
    rng = np.random.default_rng(5)
    n, dim, k = 20_000, 64, 5
    data = rng.standard_normal((n, dim)).astype(f32)
    coll = L.Collection("c", dim)
    coll.add_items(data, list(range(100, 100 + n)))
    coll.commit()
    q = (data[77] + 0.01 * rng.standard_normal(dim)).astype(f32)
    want = coll.search(q, k)
    out = coll.search_profile(q, k)
    p = out["profile"]
    assert set(p) >= {"query_kind", "vector_field", "index_path", "total_vectors", "filter_expression", "filter_matches", "scanned_vectors",
                      "result_count", "filter_us", "search_us", "rerank_us", "total_us"}
    assert p["query_kind"] == "vector" and p["vector_field"] == "default" and p["index_path"] == "flat_mmap"
    assert p["total_vectors"] == n and p["scanned_vectors"] == n and p["filter_matches"] is None and p["result_count"] == k and p["rerank_us"] == 0
    assert p["search_us"] > 0 and p["total_us"] >= p["search_us"]
    assert p["device"]["scan_launches"] >= 1 and p["device"]["scan_us"] > 0 and p["device"]["pipeline_us"] >= p["device"]["scan_us"] * 0.99
    assert out["items"]["ids"] == want.ids().tolist() and out["items"]["k"] == k and out["items"]["index"] == "FLAT-IP"
    assert out["items"]["ids"][0] == 177
    sub = np.arange(0, n, 4, dtype=np.uint64)
    pf = coll.search_profile(q, k, subset=sub)["profile"]
    assert pf["index_path"] == "flat_mmap_filtered" and pf["filter_matches"] == sub.size and pf["scanned_vectors"] == sub.size
    coll.build_index("IVF-IP", {"n_clusters": 32, "nprobe": 4})
    pi = coll.search_profile(q, k, nprobe=4)["profile"]
    assert pi["index_path"] == "ann_index" and pi["result_count"] == k
