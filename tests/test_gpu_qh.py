"""`-m gpu` parity tests of k_scan_qh (lynsedb_amd/csrc/scan_qh.h; LYNSE_HIP_QH=0: k_scan_h16): the query-stationary threshold stages of
the float path over the f16 shadow at 64 / 128 columns, and of the exactness rule of integer-valued collections (k_prep_queries:
zero margin when rows and query are integers below the 2^24 bounds; BASELINE config 3 is the full-size case of both,
tests/test_gpu_baseline_configs.py::test_c3_*).

Every case goes through the C-ABI, is compared bit for bit (row ids and f32 distance bits) with the CPU oracle
(FlatMmap::search -> exact_flat_search, src/storage/flat_mmap.rs:905-1026, :2132-2256) and with the same search on the
256 x 256 tile of k_scan_h16, and pins the tiling it ran through `profile_get()["last_plan"]`.
"""
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu
f32 = np.float32

METRICS = {"l2": O.L2, "ip": O.IP, "cosine": O.COS}


def tiling_of(p):
    return (int(p["last_plan"]) >> 16) & 0xff


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1
    return L_


def check(oracle, data, queries, k, name, rows, dists, counts, picks, tag):
    for qi in picks:
        e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, METRICS[name])
        c = int(counts[qi])
        assert c == len(e_ids), (tag, qi, c, len(e_ids))
        assert np.array_equal(dists[qi][:c].view(np.uint32), e_d.view(np.uint32)), (tag, qi, dists[qi][:c], e_d)
        assert np.array_equal(rows[qi][:c].astype(np.uint64), e_ids.astype(np.uint64)), (tag, qi, rows[qi][:c], e_ids)


def run_ab(idx, queries, k, name):
    """-> results + plan of the default (k_scan_qh where its shapes apply) and the plan of the same search on k_scan_h16 (LYNSE_HIP_QH=0, read
    per call); asserts identical bits"""
    idx.profile_get(reset=True)
    r, d, c = idx.search_batch_arrays(queries, k, name)
    p = idx.profile_get(reset=True)
    os.environ["LYNSE_HIP_QH"] = "0"
    try:
        r0, d0, c0 = idx.search_batch_arrays(queries, k, name)
        p0 = idx.profile_get(reset=True)
    finally:
        del os.environ["LYNSE_HIP_QH"]
    assert np.array_equal(r, r0) and np.array_equal(d.view(np.uint32), d0.view(np.uint32)) and np.array_equal(c, c0), (name, k)
    return r, d, c, p, p0


@pytest.mark.parametrize("dim", [128, 64])
def test_qh_threshold_stages_equal_the_oracle_and_the_256x256_tile(L, oracle, dim):
    # rows NOT a multiple of the 128-row tile (the last tile re-reads the last row and masks it), Gaussian rows + a family of
    # near-duplicates of the queries (ties and near-ties around the k-th score)
    n = 300_000 + 77
    rng = np.random.default_rng(5 + dim)
    data = rng.standard_normal((n, dim)).astype(f32)
    nq_max = 256
    q_rows = np.sort(rng.choice(n, nq_max, replace=False))
    queries = (data[q_rows] + 0.05 * rng.standard_normal((nq_max, dim)).astype(f32)).astype(f32)
    data[rng.choice(n, 500, replace=False)] = queries[rng.integers(0, nq_max, 500)]      # exact copies of queries elsewhere in the shard
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    idx.profile_enable(True)
    for name in ("l2", "ip", "cosine"):
        for nq, k in ((256, 10), (256, 100), (100, 10), (40, 100), (160, 10)):
            r, d, c, p, p0 = run_ab(idx, queries[:nq], k, name)
            # (k = 100 over Gaussian rows may send the batch down the plan ladder on either tiling — whose later levels run k_scan_h16: neither
            # the fallback count nor the tiling of a batch that fell back is pinned)
            assert tiling_of(p0) != 0x82, (name, nq, k, p, p0)
            if p["fallback_queries"] == 0:      # (FLAT-IP batches over 64 / 128-column rows run the float pass too: round 5)
                assert tiling_of(p) == 0x82, (name, nq, k, p)
            if name == "ip":                    # LYNSE_HIP_IP_LOWD=i8 (read per call): the certified int8 pass (plan flag 4) — identical bits
                os.environ["LYNSE_HIP_IP_LOWD"] = "i8"
                try:
                    r8, d8, c8 = idx.search_batch_arrays(queries[:nq], k, name)
                    p8 = idx.profile_get(reset=True)
                finally:
                    del os.environ["LYNSE_HIP_IP_LOWD"]
                assert np.array_equal(r8, r) and np.array_equal(d8.view(np.uint32), d.view(np.uint32)) and np.array_equal(c8, c), (name, nq, k)
                assert tiling_of(p8) != 0x82 and (p8["fallback_queries"] != 0 or int(p8["last_plan"]) & 4), (name, nq, k, p8)
            picks = sorted({0, 1, 31, 32, 33, 63, 64, nq // 2, nq - 2, nq - 1} & set(range(nq)))
            check(oracle, data, queries, k, name, r, d, c, picks, (dim, name, nq, k))


def test_qh_inner_product_form_below_the_int8_pass(L, oracle):
    # under 65,536 rows no batch starts on the certified int8 pass: the stages behind the emit-all first stage of the contiguous plan
    # run k_scan_qh for all three metrics (the IP form has no norm ring)
    dim, n, nq = 128, 60_000, 256
    rng = np.random.default_rng(3)
    data = rng.random((n, dim), dtype=f32)
    queries = (data[rng.integers(0, n, nq)] + 0.03 * rng.standard_normal((nq, dim)).astype(f32)).astype(f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    idx.profile_enable(True)
    for name in ("ip", "l2", "cosine"):
        for k in (10, 64):
            r, d, c, p, p0 = run_ab(idx, queries, k, name)
            assert tiling_of(p0) != 0x82 and (p["fallback_queries"] != 0 or tiling_of(p) == 0x82), (name, k, p, p0)
            check(oracle, data, queries, k, name, r, d, c, (0, 31, 32, 100, 255), (name, k))


def test_qh_integer_data_with_massive_ties(L, oracle):
    # SIFT-like integer rows (benchmarks/sift_io.py:87-89): squared distances are exact integers and tie at the k-th place;
    # a shard smaller than one tile per CU (ntiles < grid) and one with few tiles per workgroup
    from lynsedb_amd.datasets import sift_like

    dim = 128
    for n in (70_000, 140_003):
        data = sift_like(n, dim, 11)
        data[1::7] = data[0]                       # every 7th row is a copy of row 0
        queries = sift_like(256, dim, 12)
        queries[3] = data[0]
        idx = L.FlatIndex(None, dim)
        idx.write(data)
        idx.finalize()
        idx.profile_enable(True)
        # the same rows in a shard that does NOT use the exactness rule of integer collections (k_prep_queries: E = 0 when rows and
        # query are integers below the 2^24 bounds; LYNSE_HIP_NO_EXACT_INT is read when the row statistics are taken)
        os.environ["LYNSE_HIP_NO_EXACT_INT"] = "1"
        try:
            idx_m = L.FlatIndex(None, dim)
            idx_m.write(data)
            idx_m.finalize()
        finally:
            del os.environ["LYNSE_HIP_NO_EXACT_INT"]
        idx_m.profile_enable(True)
        for k in (10, 100):
            r, d, c, p, p0 = run_ab(idx, queries, k, "l2")    # (10,000 copies of one row: the queries near it go down the plan ladder)
            if k == 10 and p["fallback_queries"] == 0:   # (k = 100 on a shard this small: an emit-all sample stage, whose tiles the later stages skip — k_scan_h16)
                assert tiling_of(p) == 0x82, (n, k, p)
            check(oracle, data, queries, k, "l2", r, d, c, (0, 3, 64, 200, 255), (n, k))
            idx_m.profile_get(reset=True)
            rm, dm, cm = idx_m.search_batch_arrays(queries, k, "l2")
            pm = idx_m.profile_get(reset=True)
            assert np.array_equal(rm, r) and np.array_equal(dm.view(np.uint32), d.view(np.uint32)) and np.array_equal(cm, c), (n, k)
            assert p["pool_entries"] <= pm["pool_entries"], (n, k, p, pm)     # zero margin: only real ties are rescored beside the k best
            # a query that is NOT integer-valued falls back to the certified margin (and still equals the oracle)
            qf = (queries[:40] + f32(0.25)).astype(f32)
            rf, df, cf = idx.search_batch_arrays(qf, k, "l2")
            check(oracle, data, qf, k, "l2", rf, df, cf, (0, 3, 39), (n, k, "fractional query"))


def test_qh_after_appends_and_on_an_f16_shard(L, oracle):
    # appended rows (the shadow grows, norms and thresholds follow) and an F16 shard (VectorDtype::F16, src/storage/dtype.rs:6-29:
    # the shadow IS the stored rows; exact scores are the sequential f16 sums, simd.rs:805-846)
    dim, nq, k = 128, 200, 10
    rng = np.random.default_rng(77)
    data = rng.random((260_000, dim), dtype=f32)
    queries = (data[rng.integers(0, 260_000, nq)] + 0.02 * rng.standard_normal((nq, dim)).astype(f32)).astype(f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data[:200_000])
    idx.finalize()
    idx.profile_enable(True)
    r, d, c, p, _ = run_ab(idx, queries, k, "l2")
    assert tiling_of(p) == 0x82, p
    check(oracle, data[:200_000], queries, k, "l2", r, d, c, (0, 50, 199), "before append")
    idx.write(data[200_000:])
    idx.finalize()
    r, d, c, p, _ = run_ab(idx, queries, k, "l2")
    assert tiling_of(p) == 0x82, p
    check(oracle, data, queries, k, "l2", r, d, c, (0, 50, 199), "after append")
    # the F16 shard: 128 columns of f16 bits, 160 queries, squared L2 and cosine
    n16 = 200_000 + 5
    rows16 = oracle.round_f16(rng.standard_normal((n16, dim)).astype(f32)).reshape(n16, dim)
    q16 = (rows16[rng.integers(0, n16, 160)] + 0.05 * rng.standard_normal((160, dim)).astype(f32)).astype(f32)
    idx16 = L.FlatIndex(None, dim, 0, dtype="f16")
    idx16.write_f16_bits(rows16.astype(np.float16).view(np.uint16))
    idx16.finalize()
    idx16.profile_enable(True)
    for name in ("l2", "cosine"):
        r, d, c, p, _ = run_ab(idx16, q16, k, name)
        assert p["fallback_queries"] == 0 and tiling_of(p) == 0x82, (name, p)
        for qi in (0, 64, 159):
            e_ids, e_d = oracle.canonical_topk_f16(q16[qi], rows16, k, METRICS[name])
            cc = int(c[qi])
            assert cc == len(e_ids) and np.array_equal(d[qi][:cc].view(np.uint32), e_d.view(np.uint32)) and \
                np.array_equal(r[qi][:cc].astype(np.uint64), e_ids.astype(np.uint64)), ("f16", name, qi)


def _int_case(L, oracle, data, queries, metrics, ks, tag, dtype=None, expect_exact=None, strict=()):
    """Integer-valued rows: search on a shard with the exactness rule and on one without it (LYNSE_HIP_NO_EXACT_INT), both equal to
    the oracle bit for bit; expect_exact: True -> the rule must have narrowed the rescoring pool to (about) k rows per query,
    False -> the pools must be the same (the rule did not apply)."""
    dim = data.shape[1]

    def build(no_rule):
        if no_rule:
            os.environ["LYNSE_HIP_NO_EXACT_INT"] = "1"
        try:
            if dtype == "f16":
                idx = L.FlatIndex(None, dim, 0, dtype="f16")
                idx.write_f16_bits(data.astype(np.float16).view(np.uint16))
            else:
                idx = L.FlatIndex(None, dim)
                idx.write(data)
            idx.finalize()
        finally:
            os.environ.pop("LYNSE_HIP_NO_EXACT_INT", None)
        idx.profile_enable(True)
        return idx

    a, b = build(False), build(True)
    nq = queries.shape[0]
    for name in metrics:
        for k in ks:
            a.profile_get(reset=True); b.profile_get(reset=True)
            ra, da, ca = a.search_batch_arrays(queries, k, name)
            rb, db, cb = b.search_batch_arrays(queries, k, name)
            pa, pb = a.profile_get(reset=True), b.profile_get(reset=True)
            assert np.array_equal(ra, rb) and np.array_equal(da.view(np.uint32), db.view(np.uint32)) and np.array_equal(ca, cb), (tag, name, k)
            for qi in sorted({0, 1, nq // 2, nq - 1}):
                if dtype == "f16":
                    e_ids, e_d = oracle.canonical_topk_f16(queries[qi], data, k, METRICS[name])
                else:
                    e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, METRICS[name])
                c = int(ca[qi])
                assert c == len(e_ids) and np.array_equal(da[qi][:c].view(np.uint32), e_d.view(np.uint32)) and \
                    np.array_equal(ra[qi][:c].astype(np.uint64), e_ids.astype(np.uint64)), (tag, name, k, qi)
            if expect_exact is True and name != "cosine":
                assert pa["pool_entries"] <= pb["pool_entries"], (tag, name, k, pa, pb)
            if name in strict:     # (the rule must have applied: fewer rows inside the margin than with it)
                assert pa["pool_entries"] < pb["pool_entries"], (tag, name, k, pa, pb)
            if expect_exact is False:
                assert pa["pool_entries"] == pb["pool_entries"], (tag, name, k, pa, pb)


def test_exactness_rule_of_integer_collections_and_its_limits(L, oracle):
    """k_prep_queries' exactness rule: E = 0 only when rows and query are integers, exact in f16 (<= 2048) and
    D aq av, D aq^2, D av^2, D dmax^2 < 2^24.  Inside the bounds the zero-margin search equals the oracle (ties everywhere: small
    integer alphabets); outside them the rule must not apply (same pools as a shard built without it)."""
    rng = np.random.default_rng(99)
    nq = 64
    for n in (70_000, 60_000):    # (60,000 rows: under the int8 pass's 65,536 — IP batches run the float path too, contiguous plan)
        # (a) signed integers, dmax = aq + av: D = 64, |x| <= 100: 64 * 200^2 = 2.56M
        data = rng.integers(-100, 101, (n, 64)).astype(f32)
        queries = rng.integers(-100, 101, (nq, 64)).astype(f32)
        _int_case(L, oracle, data, queries, ("l2", "ip", "cosine"), (10, 100), ("signed", n), expect_exact=True, strict=("l2",))
        # (b) tiny alphabet {0, 1, 2}: massive ties at every rank
        data = rng.integers(0, 3, (n, 128)).astype(f32)
        queries = rng.integers(0, 3, (nq, 128)).astype(f32)
        _int_case(L, oracle, data, queries, ("l2", "ip"), (10, 64), ("ternary", n), expect_exact=True)
    n = 70_000
    # (c) just inside the bound: D = 128, non-negative, max 361: 128 * 361^2 = 16.68M < 2^24
    data = rng.integers(0, 362, (n, 128)).astype(f32)
    queries = rng.integers(0, 362, (nq, 128)).astype(f32)
    _int_case(L, oracle, data, queries, ("l2", "ip"), (10,), "inside", expect_exact=True, strict=("l2",))
    # (d) just outside: max 363: 128 * 363^2 = 16.87M > 2^24 -> the certified margin (same pools as without the rule)
    data = rng.integers(0, 364, (n, 128)).astype(f32)
    data[0, 0] = 363.0
    queries = rng.integers(0, 364, (nq, 128)).astype(f32)
    queries[0, 0] = 363.0
    _int_case(L, oracle, data, queries, ("l2", "ip"), (10,), "outside", expect_exact=False)
    # (e) one non-integer element in the shard, or beyond the exact f16 integers (4097): the rule is off for the shard
    data = rng.integers(0, 100, (n, 64)).astype(f32)
    queries = rng.integers(0, 100, (nq, 64)).astype(f32)
    d2 = data.copy(); d2[n - 1, 63] = 0.5
    _int_case(L, oracle, d2, queries, ("l2",), (10,), "one fraction", expect_exact=False)
    d3 = data.copy(); d3[5, 5] = 4097.0
    _int_case(L, oracle, d3, queries, ("l2", "ip"), (10,), "beyond f16", expect_exact=False)
    # (f) an F16 shard of integers (VectorDtype::F16: sequential f32 sums in the reference — exact all the same)
    data = rng.integers(0, 200, (n, 128)).astype(f32)
    queries = rng.integers(0, 200, (nq, 128)).astype(f32)
    _int_case(L, oracle, data, queries, ("l2", "ip"), (10,), "f16 shard", dtype="f16", expect_exact=True, strict=("l2",))
    # (g) the flag follows appends: integers first (rule on), then a block with fractions (rule off from then on)
    idx = L.FlatIndex(None, 64)
    base = rng.integers(0, 50, (n, 64)).astype(f32)
    idx.write(base); idx.finalize(); idx.profile_enable(True)
    q = rng.integers(0, 50, (nq, 64)).astype(f32)
    idx.profile_get(reset=True)
    r1, dd1, c1 = idx.search_batch_arrays(q, 10, "l2")
    p1 = idx.profile_get(reset=True)
    extra = (rng.integers(0, 50, (5000, 64)) + 0.25).astype(f32)
    idx.write(extra); idx.finalize()
    r2, dd2, c2 = idx.search_batch_arrays(q, 10, "l2")
    p2 = idx.profile_get(reset=True)
    allrows = np.concatenate([base, extra])
    for qi in (0, 31, 63):
        e_ids, e_d = oracle.canonical_topk(q[qi], base, 10, O.L2)
        assert np.array_equal(r1[qi].astype(np.uint64), e_ids.astype(np.uint64)) and np.array_equal(dd1[qi].view(np.uint32), e_d.view(np.uint32))
        e_ids, e_d = oracle.canonical_topk(q[qi], allrows, 10, O.L2)
        assert np.array_equal(r2[qi].astype(np.uint64), e_ids.astype(np.uint64)) and np.array_equal(dd2[qi].view(np.uint32), e_d.view(np.uint32))
    assert p1["pool_entries"] <= p2["pool_entries"], (p1, p2)


@pytest.mark.parametrize("dim", [100, 56])
def test_padded_shadow_puts_96_to_127_and_48_to_63_columns_on_the_whole_slab_kernels(L, oracle, dim):
    """48..63 / 96..127 columns: the f16 shadow is padded to a whole 64-element slab (zeros), so float batches of such a shard run
    k_scan_qh (tiling 0x82); LYNSE_HIP_SHADOW_PAD=0 (read when the handle is created) keeps the 8-element pitch and the ragged
    k_scan_h16 variants.  Same bits either way, and the oracle's."""
    n, nq = 200_003, 200
    rng = np.random.default_rng(dim)
    data = rng.standard_normal((n, dim)).astype(f32)
    queries = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim)).astype(f32)).astype(f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    idx.profile_enable(True)
    os.environ["LYNSE_HIP_SHADOW_PAD"] = "0"
    try:
        idx0 = L.FlatIndex(None, dim)
    finally:
        del os.environ["LYNSE_HIP_SHADOW_PAD"]
    idx0.write(data)
    idx0.finalize()
    idx0.profile_enable(True)
    for name in ("l2", "cosine"):
        for k in (10, 100):
            idx.profile_get(reset=True); idx0.profile_get(reset=True)
            r, d, c = idx.search_batch_arrays(queries, k, name)
            r0, d0, c0 = idx0.search_batch_arrays(queries, k, name)
            p, p0 = idx.profile_get(reset=True), idx0.profile_get(reset=True)
            assert np.array_equal(r, r0) and np.array_equal(d.view(np.uint32), d0.view(np.uint32)) and np.array_equal(c, c0), (dim, name, k)
            assert tiling_of(p0) != 0x82, (dim, name, k, p0)
            if k == 10 and p["fallback_queries"] == 0:   # (k = 100 on a shard this small: an emit-all sample stage, whose tiles the later stages skip — k_scan_h16)
                assert tiling_of(p) == 0x82, (dim, name, k, p)
            check(oracle, data, queries, k, name, r, d, c, (0, 1, 99, 199), (dim, name, k))
