"""Deterministic `-m gpu` oracle tests of the BASELINE.json configurations at sizes that reach the SAME kernel
instantiations and stage plans the benchmark times (VERDICT r1 "next round" item 1).  Every case goes through the C-ABI,
is compared bit for bit (ids and f32 distance bits) with the CPU oracle on a fixed set of queries, and pins the plan it
ran through `profile_get()["last_plan"]` (include/lynse_hip.h):

  C2  FLAT-IP 2.4M x 768, 256 queries, k=10 : certified int8 coarse pass on the <2,4,4,2> tiling, sampled plan with the
      threshold-only (lane-max) sample stage + 2 threshold stages = 3 scan launches, no fallback — the kernels
      bench.py times (flat_mmap.rs:4845-4982)
  C3  FLAT-L2 SIFT-like 1M x 128, 256 queries, k=100 : k_scan_qh threshold stages (scan_qh.h), sampled plan, zero margin on the integer rows; A/B against the <4,2,2,4> tiling (benchmarks/sift_io.py:87-89)
  C4  IVF-IP 768-d, nlist=4096, nprobe=32, 520k rows: centroid store right at the 4096-row single / batch-8 boundary,
      nprobe >= nlist, subset-filtered (src/index/ivf.rs:181-348)
  C5  Hamming 10M x 1024-bit, k=50, nq in {1, 256}: massive ties at the k-th distance (flat_mmap.rs:1345-1409)
"""
import os

import numpy as np
import pytest

import oracle as O
from conftest import oracle_for_every_query

pytestmark = pytest.mark.gpu
f32 = np.float32

PLAN_SAMPLED, PLAN_THRESHOLD_ONLY, PLAN_I8C, PLAN_SEGMENTS, PLAN_SMALL, PLAN_FUSED_SAMPLE = 1, 2, 4, 8, 16, 128
PLAN_STS = 1 << 24   # the self-tightening single-launch scan (k_scan_qs<.., STS>)
PLAN_QS_SAMPLE = 1 << 25   # the threshold-only sample stage ran on the query-stationary tiling (k_scan_qs<.., SMP>)


def plan_fields(p):
    lp = int(p["last_plan"])
    return lp & 0xff, (lp >> 8) & 0xff, (lp >> 16) & 0xff   # flags, stages, tiling


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1
    return L_


def assert_rows_equal(oracle_res, rows, dists, count, tag):
    e_ids, e_d = oracle_res
    c = int(count)
    assert c == len(e_ids), (tag, c, len(e_ids))
    assert np.array_equal(dists[:c].view(np.uint32), e_d.view(np.uint32)), (tag, dists[:c], e_d)
    assert np.array_equal(rows[:c].astype(np.uint64), e_ids.astype(np.uint64)), (tag, rows[:c], e_ids)


def test_c2_flat_ip_768_batch256_runs_the_benchmarked_kernels(L, oracle):
    n, dim, nq, k = 2_400_000, 768, 256, 10
    rng = np.random.default_rng(42)
    idx = L.FlatIndex(None, dim)
    idx.reserve(n)
    data = np.empty((n, dim), f32)
    for b in range(0, n, 200_000):        # uniform[0,1) f32 in blocks, like flat_search_bench.py:71-77 / bench.py
        e = min(n, b + 200_000)
        rng.random(out=data[b:e], dtype=f32)
        idx.write(data[b:e])
    q_rows = np.sort(rng.integers(0, n, nq))
    queries = (data[q_rows] + 0.03 * rng.standard_normal((nq, dim)).astype(f32)).astype(f32)
    idx.finalize()
    idx.search_batch_arrays(queries, k, "ip")          # first call builds the SQ8 codes of the int8 coarse pass
    idx.profile_enable(True)
    idx.profile_get(reset=True)
    rows, dists, counts = idx.search_batch_arrays(queries, k, "ip")
    p = idx.profile_get(reset=True)
    flags, stages, tiling = plan_fields(p)
    assert p["fallback_queries"] == 0
    assert p["scan_launches"] == 3 and stages == 3, p
    assert tiling == 0x81, hex(tiling)      # the threshold stages ran the query-stationary tiling (k_scan_qs, scan_qs.h)
    assert flags & PLAN_SAMPLED and flags & PLAN_THRESHOLD_ONLY and flags & PLAN_I8C and flags & PLAN_SEGMENTS, bin(flags)
    assert not (flags & PLAN_FUSED_SAMPLE) and not (int(p["last_plan"]) & PLAN_STS), hex(int(p["last_plan"]))
    assert int(p["last_plan"]) & PLAN_QS_SAMPLE, hex(int(p["last_plan"]))      # ... and so did the sample stage (LYNSE_HIP_QS_SAMPLE=0 below: the 256 x 256 sample tiles)
    # the full ranking of the default run against the oracle's exact_flat_search (ids + f32 distance bits), wave boundaries included
    # — ALL 256 queries, every wave / query-column position of the tilings (the variants below are then compared with this run)
    want = oracle_for_every_query(lambda qi: oracle.canonical_topk(queries[qi], data, k, O.IP), nq)
    for qi in range(nq):
        assert_rows_equal(want[qi], rows[qi], dists[qi], counts[qi], ("c2", qi))
        assert rows[qi, 0] == q_rows[qi]
    # the same batch (a) with the sample stage INSIDE the launch of the first threshold stage (k_scan_h16<.., FS>: grid-wide
    # threshold hand-over; off by default — measured slower than the two launches) and (b) on the three separate tail kernels
    # instead of k_select_final, and (c) with the threshold stages on the one-wave-per-SIMD tiling (4 waves x 4 x 4 blocks,
    # accumulators in fixed AGPR tuples, the last k-step's MFMAs deferred behind the next slab's barrier; off by default —
    # measured 11 % slower): identical bits
    import os
    # ... and (d) with the threshold stages on the 256 x 256 tile of k_scan_h16 (LYNSE_HIP_QS=0: the round-3 default)
    # ... (d) on the 256 x 256 tile of k_scan_h16 (LYNSE_HIP_QS=0: the round-3 default) and the other instantiations of k_scan_qs, and
    # (e) as ONE scan launch with self-tightening thresholds (k_scan_qs<.., STS>; LYNSE_HIP_STS=1, off by default: measured no
    # faster — a query that overflows there goes down the ladder to the staged plan, results identical either way)
    for env, launches, fused in (({"LYNSE_HIP_FUSED_SAMPLE": "1"}, 2, True), ({"LYNSE_HIP_FUSED_TAIL": "0"}, 3, False), ({"LYNSE_HIP_AG": "1"}, 3, False),
                                 ({"LYNSE_HIP_QS": "0"}, 3, False), ({"LYNSE_HIP_QS": "2"}, 3, False), ({"LYNSE_HIP_QS": "3"}, 3, False), ({"LYNSE_HIP_QS_SAMPLE": "0"}, 3, False),
                                 ({"LYNSE_HIP_STS": "1"}, None, False)):
        os.environ.update(env)
        try:
            r_u, d_u, c_u = idx.search_batch_arrays(queries, k, "ip")
            p_u = idx.profile_get(reset=True)
        finally:
            for name in env:
                del os.environ[name]
        if launches is not None:
            assert p_u["scan_launches"] == launches and bool(plan_fields(p_u)[0] & PLAN_FUSED_SAMPLE) == fused and p_u["fallback_queries"] == 0, p_u
        if env.get("LYNSE_HIP_QS") == "0":
            assert plan_fields(p_u)[2] == 0x24, hex(plan_fields(p_u)[2])
        if env.get("LYNSE_HIP_QS") in ("0", "2") or "LYNSE_HIP_QS_SAMPLE" in env:
            assert not (int(p_u["last_plan"]) & PLAN_QS_SAMPLE), hex(int(p_u["last_plan"]))
        if "LYNSE_HIP_STS" in env:   # (the plan on record is the last one run: the single launch, or the staged rerun of the queries that overflowed)
            assert p_u["scan_launches"] >= 1 and (int(p_u["last_plan"]) & PLAN_STS or p_u["fallback_queries"] > 0), p_u
        assert np.array_equal(r_u, rows) and np.array_equal(d_u.view(np.uint32), dists.view(np.uint32)) and np.array_equal(c_u, counts)
    # other batch sizes give the same answers: 40 queries (same kernels, mostly empty query columns) and 8 queries
    # (the <= 32-query tiling; from 256K rows on it streams the SQ8 codes too)
    r40, d40, c40 = idx.search_batch_arrays(queries[:40], k, "ip")
    p40 = idx.profile_get(reset=True)
    assert plan_fields(p40)[2] == 0x81 and p40["fallback_queries"] == 0, p40          # 33..256 queries over 768-column codes: k_scan_qs
    assert np.array_equal(r40, rows[:40]) and np.array_equal(d40.view(np.uint32), dists[:40].view(np.uint32))
    for nqs, want in ((40, 0x14), (100, 0x24)):                                         # LYNSE_HIP_QS_MID=0: the mid tilings of k_scan_h16
        os.environ["LYNSE_HIP_QS_MID"] = "0"
        try:
            rm, dm, cm = idx.search_batch_arrays(queries[:nqs], k, "ip")
            pm = idx.profile_get(reset=True)
        finally:
            del os.environ["LYNSE_HIP_QS_MID"]
        assert plan_fields(pm)[2] == want and pm["fallback_queries"] == 0, (nqs, pm)
        assert np.array_equal(rm, rows[:nqs]) and np.array_equal(dm.view(np.uint32), dists[:nqs].view(np.uint32))
    r8, d8, c8 = idx.search_batch_arrays(queries[:8], k, "ip")
    p8 = idx.profile_get(reset=True)
    assert plan_fields(p8)[0] & PLAN_SMALL and plan_fields(p8)[2] == 0x14
    assert np.array_equal(r8, rows[:8]) and np.array_equal(d8.view(np.uint32), dists[:8].view(np.uint32))


def test_l2_768_batch256_runs_the_certified_int8_pass(L, oracle):
    """FLAT-L2 2.4M x 768, 256 queries, k=10 (north_star: "batched cosine/IP/L2"): the certified int8 pass in its L2 forms — the
    PLAIN SQ8 codes with the exact f32 row norms in a float epilogue (k_scan_h16<..,L2,..,I8Q=4>) on every tiling: 256, 100, 48 and 8
    queries — plans pinned, ids and
    distance bits equal the oracle's exact_flat_search with the difference-form L2 kernel (simd.rs:1529-1581)."""
    n, dim, nq, k = 2_400_000, 768, 256, 10
    rng = np.random.default_rng(43)
    idx = L.FlatIndex(None, dim)
    idx.reserve(n)
    data = np.empty((n, dim), f32)
    for b in range(0, n, 200_000):
        e = min(n, b + 200_000)
        rng.random(out=data[b:e], dtype=f32)
        idx.write(data[b:e])
    q_rows = np.sort(rng.integers(0, n, nq))
    queries = (data[q_rows] + 0.03 * rng.standard_normal((nq, dim)).astype(f32)).astype(f32)
    idx.finalize()
    idx.prepare("l2", nq)
    idx.profile_enable(True)
    idx.profile_get(reset=True)
    rows, dists, counts = idx.search_batch_arrays(queries, k, "l2")
    p = idx.profile_get(reset=True)
    flags, stages, tiling = plan_fields(p)
    # threshold stages on the query-stationary tiling in its L2 form (k_scan_qs<.., MET = 1>: int8 dot products, exact f32 row norms)
    assert p["fallback_queries"] == 0 and tiling == 0x81 and flags & PLAN_I8C and flags & PLAN_SAMPLED and flags & PLAN_THRESHOLD_ONLY, (p, bin(flags))
    want = oracle_for_every_query(lambda qi: oracle.canonical_topk(queries[qi], data, k, O.L2), nq)      # all 256 queries
    for qi in range(nq):
        assert_rows_equal(want[qi], rows[qi], dists[qi], counts[qi], ("l2", qi))
        assert rows[qi, 0] == q_rows[qi]
    import os
    os.environ["LYNSE_HIP_QS"] = "0"     # ... and on the <4,2,2,4> tiling of k_scan_h16<.., I8Q = 4> (the round-3 default): identical bits
    try:
        r0, d0, c0 = idx.search_batch_arrays(queries, k, "l2")
        p0 = idx.profile_get(reset=True)
    finally:
        del os.environ["LYNSE_HIP_QS"]
    assert plan_fields(p0)[2] == 0x42 and plan_fields(p0)[0] & PLAN_I8C and p0["fallback_queries"] == 0, p0
    assert np.array_equal(r0, rows) and np.array_equal(d0.view(np.uint32), dists.view(np.uint32)) and np.array_equal(c0, counts)
    # 100 and 48 queries: the query-stationary tiling too (waves without queries skip their MFMAs; round 4); LYNSE_HIP_QS_MID=0: the 256 x 128 /
    # 128 x 64 tilings of the plain-code form, and with the mid tilings off as well the 256-query tilings
    for nqs, env, want in ((100, {}, 0x81), (100, {"LYNSE_HIP_QS_MID": "0"}, 0x24), (100, {"LYNSE_HIP_QS_MID": "0", "LYNSE_HIP_MID_TILINGS": "0"}, 0x81),
                           (100, {"LYNSE_HIP_MID_TILINGS": "0", "LYNSE_HIP_QS": "0"}, 0x42), (48, {}, 0x81), (48, {"LYNSE_HIP_QS_MID": "0"}, 0x14)):
        os.environ.update(env)
        try:
            rs, ds, cs = idx.search_batch_arrays(queries[:nqs], k, "l2")
            ps = idx.profile_get(reset=True)
        finally:
            for name in env:
                del os.environ[name]
        assert plan_fields(ps)[2] == want and plan_fields(ps)[0] & PLAN_I8C and ps["fallback_queries"] == 0, (nqs, env, ps)
        assert np.array_equal(rs, rows[:nqs]) and np.array_equal(ds.view(np.uint32), dists[:nqs].view(np.uint32))
    r8, d8, c8 = idx.search_batch_arrays(queries[:8], k, "l2")      # the <= 32-query tiling: same answers
    assert np.array_equal(r8, rows[:8]) and np.array_equal(d8.view(np.uint32), dists[:8].view(np.uint32))


def test_c3_flat_l2_sift_like_1m_k100(L, oracle):
    from lynsedb_amd.datasets import sift_like

    n, dim, nq, k = 1_000_000, 128, 256, 100
    data = sift_like(n, dim, 42)
    queries = sift_like(nq, dim, 43)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    idx.profile_enable(True)
    idx.profile_get(reset=True)
    rows, dists, counts = idx.search_batch_arrays(queries, k, "l2")
    p = idx.profile_get(reset=True)
    flags, stages, tiling = plan_fields(p)
    # threshold stages on the query-stationary tiling of the low-dimensional f16 shadow (k_scan_qh, scan_qh.h; round 5)
    assert p["fallback_queries"] == 0 and tiling == 0x82 and flags & PLAN_SAMPLED and stages >= 2, (p, bin(flags))
    want = oracle_for_every_query(lambda qi: oracle.canonical_topk(queries[qi], data, k, O.L2), nq)      # all 256 queries (integer rows: ties decided by the canonical order)
    for qi in range(nq):
        assert_rows_equal(want[qi], rows[qi], dists[qi], counts[qi], ("c3", qi))
    # integer rows x integer queries below the 2^24 bounds: the coarse pass is exact, the margin zero (k_prep_queries' exactness rule,
    # round 5) — exactly k rows per query reach the final rescoring unless the k-th distance ties
    assert p["pool_entries"] <= int(1.2 * nq * k), p
    os.environ["LYNSE_HIP_QH"] = "0"     # ... and on the <4,2,2,4> tiling of k_scan_h16 (the default up to round 4): identical bits
    try:
        r0, d0, c0 = idx.search_batch_arrays(queries, k, "l2")
        p0 = idx.profile_get(reset=True)
    finally:
        del os.environ["LYNSE_HIP_QH"]
    assert plan_fields(p0)[2] == 0x42 and p0["fallback_queries"] == 0, p0
    assert np.array_equal(r0, rows) and np.array_equal(d0.view(np.uint32), dists.view(np.uint32)) and np.array_equal(c0, counts)
    # integer-valued data: squared distances are exact integers and tie often — the canonical (distance, row) order decides
    assert np.all(dists[0] == np.round(dists[0]))
    # IP and cosine over the same store, k=100 (IP: lane-max sample up to k = 128 on the <2,4,4,2> tiling)
    for name, metric in (("ip", O.IP), ("cosine", O.COS)):
        r, d, c = idx.search_batch_arrays(queries, k, name)
        pp = idx.profile_get(reset=True)
        assert pp["fallback_queries"] == 0
        want = oracle_for_every_query(lambda qi: oracle.canonical_topk(queries[qi], data, k, metric), nq)
        for qi in range(nq):
            assert_rows_equal(want[qi], r[qi], d[qi], c[qi], ("c3", name, qi))


def test_c4_ivf_ip_768_nlist4096_nprobe32(L, oracle):
    n, dim, nlist, nprobe, k = 520_000, 768, 4096, 32, 10
    rng = np.random.default_rng(7)        # benchmarks/ivf_kmeans_baseline.py:45-55 recipe: unit centers + sigma 0.03 noise
    K = 1024
    centers = rng.standard_normal((K, dim)).astype(f32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    data = np.empty((n, dim), f32)
    for b in range(0, n, 65536):
        e = min(n, b + 65536)
        data[b:e] = centers[np.arange(b, e) % K] + 0.03 * rng.standard_normal((e - b, dim)).astype(f32)
    built = L.IvfFlatIndex.build(None, data, dim, nlist, 2, "ip", l2_partitions=False)   # device k-means, 2 Lloyd rounds
    cen, asg, off_slab, orig = built.export()
    assert cen.shape == (nlist, dim) and asg.size == n
    del built
    # parity definition: the same centroids + assignments in -> the same probes / candidates / results out
    idx = L.IvfFlatIndex.load(data, cen, asg, "ip")
    off, rows_l = oracle.lists_from_assignments(asg, nlist)
    nq = 64
    queries = (data[rng.integers(0, n, nq)] + 0.02 * rng.standard_normal((nq, dim)).astype(f32)).astype(f32)
    g_rows, g_d, g_c = idx.search_batch_arrays(queries, k, nprobe)
    want = oracle_for_every_query(lambda qi: oracle.ivf_search(queries[qi], data, cen, off, rows_l, nprobe, k, O.IP)[:2], nq)      # every query of the batch
    for qi in range(nq):
        assert_rows_equal(want[qi], g_rows[qi], g_d[qi], g_c[qi], ("c4", qi))
    # single query (config 4 latency shape) and nprobe >= nlist (every list probed = exact search in IVF order)
    r1, d1, c1 = idx.search_batch_arrays(queries[5:6], k, nprobe)
    e_ids, e_d, _ = oracle.ivf_search(queries[5], data, cen, off, rows_l, nprobe, k, O.IP)
    assert_rows_equal((e_ids, e_d), r1[0], d1[0], c1[0], ("c4 single", 5))
    for np_all in (nlist, nlist + 17):
        ra, da, ca = idx.search_batch_arrays(queries[:3], k, np_all)
        for qi in range(3):
            e_ids, e_d, _ = oracle.ivf_search(queries[qi], data, cen, off, rows_l, np_all, k, O.IP)
            assert_rows_equal((e_ids, e_d), ra[qi], da[qi], ca[qi], ("c4 all", np_all, qi))
            assert_rows_equal(oracle.canonical_topk(queries[qi], data, k, O.IP, ip_form=O.IPFORM_SINGLE), ra[qi], da[qi], ca[qi],
                              ("c4 all vs flat", qi))
    # subset-filtered (SearchParams.subset, ivf.rs:251-265): 10 % of the rows
    subset = np.sort(rng.choice(n, n // 10, replace=False)).astype(np.uint64)
    rf, df, cf = idx.search_filtered_batch_arrays(queries[:12], k, nprobe, subset)
    for qi in (0, 5, 11):
        e_ids, e_d = oracle.ivf_search_filtered(queries[qi], data, cen, off, rows_l, nprobe, k, O.IP, subset)
        assert_rows_equal((e_ids, e_d), rf[qi], df[qi], cf[qi], ("c4 filtered", qi))


@pytest.mark.parametrize("nq", [1, 256])
def test_c5_hamming_10m_1024bit_k50(L, oracle, nq):
    from lynsedb_amd.datasets import packed_bernoulli

    n, bits, k = 10_000_000, 1024, 50
    words = packed_bernoulli(n, bits, 0.5, 42)
    idx = L.FlatIndex(None, bits)
    for b in range(0, n, 2_500_000):
        idx.write_packed(words[b:b + 2_500_000])
    idx.finalize()
    rng = np.random.default_rng(9)
    qw = words[rng.integers(0, n, nq)].copy()
    flip = packed_bernoulli(nq, bits, 0.5, 10) & packed_bernoulli(nq, bits, 0.5, 11) & packed_bernoulli(nq, bits, 0.5, 12)
    qw ^= flip                                              # ~12 % of the bits flipped: the source row stays the best hit
    idx.profile_enable(True)
    rows, dists, counts = idx.search_packed_arrays(qw, k, "hamming")
    p = idx.profile_get(reset=True)
    assert p["fallback_queries"] == 0
    want = oracle_for_every_query(lambda qi: oracle.canonical_topk_packed(qw[qi], words, k, O.HAMMING), nq)      # every query of the batch
    for qi in range(nq):
        assert_rows_equal(want[qi], rows[qi], dists[qi], counts[qi], ("c5", nq, qi))
    assert np.all(dists == np.round(dists))
    if nq == 256:   # the +-1 FP4 GEMM on the query-stationary tiling (k_scan_qs<.., F4>, round 4); LYNSE_HIP_QS_F4=0: the 256 x 256 tile — identical bits
        import os

        assert plan_fields(p)[2] == 0x81, hex(plan_fields(p)[2])
        for nqs, env in ((256, {"LYNSE_HIP_QS_F4": "0"}), (100, {}), (100, {"LYNSE_HIP_QS_F4": "0"})):
            os.environ.update(env)
            try:
                r0, d0, c0 = idx.search_packed_arrays(qw[:nqs], k, "hamming")
                p0 = idx.profile_get(reset=True)
            finally:
                for name in env:
                    del os.environ[name]
            assert (plan_fields(p0)[2] == 0x81) == (not env) and p0["fallback_queries"] == 0, (nqs, env, p0)
            assert np.array_equal(r0, rows[:nqs]) and np.array_equal(d0, dists[:nqs]) and np.array_equal(c0, counts[:nqs]), (nqs, env)
    if nq == 1:   # Tanimoto / Dice on the same fingerprints (the sparse-fingerprint use of config 5)
        for name, metric in (("tanimoto", O.JACCARD), ("dice", O.DICE)):
            r, d, c = idx.search_packed_arrays(qw, k, name)
            assert_rows_equal(oracle.canonical_topk_packed(qw[0], words, k, metric), r[0], d[0], c[0], ("c5", name))


@pytest.mark.parametrize("dim,metric", [(256, "ip"), (384, "cosine"), (512, "ip"), (640, "ip"), (500, "ip"), (1024, "ip"), (1024, "cosine")])
def test_query_stationary_tiling_on_narrower_code_rows(L, oracle, dim, metric):
    """k_scan_qs<NSLAB, ...> for 2..5 and 8 slabs of 128 code columns (round 4): batches of 65..256 queries take it (tiling 0x81) with the
    sample stage on the same tiling; 33..64 queries keep the 128 x 64 tiling; LYNSE_HIP_QS_WIDTHS=0 is the round-3 path — identical bits,
    and the oracle's answers.  (500 dimensions: codes padded to 512 columns.)"""
    import os

    rng = np.random.default_rng(600 + dim)
    n, k = 400_000, 10
    data = rng.random((n, dim), dtype=f32) if metric == "ip" else rng.standard_normal((n, dim)).astype(f32)
    queries = (data[rng.integers(0, n, 256)] + 0.03 * rng.standard_normal((256, dim))).astype(f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    idx.profile_enable(True)
    m = {"ip": O.IP, "cosine": O.COS}[metric]
    for nq, want in ((256, 0x81), (100, 0x81), (48, 0x14)):
        idx.profile_get(reset=True)
        rows, dists, counts = idx.search_batch_arrays(queries[:nq], k, metric)
        p = idx.profile_get(reset=True)
        flags, stages, tiling = plan_fields(p)
        assert tiling == want and flags & PLAN_I8C and p["fallback_queries"] == 0, (nq, hex(tiling), p)
        assert bool(int(p["last_plan"]) & PLAN_QS_SAMPLE) == (want == 0x81 and dim != 1024), hex(int(p["last_plan"]))   # (1024 columns: 32-row tiles, the sample stays on k_scan_h16)
        os.environ["LYNSE_HIP_QS_WIDTHS"] = "0"
        try:
            r0, d0, c0 = idx.search_batch_arrays(queries[:nq], k, metric)
            p0 = idx.profile_get(reset=True)
        finally:
            del os.environ["LYNSE_HIP_QS_WIDTHS"]
        assert plan_fields(p0)[2] != 0x81 and p0["fallback_queries"] == 0, p0
        assert np.array_equal(r0, rows) and np.array_equal(d0.view(np.uint32), dists.view(np.uint32)) and np.array_equal(c0, counts)
        for qi in sorted({0, nq // 2, nq - 1}):
            e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, m)
            assert np.array_equal(rows[qi].astype(np.uint32), e_ids) and np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32)), (nq, qi)


@pytest.mark.parametrize("dim", [256, 512, 640])
def test_plain_l2_on_the_query_stationary_tiling_at_other_widths(L, oracle, dim):
    """k_scan_qs<NSLAB, .., MET = 1> for 2..5 slabs: 65..128 queries at every width, 129..256 queries from 512 columns on (below, the <4,2,2,4>
    tiling is as fast).  Oracle parity and identical bits with LYNSE_HIP_QS_WIDTHS=0."""
    import os

    rng = np.random.default_rng(700 + dim)
    n, k = 400_000, 10
    data = rng.standard_normal((n, dim)).astype(f32)
    queries = (data[rng.integers(0, n, 256)] + 0.05 * rng.standard_normal((256, dim))).astype(f32)
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    idx.profile_enable(True)
    for nq in (256, 100):
        idx.profile_get(reset=True)
        rows, dists, counts = idx.search_batch_arrays(queries[:nq], k, "l2")
        p = idx.profile_get(reset=True)
        flags, stages, tiling = plan_fields(p)
        assert tiling == (0x81 if (nq == 100 or dim >= 512) else 0x42) and flags & PLAN_I8C and p["fallback_queries"] == 0, (nq, hex(tiling), p)
        os.environ["LYNSE_HIP_QS_WIDTHS"] = "0"
        try:
            r0, d0, c0 = idx.search_batch_arrays(queries[:nq], k, "l2")
            p0 = idx.profile_get(reset=True)
        finally:
            del os.environ["LYNSE_HIP_QS_WIDTHS"]
        assert plan_fields(p0)[2] != 0x81 and p0["fallback_queries"] == 0, p0
        assert np.array_equal(r0, rows) and np.array_equal(d0.view(np.uint32), dists.view(np.uint32)) and np.array_equal(c0, counts)
        for qi in sorted({0, nq // 2, nq - 1}):
            e_ids, e_d = oracle.canonical_topk(queries[qi], data, k, O.L2)
            assert np.array_equal(rows[qi].astype(np.uint32), e_ids) and np.array_equal(dists[qi].view(np.uint32), e_d.view(np.uint32)), (nq, qi)


@pytest.mark.parametrize("bits", [512, 2048])
def test_batched_hamming_on_the_query_stationary_tiling_at_other_widths(L, oracle, bits):
    """k_scan_qs<.., F4> for 512- and 2048-bit fingerprints (2 / 8 slabs of FP4 nibbles): 100 and 256 packed queries, bit-exact against the
    oracle and identical to the 256 x 256 tile (LYNSE_HIP_QS_F4=0)."""
    import os

    from lynsedb_amd.datasets import packed_bernoulli

    n, k = 600_000, 50
    words = packed_bernoulli(n, bits, 0.5, 77)
    idx = L.FlatIndex(None, bits)
    idx.write_packed(words)
    idx.finalize()
    rng = np.random.default_rng(bits)
    qw = words[rng.integers(0, n, 256)].copy()
    qw ^= packed_bernoulli(256, bits, 0.5, 78) & packed_bernoulli(256, bits, 0.5, 79) & packed_bernoulli(256, bits, 0.5, 80)
    idx.profile_enable(True)
    for nq in (256, 100):
        idx.profile_get(reset=True)
        rows, dists, counts = idx.search_packed_arrays(qw[:nq], k, "hamming")
        p = idx.profile_get(reset=True)
        assert plan_fields(p)[2] == 0x81 and p["fallback_queries"] == 0, (nq, p)
        os.environ["LYNSE_HIP_QS_F4"] = "0"
        try:
            r0, d0, c0 = idx.search_packed_arrays(qw[:nq], k, "hamming")
            p0 = idx.profile_get(reset=True)
        finally:
            del os.environ["LYNSE_HIP_QS_F4"]
        assert plan_fields(p0)[2] != 0x81 and p0["fallback_queries"] == 0, p0
        assert np.array_equal(r0, rows) and np.array_equal(d0, dists) and np.array_equal(c0, counts)
        for qi in sorted({0, nq // 2, nq - 1}):
            assert_rows_equal(oracle.canonical_topk_packed(qw[qi], words, k, O.HAMMING), rows[qi], dists[qi], counts[qi], (bits, nq, qi))
