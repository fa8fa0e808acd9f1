"""Pin the CPU oracle against the reference's OWN known-answer tests.

Each test names the reference test it transcribes (file:line relative to the upstream tree).
The inputs/expected values are data transcribed from those tests; the arithmetic under test is
oracle/lynse_oracle.c.  CPU only.
"""
import json

import numpy as np
import pytest

import oracle as O

IP, L2, COS, HAM, JAC, DICE, TANI = O.IP, O.L2, O.COS, O.HAMMING, O.JACCARD, O.DICE, O.TANIMOTO
f32 = np.float32


def arr(x):
    return np.asarray(x, dtype=np.float32)


# ----------------------------------------------------------------- src/distance/simd.rs tests

def test_simd_inner_product(oracle):  # simd.rs:2906-2912
    assert abs(oracle.ip_single(arr([1, 2, 3, 4]), arr([4, 3, 2, 1])) - 20.0) < 1e-5


def test_simd_l2_squared(oracle):  # simd.rs:2914-2920
    assert abs(oracle.l2_single(arr([1, 0, 0]), arr([0, 1, 0])) - 2.0) < 1e-5


def test_simd_l2_batch8_matches_single(oracle):  # simd.rs:2922-2941 (x86 batch8 == 8 singles)
    dim = 128
    q = arr([(i - 64.0) * 0.01 for i in range(dim)])
    rows = [arr([(((row * 13 + i * 7) % 31) - 15.0) * 0.02 for i in range(dim)]) for row in range(8)]
    for r in rows:
        assert abs(oracle.l2_single(q, r) - oracle.l2_scalar(q, r)) < 1e-4


def test_simd_cosine(oracle):  # simd.rs:2943-2957
    assert abs(oracle.cos_single(arr([1, 0, 0]), arr([1, 0, 0]))) < 1e-5
    assert abs(oracle.cos_single(arr([1, 0]), arr([0, 1])) - 1.0) < 1e-5


def test_simd_high_dim_ip(oracle):  # simd.rs:3090-3098
    dim = 768
    a = arr([i * f32(0.001) for i in map(f32, range(dim))])
    b = arr([f32(dim - i) * f32(0.001) for i in range(dim)])
    assert abs(oracle.ip_single(a, b) - oracle.ip_scalar(a, b)) < 1e-2


def test_simd_ip_batch_forms_agree(oracle):  # simd.rs:3100-3113 (batch4 vs single <1e-3); here batch8
    dim = 128
    q = arr([f32(i) * f32(0.01) for i in range(dim)])
    rows = np.stack([arr([f32(i + j + 1) * f32(0.01) for i in range(dim)]) for j in range(8)])
    b8 = oracle.ip_batch8(q, rows)
    for j in range(8):
        assert abs(b8[j] - oracle.ip_single(q, rows[j])) < 1e-3
        assert b8[j] == f32(oracle.ip_batch8_row(q, rows[j]))  # batch8 == its per-row restatement, bitwise


def test_zero_norm_cosine_rule(oracle):  # simd.rs:1629-1634 denom<1e-30 -> 1.0
    assert oracle.cos_single(arr([0, 0, 0, 0]), arr([1, 2, 3, 4])) == 1.0
    assert oracle.cos_scalar(arr([0, 0]), arr([0, 0])) == 1.0


# ------------------------------------------- tests/standard_tests/test_backend.py (distance KATs)

@pytest.mark.parametrize("a,b,m,exp,tol", [
    ([1, 0, 0], [0, 1, 0], IP, 0.0, 1e-5),      # test_backend.py:25-29
    ([1, 0, 0], [1, 0, 0], IP, 1.0, 1e-5),      # :31-34
    ([1, 2, 3], [1, 2, 3], L2, 0.0, 1e-5),      # :36-39
    ([0, 0], [3, 4], L2, 25.0, 1e-4),           # :41-45
    ([1, 0], [1, 0], COS, 0.0, 1e-5),           # :47-50
    ([1, 0], [0, 1], COS, 1.0, 1e-5),           # :52-56
    ([1, 1, 0], [1, 0, 1], DICE, 0.5, 1e-5),    # :76
    ([1, 1, 0], [1, 0, 1], TANI, 2.0 / 3.0, 1e-5),  # :77
    ([3, 4], [3, 4], IP, 25.0, 1e-4),           # :218-221
    ([1, 2, 3, 4], [1, 2, 3, 4], COS, 0.0, 1e-5),   # :223-225
])
def test_backend_distance_kats(oracle, a, b, m, exp, tol):
    assert abs(oracle.compute_distance(arr(a), arr(b), m) - exp) < tol


def test_backend_matches_numpy(oracle):  # test_backend.py:235-249
    np.random.seed(5)
    a = np.random.rand(16).astype(f32)
    b = np.random.rand(16).astype(f32)
    assert abs(oracle.compute_distance(a, b, IP) - float(np.dot(a, b))) < 1e-4
    np.random.seed(6)
    a = np.random.rand(16).astype(f32)
    b = np.random.rand(16).astype(f32)
    assert abs(oracle.compute_distance(a, b, L2) - float(np.sum((a - b) ** 2))) < 1e-4


def test_backend_topk(oracle):  # test_backend.py:107-190 (N=200, D=16, seeds 7 / 1)
    np.random.seed(7)
    vecs = np.random.rand(200, 16).astype(f32)
    np.random.seed(1)
    q = np.random.rand(16).astype(f32)
    ids, d = oracle.top_k_search(q, vecs, 10, IP)
    assert len(ids) == 10 and np.all(np.isfinite(d))
    ids, _ = oracle.top_k_search(q, vecs, 300, IP)  # :139-141 k>N -> N
    assert len(ids) == 200
    eye = np.eye(16, dtype=f32)
    for i in range(16):  # :123-135, :156-160
        ids, d = oracle.top_k_search(eye[i], eye, 1, IP)
        assert ids[0] == i and abs(d[0] - 1.0) < 1e-5
    ids, d = oracle.top_k_search(eye[3], eye, 1, L2)
    assert ids[0] == 3 and abs(d[0]) < 1e-5
    ids, _ = oracle.top_k_search(q, vecs, 1, IP)  # :164-168
    assert ids[0] == int(np.argmax(vecs @ q))
    ids, _ = oracle.top_k_search(q, vecs, 1, L2)  # :170-175
    assert ids[0] == int(np.argmin(np.sum((vecs - q) ** 2, axis=1)))
    np.random.seed(42)  # :177-186
    v2 = np.random.rand(50, 16).astype(f32)
    q2 = np.random.rand(16).astype(f32)
    ids, _ = oracle.top_k_search(q2, v2, 1, COS)
    qn = q2 / (np.linalg.norm(q2) + 1e-9)
    vn = v2 / (np.linalg.norm(v2, axis=1, keepdims=True) + 1e-9)
    assert ids[0] == int(np.argmax(vn @ qn))
    # same answers from the FlatMmap path and the canonical order
    for m in (IP, L2, COS):
        a = oracle.top_k_search(q, vecs, 20, m)
        b = oracle.flat_search(q, vecs, 20, m)
        c = oracle.canonical_topk(q, vecs, 20, m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(b[0], c[0])
        assert np.array_equal(a[1], b[1]) and np.array_equal(b[1], c[1])


# ------------------------------------------------------------- src/distance/mod.rs tests

def test_top_k_ip(oracle):  # distance/mod.rs:502-515
    c = arr([[1, 0, 0, 0], [0.5, 0.5, 0, 0], [0, 1, 0, 0]])
    ids, d = oracle.top_k_search(arr([1, 0, 0, 0]), c, 2, IP)
    assert len(ids) == 2 and ids[0] == 0 and abs(d[0] - 1.0) < 1e-6


def test_top_k_l2(oracle):  # distance/mod.rs:517-527
    c = arr([[1, 0, 0], [0.1, 0, 0], [2, 0, 0]])
    ids, _ = oracle.top_k_search(arr([0, 0, 0]), c, 2, L2)
    assert ids[0] == 1


def test_top_k_larger(oracle):  # distance/mod.rs:529-550
    dim, n = 16, 1000
    c = (np.arange(n * dim, dtype=np.float32) * f32(0.001)).reshape(n, dim)
    ids, d = oracle.top_k_search(c[0], c, 5, L2)
    assert len(ids) == 5 and ids[0] == 0 and d[0] < 1e-6
    assert np.all(np.diff(d) >= 0)


def test_top_k_empty_and_zero_k(oracle):  # distance/mod.rs:571-584
    ids, d = oracle.top_k_search(arr([1, 2]), np.zeros((0, 2), f32), 5, L2)
    assert len(ids) == 0 and len(d) == 0
    ids, d = oracle.top_k_search(arr([1, 2]), arr([[1, 2], [3, 4]]), 0, L2)
    assert len(ids) == 0 and len(d) == 0
    ids, d = oracle.flat_search(arr([1, 2]), np.zeros((0, 2), f32), 5, L2)  # flat_mmap.rs:832-835
    assert len(ids) == 0


def test_top_k_clamps_k(oracle):  # distance/mod.rs:586-599
    ids, d = oracle.top_k_search(arr([0, 0]), arr([[2, 0], [1, 0]]), 10, L2)
    assert list(ids) == [1, 0] and len(d) == 2 and d[0] <= d[1]
    ids, d = oracle.flat_search(arr([0, 0]), arr([[2, 0], [1, 0]]), 10, L2)
    assert list(ids) == [1, 0]


def test_top_k_binary_metrics(oracle):  # distance/mod.rs:601-621
    q = arr([1, 0, 1, 0])
    c = arr([[1, 0, 1, 0], [1, 1, 1, 0], [0, 1, 0, 1]])
    for fn in (oracle.top_k_search, oracle.flat_search, oracle.canonical_topk):
        ids, d = fn(q, c, 3, HAM)
        assert list(ids) == [0, 1, 2] and list(d) == [0.0, 1.0, 4.0]
        ids, d = fn(q, c, 3, JAC)
        assert list(ids) == [0, 1, 2]
        assert abs(d[0]) < 1e-6 and abs(d[1] - 1 / 3) < 1e-6 and abs(d[2] - 1.0) < 1e-6


def test_metric_aliases(oracle):  # distance/mod.rs:645-692 (in-scope rows)
    assert oracle.metric_from_str("DOT") == IP
    assert oracle.metric_from_str("euclidean") == L2
    assert oracle.metric_from_str("cosine_distance") == COS
    assert oracle.metric_from_str("unknown") == -1
    assert oracle.metric_from_index_mode("FLAT-TANIMOTO-BINARY") == TANI
    assert oracle.metric_from_index_mode("FLAT-BOGUS") == -1
    assert oracle.metric_from_index_mode("FLAT-HAMMING-BINARY") == HAM
    assert oracle.metric_from_index_mode("IVF-IP") == IP
    assert oracle.is_binary(DICE) and not oracle.is_binary(COS)
    assert not oracle.is_ascending(IP) and all(oracle.is_ascending(m) for m in (L2, COS, HAM, JAC, DICE, TANI))


# ------------------------------------------------------------- src/storage/flat_mmap.rs tests

def test_flat_mmap_write_search(oracle):  # flat_mmap.rs:6022-6054
    data = arr([[1, 0, 0, 0], [0, 1, 0, 0], [0.5, 0.5, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    ids, d = oracle.flat_search(arr([1, 0, 0, 0]), data, 2, IP)
    assert len(ids) == 2 and ids[0] == 0 and abs(d[0] - 1.0) < 1e-6
    ids, _ = oracle.flat_search(arr([0, 0, 0, 0]), data, 1, L2)
    assert ids[0] == 2


def test_packed_binary_cache_matches_reference_metrics(oracle):  # flat_mmap.rs:6386-6421
    dim = 130
    rows = np.zeros((3, dim), f32)
    for i in (0, 1, 64, 129):
        rows[0, i] = 1.0
        rows[1, i] = 1.0
    rows[1, 5] = 1.0
    for i in (2, 3, 65):
        rows[2, i] = 1.0
    q = rows[0].copy()
    packed = oracle.pack_binary(rows)
    assert packed.shape == (3, 3) and packed.nbytes == 3 * ((dim + 63) // 64) * 8
    # little-endian bit order inside each u64 word (simd.rs:750-757)
    assert int(packed[0, 0]) == (1 << 0) | (1 << 1) and int(packed[0, 1]) == 1 and int(packed[0, 2]) == 1 << 1
    for m in (HAM, JAC, TANI, DICE):
        p_ids, p_d = oracle.flat_search(q, rows, 3, m)            # packed path (flat_mmap.rs:839-845)
        r_ids, r_d = oracle.top_k_search(q, rows, 3, m)           # float-threshold path
        assert np.array_equal(p_ids, r_ids)
        assert np.all(np.abs(p_d - r_d) < 1e-6)
        c_ids, c_d = oracle.canonical_topk_packed(packed[0], packed, 3, m)
        assert np.array_equal(c_ids, p_ids) and np.array_equal(c_d, p_d)


# ------------------------------------------- vector_store.rs / cluster.rs merge KATs

def test_segmented_merge(oracle):  # vector_store.rs:1310-1328 (2 segments x 100 rows, D=4)
    data = np.arange(400, dtype=f32).reshape(100, 4)
    q = arr([0, 1, 2, 3])
    merged_ids, merged_d = [], []
    for seg, base in ((data, 0), (data, 100)):
        ids, d = oracle.flat_search(q, seg, 1, L2)
        merged_ids += [base + int(i) for i in ids]
        merged_d += list(d)
    ids, d = oracle.merge_results(merged_ids, merged_d, 1, L2)
    assert list(ids) == [0]  # tie (0.0, 0.0) broken by row ascending (vector_store.rs:959-967)


def test_cluster_merge_blocks(oracle):  # cluster.rs:674-692, :695-719
    ids, d = oracle.merge_results([1, 2, 3], [0.4, 0.9, 0.8], 2, IP)
    assert list(ids) == [2, 3] and np.allclose(d, [0.9, 0.8])
    ids, d = oracle.merge_results([1, 2, 3], [4.0, 1.0, 2.0], 2, L2)
    assert list(ids) == [2, 3] and np.allclose(d, [1.0, 2.0])
    ids, d = oracle.merge_results([1, 2, 3, 4, 5], [0.1, 0.2, 0.3, 0.4, 0.15], 2, L2)
    assert list(ids) == [1, 5] and np.allclose(d, [0.1, 0.15])
    ids, d = oracle.merge_results([20, 90, 40], [0.2, 0.9, 0.4], 2, IP)  # distance/mod.rs:637-643
    assert list(ids) == [90, 40] and np.allclose(d, [0.9, 0.4])


def test_python_reference_golden_merge(oracle, golden_dir):
    """cluster._merge_pairs outputs captured by importing the reference (tests/golden/make_*.py)."""
    g = json.loads((golden_dir / "python_reference_vectors.json").read_text())
    for case in g["merge_pairs"]:
        ids = [i for blk in case["blocks"] for i in blk[0]]
        sc = [s for blk in case["blocks"] for s in blk[1]]
        metric = L2 if case["ascending"] else IP
        got_ids, got_sc = oracle.merge_results(ids, sc, case["k"], metric)
        assert [int(x) for x in got_ids] == case["ids"]
        assert np.allclose(got_sc, np.asarray(case["scores"], f32))
    for row in g["is_ascending_index"]:
        if row["mode"] is None:
            continue
        m = oracle.metric_from_index_mode(row["mode"])
        assert m >= 0 and oracle.is_ascending(m) == row["ascending"]


# ------------------------------------------------------------- kmeans.rs / ivf.rs tests

def test_fastrng_first_values(oracle):  # kmeans.rs:21-35: x*6364136223846793005+1442695040888963407, >>33 / 2^31
    s = 42
    exp = []
    for _ in range(4):
        s = (s * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        exp.append((s >> 33) / float(1 << 31))
    assert list(oracle.fastrng_stream(42, 4)) == exp
    assert all(0.0 <= x < 1.0 for x in exp)


def _ivf_build(oracle, data, nlist, metric, iters=20):
    routing = L2 if oracle.is_binary(metric) else metric  # ivf.rs:81-87
    cen, asg = oracle.kmeans_train(data, nlist, iters, routing)
    offsets, rows = oracle.lists_from_assignments(asg, cen.shape[0])
    return cen, asg, offsets, rows


def test_ivf_two_clusters(oracle):  # ivf.rs:546-575 (unfiltered part: 2 well-separated clusters)
    data = arr([[0, 0], [0.1, 0], [10, 10], [10.1, 10]])
    cen, asg, off, rows = _ivf_build(oracle, data, 2, L2)
    assert asg[0] == asg[1] and asg[2] == asg[3] and asg[0] != asg[2]
    ids, d, probed = oracle.ivf_search(arr([0, 0]), data, cen, off, rows, 1, 2, L2)
    assert sorted(ids) == [0, 1]


def test_ivf_ip_recall_improves_with_nprobe(oracle):  # ivf.rs:578-638
    n, dim = 800, 32
    i = np.arange(n)[:, None]
    j = np.arange(dim)[None, :]
    data = (((i * 131 + j * 17 + 1) % 997).astype(f32) / f32(997.0) + f32(0.01)).astype(f32)
    cen, asg, off, rows = _ivf_build(oracle, data, 32, IP)
    q = data[0]
    exact = set(np.argsort(-(data.astype(np.float64) @ q.astype(np.float64)), kind="stable")[:10])
    ids_low, _, _ = oracle.ivf_search(q, data, cen, off, rows, 2, 10, IP)
    ids_high, d_high, _ = oracle.ivf_search(q, data, cen, off, rows, 32, 10, IP)
    rec_low = len(exact & set(int(x) for x in ids_low)) / 10
    rec_high = len(exact & set(int(x) for x in ids_high)) / 10
    assert rec_high >= rec_low and rec_high == 1.0
    f_ids, f_d = oracle.canonical_topk(q, data, 10, IP, O.IPFORM_SINGLE)
    assert np.array_equal(np.sort(ids_high), np.sort(f_ids))


def test_ivf_hamming_full_probe_matches_flat(oracle):  # ivf.rs:641-679
    n, dim = 256, 32
    i = np.arange(n)[:, None]
    j = np.arange(dim)[None, :]
    data = (((i * 17 + j * 3) % 2) == 0).astype(f32)
    cen, asg, off, rows = _ivf_build(oracle, data, 16, HAM)
    packed = oracle.pack_binary(data)
    q = data[0]
    exact = np.sort(oracle.all_distances(q, data, HAM), kind="stable")[:10]
    _, got, _ = oracle.ivf_search(q, data, cen, off, rows, 16, 10, HAM, packed=packed)
    assert np.array_equal(got, exact)


# ------------------------------------------------------------- ivf_flat_mmap.rs tests

def _ivf_flat_build(oracle, data, nlist, iters):
    cen, asg = oracle.kmeans_train(data, nlist, iters, L2)  # ivf_flat_mmap.rs:98 train_l2
    offsets, orig = oracle.ivf_flat_layout(asg, cen.shape[0])
    slab = data[orig]
    return cen, offsets, orig, slab, oracle.ivf_routing_dims(cen)


def test_ivf_flat_build_and_search(oracle):  # ivf_flat_mmap.rs:675-719
    data = arr([
        1.0, 0.1, 0.0, 0.0, 0.9, 0.0, 0.1, 0.0, 1.0, 0.0, 0.0, 0.1, 0.8, 0.1, 0.1, 0.0,
        0.0, 1.0, 0.1, 0.0, 0.1, 0.9, 0.0, 0.0, 0.0, 1.0, 0.0, 0.1, 0.1, 0.8, 0.1, 0.0,
        0.0, 0.0, 1.0, 0.1, 0.0, 0.1, 0.9, 0.0, 0.1, 0.0, 1.0, 0.0, 0.0, 0.0, 0.8, 0.1,
    ]).reshape(12, 4)
    cen, off, orig, slab, rd = _ivf_flat_build(oracle, data, 3, 10)
    q = arr([1, 0, 0, 0])
    ids, d = oracle.ivf_flat_search(q, slab, cen, off, orig, 1, 3, IP, rd)
    assert len(ids) == 3 and ids[0] <= 3
    ids0, d0 = oracle.ivf_flat_search(q, slab, cen, off, orig, 0, 3, IP, rd)  # nprobe 0 == 1
    assert np.array_equal(ids0, ids) and np.array_equal(d0, d)


def test_ivf_flat_reopen_case(oracle):  # ivf_flat_mmap.rs:752-773
    data = arr([[1, 0], [0, 1], [-1, 0], [0, -1]])
    cen, off, orig, slab, rd = _ivf_flat_build(oracle, data, 2, 5)
    ids, _ = oracle.ivf_flat_search(arr([1, 0]), slab, cen, off, orig, 2, 1, IP, rd)
    assert ids[0] == 0


def test_ivf_flat_recall(oracle):  # ivf_flat_mmap.rs:776-815 (LCG seed 42, n=1000, D=8, 10 partitions)
    n, dim = 1000, 8
    rng = 42
    vals = np.zeros(n * dim, f32)
    for t in range(n * dim):
        rng = (rng * 6364136223846793005 + 1) % (1 << 64)
        vals[t] = f32(f32(f32(rng >> 33) / f32(4294967295.0)) * f32(2.0)) - f32(1.0)
    data = vals.reshape(n, dim)
    cen, off, orig, slab, rd = _ivf_flat_build(oracle, data, 10, 10)
    q = data[0].copy()
    ivf_ids, _ = oracle.ivf_flat_search(q, slab, cen, off, orig, 10, 5, IP, rd)
    bf_ids, _ = oracle.flat_search(q, data, 5, IP)
    assert ivf_ids[0] == bf_ids[0]
    assert np.array_equal(ivf_ids, bf_ids)  # nprobe = all partitions -> exact (:390-392)


# ------------------------------------------------------------- cross-checks of the restatement itself

@pytest.mark.parametrize("dim", [1, 3, 7, 8, 9, 15, 16, 17, 24, 31, 64, 100, 128, 130, 768, 771])
def test_intrinsic_and_portable_builds_bit_identical(oracle, oracle_portable, dim):
    assert oracle.lib.lo_has_avx2_fma() == 1 and oracle_portable.lib.lo_has_avx2_fma() == 0
    rng = np.random.default_rng(dim)
    a = rng.standard_normal(dim).astype(f32)
    b = rng.standard_normal(dim).astype(f32)
    for name in ("ip_single", "ip_batch8_row", "l2_single", "cos_single"):
        x, y = getattr(oracle, name)(a, b), getattr(oracle_portable, name)(a, b)
        assert f32(x).tobytes() == f32(y).tobytes(), (name, dim)


def test_reference_policy_equals_canonical_without_ties(oracle):
    """The chunked FlatMmap scan (any thread count) and the canonical (dist,id) order agree when
    distances are distinct; IP scores differ by the accumulation form only."""
    rng = np.random.default_rng(11)
    n, dim, k = 9000, 24, 25
    data = rng.random((n, dim), dtype=f32)
    q = rng.random(dim, dtype=f32)
    for m in (L2, COS):
        c = oracle.canonical_topk(q, data, k, m)
        for t in (1, 3, 8):
            r = oracle.flat_search(q, data, k, m, n_threads=t)
            assert np.array_equal(r[0], c[0]) and np.array_equal(r[1], c[1])
        r = oracle.flat_search(q, data, k, m, n_threads=4, mt=True)
        assert np.array_equal(r[0], c[0]) and np.array_equal(r[1], c[1])
    c = oracle.canonical_topk(q, data, k, IP)  # AUTO -> batch8 form for n >= 4096
    for t in (1, 3, 8):
        r = oracle.flat_search(q, data, k, IP, n_threads=t)
        assert np.array_equal(r[0], c[0])
        assert np.allclose(r[1], c[1], rtol=1e-6, atol=0)
    r8 = oracle.flat_search(q, data[:8192], k, IP, n_threads=8)  # 1024-row chunks, all full 8-blocks
    c8 = oracle.canonical_topk(q, data[:8192], k, IP, O.IPFORM_BATCH8)
    assert np.array_equal(r8[0], c8[0]) and np.array_equal(r8[1], c8[1])


def test_packed_search_matches_canonical(oracle):
    rng = np.random.default_rng(5)
    n, words, k = 6000, 4, 50
    rows = rng.integers(0, 2**63, size=(n, words), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, words), dtype=np.uint64)
    q = rows[17].copy()
    for m in (HAM, JAC, DICE):
        c = oracle.canonical_topk_packed(q, rows, k, m)
        for t in (1, 4):
            r = oracle.packed_binary_search(q, rows, k, m, n_threads=t)
            assert np.array_equal(r[1], c[1])  # distances always; ids where untied
            untied = np.ones(k, bool)
            untied[1:] &= c[1][1:] != c[1][:-1]
            untied[:-1] &= c[1][:-1] != c[1][1:]
            assert np.array_equal(r[0][untied], c[0][untied])


def test_binary_quantizer_kat(oracle):  # quantizer/mod.rs:738-773
    f32 = np.float32
    # default threshold 0.5
    enc = oracle.binary_quantize(np.array([0.1, 0.9, 0.3, 0.7], f32), np.full(4, 0.5, f32))
    assert enc[0].tolist() == [0.0, 1.0, 0.0, 1.0]
    # fitted: col0 median==min -> midrange 5, col1 median 2, col2 median 5, col3 constant -> midrange 0
    train = np.array([[0, 1, 0, 0], [0, 2, 5, 0], [10, 3, 10, 0]], f32)
    ab, thr = oracle.binary_fit(train)
    assert not ab and thr.tolist() == [5.0, 2.0, 5.0, 0.0]
    assert oracle.binary_quantize(np.array([1.0, 2.5, 4.0, 1.0], f32), thr)[0].tolist() == [0.0, 1.0, 0.0, 1.0]
    # balanced {0,1} data keeps the 0.5 cut
    bits = np.array([[0, 1, 1, 0], [0, 1, 1, 0]], f32)
    ab, thr = oracle.binary_fit(bits)
    assert ab and np.all(thr == 0.5)
    assert oracle.binary_quantize(bits[0], thr)[0].tolist() == [0.0, 1.0, 1.0, 0.0]


def test_filtered_search_kat_and_policies(oracle):  # vector_store.rs:1309-1329, flat_mmap.rs:491-815
    f32 = np.float32
    data = np.concatenate([np.arange(400, dtype=f32)] * 2).reshape(200, 4)
    q = np.array([0, 1, 2, 3], f32)
    ids, d = oracle.flat_search_filtered(q, data, 1, O.L2, [100])
    assert ids.tolist() == [100] and d.tolist() == [0.0]
    # direct path (<= 50,000 ids, subset order) and bitset path (> 50,000, row order) agree with the canonical answer
    rng = np.random.default_rng(3)
    big = rng.standard_normal((60000, 8)).astype(f32)
    qq = rng.standard_normal(8).astype(f32)
    for m in (10, 4000, 55000):
        sub = np.sort(rng.choice(60000, m, replace=False)).astype(np.uint64)
        for metric in (O.IP, O.L2, O.COS):
            a = oracle.flat_search_filtered(qq, big, 7, metric, sub)
            b = oracle.canonical_topk_filtered(qq, big, 7, metric, sub)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    # k is clamped to the subset length; rows >= n are skipped
    ids, d = oracle.flat_search_filtered(q, data, 5, O.L2, [3, 1000])
    assert ids.tolist() == [3]


def test_f16_storage_kernels(oracle, oracle_portable):  # simd.rs:805-846, dtype.rs encode/decode
    f32 = np.float32
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.standard_normal(5000).astype(f32) * 50, np.array([0, -0.0, 65504, 65520, 1e-8, 6e-8, 2.98e-8, 3e-8], f32)])
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(f32)
    assert np.array_equal(oracle.round_f16(x), ref) and np.array_equal(oracle_portable.round_f16(x), ref)
    q = rng.standard_normal(37).astype(f32)
    c = oracle.round_f16(rng.standard_normal(37).astype(f32))
    # sequential f32 accumulation, separate multiply and add (numpy float32 scalars round after every operation)
    s = f32(0)
    for a, b in zip(q, c):
        s = f32(s + f32(a * b))
    assert oracle.distance_f16(q, c, O.IP) == float(s) == oracle_portable.distance_f16(q, c, O.IP)
    s = f32(0)
    for a, b in zip(q, c):
        d = f32(a - b)
        s = f32(s + f32(d * d))
    assert oracle.distance_f16(q, c, O.L2) == float(s)
    dot = nq = nc = f32(0)
    for a, b in zip(q, c):
        dot = f32(dot + f32(a * b)); nq = f32(nq + f32(a * a)); nc = f32(nc + f32(b * b))
    expect = f32(1) - f32(dot / f32(np.sqrt(nq) * np.sqrt(nc)))
    assert oracle.distance_f16(q, c, O.COS) == float(expect)
    assert oracle.distance_f16(q, np.zeros(37, f32), O.COS) == 1.0 and oracle.distance_f16(np.zeros(37, f32), c, O.COS) == 1.0


def test_sq8_fit_and_codes(oracle, oracle_portable):  # flat_mmap.rs:5685-5750
    f32 = np.float32
    rng = np.random.default_rng(4)
    data = rng.standard_normal((500, 9)).astype(f32)
    data[:, 2] = 0.25          # constant dimension -> scale 0 -> code 0
    mins, scales, codes = oracle.sq8_fit(data)
    assert np.array_equal(mins, data.min(0)) and scales[2] == 0.0 and np.all(codes[:, 2] == 0)
    rngs = data.max(0) - data.min(0)
    with np.errstate(divide="ignore"):
        exp_scales = np.where(rngs > 1e-30, f32(255.0) / rngs, f32(0)).astype(f32)
    assert np.array_equal(scales, exp_scales)
    t = ((data - mins) * scales).astype(f32)
    exp = np.clip(np.where(t >= 0, np.floor(t + f32(0.5)), np.ceil(t - f32(0.5))), 0, 255).astype(np.uint8)  # round half away from zero
    assert np.array_equal(codes, exp)
    assert codes.min() == 0 and codes.max() == 255
    m2, s2, c2 = oracle_portable.sq8_fit(data)
    assert np.array_equal(c2, codes) and np.array_equal(s2, scales)
    # two-pass search: with n <= n_cand every row is a candidate -> the exact answer
    q = data[3] + f32(0.01)
    ids, d = oracle.sq8_search(q, data[:150], *oracle.sq8_fit(data[:150]), 5, O.L2)
    e_ids, e_d = oracle.canonical_topk(q, data[:150], 5, O.L2, O.IPFORM_SINGLE)
    assert np.array_equal(ids, e_ids) and np.array_equal(d, e_d)
