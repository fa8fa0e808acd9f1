"""The API-level known-answer tests of the reference's own Python suite that lie ON the hot path (`tests/standard_tests/test_search.py`
with the fixture recipe of `conftest.py:37-55`, and `flat_mmap.rs:6109-6125`), driven through this build's `lynse._core`-shaped mirror
(`lynsedb_amd.Collection` / `FlatIndex`) on the GPU.  Field filters (`where=`), the storage engine and the approximate modes are outside the path;
what the tests below transcribe is the part of each case that reaches `Collection::search`: results, counts, ids, distances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
DIM, N = 8, 20            # tests/standard_tests/conftest.py:7-8


@pytest.fixture(scope="module")
def L():
    import lynsedb_amd as L_

    assert L_._lib.device_count() >= 1
    return L_


def populated(L):
    """conftest.py:44-55: N vectors of np.random.rand(DIM) under seed 42, ids 0..N-1."""
    np.random.seed(42)
    vectors = np.stack([np.random.rand(DIM).astype(f32) for _ in range(N)])
    coll = L.Collection("test_col", DIM)
    coll.add_items(vectors, list(range(N)))
    coll.commit()
    return coll, vectors


def query_vec():
    np.random.seed(0)         # conftest.py:58-62
    return np.random.rand(DIM).astype(f32)


def test_approx_is_ignored_for_hamming_and_jaccard(L):
    """test_search.py:45-74: a 3 x 4 binary collection under FLAT-HAMMING-BINARY / FLAT-JACCARD-BINARY; approx / eps do not change the answer
    (this build has no approximate mode on the path: the flags are accepted and ignored, like the reference does for these metrics)."""
    vectors = np.array([[1.0, 0.0, 1.0, 0.0], [1.0, 1.0, 1.0, 0.0], [0.0, 0.0, 0.0, 0.0]], dtype=f32)
    query = np.array([1.0, 0.0, 1.0, 0.0], dtype=f32)
    for mode, eps, want_ids, want_d in (("FLAT-HAMMING-BINARY", 2.0, [0, 1, 2], [0.0, 1.0, 2.0]),
                                        ("FLAT-JACCARD-BINARY", 0.5, [0, 1, 2], [0.0, 1.0 / 3.0, 1.0])):
        coll = L.Collection(mode.lower(), 4)
        coll.add_items(vectors, list(range(len(vectors))))
        coll.commit()
        coll.build_index(mode, None)
        exact = coll.search(query, 3, approx=False)
        approx = coll.search(query, 3, approx=True, eps=eps)
        assert approx.ids().tolist() == exact.ids().tolist() == want_ids
        assert np.allclose(approx.distances(), exact.distances()) and np.allclose(exact.distances(), want_d, atol=1e-6)
        assert not np.allclose(exact.distances(), np.round(exact.distances() / eps) * eps)


@pytest.mark.parametrize("index_mode", ["FLAT-HAMMING-BINARY", "FLAT-JACCARD-BINARY", "FLAT-TANIMOTO-BINARY", "FLAT-DICE-BINARY"])
def test_binary_flat_metrics_find_exact_self(L, index_mode):
    """test_search.py:191-212 (the metrics of this path): 32 x 16 bit vectors under rng 20260620, the stored row 7 finds itself at distance 0."""
    rng = np.random.default_rng(20260620)
    vectors = rng.integers(0, 2, size=(32, 16)).astype(f32)
    coll = L.Collection("domain_" + index_mode.lower().replace("-", "_"), 16)
    coll.add_items(vectors, list(range(32)))
    coll.commit()
    coll.build_index(index_mode, None)
    result = coll.search(vectors[7], 1)
    # (several stored rows may equal row 7 bit for bit under a 16-bit alphabet: the canonical order returns the smallest such id)
    first = int(np.nonzero((vectors == vectors[7]).all(axis=1))[0][0])
    assert result.ids().tolist() == [first]
    assert result.distances()[0] == pytest.approx(0.0, abs=1e-5)


@pytest.mark.parametrize("index_mode", ["FLAT-IP", "FLAT-L2", "FLAT-COS"])
def test_float_flat_metrics_find_exact_self(L, index_mode):
    """The float twin of the case above (`rng.random((32, 16)) + 0.01`): L2 / cosine distance 0 at the row itself; IP: the row is its own best
    match only by norm, so the check is the score = |v|^2 of the returned row."""
    rng = np.random.default_rng(20260620)
    vectors = rng.random((32, 16), dtype=f32) + f32(0.01)
    coll = L.Collection("float_" + index_mode.lower(), 16)
    coll.add_items(vectors, list(range(32)))
    coll.commit()
    coll.build_index(index_mode, None)
    result = coll.search(vectors[7], 1)
    if index_mode == "FLAT-IP":
        best = int(np.argmax(vectors @ vectors[7]))
        assert result.ids().tolist() == [best]
    else:
        assert result.ids().tolist() == [7] and result.distances()[0] == pytest.approx(0.0, abs=1e-5)


def test_edge_cases_of_search(L):
    """test_search.py:691-718: empty collection -> no results; k > N -> N results; k = 1; everything deleted -> empty; restore brings ids back."""
    empty = L.Collection("empty", DIM)
    assert len(empty.search(query_vec(), 5)) == 0
    coll, _ = populated(L)
    q = query_vec()
    assert len(coll.search(q, N * 10).ids()) == N
    r1 = coll.search(q, 1)
    assert len(r1.ids()) == 1 and len(r1.distances()) == 1
    coll.delete_items(list(range(N)))
    assert len(coll.search(q, 5).ids()) == 0
    coll.restore_items([0, 1, 2])
    got = coll.search(q, 5).ids().tolist()
    assert len(got) == 3 and all(rid in (0, 1, 2) for rid in got)


def test_search_list_input_flat_l2_and_batch(L):
    """test_search.py:168-177, :727-737: a plain list as the query; FLAT-L2 after build_index; two different queries give different results."""
    coll, vectors = populated(L)
    assert len(coll.search([0.1] * DIM, 3).ids()) == 3
    coll.build_index("FLAT-L2", None)
    res = coll.search(query_vec(), 5)
    assert len(res.ids()) == 5 and np.all(np.diff(res.distances()) >= 0)
    d = ((vectors - query_vec()) ** 2).sum(1)
    assert res.ids().tolist() == np.argsort(d, kind="stable")[:5].tolist()
    np.random.seed(10)
    q1 = np.random.rand(DIM).astype(f32)
    np.random.seed(20)
    q2 = np.random.rand(DIM).astype(f32)
    out = coll.batch_search(np.stack([q1, q2]), 5)
    assert len(out) == 2 and out[0].ids().tolist() != out[1].ids().tolist()


def test_flat_mmap_reopen(L, tmp_path):
    """flat_mmap.rs:6109-6125: two rows written to a raw f32 segment file, the store reopened from the file: len 2, L2 search of row 0 -> id 0."""
    path = tmp_path / "vectors.bin"
    np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], dtype="<f4").tofile(path)
    store = L.FlatIndex(str(path), 3)
    assert len(store) == 2
    ids, dists = store.search(np.array([1.0, 2.0, 3.0], f32), k=1, metric="l2")
    assert ids.tolist() == [0] and dists[0] == 0.0
