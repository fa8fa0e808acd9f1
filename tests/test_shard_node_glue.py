"""Host-only shard-node glue: the C-ABI entry points run without a GPU.  Cases follow the reference's
code paths (src/engine.rs:3286-3418, src/rpc.rs:1156-1177, src/cluster.rs:404-435, test :738-749)."""
import numpy as np
import pytest

from lynsedb_amd import shard_node as N

f32 = np.float32


def test_filter_tombstoned_limit():
    ids, d = [5, 9, 2, 7, 1], [0.1, 0.2, 0.3, 0.4, 0.5]
    i, x = N.filter_tombstoned_limit(ids, d, [], 3)           # empty set: plain take(limit)
    assert i.tolist() == [5, 9, 2] and np.array_equal(x, np.array([0.1, 0.2, 0.3], f32))
    i, x = N.filter_tombstoned_limit(ids, d, [9, 1, 100], 3)  # order preserved, tombstoned ids dropped before the cut
    assert i.tolist() == [5, 2, 7] and np.array_equal(x, np.array([0.1, 0.3, 0.4], f32))
    i, x = N.filter_tombstoned_limit(ids, d, [5, 9, 2, 7, 1], 3)
    assert i.size == 0
    i, x = N.filter_tombstoned_limit([], [], [1], 3)
    assert i.size == 0
    i, x = N.filter_tombstoned_limit(ids, d, [], 0)
    assert i.size == 0


def test_merge_row_results():
    # ascending metric: duplicates keep the smaller distance, order (distance, id)
    i, d = N.merge_row_results([1, 2, 3], [0.5, 0.2, 0.9], [3, 4, 2], [0.1, 0.2, 0.7], 10, "l2")
    assert i.tolist() == [3, 2, 4, 1] and np.array_equal(d, np.array([0.1, 0.2, 0.2, 0.5], f32))
    # descending metric (ip): the larger distance wins, ties by id
    i, d = N.merge_row_results([1, 2], [0.5, 0.9], [2, 3], [0.95, 0.5], 2, "ip")
    assert i.tolist() == [2, 1] and np.array_equal(d, np.array([0.95, 0.5], f32))
    # one side empty: the other side comes back untouched, even beyond `limit` (engine.rs:3372-3377)
    i, d = N.merge_row_results([4, 1, 9], [3.0, 1.0, 2.0], [], [], 2, "l2")
    assert i.tolist() == [4, 1, 9]
    i, d = N.merge_row_results([], [], [8, 7], [1.0, 0.5], 1, "l2")
    assert i.tolist() == [8, 7]
    with pytest.raises(ValueError):
        N.merge_row_results([1], [1.0], [2], [2.0], 1, "no-such-metric")


def test_result_block_codec_roundtrip():  # cluster.rs:738-749
    enc = N.encode_search_result([7], [0.5], [{"tag": "x"}])
    assert enc[:4] == (1).to_bytes(4, "little") and enc[4:12] == (7).to_bytes(8, "little")
    assert enc[12:16] == np.array([0.5], "<f4").tobytes()
    assert enc[16:20] == len(b'[{"tag":"x"}]').to_bytes(4, "little") and enc[20:] == b'[{"tag":"x"}]'
    ids, d, fields, off = N.decode_search_result(enc, 0)
    assert off == len(enc) and ids.tolist() == [7] and d.tolist() == [0.5] and fields == [{"tag": "x"}]
    # no fields: fields_len = 0
    enc = N.encode_search_result([1, 2, 3], [1.5, 2.5, 3.5])
    assert len(enc) == 4 + 3 * 12 + 4 and enc[-4:] == bytes(4)
    # a batch frame holds several blocks back to back (rpc.rs:643-656 / cluster.rs:140-156)
    frame = N.encode_batch([(np.array([1, 2]), np.array([0.1, 0.2])), (np.array([], np.uint64), np.array([], f32)), (np.array([9]), np.array([3.0]))])
    blocks = N.decode_batch(frame)
    assert [b[0].tolist() for b in blocks] == [[1, 2], [], [9]] and blocks[2][1].tolist() == [3.0]
    # truncated frames and oversized field lengths are errors, not crashes
    with pytest.raises(Exception):
        N.decode_search_result(enc[:10], 0)
    bad = bytearray(N.encode_search_result([7], [0.5], [{"a": 1}]))
    bad[16:20] = (10_000).to_bytes(4, "little")
    with pytest.raises(Exception, match="exceeds frame"):
        N.decode_search_result(bytes(bad), 0)


def test_merge_row_results_with_nan_distances_is_a_total_order():
    """NaN distances (vectors holding NaN) must not reach std::sort through a comparator that is not a strict weak order:
    real distances come first in metric order, NaN entries last, ids ascending inside equal keys."""
    import numpy as np

    from lynsedb_amd import shard_node as N

    nan = float("nan")
    rng = np.random.default_rng(0)
    for asc_metric in ("l2", "ip"):
        ids_l = np.arange(0, 400, dtype=np.uint64)
        d_l = rng.random(400).astype(np.float32)
        d_l[::7] = nan
        ids_r = np.arange(300, 700, dtype=np.uint64)
        d_r = rng.random(400).astype(np.float32)
        d_r[::5] = nan
        ids, d = N.merge_row_results(ids_l, d_l, ids_r, d_r, 10_000, asc_metric)
        assert len(set(ids.tolist())) == ids.size == 700
        real = ~np.isnan(d)
        k = int(real.sum())
        assert real[:k].all() and not real[k:].any()                      # NaN last
        key = d[:k] if asc_metric == "l2" else -d[:k]
        assert np.all(np.diff(key) >= 0)
        assert np.all(np.diff(ids[k:].astype(np.int64)) > 0)              # ids ascending among the NaN entries
