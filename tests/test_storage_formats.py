"""On-disk formats of the reference: manifest / segments / id_map / ivf_meta — host logic, CPU only.
Known-answer cases transcribed from the reference's own tests (src/storage/vector_store.rs:1269-1399)."""
import json

import numpy as np
import pytest

from lynsedb_amd import storage as S

f32 = np.float32


def test_manifest_rejects_escaping_and_missing_segment_paths(tmp_path):  # vector_store.rs:1269-1307
    col = tmp_path / "collection"
    col.mkdir()
    (tmp_path / "outside.bin").write_bytes(bytes(16))
    m = {"version": 1, "generation": 1, "id_map_file": "id_map.bin", "segments": [{"file": "../outside.bin", "rows": 1}]}
    (col / S.VECTOR_MANIFEST_FILE).write_text(json.dumps(m))
    with pytest.raises(S.StorageError, match="safe relative path"):
        S.load_manifest(col, 4)
    m["segments"][0]["file"] = "vector_segments/missing.bin"
    (col / S.VECTOR_MANIFEST_FILE).write_text(json.dumps(m))
    with pytest.raises(S.StorageError, match="is unavailable"):
        S.load_manifest(col, 4)
    for bad in ("/abs.bin", "", "a/./b.bin", "a//b.bin"):
        with pytest.raises(S.StorageError):
            S.validate_manifest_path(bad, "segment path")
    m["version"] = 2
    (col / S.VECTOR_MANIFEST_FILE).write_text(json.dumps(m))
    with pytest.raises(S.StorageError, match="newer than supported"):
        S.load_manifest(col, 4)


def test_legacy_partial_row_is_trimmed(tmp_path):  # vector_store.rs:1386-1398
    (tmp_path / "vectors.bin").write_bytes(np.ones(8, "<f4").tobytes() + bytes([1, 2, 3]))
    m = S.load_manifest(tmp_path, 4)
    assert [(s.file, s.rows) for s in m.segments] == [("vectors.bin", 2)]
    segs = list(S.read_segments(tmp_path, 4))
    assert len(segs) == 1 and segs[0][0] == 0 and segs[0][1].shape == (2, 4)
    assert S.load_manifest(tmp_path / "nothing_here", 4).segments == []


def test_segmented_write_layout_matches_reference_naming(tmp_path):  # vector_store.rs:1309-1318 (target 1024 B in tests)
    data = np.arange(400, dtype=f32).reshape(100, 4)
    m = S.write_flat_collection(tmp_path, [data, data], segment_target_bytes=1024)
    # 100 rows x 16 B = 1600 B > 1024: every write opens a segment; the first one is vectors.bin, the manifest appears with the second
    assert [s.file for s in m.segments] == ["vectors.bin", "vector_segments/seg-00000000000000000002-000001.bin"]
    assert [s.rows for s in m.segments] == [100, 100] and m.generation == 2
    on_disk = json.loads((tmp_path / S.VECTOR_MANIFEST_FILE).read_text())
    assert on_disk["version"] == 1 and on_disk["id_map_file"] == "id_map.bin" and len(on_disk["segments"]) == 2
    back = S.load_manifest(tmp_path, 4)
    assert [(s.file, s.rows) for s in back.segments] == [(s.file, s.rows) for s in m.segments]
    rows = np.concatenate([a for _, a in S.read_segments(tmp_path, 4)])
    assert rows.shape == (200, 4) and np.array_equal(rows[100:], data)
    # small batches append to the current segment until the target size is reached
    m2 = S.write_flat_collection(tmp_path / "b", [data[:10], data[10:30], data[30:]], segment_target_bytes=1024)
    assert [(s.file, s.rows) for s in m2.segments] == [("vectors.bin", 30), ("vector_segments/seg-00000000000000000002-000001.bin", 70)]


def test_id_map_roundtrip_and_fallback(tmp_path):  # engine.rs:2588-2617, :3071-3074
    ids = np.array([10, 11, 500, 7], np.uint64)
    S.write_flat_collection(tmp_path, [np.zeros((4, 2), f32)], ids=ids)
    raw = (tmp_path / "id_map.bin").read_bytes()
    assert raw == ids.astype("<u8").tobytes()
    (tmp_path / "id_map.bin").write_bytes(raw + b"\x01\x02")  # partial trailing bytes are ignored
    got = S.load_id_map(tmp_path / "id_map.bin")
    assert np.array_equal(got, ids)
    assert S.rows_to_user_ids(np.array([2, 0, 9]), got).tolist() == [500, 10, 9]  # rows past the map are their own id
    assert S.load_id_map(tmp_path / "missing.bin").size == 0


def test_ivf_meta_roundtrip(tmp_path):  # ivf_flat_mmap.rs:448-530
    rng = np.random.default_rng(1)
    meta = S.IvfMeta(3, 5, 2, rng.standard_normal((2, 3)).astype(f32), np.array([0, 2, 5], np.uint64), np.array([4, 1, 0, 2, 3], np.uint32))
    p = S.ivf_meta_path(tmp_path / "vecs.bin")
    assert p.name == "vecs.ivf_meta.bin"
    S.save_ivf_meta(p, meta)
    raw = p.read_bytes()
    assert len(raw) == 24 + 2 * 3 * 4 + 3 * 8 + 5 * 4
    assert np.frombuffer(raw[:24], "<u8").tolist() == [3, 5, 2]
    back = S.load_ivf_meta(p)
    assert (back.dim, back.n_vectors, back.n_partitions) == (3, 5, 2)
    assert np.array_equal(back.centroids, meta.centroids) and np.array_equal(back.partition_offsets, meta.partition_offsets)
    assert np.array_equal(back.original_ids, meta.original_ids)
    assert S.ivf_assignments_from_meta(back).tolist() == [1, 0, 1, 1, 0]
    p.write_bytes(raw[:-3])
    with pytest.raises(IOError):
        S.load_ivf_meta(p)


def test_pending_update_journal_is_refused(tmp_path):  # vector_store.rs:28, :682-687
    data = np.arange(40, dtype=f32).reshape(10, 4)
    S.write_flat_collection(tmp_path, [data])
    assert S.load_manifest(tmp_path, 4).segments[0].rows == 10
    (tmp_path / S.UPDATE_JOURNAL_FILE).write_bytes(b"\x00" * 16)
    with pytest.raises(S.StorageError, match="pending row updates"):
        S.load_manifest(tmp_path, 4)
    with pytest.raises(S.StorageError):
        list(S.read_segments(tmp_path, 4))


def test_write_appends_to_an_existing_collection(tmp_path):  # VectorStore::write appends (vector_store.rs:379-445)
    a = np.arange(40, dtype=f32).reshape(10, 4)
    b = a + 100
    S.write_flat_collection(tmp_path, [a], ids=np.arange(10, dtype=np.uint64))
    m = S.write_flat_collection(tmp_path, [b], ids=np.arange(50, 60, dtype=np.uint64))   # second call: must not truncate
    assert sum(s.rows for s in m.segments) == 20
    rows = np.concatenate([x for _, x in S.read_segments(tmp_path, 4)])
    assert np.array_equal(rows, np.concatenate([a, b]))
    assert S.load_id_map(tmp_path / "id_map.bin").tolist() == list(range(10)) + list(range(50, 60))
    # ... also across the segment boundary, with the manifest on disk
    c = a + 1000
    m = S.write_flat_collection(tmp_path, [c], segment_target_bytes=64)
    assert [s.rows for s in m.segments] == [20, 10] and (tmp_path / S.VECTOR_MANIFEST_FILE).exists()
    rows = np.concatenate([x for _, x in S.read_segments(tmp_path, 4)])
    assert np.array_equal(rows, np.concatenate([a, b, c]))


def test_open_ivf_flat_reports_a_short_data_file(tmp_path):  # ivf_flat_mmap.rs:161-223
    meta = S.IvfMeta(3, 5, 2, np.zeros((2, 3), f32), np.array([0, 2, 5], np.uint64), np.array([4, 1, 0, 2, 3], np.uint32))
    data = tmp_path / "vecs.bin"
    S.save_ivf_meta(S.ivf_meta_path(data), meta)
    data.write_bytes(np.zeros(4 * 3, "<f4").tobytes())   # 4 rows instead of 5
    with pytest.raises(IOError, match="shorter than its metadata"):
        S.open_ivf_flat(data)


def _materialise_reference_layout(tmp_path, golden_dir):
    import json

    fx = json.loads((golden_dir / "reference_layout_fixture.json").read_text())
    for rel, hx in fx["files"].items():
        f = tmp_path / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_bytes(bytes.fromhex(hx))
    return fx


def test_hand_assembled_reference_layout_is_read_byte_for_byte(tmp_path, golden_dir):
    """A collection directory assembled BY HAND from the reference's struct definitions and serde_json's pretty format
    (tests/golden/make_reference_layout_fixture.py: vector_store.rs:24-66, :294-298, :370-445; engine.rs:2588-2617) — not by this
    package's writer: manifest field order / segment naming after three appends / raw LE f32 rows / a torn trailing append /
    a short id map with stray bytes."""
    fx = _materialise_reference_layout(tmp_path, golden_dir)
    dim, n = fx["dim"], fx["rows"]
    m = S.load_manifest(tmp_path, dim)
    assert (m.version, m.generation, m.id_map_file) == (1, 3, "id_map.bin")
    assert [s.file for s in m.segments] == ["vectors.bin", "vector_segments/seg-00000000000000000002-000001.bin",
                                            "vector_segments/seg-00000000000000000003-000002.bin"]
    assert [s.rows for s in m.segments] == fx["segment_rows"]            # the 7 torn bytes of the last segment are ignored
    got = np.concatenate([rows for _, rows in S.read_segments(tmp_path, dim)])
    want = np.array([[((r * 37 + d * 11) % 101 - 50) * 0.125 for d in range(dim)] for r in range(n)], np.float32)
    assert got.shape == (n, dim) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ids = S.load_id_map(tmp_path / m.id_map_file)
    assert ids.dtype == np.uint64 and ids.tolist() == fx["mapped_ids"]   # 3 stray bytes ignored
    assert S.rows_to_user_ids(np.array([0, 59, 60, 92]), ids).tolist() == [10_000_000_000, 10_000_000_177, 60, 92]
    # and this package's own writer, continuing the SAME directory, names the next segment as the reference would
    S.write_flat_collection(tmp_path, [np.zeros((50, dim), np.float32)], segment_target_bytes=1024)
    m2 = S.load_manifest(tmp_path, dim)
    assert m2.generation == 4 and m2.segments[-1].file == "vector_segments/seg-00000000000000000004-000003.bin"
