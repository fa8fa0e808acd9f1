#!/usr/bin/env python3
"""Hand-assembles a collection directory in the reference's ON-DISK layout, byte by byte, from the reference's struct
definitions — NOT through lynsedb_amd/storage.py's writer (the reader had only ever met bytes its own
writer produced; no Rust toolchain here to let the reference write them).

Derivation (BirchKwok/lynsedb v0.8.0):
* `VectorManifest { version: u32, generation: u64, id_map_file: String, segments: Vec<SegmentEntry { file: String, rows: u64 }> }`
  (src/storage/vector_store.rs:36-48) serialised with `serde_json::to_vec_pretty` (:294-298): fields in DECLARATION order,
  two-space indent, `"key": value`, no trailing newline;
* state after three `append_encoded_bytes` calls with the test build's 1024-byte segment target (:32-33, :379-445): the first
  segment is `vectors.bin` (generation 1, manifest not yet written), the second and third are
  `vector_segments/seg-{generation + 1:020}-{segments.len():06}.bin` (:370-377) = generation 2 / index 1 and generation 3 /
  index 2, and the manifest on disk is the one persisted by the LAST new segment (generation 3);
* segment files are raw little-endian f32, row-major, no header (flat_mmap.rs:89-109); the last one carries 7 stray bytes of a
  torn append (row counts come from the file length, vector_store.rs:190-205, a partial row is ignored);
* `id_map.bin`: one little-endian u64 user id per row (src/engine.rs:2588-2603), here shorter than the row count (rows past
  its end map to themselves, :3071-3074) and with 3 stray trailing bytes.

Writes tests/golden/reference_layout_fixture.json: {relative path: hex bytes} + the expected rows / ids."""
import json
import struct
from pathlib import Path

DIM = 6
ROWS = [40, 42, 11]            # 40 rows x 24 B = 960 B (<= 1024), 42 rows x 24 B = 1008 B, 11 rows


def value(r, d):               # a closed form: the test recomputes it instead of trusting these bytes' round trip
    return ((r * 37 + d * 11) % 101 - 50) * 0.125


def main():
    files = {}
    r = 0
    names = ["vectors.bin", "vector_segments/seg-%020d-%06d.bin" % (2, 1), "vector_segments/seg-%020d-%06d.bin" % (3, 2)]
    for name, n in zip(names, ROWS):
        b = bytearray()
        for _ in range(n):
            for d in range(DIM):
                b += struct.pack("<f", value(r, d))
            r += 1
        files[name] = bytes(b)
    files[names[2]] += bytes([0xde, 0xad, 0xbe, 0xef, 0x01, 0x02, 0x03])          # torn append: ignored
    manifest = ('{\n  "version": 1,\n  "generation": 3,\n  "id_map_file": "id_map.bin",\n  "segments": [\n'
                '    {\n      "file": "%s",\n      "rows": 40\n    },\n'
                '    {\n      "file": "%s",\n      "rows": 42\n    },\n'
                '    {\n      "file": "%s",\n      "rows": 11\n    }\n  ]\n}') % tuple(names)
    files["vector_manifest.json"] = manifest.encode()
    ids = [10_000_000_000 + 3 * i for i in range(60)]                            # u64 ids beyond u32; only the first 60 rows mapped
    files["id_map.bin"] = b"".join(struct.pack("<Q", i) for i in ids) + b"\x07\x07\x07"
    files["info.json"] = json.dumps({"total_shape": [sum(ROWS), DIM]}, separators=(",", ":")).encode()   # persist_metadata (:312-318); not read
    out = {"dim": DIM, "rows": sum(ROWS), "segment_rows": ROWS, "mapped_ids": ids, "files": {k: v.hex() for k, v in files.items()}}
    Path(__file__).with_name("reference_layout_fixture.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
