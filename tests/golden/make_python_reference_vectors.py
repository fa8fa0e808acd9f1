"""Generate golden vectors by IMPORTING the reference's pure-Python modules.

Run in the build container only (needs /root/reference; the reference cannot travel):
    python tests/golden/make_python_reference_vectors.py
Writes tests/golden/python_reference_vectors.json (inputs + expected outputs — data only).

Covered reference functions (python/lynse/cluster.py): _hash_u64 (:156-158) and the bucket rule
(:1273, :1364-1370), _merge_pairs (:535-556), _is_ascending_index (:182), and
result_view._parse_index_mode; benchmarks/sift_io.py read_fvecs/read_ivecs (:10-53); the search-result block codec
_encode_search_binary / _split_search_binary (:230-283).
The Rust core (`lynse._core`) cannot be built/imported here, so no reference arithmetic is run.
"""
import json
import struct
import sys
import tempfile
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
sys.path.insert(0, str(REF / "python"))
sys.path.insert(0, str(REF / "benchmarks"))

from lynse import cluster  # noqa: E402
from lynse import result_view  # noqa: E402
import sift_io  # noqa: E402

out = {}

# --- shard routing hash
keys = [f"db/coll/{i}" for i in list(range(0, 40)) + [12345, 999999, 2**31, 2**40 + 7]]
keys += ["bench_db/bench_vectors/0", "bench_db/bench_vectors/1", "a/b/int:5", "x"]
out["hash_u64"] = [{"key": k, "hash": str(cluster._hash_u64(k)), "bucket4096": cluster._hash_u64(k) % 4096}
                   for k in keys]

# --- merge of per-shard (ids, scores) blocks
rng = np.random.default_rng(3)
merge_cases = []
fixed = [
    ([([1, 2], [0.9, 0.5]), ([3, 4], [0.8, 0.1])], 3, False),
    ([([1, 2], [0.4, 0.9]), ([3], [0.8])], 2, False),
    ([([1, 2], [4.0, 1.0]), ([3], [2.0])], 2, True),
    ([([1, 2, 3, 4], [0.1, 0.2, 0.3, 0.4]), ([5], [0.15])], 2, True),
    ([([], []), ([7], [1.0])], 5, True),
    ([([1], [1.0])], 0, True),
]
for blocks, k, asc in fixed:
    res = cluster._merge_pairs([(list(i), list(s), []) for i, s in blocks], k, asc, False)
    merge_cases.append({"blocks": [[list(i), list(s)] for i, s in blocks], "k": k, "ascending": asc,
                        "ids": list(res[0]), "scores": list(res[1])})
for case in range(12):
    nblocks = int(rng.integers(1, 9))
    k = int(rng.integers(1, 40))
    asc = bool(case % 2)
    blocks, next_id = [], 0
    for _ in range(nblocks):
        m = int(rng.integers(0, 50))
        # distinct scores so the (unpinned) tie order cannot matter
        scores = sorted((float(np.float32(x)) for x in rng.random(m) + 1e-3 * np.arange(m)), reverse=not asc)
        ids = list(range(next_id, next_id + m))
        next_id += m
        blocks.append((ids, scores))
    res = cluster._merge_pairs([(i, s, []) for i, s in blocks], k, asc, False)
    merge_cases.append({"blocks": [[i, s] for i, s in blocks], "k": k, "ascending": asc,
                        "ids": list(res[0]), "scores": list(res[1])})
out["merge_pairs"] = merge_cases

# --- index-mode parsing / ordering
modes = ["FLAT-IP", "FLAT-L2", "FLAT-COS", "FLAT-HAMMING-BINARY", "FLAT-JACCARD-BINARY",
         "FLAT-TANIMOTO-BINARY", "FLAT-DICE-BINARY", "IVF-IP", "IVF-L2", "IVF-COS", "IVF-HAMMING-BINARY",
         "flat-ip", None]
out["is_ascending_index"] = [{"mode": m, "ascending": bool(cluster._is_ascending_index(m))} for m in modes]
out["parse_index_mode"] = []
for m in modes:
    if m is None:
        continue
    try:
        parsed = result_view._parse_index_mode(m)
        out["parse_index_mode"].append({"mode": m, "parsed": list(parsed)})
    except Exception as e:  # pragma: no cover
        out["parse_index_mode"].append({"mode": m, "error": type(e).__name__})

# --- fvecs / ivecs reader round trip (tiny synthetic file, bytes hex-encoded)
vec = (np.arange(3 * 5, dtype=np.float32).reshape(3, 5) * 0.5) - 1.0
raw = b"".join(struct.pack("<i", 5) + row.astype("<f4").tobytes() for row in vec)
iv = np.arange(2 * 4, dtype=np.int32).reshape(2, 4) * 3
raw_i = b"".join(struct.pack("<i", 4) + row.astype("<i4").tobytes() for row in iv)
with tempfile.TemporaryDirectory() as td:
    p = Path(td) / "t.fvecs"
    p.write_bytes(raw)
    got = np.asarray(sift_io.read_fvecs(p))
    pi = Path(td) / "t.ivecs"
    pi.write_bytes(raw_i)
    goti = np.asarray(sift_io.read_ivecs(pi))
out["fvecs"] = {"hex": raw.hex(), "shape": list(got.shape), "values": got.astype(float).ravel().tolist()}
out["ivecs"] = {"hex": raw_i.hex(), "shape": list(goti.shape), "values": goti.astype(int).ravel().tolist()}

# --- the search-result block codec (cluster.py:230-241 _split_search_binary, :270-283 _encode_search_binary — the Python twins of
# encode_search_result_binary, src/rpc.rs:1156-1177, and decode_search_result_binary, src/cluster.rs:404-435): blocks ENCODED by the
# reference, and what the reference DECODES from them (single blocks, and several blocks back to back in one frame)
crng = np.random.default_rng(11)
codec = []
codec_cases = [
    ([], [], None),
    ([7], [0.5], None),
    ([0, 2**32 + 5, 2**64 - 1], [1.0, -0.0, float(np.float32(3.4e38))], None),
    ([3, 1, 2], [float("inf"), float("-inf"), 1e-45], None),
    ([10, 11], [0.25, 0.75], [{"name": "a", "tag": 1}, {"name": "b", "nested": {"x": [1, 2, 3]}}]),
    ([5], [2.0], [{"text": "snow\u2603 / quote\" / backslash\\"}]),
]
for _ in range(6):
    m = int(crng.integers(1, 40))
    codec_cases.append(([int(x) for x in crng.integers(0, 2**63, m, dtype=np.uint64)], [float(np.float32(x)) for x in crng.standard_normal(m)], None))
for ids, dists, fields in codec_cases:
    blob = cluster._encode_search_binary(ids, dists, fields)
    d_ids, d_dists, d_fields, d_off = cluster._split_search_binary(blob, 0)
    codec.append({"ids": [str(i) for i in ids], "dists_f32_bits": [int(np.float32(d).view(np.uint32)) for d in dists], "fields": fields,
                  "hex": blob.hex(), "decoded_ids": [str(i) for i in d_ids],
                  "decoded_dists_f32_bits": [int(np.float32(d).view(np.uint32)) for d in d_dists], "decoded_fields": d_fields, "next_offset": d_off})
out["search_block_codec"] = codec
frame = b"".join(bytes.fromhex(c["hex"]) for c in codec[:6])
offs, off = [], 0
for _ in range(6):
    _i, _d, _f, off = cluster._split_search_binary(frame, off)
    offs.append(off)
out["search_block_frame"] = {"hex": frame.hex(), "next_offsets": offs}

dst = Path(__file__).resolve().parent / "python_reference_vectors.json"
dst.write_text(json.dumps(out, indent=1))
print("wrote", dst, {k: len(v) for k, v in out.items()})
