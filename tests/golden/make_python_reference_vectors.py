"""Generate golden vectors by IMPORTING the reference's pure-Python modules.

Run in the build container only (needs /root/reference; the reference cannot travel):
    python tests/golden/make_python_reference_vectors.py
Writes tests/golden/python_reference_vectors.json (inputs + expected outputs — data only).

Covered reference functions (python/lynse/cluster.py): _hash_u64 (:156-158) and the bucket rule
(:1273, :1364-1370), _merge_pairs (:535-556), _is_ascending_index (:182), and
result_view._parse_index_mode; benchmarks/sift_io.py read_fvecs/read_ivecs (:10-53).
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

dst = Path(__file__).resolve().parent / "python_reference_vectors.json"
dst.write_text(json.dumps(out, indent=1))
print("wrote", dst, {k: len(v) for k, v in out.items()})
