"""Debug of the fused sample stage (k_scan_h16<.., FS>): thresholds the grid agrees on vs the separate sample launch + k_select."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L  # noqa: E402

n, dim, nq, k = int(os.environ.get("N", 1_250_000)), 768, 256, 10
rng = np.random.default_rng(3)
idx = L.FlatIndex(None, dim)
idx.reserve(n)
data = np.empty((n, dim), np.float32)
for b in range(0, n, 200_000):
    e = min(n, b + 200_000)
    rng.random(out=data[b:e], dtype=np.float32)
    idx.write(data[b:e])
q_rows = np.sort(rng.integers(0, n, nq))
queries = (data[q_rows] + 0.03 * rng.standard_normal((nq, dim)).astype(np.float32)).astype(np.float32)
idx.finalize()
lib = L._lib.lib
lib.lynse_hip_debug_workspace.restype = C.c_int
lib.lynse_hip_debug_workspace.argtypes = [C.c_void_p] * 5 + [C.c_uint32]


def dump(tag):
    thr, cnt, ovf, gs = np.zeros(nq, np.float32), np.zeros(nq, np.uint32), np.zeros(nq, np.uint32), np.zeros(4, np.uint32)
    rc = lib.lynse_hip_debug_workspace(idx._h, thr.ctypes.data, cnt.ctypes.data, ovf.ctypes.data, gs.ctypes.data, nq)
    print(tag, "rc", rc, "gsync", gs, "thr[:6]", thr[:6], "thr min/max", thr.min(), thr.max(), "count sum/max", cnt.sum(), cnt.max(), "ovf", ovf.sum(), flush=True)
    return thr


def search(tag):
    idx.profile_enable(True)
    idx.profile_get(reset=True)
    r, d, c = idx.search_batch_arrays(queries, k, "ip")
    p = idx.profile_get(reset=True)
    print(tag, {kk: p[kk] for kk in ("scan_launches", "fallback_queries", "pool_entries", "last_plan")}, hex(p["last_plan"]), "counts", c.min(), c.max(), flush=True)
    return r, d, c


os.environ["LYNSE_HIP_FUSED_SAMPLE"] = "0"
os.environ["LYNSE_HIP_FUSED_TAIL"] = "0"
search("build")            # builds the SQ8 codes
r0, d0, c0 = search("unfused")
dump("after unfused search")
os.environ["LYNSE_HIP_FUSED_SAMPLE"] = "1"
os.environ["LYNSE_HIP_DEBUG_FS"] = "1"
search("fused (stopped behind the scan)")
t_f = dump("fused stage")
del os.environ["LYNSE_HIP_DEBUG_FS"]
r1, d1, c1 = search("fused")
dump("after fused search")
print("fused == unfused:", np.array_equal(r0, r1), np.array_equal(d0.view(np.uint32), d1.view(np.uint32)), np.array_equal(c0, c1))
bad = np.nonzero((r0 != r1).any(axis=1))[0]
print("queries that differ:", len(bad), bad[:20])
lost = np.array([r for qi in range(nq) for r in r0[qi] if r not in set(r1[qi].tolist())], dtype=np.int64)
print("lost rows:", len(lost), "min", lost.min() if len(lost) else None, "max", lost.max() if len(lost) else None)
print("lost by tile round (tile // 256):", np.bincount((lost // 256) // 256, minlength=n // 65536 + 1))
print("found by tile round:", np.bincount((np.intersect1d(r0.ravel(), r1.ravel()).astype(np.int64) // 256) // 256, minlength=n // 65536 + 1))
import oracle as O
orc = O.get()
for qi in bad[:1]:
    e_ids, e_d = orc.canonical_topk(queries[qi], data, k, O.IP)
    print("q", qi, "thr_fused_stage", t_f[qi])
    print("  oracle ", e_ids, e_d)
    print("  unfused", r0[qi], d0[qi])
    print("  fused  ", r1[qi], d1[qi])
os.environ["LYNSE_HIP_FUSED_TAIL"] = "1"
r2, d2, c2 = search("fused + fused tail")
print("fused+tail == unfused:", np.array_equal(r0, r2), np.array_equal(d0.view(np.uint32), d2.view(np.uint32)), np.array_equal(c0, c2))
print("top rows ok:", np.array_equal(r0[:, 0], q_rows.astype(np.uint64)))
