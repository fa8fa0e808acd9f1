import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import lynsedb_amd as L
from lynsedb_amd._lib import lib
rng = np.random.default_rng(3)
n, dim, nq, k = 100000, 128, 2, 10
data = rng.standard_normal((n, dim)).astype(np.float32)
queries = rng.standard_normal((nq, dim)).astype(np.float32)
queries[-1, 7] = np.nan
idx = L.FlatIndex(None, dim); idx.write(data); idx.finalize()
idx.profile_enable(True)
for name in ("cosine", "l2"):
    idx.profile_get(reset=True)
    rows, dists, counts = idx.search_batch_arrays(queries, k, name)
    p = idx.profile_get(reset=True)
    print(name, counts, rows[1][:5], dists[1][:5], {k_: (hex(v) if k_ == "last_plan" else v) for k_, v in p.items()})
    thr = np.zeros(nq, np.float32); cnt = np.zeros(nq, np.uint32); ovf = np.zeros(nq, np.uint32); gs = np.zeros(64, np.uint32)
    lib.lynse_hip_debug_workspace(idx._h, thr.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), ovf.ctypes.data_as(C.c_void_p), gs.ctypes.data_as(C.c_void_p), nq)
    print("   thr", thr, "count", cnt, "overflow", ovf)
