"""s_memtime stamps of the fused sample stage: where the in-launch threshold hand-over spends its time."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L  # noqa: E402

n, dim, nq, k = int(os.environ.get("N", 1_250_000)), 768, 256, 10
dev = torch.device("cuda", 0)
idx = L.FlatIndex(None, dim)
idx.reserve(n)
g = torch.Generator(device=dev); g.manual_seed(5)
for b in range(0, n, 250_000):
    e = min(n, b + 250_000)
    idx.write_device(torch.rand((e - b, dim), generator=g, device=dev))
idx.finalize()
queries = torch.rand((nq, dim), generator=g, device=dev).cpu().numpy()
lib = L._lib.lib
lib.lynse_hip_debug_fs_stamps.restype = C.c_int
lib.lynse_hip_debug_fs_stamps.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(3):
    idx.search_batch_arrays(queries, k, "ip")
os.environ["LYNSE_HIP_FS_STAMPS"] = "1"
for rep in range(3):
    idx.search_batch_arrays(queries, k, "ip")
    st = np.zeros((512, 8), np.uint64)
    assert lib.lynse_hip_debug_fs_stamps(idx._h, st.ctypes.data) == 0
    st = st[:256].astype(np.int64)
    t0 = st[:, 0].min()
    names = ["kernel start", "first tile done", "after sync 1", "after select", "after sync 2", "thresholds loaded", "kernel end"]
    print("rep", rep, "(s_memtime ticks relative to the earliest workgroup start; min / median / max over 256 workgroups)")
    for i, nm in enumerate(names):
        c = st[:, i] - t0
        print("  %-18s %9d %9d %9d" % (nm, c.min(), np.median(c), c.max()))
