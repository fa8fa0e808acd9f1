"""s_memtime stamps of k_select / k_select_final: where the ~23 us of a select go (1.25M x 768, 256 queries, k = 10)."""
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
lib.lynse_hip_debug_sel_stamps.restype = C.c_int
lib.lynse_hip_debug_sel_stamps.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(3):
    idx.search_batch_arrays(queries, k, "ip")
os.environ["LYNSE_HIP_SEL_STAMPS"] = "1"
names = ["start", "keys in LDS", "radix select done", "tighten done", "count pass done", "write-back done", "final done"]
for rep in range(2):
    idx.search_batch_arrays(queries, k, "ip")
    st = np.zeros((4, 256, 8), np.uint64)
    assert lib.lynse_hip_debug_sel_stamps(idx._h, st.ctypes.data) == 0
    st = st.astype(np.int64)
    for stage in range(4):
        s = st[stage]
        if s[:, 0].max() == 0:
            continue
        print("rep", rep, "select behind stage", stage, "(ticks since the workgroup's start; median / max over queries)")
        for i in range(1, 7):
            d = s[:, i] - s[:, 0]
            if s[:, i].max() == 0:
                continue
            print("   %-20s %8d %8d" % (names[i], np.median(d), d.max()))
