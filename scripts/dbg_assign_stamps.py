"""s_memtime stamps of k_select_final in the shape of the k-means assignment: 4096 centroids x 768, one emit-all stage (4096 key
slots per query), k = 1, 256 queries (the stamp buffer holds 256 queries; the assignment launches 8192 per pass)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L  # noqa: E402

n, dim, nq, k = 4096, 768, 256, 1
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
cen = torch.nn.functional.normalize(torch.randn((n, dim), generator=g, device=dev), dim=1)
idx = L.FlatIndex(None, dim)
idx.write_device(cen)
idx.finalize()
idx.set_ip_form(1)
idx.set_plan(4096, 8, 4096)
pick = torch.randint(0, n, (nq,), generator=g, device=dev)
queries = (cen[pick] + 0.03 * torch.randn((nq, dim), generator=g, device=dev)).cpu().numpy()
lib = L._lib.lib
lib.lynse_hip_debug_sel_stamps.restype = C.c_int
lib.lynse_hip_debug_sel_stamps.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(3):
    r = idx.search_batch_arrays(queries, k, "ip")
print("top-1 is the picked centroid:", bool((np.asarray(r[0])[:, 0] == pick.cpu().numpy()).all()))
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
        print("rep", rep, "select behind stage", stage, "(ticks of 10 ns since the workgroup's start; median / max over queries)")
        for i in range(1, 7):
            d = s[:, i] - s[:, 0]
            if s[:, i].max() == 0:
                continue
            print("   %-20s %8d %8d" % (names[i], np.median(d), d.max()))
