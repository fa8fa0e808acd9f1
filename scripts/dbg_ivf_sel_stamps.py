"""s_memtime stamps of the three k_select launches of an IVF batch (C4 share: 6.25M x 768 clustered rows, nlist 4096, nprobe 32, 256 queries)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L  # noqa: E402

n, dim, nlist, nprobe, k = int(os.environ.get("N", 6_250_000)), 768, 4096, 32, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
centers = torch.randn((4096, dim), generator=g, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
rows_d = torch.empty((n, dim), device=dev, dtype=torch.float32)
for b0 in range(0, n, 250_000):
    e = min(n, b0 + 250_000)
    ids = torch.arange(b0, e, device=dev) % 4096
    rows_d[b0:e] = centers[ids] + 0.03 * torch.randn((e - b0, dim), generator=g, device=dev)
ivf = L.IvfFlatIndex.build_device(rows_d, dim, nlist, 2, "ip", l2_partitions=False)
qsel = torch.randint(0, n, (256,), generator=g, device=dev)
dq = (rows_d[qsel] + 0.01 * torch.randn((256, dim), generator=g, device=dev)).contiguous()
rows = torch.zeros((256, k), dtype=torch.int64, device=dev)
d = torch.zeros((256, k), dtype=torch.float32, device=dev)
c = torch.zeros(256, dtype=torch.int32, device=dev)
for _ in range(3):
    ivf.search_device(dq, k, nprobe, rows, d, c)
torch.cuda.synchronize()
lib = L._lib.lib
lib.lynse_hip_debug_ivf_sel_stamps.restype = C.c_int
lib.lynse_hip_debug_ivf_sel_stamps.argtypes = [C.c_void_p, C.c_void_p]
os.environ["LYNSE_HIP_SEL_STAMPS"] = "1"
names = ["start", "keys in LDS", "radix select done", "tighten done", "count pass done", "write-back done"]
for rep in range(2):
    ivf.search_device(dq, k, nprobe, rows, d, c)
    torch.cuda.synchronize()
    st = np.zeros((4, 256, 8), np.uint64)
    assert lib.lynse_hip_debug_ivf_sel_stamps(ivf._h, st.ctypes.data) == 0
    st = st.astype(np.int64)
    for win in range(4):
        s = st[win]
        if s[:, 0].max() == 0:
            continue
        print("rep", rep, "select behind window", win, "(cycles since the workgroup's start; median / max over queries)")
        for i in range(1, 6):
            if s[:, i].max() == 0:
                continue
            dd = (s[:, i] - s[:, 0])[s[:, i] > 0]
            print("   %-20s %8d %8d   (%d queries)" % (names[i], np.median(dd), dd.max(), len(dd)))
