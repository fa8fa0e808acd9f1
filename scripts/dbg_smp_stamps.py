"""Phase stamps of the sample stage (k_scan_qs<.., SMP>; LYNSE_HIP_DEBUG_FLAGS=64): where its ~35 us go.
N rows x 768, 256 queries, k = 10.  Per workgroup: entry -> fragments + constants loaded -> ring primed -> first stage landed ->
tiles done -> keys written (s_memtime ticks), and the launch skew over the grid (s_memrealtime, 100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["LYNSE_HIP_DEBUG_FLAGS"] = "64"
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
for _ in range(5):
    idx.search_batch_arrays(queries, k, "ip")
names = ["entry", "ring primed (fragment loads issued in front of it)", "fragments + constants in registers", "first stage landed", "tiles done", "keys written"]
for rep in range(3):
    idx.search_batch_arrays(queries, k, "ip")
    out = np.zeros(16 * 8192, np.uint64)
    assert lib.lynse_hip_debug_phase_cycles(out.ctypes.data_as(C.c_void_p), out.size) == 0
    for region in range(4):
        st = out[region * 8192: region * 8192 + 256 * 8].reshape(256, 8).astype(np.int64)
        if st[:, 6].max() == 0 or st[:, 5].max() == 0:
            continue
        rt = st[:, 6] - st[:, 6].min()
        print("rep", rep, "region", region, "(sample stage): launch skew over the 256 workgroups: median %.2f us, max %.2f us" % (np.median(rt) / 100.0, rt.max() / 100.0))
        for i in range(1, 6):
            d = st[:, i] - st[:, 0]
            print("   %-52s median %8d   max %8d ticks" % (names[i], np.median(d), d.max()))
