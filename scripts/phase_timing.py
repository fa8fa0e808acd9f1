import ctypes as C, os, sys, numpy as np, torch
# PHASE_DBG: compile-time experiment variant (DBG << 8); | 2 = no emission
os.environ["LYNSE_HIP_DEBUG_FLAGS"] = str(64 | (int(os.environ.get("PHASE_DBG", "0")) << 8) | (2 if os.environ.get("PHASE_NOEMIT") else 0))
sys.path.insert(0, '.')
import lynsedb_amd as L
dev = torch.device('cuda', 0)
N, D = 4_000_000, 768
idx = L.FlatIndex(None, D, 0); idx.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for b in range(0, N, 500_000):
    idx.write_device(torch.rand((500_000, D), generator=g, device=dev))
idx.finalize()
q = torch.rand((256, D), generator=g, device=dev)
rows = torch.zeros((256, 10), dtype=torch.int64, device=dev); d = torch.zeros((256, 10), device=dev); c = torch.zeros(256, dtype=torch.int32, device=dev)
for _ in range(3): idx.search_device(q, 10, "ip", rows, d, c)
out = np.zeros(256 * 8 * 4, np.uint64)
rc = L._lib.lib.lynse_hip_debug_phase_cycles(out.ctypes.data_as(C.c_void_p), out.size)
raw = out.reshape(256, 8, 4)
a = raw.astype(np.float64)
iq, iv = (raw[:, :, 2] >> np.uint64(32)).astype(np.float64), (raw[:, :, 2] & np.uint64(0xffffffff)).astype(np.float64)
tiles = (N - 2097152 + 255) // 256
i8c = os.environ.get("LYNSE_HIP_COARSE", "") != "f16"
iters = tiles / 256 * (6 if i8c else 12)  # slab steps per workgroup: 768 / 128 (int8) or 768 / 64 (f16) per 256-row tile
print("rc", rc, "coarse", "i8" if i8c else "f16", "iters/block ~", iters, "(cycles per slab step, s_memtime)")
names = ["wait_vmcnt", "barrier", "issue", "compute"]
for w in (0, 3, 7):
    print("wave", w, {n: round(a[:, w, i].mean() / iters, 1) for i, n in enumerate(names)})
print("DMA issue cycles per slab step and wave: query-image pieces", round(iq.mean() / iters, 1), " row pieces", round(iv.mean() / iters, 1),
      "(4 instructions each; includes two s_memtime reads per instruction)")
print("all  ", {n: round(a[:, :, i].mean() / iters, 1) for i, n in enumerate(names)}, "sum", round(a.sum(2).mean() / iters, 1))
