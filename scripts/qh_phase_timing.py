"""s_memtime phase sums of k_scan_qh (scan_qh.h, debug_flags & 64) on the C3 shape: cycles per 128-row tile step and wave.
QH_NOEMIT=1: no emission (debug_flags & 2); QH_K: k (default 100)."""
import ctypes as C, os, sys, numpy as np, torch
os.environ["LYNSE_HIP_DEBUG_FLAGS"] = str(64 | (2 if os.environ.get("QH_NOEMIT") else 0))
sys.path.insert(0, '.')
import lynsedb_amd as L
from lynsedb_amd.datasets import sift_like
dev = torch.device('cuda', 0)
N, D, K = 1_000_000, 128, int(os.environ.get("QH_K", "100"))
data = sift_like(N, D, 42); qs = sift_like(256, D, 43)
idx = L.FlatIndex(None, D, 0); idx.write(data); idx.finalize()
q = torch.as_tensor(qs, device=dev)
rows = torch.zeros((256, K), dtype=torch.int64, device=dev); d = torch.zeros((256, K), device=dev); c = torch.zeros(256, dtype=torch.int32, device=dev)
for _ in range(3): idx.search_device(q, K, "l2", rows, d, c)
torch.cuda.synchronize()
out = np.zeros(256 * 8 * 4, np.uint64)
rc = L._lib.lib.lynse_hip_debug_phase_cycles(out.ctypes.data_as(C.c_void_p), out.size)
a = out.reshape(256, 8, 4).astype(np.float64)
idx.profile_enable(True); idx.search_device(q, K, "l2", rows, d, c); torch.cuda.synchronize(); p = idx.profile_get()
stages = (int(p["last_plan"]) >> 8) & 0xff
print("rc", rc, "plan %#x" % int(p["last_plan"]), "scan_us", p.get("scan_us"))
names = ["wait+barrier", "mfma", "epilogue", "issue"]
tot = a.sum(2)
print("LAST stage: ticks per workgroup (sum of phases), mean over waves:", round(tot.mean(), 0))
for w in range(8):
    print("wave", w, {n: round(a[:, w, i].mean(), 0) for i, n in enumerate(names)}, "sum", round(tot[:, w].mean(), 0))
