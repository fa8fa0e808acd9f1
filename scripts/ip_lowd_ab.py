"""FLAT-IP at low dimension: the certified int8 pass (default) against the float pass on the f16 shadow (LYNSE_HIP_COARSE=f16: k_scan_qh at 64 / 128 columns)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lynsedb_amd as L
dev = torch.device('cuda', 0)
for dim in (128, 64, 100):
    rng = np.random.default_rng(dim)
    data = rng.standard_normal((1_000_000, dim)).astype(np.float32)
    qs = (data[rng.integers(0, 1_000_000, 256)] + 0.05 * rng.standard_normal((256, dim))).astype(np.float32)
    idx = L.FlatIndex(None, dim, 0); idx.write(data); idx.finalize()
    for nq, k in ((256, 10), (256, 100), (100, 10)):
        dq = torch.as_tensor(qs[:nq], device=dev)
        rows = torch.zeros((nq, k), dtype=torch.int64, device=dev); d = torch.zeros((nq, k), device=dev); c = torch.zeros(nq, dtype=torch.int32, device=dev)
        for _ in range(4): idx.search_device(dq, k, "ip", rows, d, c)
        torch.cuda.synchronize(); ts = []
        for _ in range(15):
            t0 = time.perf_counter(); idx.search_device(dq, k, "ip", rows, d, c); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        idx.profile_enable(True); idx.search_device(dq, k, "ip", rows, d, c); torch.cuda.synchronize(); p = idx.profile_get(reset=True); idx.profile_enable(False)
        print("coarse", os.environ.get("LYNSE_HIP_COARSE", "default"), "dim", dim, "nq", nq, "k", k, "median ms %.4f" % (ts[7] * 1e3), "plan %#x" % int(p["last_plan"]), "rescored/q", round(p["pool_entries"] / nq, 1))
