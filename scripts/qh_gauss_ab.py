"""k_scan_qh against k_scan_h16 (LYNSE_HIP_QH=0 / 1, read per call) on NON-integer rows: 1M x 128 and 1M x 64 Gaussian, L2 / cosine, k = 10 / 100, 256 and 100 queries."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lynsedb_amd as L
dev = torch.device('cuda', 0)
for dim in (128, 64):
    rng = np.random.default_rng(dim)
    data = rng.standard_normal((1_000_000, dim)).astype(np.float32)
    qs = (data[rng.integers(0, 1_000_000, 256)] + 0.05 * rng.standard_normal((256, dim))).astype(np.float32)
    idx = L.FlatIndex(None, dim, 0); idx.write(data); idx.finalize()
    for metric in ("l2", "cosine"):
        for nq, k in ((256, 10), (256, 100), (100, 10)):
            dq = torch.as_tensor(qs[:nq], device=dev)
            rows = torch.zeros((nq, k), dtype=torch.int64, device=dev); d = torch.zeros((nq, k), device=dev); c = torch.zeros(nq, dtype=torch.int32, device=dev)
            out = {}
            for qh in ("0", "1"):
                os.environ["LYNSE_HIP_QH"] = qh
                for _ in range(4): idx.search_device(dq, k, metric, rows, d, c)
                torch.cuda.synchronize(); ts = []
                for _ in range(15):
                    t0 = time.perf_counter(); idx.search_device(dq, k, metric, rows, d, c); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                ts.sort(); out[qh] = ts[7] * 1e3
            print("dim", dim, metric, "nq", nq, "k", k, "k_scan_h16 %.4f ms  k_scan_qh %.4f ms" % (out["0"], out["1"]))
