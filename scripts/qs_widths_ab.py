#!/usr/bin/env python3
"""The query-stationary tiling on narrower code rows (256 / 384 / 512 / 640 columns) against the round-3 tilings (LYNSE_HIP_QS_WIDTHS=0):
median ms per batch over ROWS x dim, 256 / 100 / 40 queries, IP and cosine; results of the two must be identical."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
dev = torch.device("cuda", 0)
n = int(os.environ.get("ROWS", 6_000_000))
for dim in [int(x) for x in os.environ.get("DIMS", "256,384,512,640").split(",")]:
    idx = L.FlatIndex(None, dim, 0); idx.reserve(n)
    g = torch.Generator(device=dev); g.manual_seed(dim)
    for b in range(0, n, 500_000):
        idx.write_device(torch.rand((min(500_000, n - b), dim), generator=g, device=dev))
    idx.finalize()
    q_all = torch.rand((256, dim), generator=g, device=dev)
    for metric in os.environ.get("METRICS", "ip,cosine").split(","):
        for nq in (256, 100, 40):
            dq = q_all[:nq].contiguous()
            res = {}
            for w in ("1", "0"):
                os.environ["LYNSE_HIP_QS_WIDTHS"] = w
                rows = torch.zeros((nq, 10), dtype=torch.int64, device=dev); d = torch.zeros((nq, 10), dtype=torch.float32, device=dev); c = torch.zeros(nq, dtype=torch.int32, device=dev)
                fn = lambda: idx.search_device(dq, 10, metric, rows, d, c)  # noqa: E731
                for _ in range(3): fn()
                torch.cuda.synchronize(); ts = []
                for _ in range(10):
                    t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
                idx.profile_enable(True); idx.profile_get(reset=True); fn(); p = idx.profile_get(reset=True); idx.profile_enable(False)
                res[w] = (float(np.median(ts)) * 1e3, hex((int(p["last_plan"]) >> 16) & 0xff), p["fallback_queries"], rows.cpu().numpy().copy(), d.cpu().numpy().copy())
            same = np.array_equal(res["1"][3], res["0"][3]) and np.array_equal(res["1"][4].view(np.uint32), res["0"][4].view(np.uint32))
            print("dim", dim, metric, "nq", nq, "qs %.3f ms (%s)" % res["1"][:2], "old %.3f ms (%s)" % res["0"][:2], "identical", same, "fallback", res["1"][2], res["0"][2], flush=True)
    del idx
    torch.cuda.empty_cache()
