"""C3 shape (SIFT-like 1M x 128, squared L2, k = 100) against the batch size: ms per call for 32 / 64 / 128 / 256 queries, and k = 10."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L  # noqa: E402
from lynsedb_amd.datasets import sift_like  # noqa: E402

dev = torch.device("cuda", 0)
data = sift_like(1_000_000, 128, 42)
qs = sift_like(256, 128, 43)
idx = L.FlatIndex(None, 128, 0)
idx.write(data)
idx.finalize()
for k in (100, 10):
    for nq in (32, 64, 128, 256):
        dq = torch.as_tensor(qs[:nq], device=dev)
        rows = torch.zeros((nq, k), dtype=torch.int64, device=dev)
        d = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        c = torch.zeros(nq, dtype=torch.int32, device=dev)
        for _ in range(5):
            idx.search_device(dq, k, "l2", rows, d, c)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            idx.search_device(dq, k, "l2", rows, d, c)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        idx.profile_enable(True)
        idx.search_device(dq, k, "l2", rows, d, c)
        torch.cuda.synchronize()
        p = idx.profile_get()
        idx.profile_enable(False)
        print("k %3d nq %3d  median %.4f ms  (%.0f q/s)  scan_us %.1f plan %#x" % (k, nq, ts[10] * 1e3, nq / ts[10], p.get("scan_us", 0.0), int(p.get("last_plan", 0))))
