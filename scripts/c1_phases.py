"""Development: where the time of the fused single-query search (C1: 100k x 128, k = 10) goes — LYNSE_HIP_SMALL_DBG=1 stamps."""
import os
import sys
import time

os.environ["LYNSE_HIP_SMALL_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L

rng = np.random.default_rng(42)
n, dim, k = 100_000, 128, 10
data = rng.random((n, dim), dtype=np.float32)
q = rng.random(dim, dtype=np.float32)
dev = torch.device("cuda", 0)
idx = L.FlatIndex(None, dim, 0)
idx.write(data)
idx.finalize()
dq = torch.as_tensor(q.reshape(1, -1), device=dev)
rows = torch.zeros((1, k), dtype=torch.int64, device=dev)
d = torch.zeros((1, k), dtype=torch.float32, device=dev)
c = torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(8):
    idx.search_device(dq, k, "ip", rows, d, c)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    t0 = time.perf_counter()
    idx.search_device(dq, k, "ip", rows, d, c)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("median wall ms", ts[len(ts) // 2] * 1e3, file=sys.stderr)
