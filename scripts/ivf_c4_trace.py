"""C4 share (IVF-Flat IP 6.25M x 768, nlist 4096, nprobe 32, k 10): single query and batch 256 through the device API — run under
rocprofv3 --kernel-trace --stats to see which kernels the call spends its time in."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L  # noqa: E402

dev = torch.device("cuda", 0)
n, dim, nlist, nprobe, k = int(os.environ.get("N", 6_250_000)), 768, 4096, 32, 10
g = torch.Generator(device=dev); g.manual_seed(7)
centers = torch.randn((4096, dim), generator=g, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
rows_d = torch.empty((n, dim), device=dev)
for b0 in range(0, n, 250_000):
    e = min(n, b0 + 250_000)
    rows_d[b0:e] = centers[torch.arange(b0, e, device=dev) % 4096] + 0.03 * torch.randn((e - b0, dim), generator=g, device=dev)
ivf = L.IvfFlatIndex.build_device(rows_d, dim, nlist, 2, "ip", l2_partitions=False)
qsel = torch.randint(0, n, (256,), generator=g, device=dev)
queries = (rows_d[qsel] + 0.01 * torch.randn((256, dim), generator=g, device=dev)).contiguous()
for nq, reps in ((1, 50), (8, 30), (64, 20), (256, 20)):
    dq = queries[:nq].contiguous()
    rows = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    d = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    c = torch.zeros(nq, dtype=torch.int32, device=dev)
    for _ in range(3):
        ivf.search_device(dq, k, nprobe, rows, d, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ivf.search_device(dq, k, nprobe, rows, d, c)
    torch.cuda.synchronize()
    print("nq", nq, "ms", round((time.perf_counter() - t0) / reps * 1e3, 4), flush=True)
