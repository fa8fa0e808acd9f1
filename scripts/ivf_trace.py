"""Tiny IVF workload for a rocprofv3 kernel trace: 2M x 768, nlist 1024, nprobe 32, single query and 256 queries."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L
rng = np.random.default_rng(7)
n, dim, K = 1_000_000, 768, 1024
centers = rng.standard_normal((K, dim)).astype(np.float32)
data = (centers[np.arange(n) % K] + 0.03 * rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
ivf = L.IvfFlatIndex.build(None, data, dim, K, 3, "ip", l2_partitions=False)
qs = np.ascontiguousarray(data[rng.integers(0, n, 256)])
for nq in (1, 256):
    for _ in range(3):
        ivf.search_batch_arrays(qs[:nq], 10, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ivf.search_batch_arrays(qs[:nq], 10, 32)
    print(nq, "ms", (time.perf_counter() - t0) * 100)
