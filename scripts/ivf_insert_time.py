"""IVFIndex::insert on the device (lynse_hip_ivf_insert_f32): time to add 10,000 rows to a 2M x 128 index (the indexed rows are gathered
inside HBM into the re-assembled slab store; round 3 read them back to the host and uploaded them again)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lynsedb_amd as L
rng = np.random.default_rng(1)
n, dim, nlist = 2_000_000, 128, 1024
data = rng.random((n, dim), dtype=np.float32)
idx = L.IvfFlatIndex.build(None, data, dim, nlist, 2, "l2", l2_partitions=False)
new = rng.random((10_000, dim), dtype=np.float32)
q = data[:64] + 0.01
idx.search_batch_arrays(q, 10, 8)
for rep in range(3):
    t0 = time.perf_counter()
    idx.insert(new)
    t1 = time.perf_counter()
    idx.search_batch_arrays(q, 10, 8)
    t2 = time.perf_counter()
    print("insert of 10,000 rows into %d x %d (%d lists): %.1f ms, first search after it %.1f ms" % (len(idx) - 10_000, dim, nlist, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
