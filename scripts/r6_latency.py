#!/usr/bin/env python3
"""Where the wall time of the latency-shaped configurations goes (round 6): python scripts/r6_latency.py [c1] [c3] [c4]

For each configuration: median wall time of (a) the Python wrapper + a trailing torch.cuda.synchronize() (what bench.py timed up to
round 5), (b) the Python wrapper alone (the call is blocking: its results are final when it returns), (c) the bare C-ABI call through
ctypes with prebuilt arguments (what a Rust FFI caller pays), and the HIP-event time of the library's own kernels."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lynsedb_amd as L  # noqa: E402
from lynsedb_amd._lib import lib  # noqa: E402

dev = torch.device("cuda", 0)


def med(fn, warm, reps, sync):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        if sync:
            torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return round(ts[len(ts) // 2] * 1e6, 2), round(ts[len(ts) // 10] * 1e6, 2)


def three(name, wrapper, raw, warm=20, reps=200):
    out = {"config": name}
    out["wrapper_plus_torch_sync_us"] = med(wrapper, warm, reps, True)
    out["wrapper_us"] = med(wrapper, warm, reps, False)
    out["raw_cabi_us"] = med(raw, warm, reps, False)
    print(json.dumps(out), flush=True)


def c1():
    rng = np.random.default_rng(42)
    data = rng.random((100_000, 128), dtype=np.float32)
    q = rng.random(128, dtype=np.float32)
    idx = L.FlatIndex(None, 128, 0)
    idx.write(data)
    idx.finalize()
    dq = torch.as_tensor(q.reshape(1, -1), device=dev)
    rows = torch.zeros((1, 10), dtype=torch.int64, device=dev)
    d = torch.zeros((1, 10), dtype=torch.float32, device=dev)
    c = torch.zeros(1, dtype=torch.int32, device=dev)
    args = (idx._h, C.c_void_p(dq.data_ptr()), 1, 10, 0, C.c_void_p(rows.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(c.data_ptr()), None)
    f = lib.lynse_hip_flat_search_f32_device
    three("c1", lambda: idx.search_device(dq, 10, "ip", rows, d, c), lambda: f(*args))


def c3():
    from lynsedb_amd.datasets import sift_like

    data = sift_like(1_000_000, 128, 42)
    qs = sift_like(256, 128, 43)
    idx = L.FlatIndex(None, 128, 0)
    idx.write(data)
    idx.finalize()
    import os
    shapes = ((256, 100), (32, 100), (256, 10))
    if os.environ.get("LAT_C3_ONLY"):      # e.g. LAT_C3_ONLY=0: the first shape only (kernel timelines)
        shapes = (shapes[int(os.environ["LAT_C3_ONLY"])],)
    for nq, k in shapes:
        dq = torch.as_tensor(qs[:nq], device=dev)
        rows = torch.zeros((nq, k), dtype=torch.int64, device=dev)
        d = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        c = torch.zeros(nq, dtype=torch.int32, device=dev)
        args = (idx._h, C.c_void_p(dq.data_ptr()), nq, k, 1, C.c_void_p(rows.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(c.data_ptr()), None)
        f = lib.lynse_hip_flat_search_f32_device
        three("c3 nq=%d k=%d" % (nq, k), lambda: idx.search_device(dq, k, "l2", rows, d, c), lambda: f(*args), 5, 60)


def c4():
    n, dim, nlist, nprobe, k = 1_600_000, 768, 1024, 32, 10   # the list length of the C4 share (1,526 rows per list), fewer lists: builds in seconds
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    centers = torch.randn((nlist, dim), generator=g, device=dev)
    centers /= centers.norm(dim=1, keepdim=True) + 1e-12
    rows_d = torch.empty((n, dim), device=dev, dtype=torch.float32)
    for b0 in range(0, n, 200_000):
        e = min(n, b0 + 200_000)
        ids = torch.arange(b0, e, device=dev) % nlist
        blk = centers[ids] + 0.03 * torch.randn((e - b0, dim), generator=g, device=dev)
        rows_d[b0:e] = blk / (blk.norm(dim=1, keepdim=True) + 1e-12)
    ivf = L.IvfFlatIndex.build_device(rows_d, dim, nlist, 3, "ip", l2_partitions=False)
    torch.cuda.synchronize()
    q = rows_d[12345:12346] + 0.01 * torch.randn((1, dim), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    rows = torch.zeros((1, k), dtype=torch.int64, device=dev)
    d = torch.zeros((1, k), dtype=torch.float32, device=dev)
    c = torch.zeros(1, dtype=torch.int32, device=dev)
    args = (ivf._h, C.c_void_p(q.data_ptr()), 1, k, nprobe, C.c_void_p(rows.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(c.data_ptr()))
    f = lib.lynse_hip_ivf_search_f32_device
    three("c4-like nq=1 (1.6M x 768, nlist 1024, nprobe 32)", lambda: ivf.search_device(q, k, nprobe, rows, d, c), lambda: f(*args))


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "c3", "c4"]
    for w in which:
        {"c1": c1, "c3": c3, "c4": c4}[w]()
