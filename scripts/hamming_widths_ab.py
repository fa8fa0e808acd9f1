#!/usr/bin/env python3
"""Batched Hamming (FP4 +-1 GEMM) on the query-stationary tiling against the 256 x 256 tile (LYNSE_HIP_QS_F4=0): 512 / 1024 / 2048-bit rows,
256 and 100 packed queries, k = 50; median ms per batch, identical results required, two queries against the oracle on a 200k-row prefix index."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402
orc = O.get()
dev = torch.device("cuda", 0)
n, k = int(os.environ.get("ROWS", 8_000_000)), 50
for bits in (512, 1024, 2048):
    W = bits // 64
    idx = L.FlatIndex(None, bits, 0); idx.reserve(n)
    g = torch.Generator(device=dev); g.manual_seed(bits)
    first = None
    for b in range(0, n, 2_000_000):
        w = torch.randint(-2**63, 2**63 - 1, (min(2_000_000, n - b), W), generator=g, device=dev, dtype=torch.int64)
        if first is None: first = w[:200_000].clone()
        idx.write_packed_device(w); del w
    idx.finalize(); idx.prepare("hamming", 256)
    q = first[torch.arange(256, device=dev) * 701 % first.shape[0]].clone(); q[:, 0] ^= 0xFFFF
    small = L.FlatIndex(None, bits, 0); small.write_packed_device(first); small.finalize()
    for nq in (256, 100):
        dq = q[:nq].contiguous(); res = {}
        for f in ("1", "0"):
            os.environ["LYNSE_HIP_QS_F4"] = f
            rows = torch.zeros((nq, k), dtype=torch.int64, device=dev); d = torch.zeros((nq, k), dtype=torch.float32, device=dev); c = torch.zeros(nq, dtype=torch.int32, device=dev)
            fn = lambda: idx.search_packed_device(dq, k, "hamming", rows, d, c)  # noqa: E731
            for _ in range(3): fn()
            torch.cuda.synchronize(); ts = []
            for _ in range(10):
                t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            sr = torch.zeros((nq, k), dtype=torch.int64, device=dev); sd = torch.zeros((nq, k), dtype=torch.float32, device=dev); sc = torch.zeros(nq, dtype=torch.int32, device=dev)
            small.search_packed_device(dq, k, "hamming", sr, sd, sc); torch.cuda.synchronize()
            ok = True
            hr, hq = first.cpu().numpy().view(np.uint64), dq.cpu().numpy().view(np.uint64)
            for qi in (0, nq - 1):
                e_ids, e_d = orc.canonical_topk_packed(hq[qi], hr, k, O.HAMMING)
                ok = ok and np.array_equal(sr[qi].cpu().numpy().astype(np.uint32), e_ids) and np.array_equal(sd[qi].cpu().numpy(), e_d)
            res[f] = (float(np.median(ts)) * 1e3, rows.cpu().numpy().copy(), d.cpu().numpy().copy(), ok)
        same = np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][2], res["0"][2])
        print("bits", bits, "nq", nq, "qs %.3f ms" % res["1"][0], "old %.3f ms" % res["0"][0], "identical", same, "oracle (200k-row index)", res["1"][3], res["0"][3], flush=True)
    del idx, small
    torch.cuda.empty_cache()
