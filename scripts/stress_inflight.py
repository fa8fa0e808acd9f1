#!/usr/bin/env python3
"""Randomised sweep of the searches in flight (GPU): lynse_hip_flat_search_submit_* / _wait with random shapes, metrics and
numbers of batches in flight against the blocking entry points (bit-equal rows, distances, counts) and, for a few queries
of every case, against the oracle.  Usage: python scripts/stress_inflight.py [seconds | c<N>] [seed].
STRESS_COMM=1 sends every batch through a 1-rank RCCL communicator (the exchange half of a sharded ticket: status word in the
result block, hand-over to the exchange stream, merge kernel)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402

orc = O.get()
_a1 = sys.argv[1] if len(sys.argv) > 1 else "60"
max_cases = int(_a1[1:]) if _a1.startswith("c") else None
budget = float("inf") if max_cases is not None else float(_a1)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NAME = {O.IP: "ip", O.L2: "l2", O.COS: "cosine", O.HAMMING: "hamming"}
dev = torch.device("cuda", 0)
COMM = None
import os  # noqa: E402
if os.environ.get("STRESS_COMM") == "1":
    from lynsedb_amd.sharded import NativeComm  # noqa: E402
    COMM = NativeComm(None, 0, 1, 0)
t0, cases, bad = time.time(), 0, []


def outs(nq, k):
    return (torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev),
            torch.zeros(nq, dtype=torch.int32, device=dev))


while time.time() - t0 < budget and (max_cases is None or cases < max_cases):
    n = int(rng.choice([300, 20000, 70001, 200000, 600000]))
    metric = int(rng.choice([O.IP, O.IP, O.L2, O.COS, O.HAMMING]))
    dim = int(rng.choice([8, 64, 100, 128, 384])) if metric != O.HAMMING else int(rng.choice([64, 130, 1024]))
    nq = int(rng.choice([5, 33, 70, 200, 256]))
    k = int(rng.choice([1, 10, 64, 100]))
    depth = int(rng.choice([1, 2, 3, 4]))
    nb = int(rng.integers(depth, depth + 3))
    if n * dim > 60_000_000:
        continue
    idx = L.FlatIndex(None, dim, 0)
    if metric == O.HAMMING:
        W = (dim + 63) // 64
        words = rng.integers(0, np.iinfo(np.int64).max, size=(n, W), dtype=np.int64).view(np.uint64)
        if dim % 64:
            words[:, -1] &= np.uint64((1 << (dim % 64)) - 1)
        idx.write_packed(words)
        batches = [words[rng.integers(0, n, nq)] ^ np.uint64(int(rng.integers(0, 255))) for _ in range(nb)]
        if dim % 64:
            for b in batches:
                b[:, -1] &= np.uint64((1 << (dim % 64)) - 1)
        dq = [torch.as_tensor(np.ascontiguousarray(b).view(np.int64), device=dev) for b in batches]
        blocking = lambda q, o: idx.search_packed_device(q, k, "hamming", *o)  # noqa: E731
    else:
        kind = int(rng.integers(0, 3))
        data = rng.random((n, dim), dtype=np.float32) if kind == 0 else rng.standard_normal((n, dim)).astype(np.float32)
        if kind == 2:
            data *= np.float32(10.0 ** rng.integers(-3, 3))
        idx.write(data)
        batches = [(data[rng.integers(0, n, nq)] + np.float32(0.05) * rng.standard_normal((nq, dim)).astype(np.float32)) for _ in range(nb)]
        dq = [torch.as_tensor(np.ascontiguousarray(b, dtype=np.float32), device=dev) for b in batches]
        blocking = lambda q, o: idx.search_device(q, k, NAME[metric], *o)  # noqa: E731
    idx.finalize()
    ref = []
    for q in dq:
        o = outs(nq, k)
        blocking(q, o)
        torch.cuda.synchronize()
        ref.append([t.cpu().numpy() for t in o])
    got = [outs(nq, k) for _ in dq]
    pending = []
    for i, q in enumerate(dq):
        pending.append(idx.search_submit(q, k, NAME[metric], *got[i], comm=COMM.handle if COMM is not None else None))
        if len(pending) >= depth:
            pending.pop(0).wait()
    for t in pending:
        t.wait()
    torch.cuda.synchronize()
    ok = True
    for i in range(nb):
        r, d, c = [t.cpu().numpy() for t in got[i]]
        kk = min(k, n)
        if not (np.array_equal(c, ref[i][2]) and np.array_equal(r[:, :kk], ref[i][0][:, :kk]) and np.array_equal(d[:, :kk].view(np.uint32), ref[i][1][:, :kk].view(np.uint32))):
            ok = False
        for qi in (0, nq - 1):  # and the oracle
            if metric == O.HAMMING:
                e_ids, e_d = orc.canonical_topk_packed(batches[i][qi], words, k, metric)
            else:
                e_ids, e_d = orc.canonical_topk(np.ascontiguousarray(batches[i][qi], dtype=np.float32), data, k, metric)
            cc = int(c[qi])
            if cc != len(e_ids) or not np.array_equal(r[qi, :cc].astype(np.uint32), e_ids) or not np.array_equal(d[qi, :cc].view(np.uint32), e_d.view(np.uint32)):
                ok = False
    cases += 1
    if not ok:
        bad.append((n, dim, nq, k, metric, depth, nb))
        print("MISMATCH", bad[-1], flush=True)
    del idx
print("cases", cases, "mismatches", len(bad), bad[:10])
if COMM is not None:
    COMM.close()
sys.exit(1 if bad else 0)
