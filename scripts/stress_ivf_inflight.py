#!/usr/bin/env python3
"""Randomised sweep of the IVF searches in flight (GPU): lynse_hip_ivf_search_submit_f32_device / _wait with random index shapes,
metrics, nprobe, batch sizes and numbers of batches in flight against the blocking entry point (bit-equal rows, distances, counts)
and, for a few queries of every case, against the oracle (IVFIndex::search, ivf.rs:181-348).  Cases include lists left empty and
queries that probe only empty lists (the all-lists-empty fallback is re-answered inside wait), nprobe >= nlist (answered inside
submit) and slabs large enough for the certified int8 pass.  Usage: python scripts/stress_ivf_inflight.py [seconds | c<N>] [seed].
STRESS_COMM=1 sends every batch through a 1-rank RCCL communicator (status word in the result block, exchange stream, merge)."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lynsedb_amd as L  # noqa: E402
import oracle as O  # noqa: E402

orc = O.get()
_a1 = sys.argv[1] if len(sys.argv) > 1 else "60"
max_cases = int(_a1[1:]) if _a1.startswith("c") else None
budget = float("inf") if max_cases is not None else float(_a1)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NAME = {O.IP: "ip", O.L2: "l2", O.COS: "cosine"}
dev = torch.device("cuda", 0)
COMM = None
if os.environ.get("STRESS_COMM") == "1":
    from lynsedb_amd.sharded import NativeComm  # noqa: E402
    COMM = NativeComm(None, 0, 1, 0)
t0, cases, bad = time.time(), 0, []
totals = {"in_flight": 0, "inside_submit": 0, "redone_in_wait": 0}


def outs(nq, k):
    return (torch.zeros((nq, k), dtype=torch.int64, device=dev), torch.zeros((nq, k), dtype=torch.float32, device=dev),
            torch.zeros(nq, dtype=torch.int32, device=dev))


def host(o):
    return o[0].cpu().numpy().view(np.uint64), o[1].cpu().numpy(), o[2].cpu().numpy().view(np.uint32)


while time.time() - t0 < budget and (max_cases is None or cases < max_cases):
    n = int(rng.choice([700, 9000, 40000, 90000, 150000]))
    if os.environ.get("STRESS_N"):
        n = int(os.environ["STRESS_N"])
    metric = int(rng.choice([O.IP, O.IP, O.L2, O.COS]))
    dim = int(rng.choice([8, 48, 64, 128, 256]))
    if n * dim > 24_000_000:
        continue
    nlist = int(rng.choice([1, 7, 32, 100, 300]))
    nlist = min(nlist, n)
    nprobe = int(rng.choice([1, 2, 5, 16, 40]))
    nq = int(rng.choice([1, 4, 33, 70, 200, 256]))
    k = int(rng.choice([1, 10, 50]))
    depth = int(rng.choice([1, 2, 3]))
    nb = int(rng.integers(depth, depth + 3))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        data = rng.random((n, dim), dtype=np.float32)
    elif kind == 1:
        data = rng.standard_normal((n, dim)).astype(np.float32)
    else:
        centers = rng.standard_normal((max(nlist // 2, 3), dim)).astype(np.float32)
        data = (centers[rng.integers(0, centers.shape[0], n)] + 0.3 * rng.standard_normal((n, dim))).astype(np.float32)
    built = L.IvfFlatIndex.build(None, data, dim, nlist, 2, NAME[metric], l2_partitions=False)
    cen, asg, _, _ = built.export()
    del built
    empties = int(rng.choice([0, 0, 2]))
    if empties:   # centroids that own no row; some queries sit on them
        far = (150.0 + np.arange(empties * dim, dtype=np.float32)).reshape(empties, dim)
        cen = np.concatenate([cen, far])
    idx = L.IvfFlatIndex.load(data, cen, asg, NAME[metric])
    off, rows = orc.lists_from_assignments(asg, cen.shape[0])
    batches = []
    for _ in range(nb):
        b = (data[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
        if empties and rng.random() < 0.5:
            b[int(rng.integers(0, nq))] = cen[-1]
        batches.append(b)
    dq = [torch.as_tensor(b, device=dev) for b in batches]
    blocking = []
    case = (n, dim, nlist, nprobe, nq, k, depth, nb, NAME[metric], kind, empties)
    try:
        for q in dq:
            o = outs(nq, k)
            idx.search_device(q, k, nprobe, *o)
            blocking.append(host(o))
    except Exception as e:  # noqa: BLE001
        print("BLOCKING SEARCH FAILED", case, repr(e), flush=True)
        bad.append(case)
        cases += 1
        continue
    res = [outs(nq, k) for _ in batches]
    pending = []
    try:
        for i in range(nb):
            pending.append(idx.search_submit(dq[i], k, nprobe, *res[i], comm=COMM.handle if COMM is not None else None))
            if len(pending) >= depth:
                pending.pop(0).wait()
        for t in pending:
            t.wait()
    except Exception as e:  # noqa: BLE001
        print("TICKET FAILED", case, repr(e), flush=True)
        bad.append(case)
        cases += 1
        for t in pending:
            try:
                t.wait()
            except Exception:  # noqa: BLE001
                pass
        continue
    st = idx.ticket_stats()
    for key in totals:
        totals[key] += st[key]
    ok = True
    why = []
    for i in range(nb):
        g = host(res[i])
        if not (np.array_equal(g[0], blocking[i][0]) and np.array_equal(g[1].view(np.uint32), blocking[i][1].view(np.uint32)) and np.array_equal(g[2], blocking[i][2])):
            ok = False
            dq_ = [int(x) for x in np.nonzero((g[2] != blocking[i][2]) | np.any(g[0] != blocking[i][0], axis=1))[0][:4]]
            why.append(("ticket != blocking", i, dq_, [int(g[2][x]) for x in dq_], [int(blocking[i][2][x]) for x in dq_]))
        for qi in {0, nq - 1, int(rng.integers(0, nq))}:
            e_ids, e_d, _ = orc.ivf_search(batches[i][qi], data, cen, off, rows, nprobe, k, metric)
            for tag, gg in (("ticket", g), ("blocking", blocking[i])):
                c = int(gg[2][qi])
                if c != len(e_ids) or not np.array_equal(gg[0][qi, :c], e_ids.astype(np.uint64)) or not np.array_equal(gg[1][qi, :c].view(np.uint32), e_d.view(np.uint32)):
                    ok = False
                    why.append((tag + " != oracle", i, qi, c, len(e_ids)))
    cases += 1
    if not ok:
        bad.append((n, dim, nlist, nprobe, nq, k, depth, nb, NAME[metric], kind, empties))
        print("MISMATCH", bad[-1], why[:4], st, flush=True)
    del idx
print("cases %d mismatches %d tickets %s comm %s seconds %.0f" % (cases, len(bad), totals, COMM is not None, time.time() - t0))
sys.exit(1 if bad else 0)
