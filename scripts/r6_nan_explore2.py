import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import lynsedb_amd as L
import oracle as O
orc = O.get()
rng = np.random.default_rng(3)
def model(q, data, k, metric):
    d = np.asarray(orc.all_distances(q, data, metric), np.float32)
    asc = metric != O.IP
    worst = np.float32(np.inf if asc else -np.inf)
    d = np.where(np.isnan(d), worst, d)
    key = d if asc else -d
    order = np.lexsort((np.arange(len(d)), key))
    return order[:k].astype(np.uint64), d[order[:k]]
for n, dim, nq, k in ((64, 96, 1, 64), (64, 96, 8, 64), (64, 96, 40, 64), (5000, 96, 3, 20), (100000, 128, 2, 10), (300000, 128, 64, 10), (300000, 256, 64, 10)):
    data = rng.standard_normal((n, dim)).astype(np.float32)
    sp = rng.choice(n, 12, replace=False)
    data[sp[0:4], 3] = np.nan
    data[sp[4:6], 5] = np.inf
    data[sp[6:8], 5] = -np.inf
    data[sp[8], 1] = np.inf; data[sp[8], 2] = -np.inf
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    queries[:, 1] = np.abs(queries[:, 1]); queries[:, 2] = np.abs(queries[:, 2])
    if nq > 1:
        queries[-1, 7] = np.nan
    if nq > 2:
        queries[-2, 9] = np.inf
    idx = L.FlatIndex(None, dim)
    idx.write(data)
    idx.finalize()
    for name, m in (("ip", O.IP), ("l2", O.L2), ("cosine", O.COS)):
        try:
            rows, dists, counts = idx.search_batch_arrays(queries, k, name)
        except Exception as e:
            print(n, dim, nq, k, name, "ERROR", repr(e)); continue
        bad = []
        for qi in range(nq):
            e_r, e_d = model(queries[qi], data, k, m)
            c = int(counts[qi])
            ok = c == len(e_r) and np.array_equal(rows[qi, :c].astype(np.uint64), e_r) and np.array_equal(dists[qi, :c].view(np.uint32), e_d.view(np.uint32))
            if not ok:
                bad.append(qi)
                if len(bad) <= 2:
                    diff = [i for i in range(min(c, len(e_r))) if rows[qi, i] != e_r[i] or dists[qi, i].view(np.uint32) != e_d[i].view(np.uint32)]
                    print("  q", qi, "count", c, "want", len(e_r), "first diffs at", diff[:6])
                    for i in diff[:6]:
                        print("     pos", i, "got", int(rows[qi, i]), dists[qi, i], "want", int(e_r[i]), e_d[i])
        print(n, dim, nq, k, name, "mismatching queries:", bad[:8], "special rows", sorted(sp.tolist())[:12])
