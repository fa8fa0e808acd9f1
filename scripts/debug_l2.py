import sys, numpy as np, torch
sys.path.insert(0, '.')
import lynsedb_amd as L
dev = torch.device('cuda', 0)
N, D = int(sys.argv[1]), 768
metric = sys.argv[2] if len(sys.argv) > 2 else 'l2'
idx = L.FlatIndex(None, D, 0); idx.reserve(N)
src = []
for b in range(0, N, 500_000):
    g = torch.Generator(device=dev); g.manual_seed(b)
    t = torch.rand((min(500_000, N - b), D), generator=g, device=dev)
    src.append(t); idx.write_device(t)
idx.finalize()
qrows = [5, N // 2 + 7, N - 3]
q = torch.stack([src[r // 500_000][r % 500_000] for r in qrows]) + 0.01
rows, dists, counts = idx.search_batch_arrays(q.cpu().numpy(), 10, metric)
bs = None
for i, s in enumerate(src):
    if metric == 'l2':
        sc = (q * q).sum(1, keepdim=True) + (s * s).sum(1)[None, :] - 2 * (q @ s.T)
    else:
        sc = -(q @ s.T)
    ts, ti = torch.topk(sc, 10, dim=1, largest=False); ti = ti + i * 500_000
    if bs is None: bs, bi = ts, ti
    else:
        cs, ci = torch.cat([bs, ts], 1), torch.cat([bi, ti], 1)
        bs, tj = torch.topk(cs, 10, dim=1, largest=False); bi = torch.gather(ci, 1, tj)
print('ours ', rows[:, :5].tolist(), dists[:, :5].tolist())
print('torch', bi[:, :5].tolist(), bs[:, :5].tolist())
