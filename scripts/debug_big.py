import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
import lynsedb_amd as L
from lynsedb_amd._lib import lib, check
dev = torch.device('cuda', 0)
N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, 768
idx = L.FlatIndex(None, D, 0)
idx.reserve(N)
src = []
for b in range(0, N, 500_000):
    g = torch.Generator(device=dev); g.manual_seed(b)
    t = torch.rand((min(500_000, N - b), D), generator=g, device=dev)
    src.append(t); idx.write_device(t)
idx.finalize()
full = torch.cat(src) if N <= 4_000_000 else None
# 1. copy_rows_device vs source
buf = torch.empty((500_000, D), device=dev)
bad = 0
for i, b in enumerate(range(0, N, 500_000)):
    nr = src[i].shape[0]
    check(lib.lynse_hip_flat_copy_rows_device(idx.handle, b, nr, C.c_void_p(buf.data_ptr())))
    torch.cuda.synchronize()
    eq = torch.equal(buf[:nr], src[i])
    if not eq:
        bad += 1
        print('copy_rows_device mismatch at block', b, (buf[:nr] != src[i]).any(1).nonzero()[:5].flatten().tolist())
print('copy_rows_device bad blocks', bad)
# 2. host read
for b in (0, N // 2, N - 1000):
    h = idx.read_rows(b, 1000)
    blk = src[b // 500_000][(b % 500_000):(b % 500_000) + 1000].cpu().numpy()
    print('read_rows', b, np.array_equal(h, blk))
# 3. search vs torch on source
q = torch.cat([s[:3] for s in src[-2:]] + [src[0][:2]])
rows, dists, counts = idx.search_batch_arrays(q.cpu().numpy(), 10, 'ip')
best = None
for i, s in enumerate(src):
    sc = q @ s.T
    ts, ti = torch.topk(sc, 10, dim=1)
    ti = ti + i * 500_000
    if best is None: bs, bi = ts, ti
    else:
        cs, ci = torch.cat([bs, ts], 1), torch.cat([bi, ti], 1)
        bs, tj = torch.topk(cs, 10, dim=1); bi = torch.gather(ci, 1, tj)
    best = True
print('ours ', rows[:, :4].tolist())
print('torch', bi[:, :4].tolist())
print('dist diff', np.abs(dists - bs.cpu().numpy()).max())
