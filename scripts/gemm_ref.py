"""How fast does the vendor GEMM run the same shape?  rows16[N,768] @ q16[768,256] (f16 in, f16 out).
Only a yardstick for k_scan_h16's MFMA phase: the product path never calls it."""
import time, torch
dev = torch.device("cuda", 0)
for N in (4_000_000, 10_000_000):
    a = torch.rand((N, 768), device=dev, dtype=torch.float16)
    b = torch.rand((768, 256), device=dev, dtype=torch.float16)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c = a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(N, "ms", round(dt * 1e3, 3), "PFLOP/s", round(2 * N * 768 * 256 / dt / 1e15, 3), "in GB/s", round(N * 768 * 2 / dt / 1e9, 1))
    del a, c
