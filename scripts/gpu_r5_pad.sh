#!/bin/bash
# round 5: padded f16 shadow for 48..63 / 96..127 columns — the whole GPU suite, a randomised sweep, and the speed of a 1M x 100 / 1M x 96 L2 batch with and without
mkdir -p gpurun_out/pad
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/pad/pytest.txt 2>&1; tail -15 gpurun_out/pad/pytest.txt | cut -c1-300
(timeout 260 python scripts/stress_parity.py 200 81 2>&1 | tail -3) > gpurun_out/pad/stress_parity.log; cat gpurun_out/pad/stress_parity.log
(timeout 200 python scripts/stress_ivf.py 120 82 2>&1 | tail -3) > gpurun_out/pad/stress_ivf.log; cat gpurun_out/pad/stress_ivf.log
cat > /tmp/pad_bench.py <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import lynsedb_amd as L
dev = torch.device('cuda', 0)
for dim in (100, 96, 56):
    rng = np.random.default_rng(dim)
    data = rng.standard_normal((1_000_000, dim)).astype(np.float32)
    qs = (data[rng.integers(0, 1_000_000, 256)] + 0.05 * rng.standard_normal((256, dim))).astype(np.float32)
    idx = L.FlatIndex(None, dim, 0); idx.write(data); idx.finalize()
    dq = torch.as_tensor(qs, device=dev)
    for metric in ("l2", "cosine"):
        for k in (10, 100):
            rows = torch.zeros((256, k), dtype=torch.int64, device=dev); d = torch.zeros((256, k), device=dev); c = torch.zeros(256, dtype=torch.int32, device=dev)
            for _ in range(4): idx.search_device(dq, k, metric, rows, d, c)
            torch.cuda.synchronize(); ts = []
            for _ in range(15):
                t0 = time.perf_counter(); idx.search_device(dq, k, metric, rows, d, c); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            ts.sort(); print("pad", os.environ.get("LYNSE_HIP_SHADOW_PAD", "1"), "dim", dim, metric, "k", k, "median ms %.4f" % (ts[7] * 1e3), "hbm MB", idx.hbm_bytes() >> 20 if hasattr(idx, "hbm_bytes") else "")
PY
for p in 0 1; do LYNSE_HIP_SHADOW_PAD=$p python /tmp/pad_bench.py 2>&1 | grep -v amdgpu.ids; done
