#!/usr/bin/env python3
"""Decomposition of the low-D L2 scan (config 3: SIFT-like 1M x 128, batch 256, k = 100) — EXPERIMENTS build only.
LYNSE_HIP_DEBUG_FLAGS = DBG << 8 selects compile-time variants of the DENSE <4,2,2,4> L2 kernel with pieces removed (results are
wrong by construction; only the time of the threshold-stage scan launch matters)."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
names = {0: "everything", 16: "no epilogue", 4: "everything but the query-image DMA", 12: "MFMA + LDS reads + epilogue (no DMA)",
         28: "MFMA + LDS reads only", 30: "MFMA only", 17: "DMA + LDS reads (no MFMA, no epilogue)", 19: "DMA only",
         20: "no epilogue, no query-image DMA", 24: "no epilogue, no row DMA"}
for dbg, name in names.items():
    env = dict(os.environ, LYNSE_HIP_DEBUG_FLAGS=str(dbg << 8))
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "exp_c3.py")], capture_output=True, text=True, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
    print(dbg, name, {k: d.get(k) for k in ("scan_us", "launches", "median_ms", "error") if k in d}, flush=True)
