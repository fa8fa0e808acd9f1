"""Dataset helpers for the benchmark configs (host-side, no arithmetic of the search path).

read_fvecs / read_ivecs follow benchmarks/sift_io.py:10-53 (little-endian int32 dim prefix per row).
SIFT1M is not shipped with the image; `sift_like` generates the documented synthetic stand-in
(integers 0..218 stored as f32, D=128 — SURVEY.md §8d C3).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np


def _read_vecs(path, max_rows, kind):
    path = Path(path)
    raw = np.fromfile(path, dtype="<i4")
    if raw.size == 0:
        raise ValueError(f"empty {kind} file: {path}")
    dim = int(raw[0])
    if dim <= 0:
        raise ValueError(f"invalid {kind} dim={dim} in {path}")
    stride = dim + 1
    if raw.size % stride != 0:
        raise ValueError(f"{kind} size {raw.size} not divisible by dim+1={stride} ({path})")
    n = raw.size // stride
    if max_rows is not None:
        n = min(n, max_rows)
    mat = raw[: n * stride].reshape(n, stride)
    if not np.all(mat[:, 0] == dim):
        raise ValueError(f"inconsistent dims inside {path}")
    return mat[:, 1:]


def read_fvecs(path, *, max_rows=None) -> np.ndarray:
    return _read_vecs(path, max_rows, "fvecs").view(np.float32).copy()


def read_ivecs(path, *, max_rows=None) -> np.ndarray:
    return _read_vecs(path, max_rows, "ivecs").astype(np.int32).copy()


def sift_like(n: int, dim: int = 128, seed: int = 42) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.integers(0, 219, size=(n, dim)).astype(np.float32)


def packed_bernoulli(n: int, bits: int, p: float = 0.5, seed: int = 42) -> np.ndarray:
    """Pre-packed fingerprints: (n, ceil(bits/64)) u64, bit i of word i/64 LSB-first."""
    rng = np.random.default_rng(seed)
    words = (bits + 63) // 64
    if p == 0.5 and bits % 64 == 0:
        return rng.integers(0, np.iinfo(np.uint64).max, size=(n, words), dtype=np.uint64, endpoint=True)
    out = np.zeros((n, words), np.uint64)
    for w in range(words):
        nb = min(64, bits - w * 64)
        dense = rng.random((n, nb)) < p
        out[:, w] = (dense.astype(np.uint64) << np.arange(nb, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
    return out
