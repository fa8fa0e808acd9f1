"""Dataset helpers for the benchmark configs (host-side, no arithmetic of the search path).

read_fvecs / read_ivecs read the TEXMEX record container the reference's benchmarks load (benchmarks/sift_io.py:10-53:
little-endian int32 dim prefix per row); outputs are pinned by golden vectors captured from the reference's reader.
SIFT1M is not shipped with the image; `sift_like` generates the documented synthetic stand-in
(integers 0..218 stored as f32, D=128 — SURVEY.md §8d C3).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np


class VecsFile:
    """A TEXMEX `.fvecs` / `.ivecs` file mapped read-only: every record is `<i4 dim` followed by `dim` 4-byte values
    (little endian) — the container of SIFT1M's base / query / ground-truth files.  The file is memory-mapped and viewed
    as a structured array, so opening SIFT1M costs nothing and `max_rows` never touches the rest of the file."""

    def __init__(self, path, value_dtype: str):
        self.path = Path(path)
        nbytes = self.path.stat().st_size
        if nbytes < 4:
            raise ValueError(f"{self.path}: no record header ({nbytes} bytes)")
        with open(self.path, "rb") as fh:
            self.dim = int(np.frombuffer(fh.read(4), "<i4")[0])
        if self.dim < 1:
            raise ValueError(f"{self.path}: record header says {self.dim} components")
        record = np.dtype([("dim", "<i4"), ("values", value_dtype, (self.dim,))])
        if nbytes % record.itemsize:
            raise ValueError(f"{self.path}: {nbytes} bytes is not a whole number of {record.itemsize}-byte records "
                             f"(dim {self.dim})")
        self.records = np.memmap(self.path, dtype=record, mode="r")

    def __len__(self) -> int:
        return int(self.records.shape[0])

    def rows(self, max_rows=None) -> np.ndarray:
        recs = self.records if max_rows is None else self.records[: max(int(max_rows), 0)]
        bad = np.flatnonzero(recs["dim"] != self.dim)
        if bad.size:
            raise ValueError(f"{self.path}: record {int(bad[0])} has {int(recs['dim'][bad[0]])} components, the file "
                             f"started with {self.dim}")
        return np.array(recs["values"])  # owned, contiguous copy


def read_fvecs(path, *, max_rows=None) -> np.ndarray:
    """float32 matrix (n, dim) of an .fvecs file."""
    return VecsFile(path, "<f4").rows(max_rows).astype(np.float32, copy=False)


def read_ivecs(path, *, max_rows=None) -> np.ndarray:
    """int32 matrix (n, dim) of an .ivecs file (e.g. sift_groundtruth.ivecs: 100 neighbour ids per query)."""
    return VecsFile(path, "<i4").rows(max_rows).astype(np.int32, copy=False)


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
