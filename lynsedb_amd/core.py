"""Host-side mirror of the `lynse._core` surface for the FLAT / IVF-Flat hot path.

Same class / function names, argument meaning and error behaviour as the reference's PyO3 module
(src/python/mod.rs) for the entry points on this path, so the reference's own benchmarks and tests
(`benchmarks/flat_search_bench.py`, `tests/standard_tests/test_backend.py`) read unchanged against it:

    FlatIndex            src/python/mod.rs:1942-2047   (FlatMmap)
    IvfFlatIndex         src/python/mod.rs:2056-2156   (IvfFlatMmap)
    py_compute_distance  src/python/mod.rs:2161-2185
    py_top_k_search      src/python/mod.rs:2189-2223
    DatabaseManager / Collection / SearchResult  (the subset `flat_search_bench.py:43-96` drives,
                         src/python/mod.rs:984-1005, :1135-1150, :1171-1193, :1343, :1383-1408,
                         :1876-1921, :2235-2418)

All arithmetic happens in liblynse_hip.so on the GPU; nothing here computes distances on the host.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check, lib

_METRIC_IS_BINARY = {_lib.METRIC_HAMMING, _lib.METRIC_JACCARD, _lib.METRIC_DICE, _lib.METRIC_TANIMOTO}


def visible_devices() -> list:
    """`LYNSE_HIP_DEVICES` (the build's one addition to the reference's env-only configuration, next to `RAYON_NUM_THREADS` /
    `LYNSE_SEGMENT_TARGET_BYTES`): a comma-separated list of HIP device ordinals this process may use, e.g. "2,3" — rank r of a node-local
    job takes entry r % len.  Unset: every device the runtime shows."""
    v = os.environ.get("LYNSE_HIP_DEVICES", "").strip()
    if v:
        try:
            devs = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            raise ValueError(f"LYNSE_HIP_DEVICES must be a comma-separated list of device ordinals, got {v!r}") from None
        if not devs or min(devs) < 0:
            raise ValueError(f"LYNSE_HIP_DEVICES must name at least one non-negative device ordinal, got {v!r}")
        return devs
    return list(range(max(_lib.device_count(), 1)))


def default_device() -> int:
    """Device ordinal: LYNSE_HIP_DEVICE, else entry LOCAL_RANK % len of LYNSE_HIP_DEVICES, else LOCAL_RANK (one process per GPU), else the
    first entry of LYNSE_HIP_DEVICES, else 0."""
    v = os.environ.get("LYNSE_HIP_DEVICE")
    if v is not None and v != "":
        return int(v)
    rank = os.environ.get("LOCAL_RANK")
    if os.environ.get("LYNSE_HIP_DEVICES", "").strip():
        devs = visible_devices()
        return devs[int(rank) % len(devs)] if rank not in (None, "") else devs[0]
    if rank not in (None, ""):
        return int(rank)
    return 0


def metric_from_str(metric: str) -> int:
    """DistanceMetric::from_str (src/distance/mod.rs:39-63); unknown -> ValueError("Unknown metric: ...")."""
    out = C.c_int(-1)
    check(lib.lynse_hip_metric_from_str(str(metric).encode(), C.byref(out)))
    return out.value


def metric_from_index_mode(mode: str) -> int:
    out = C.c_int(-1)
    check(lib.lynse_hip_metric_from_index_mode(str(mode).encode(), C.byref(out)))
    return out.value


def _f32(a, ndim: int, what: str) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    if a.ndim != ndim:
        raise ValueError(f"{what} must be a {ndim}-D array")
    if not a.flags["C_CONTIGUOUS"]:
        if ndim == 2:
            raise ValueError("numpy array must be contiguous (C-order)")  # src/python/mod.rs:1393-1395
        a = np.ascontiguousarray(a)
    return a


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _sync_producer(tensor) -> None:
    """The library works on its own HIP stream: device inputs must be complete before the call.
    Synchronise the torch stream that (may have) produced `tensor`."""
    import torch

    torch.cuda.current_stream(tensor.device).synchronize()


class FlatIndex:
    """`lynse._core.FlatIndex` (src/python/mod.rs:1942-2047) on one MI355X.

    The reference opens an mmapped file at `path`; here rows live in HBM (`path` may be None).  If
    `path` names an existing raw little-endian f32 row file (the reference's segment format,
    flat_mmap.rs:89-109) it is loaded.
    """

    def __init__(self, path: Optional[str], dim: int, device: Optional[int] = None, dtype: str = "f32"):
        self._h = C.c_void_p()
        self._dim = int(dim)
        dev = default_device() if device is None else int(device)
        d = dtype.strip().lower()  # VectorDtype::parse (src/storage/dtype.rs:12-21)
        if d in ("f32", "float32", "float"):
            self.dtype = "f32"
        elif d in ("f16", "float16", "half", "fp16"):
            self.dtype = "f16"
        else:
            raise ValueError(f"unsupported vector dtype '{dtype}'; expected float32/f32 or float16/f16")
        check(lib.lynse_hip_flat_create(self._dim, dev, C.byref(self._h)))
        if self.dtype == "f16":
            check(lib.lynse_hip_flat_set_dtype(self._h, 1))
        self.path = path
        if path and os.path.exists(path) and os.path.getsize(path) > 0:
            if self.dtype == "f16":
                bits = np.fromfile(path, dtype="<u2")
                if bits.size % self._dim:
                    raise IOError("vector file size is not a multiple of the row size")
                self.write_f16_bits(bits.reshape(-1, self._dim))
            else:
                data = np.fromfile(path, dtype="<f4")
                if data.size % self._dim:
                    raise IOError("vector file size is not a multiple of the row size")
                self.write(data.reshape(-1, self._dim))

    def write_f16_bits(self, bits) -> None:
        """Append rows given as IEEE binary16 words — the bytes of an F16 segment file (flat_mmap.rs:187-221)."""
        b = np.ascontiguousarray(bits, dtype=np.uint16)
        if b.ndim != 2 or b.shape[1] != self._dim:
            raise ValueError(f"data dimension mismatch: expected {self._dim}")
        check(lib.lynse_hip_flat_append_f16_bits(self._h, _ptr(b), b.shape[0]))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.lynse_hip_flat_destroy(h)

    def __len__(self) -> int:
        return int(lib.lynse_hip_flat_len(self._h))

    @property
    def dim(self) -> int:
        return self._dim

    @property
    def handle(self):
        return self._h

    def reserve(self, rows: int) -> None:
        check(lib.lynse_hip_flat_reserve(self._h, int(rows)))

    def write(self, data) -> None:
        """Append rows from a contiguous (n, dim) float32 array (FlatMmap::write)."""
        a = _f32(data, 2, "data")
        if a.shape[1] != self._dim:
            raise ValueError(f"data dimension mismatch: expected {self._dim}, got {a.shape[1]}")
        check(lib.lynse_hip_flat_append_f32(self._h, _ptr(a), a.shape[0]))

    def write_device(self, tensor) -> None:
        """Append rows already resident in HBM (a contiguous float32 torch tensor on this device)."""
        if tensor.dim() != 2 or tensor.shape[1] != self._dim or not tensor.is_contiguous():
            raise ValueError("tensor must be a contiguous (n, dim) float32 device tensor")
        _sync_producer(tensor)
        check(lib.lynse_hip_flat_append_f32_device(self._h, C.c_void_p(tensor.data_ptr()), tensor.shape[0]))

    def write_packed(self, words) -> None:
        """Append pre-packed one-bit rows: (n, ceil(dim/64)) uint64, LSB-first (BinaryData layout)."""
        w = np.ascontiguousarray(words, dtype=np.uint64)
        if w.ndim != 2 or w.shape[1] != (self._dim + 63) // 64:
            raise ValueError("packed rows must have ceil(dim/64) u64 words")
        check(lib.lynse_hip_flat_append_packed_u64(self._h, _ptr(w), w.shape[0]))

    def write_packed_device(self, tensor) -> None:
        _sync_producer(tensor)
        check(lib.lynse_hip_flat_append_packed_u64_device(self._h, C.c_void_p(tensor.data_ptr()), tensor.shape[0]))

    def finalize(self) -> None:
        check(lib.lynse_hip_flat_finalize(self._h))

    def set_row_map(self, stride: int, offset: int) -> None:
        check(lib.lynse_hip_flat_set_row_map(self._h, int(stride), int(offset)))

    def set_ip_form(self, form: int) -> None:
        check(lib.lynse_hip_flat_set_ip_form(self._h, int(form)))

    def set_fused_search(self, on: bool = True) -> None:
        """on=False forces the staged pipeline for small shards / few queries (the fused single-launch search is the default)."""
        check(lib.lynse_hip_flat_set_fused_search(self._h, 1 if on else 0))

    def set_plan(self, stage0_rows: int = 4096, growth: int = 8, cap: int = 8192) -> None:
        check(lib.lynse_hip_flat_set_plan(self._h, stage0_rows, growth, cap))

    def read_rows(self, first: int, n: int) -> np.ndarray:
        out = np.empty((n, self._dim), np.float32)
        check(lib.lynse_hip_flat_read_rows(self._h, first, n, _ptr(out)))
        return out

    def read_packed(self, first: int, n: int) -> np.ndarray:
        out = np.empty((n, (self._dim + 63) // 64), np.uint64)
        check(lib.lynse_hip_flat_read_packed(self._h, first, n, _ptr(out)))
        return out

    # -- search -------------------------------------------------------------------------------
    def search_batch_arrays(self, queries, k: int, metric):
        """Batched search returning padded arrays (rows u64[nq,k], dists f32[nq,k], counts u32[nq])."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        q = _f32(queries, 2, "queries")
        if q.shape[1] != self._dim:
            raise ValueError(f"query dimension mismatch: expected {self._dim}, got {q.shape[1]}")
        nq, k = q.shape[0], int(k)
        rows = np.empty((nq, max(k, 1)), np.uint64)
        dists = np.empty((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        check(lib.lynse_hip_flat_search_f32(self._h, _ptr(q), nq, k, m, _ptr(rows), _ptr(dists), _ptr(counts)))
        return rows[:, :k], dists[:, :k], counts

    def search_sq8_batch_arrays(self, queries, k: int, metric):
        """`FlatMmap::search(.., use_sq8 = true, ..)` — the FLAT-*-SQ8 index modes (flat_mmap.rs:891-905, :5868-5926)."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        q = _f32(queries, 2, "queries")
        if q.shape[1] != self._dim:
            raise ValueError(f"query dimension mismatch: expected {self._dim}, got {q.shape[1]}")
        nq, k = q.shape[0], int(k)
        rows = np.empty((nq, max(k, 1)), np.uint64)
        dists = np.empty((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        check(lib.lynse_hip_flat_search_sq8_f32(self._h, _ptr(q), nq, k, m, _ptr(rows), _ptr(dists), _ptr(counts)))
        return rows[:, :k], dists[:, :k], counts

    def sq8_params(self):
        mins, scales = np.empty(self._dim, np.float32), np.empty(self._dim, np.float32)
        check(lib.lynse_hip_flat_sq8_params(self._h, _ptr(mins), _ptr(scales)))
        return mins, scales

    def search_filtered_batch_arrays(self, queries, k: int, metric, subset_rows):
        """`FlatMmap::search_filtered` (flat_mmap.rs:491-815) for a batch sharing one subset of row indices."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        q = _f32(queries, 2, "queries")
        if q.shape[1] != self._dim:
            raise ValueError(f"query dimension mismatch: expected {self._dim}, got {q.shape[1]}")
        sub = np.ascontiguousarray(np.asarray(subset_rows).reshape(-1), dtype=np.uint64)
        nq, k = q.shape[0], int(k)
        rows = np.empty((nq, max(k, 1)), np.uint64)
        dists = np.empty((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        check(lib.lynse_hip_flat_search_filtered_f32(self._h, _ptr(q), nq, k, m, _ptr(sub) if sub.size else None, sub.size,
                                                     _ptr(rows), _ptr(dists), _ptr(counts)))
        return rows[:, :k], dists[:, :k], counts

    def search_filtered_bitset_batch_arrays(self, queries, k: int, metric, bitset_words):
        """Same with the subset given as the reference's BitSet words (u64, bit r of word r // 64)."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        q = _f32(queries, 2, "queries")
        if q.shape[1] != self._dim:
            raise ValueError(f"query dimension mismatch: expected {self._dim}, got {q.shape[1]}")
        words = np.ascontiguousarray(np.asarray(bitset_words).reshape(-1), dtype=np.uint64)
        nq, k = q.shape[0], int(k)
        rows = np.empty((nq, max(k, 1)), np.uint64)
        dists = np.empty((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        check(lib.lynse_hip_flat_search_filtered_bitset_f32(self._h, _ptr(q), nq, k, m, _ptr(words) if words.size else None,
                                                            words.size, _ptr(rows), _ptr(dists), _ptr(counts)))
        return rows[:, :k], dists[:, :k], counts

    def search_filtered(self, query, k: int, metric, subset_rows):
        """-> (rows u32[], distances f32[]) among `subset_rows` only."""
        q = _f32(query, 1, "query")
        rows, dists, counts = self.search_filtered_batch_arrays(q.reshape(1, -1), k, metric, subset_rows)
        c = int(counts[0])
        return rows[0, :c].astype(np.uint32), dists[0, :c].copy()

    def search_packed_arrays(self, query_words, k: int, metric):
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        qw = np.ascontiguousarray(query_words, dtype=np.uint64)
        if qw.ndim == 1:
            qw = qw.reshape(1, -1)
        nq, k = qw.shape[0], int(k)
        rows = np.empty((nq, max(k, 1)), np.uint64)
        dists = np.empty((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        check(lib.lynse_hip_flat_search_packed_u64(self._h, _ptr(qw), nq, k, m, _ptr(rows), _ptr(dists), _ptr(counts)))
        return rows[:, :k], dists[:, :k], counts

    def search(self, query, k: int = 10, metric: str = "ip"):
        """Brute-force top-k -> (indices u32[k'], distances f32[k']) (src/python/mod.rs:1990-2006)."""
        m = metric_from_str(metric)
        q = _f32(query, 1, "query")
        rows, dists, counts = self.search_batch_arrays(q.reshape(1, -1), k, m)
        c = int(counts[0])
        return rows[0, :c].astype(np.uint32), dists[0, :c].copy()

    def batch_search(self, queries, k: int = 10, metric: str = "ip"):
        """list of (indices, distances) per query (src/python/mod.rs:2018-2046).  The reference loops
        queries sequentially, re-reading the collection each time; here the batch shares one pass."""
        m = metric_from_str(metric)
        rows, dists, counts = self.search_batch_arrays(queries, k, m)
        return [(rows[i, :int(c)].astype(np.uint32), dists[i, :int(c)].copy()) for i, c in enumerate(counts)]

    def search_device(self, d_queries, k: int, metric, d_rows, d_dists, d_counts, stream=None):
        """All buffers are torch tensors resident on this device (bench path: no PCIe in the timed region)."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        nq = d_queries.shape[0]
        _sync_producer(d_queries)
        check(lib.lynse_hip_flat_search_f32_device(
            self._h, C.c_void_p(d_queries.data_ptr()), nq, int(k), m, C.c_void_p(d_rows.data_ptr()),
            C.c_void_p(d_dists.data_ptr()), C.c_void_p(d_counts.data_ptr()),
            C.c_void_p(stream) if stream else None))

    def search_packed_device(self, d_qwords, k: int, metric, d_rows, d_dists, d_counts, stream=None):
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        nq = d_qwords.shape[0]
        _sync_producer(d_qwords)
        check(lib.lynse_hip_flat_search_packed_u64_device(
            self._h, C.c_void_p(d_qwords.data_ptr()), nq, int(k), m, C.c_void_p(d_rows.data_ptr()),
            C.c_void_p(d_dists.data_ptr()), C.c_void_p(d_counts.data_ptr()),
            C.c_void_p(stream) if stream else None))

    # -- searches in flight (lynse_hip_flat_search_submit_* / _wait) --------------------------------
    def search_submit(self, d_queries, k: int, metric, d_rows, d_dists, d_counts, comm=None) -> "SearchTicket":
        """Enqueue one batch (<= 256 queries, torch tensors on this device) on a search context of the index and return a
        ticket; `wait()` makes the results final in d_rows / d_dists / d_counts.  Up to LYNSE_HIP_CONTEXTS batches overlap on
        the device.  `comm`: the communicator handle of a row-sharded collection (a collective then)."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        if m >= 3 and d_queries.is_floating_point():
            # float queries of a binary metric are packed by the blocking entry point (pack_binary_query); batches in flight
            # take packed words: answer this one now (only without a communicator: the sharded entry points are packed-only too)
            if comm is not None:
                raise ValueError("sharded searches of a binary metric take packed u64 query words")
            self.search_device(d_queries, k, m, d_rows, d_dists, d_counts)
            return SearchTicket(None, None)
        _sync_producer(d_queries)
        t = C.c_void_p()
        fn = lib.lynse_hip_flat_search_submit_packed_u64_device if m >= 3 else lib.lynse_hip_flat_search_submit_f32_device
        check(fn(self._h, comm, C.c_void_p(d_queries.data_ptr()), d_queries.shape[0], int(k), m, C.c_void_p(d_rows.data_ptr()),
                 C.c_void_p(d_dists.data_ptr()), C.c_void_p(d_counts.data_ptr()), C.byref(t)))
        return SearchTicket(t, (d_queries, d_rows, d_dists, d_counts))

    # -- profiling ----------------------------------------------------------------------------
    def profile_enable(self, on=True) -> None:
        """True / 1: time every search; n > 1: every n-th search (HIP events between the kernels cost microseconds); False: off."""
        check(lib.lynse_hip_flat_profile_enable(self._h, int(on)))

    def profile_get(self, reset: bool = True) -> dict:
        p = _lib.Profile()
        check(lib.lynse_hip_flat_profile_get(self._h, C.byref(p), 1 if reset else 0))
        return {f: getattr(p, f) for f, _ in _lib.Profile._fields_}

    def prepare(self, metric, nq: int = 256) -> None:
        """Build the derived copies a batch of `nq` queries of `metric` reads now (SQ8 codes, +-1 bytes, packed words, shadow)."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        check(lib.lynse_hip_flat_prepare(self._h, m, int(nq)))

    def hbm_bytes(self) -> int:
        return int(lib.lynse_hip_flat_hbm_bytes(self._h))

    def coarse_scores(self, queries, metric, coarse: str = "i8"):
        """Diagnostics of the certified coarse pass (`lynse_hip_flat_coarse_scores`): for a shard of <= 16,384 rows and <= 256 queries the
        coarse score / distance of every (query, row) as the scan kernels compute it, the certified bound E per query, and the form bits."""
        q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        scores = np.empty((q.shape[0], len(self)), np.float32)
        bound = np.empty(q.shape[0], np.float32)
        form = C.c_int(0)
        check(lib.lynse_hip_flat_coarse_scores(self._h, _ptr(q), q.shape[0], m, 1 if coarse in ("i8", "int8", 1, True) else 0, _ptr(scores), _ptr(bound), C.byref(form)))
        return scores, bound, form.value

    def coarse_state(self) -> dict:
        """State of the coarse-pass selection: overflow strikes of the certified int8 pass (3 = switched off, -1 = off because the
        rows are not finite) and the rows covered by the SQ8 codes built so far."""
        strikes, rows = C.c_int(0), C.c_uint64(0)
        check(lib.lynse_hip_flat_coarse_state(self._h, C.byref(strikes), C.byref(rows)))
        return {"i8c_strikes": strikes.value, "sq8_rows": rows.value, "bpm_rows": int(lib.lynse_hip_flat_bpm_rows(self._h))}


class SearchTicket:
    """A batch in flight (FlatIndex.search_submit).  Keeps the tensors of the batch alive until it is waited for."""

    def __init__(self, handle, keep, ivf: bool = False):
        self._t, self._keep, self._ivf = handle, keep, ivf

    def wait(self) -> None:
        t, self._t = self._t, None
        if t is not None:
            try:
                check((lib.lynse_hip_ivf_search_wait if self._ivf else lib.lynse_hip_flat_search_wait)(t))
            finally:
                self._keep = None

    def __del__(self):  # a ticket must not die with its batch in flight: it holds a search context and the reader lock
        try:
            self.wait()
        except Exception:  # noqa: BLE001
            pass


class IvfFlatIndex:
    """`lynse._core.IvfFlatIndex` (src/python/mod.rs:2056-2156): k-means partitions, rows stored as
    contiguous per-partition slabs in HBM, search scans the nprobe nearest slabs."""

    def __init__(self, handle, dim: int):
        self._h = handle
        self._dim = dim

    def profile_enable(self, on=True) -> None:
        check(lib.lynse_hip_ivf_profile_enable(self._h, int(on)))

    def profile_get(self, reset: bool = True) -> dict:
        """The slab store's profile; `last_plan` bit 2 / bit 6: the last staged chunk ran / started on the certified int8 pass."""
        p = _lib.Profile()
        check(lib.lynse_hip_ivf_profile_get(self._h, C.byref(p), 1 if reset else 0))
        return {f: getattr(p, f) for f, _ in _lib.Profile._fields_}

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.lynse_hip_ivf_destroy(h)

    @staticmethod
    def build(path, data, dim: int, n_partitions: int = 256, n_iters: int = 20, metric: str = "ip",
              device: Optional[int] = None, l2_partitions: bool = True) -> "IvfFlatIndex":
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        a = _f32(data, 2, "data")
        if a.shape[1] != dim:
            raise ValueError(f"data dimension mismatch: expected {dim}, got {a.shape[1]}")
        if n_partitions <= 0:
            raise IOError("IVF partition count must be greater than zero")
        if a.shape[0] < n_partitions:
            raise IOError("IVF requires at least as many vectors as partitions")
        h = C.c_void_p()
        dev = default_device() if device is None else int(device)
        if m >= 3:  # IVF-HAMMING/JACCARD-BINARY (src/index/mod.rs:376-385) is an IVFIndex mode: no IvfFlat L2 cells
            l2_partitions = False
        check(lib.lynse_hip_ivf_build(_ptr(a), a.shape[0], dim, n_partitions, n_iters, m,
                                      1 if l2_partitions else 0, dev, C.byref(h)))
        return IvfFlatIndex(h, dim)

    @staticmethod
    def build_device(d_rows, dim: int, n_partitions: int = 256, n_iters: int = 20, metric: str = "ip",
                     l2_partitions: bool = True) -> "IvfFlatIndex":
        """`build` over rows already resident in HBM (a contiguous float32 torch tensor on the target device): k-means, slab
        reordering and the store never stage row data through host memory.  Float metrics."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        if d_rows.dim() != 2 or d_rows.shape[1] != dim or not d_rows.is_contiguous():
            raise ValueError("rows must be a contiguous (n, dim) float32 device tensor")
        _sync_producer(d_rows)
        h = C.c_void_p()
        check(lib.lynse_hip_ivf_build_device(C.c_void_p(d_rows.data_ptr()), d_rows.shape[0], dim, n_partitions, n_iters, m,
                                             1 if l2_partitions else 0, d_rows.device.index or 0, C.byref(h)))
        return IvfFlatIndex(h, dim)

    @staticmethod
    def load_device(d_rows, centroids, assignments, metric: str = "ip", ivfflat_routing: bool = False) -> "IvfFlatIndex":
        """`load` with the rows in HBM; centroids / assignments are host arrays."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        c = _f32(centroids, 2, "centroids")
        asg = np.ascontiguousarray(assignments, dtype=np.uint32)
        _sync_producer(d_rows)
        h = C.c_void_p()
        check(lib.lynse_hip_ivf_load_device(C.c_void_p(d_rows.data_ptr()), d_rows.shape[0], d_rows.shape[1], _ptr(c), c.shape[0], _ptr(asg), m,
                                            d_rows.device.index or 0, C.byref(h)))
        idx = IvfFlatIndex(h, d_rows.shape[1])
        if ivfflat_routing:
            check(lib.lynse_hip_ivf_set_routing(h, 1))
        return idx

    def search_submit(self, d_queries, k: int, nprobe: int, d_rows, d_dists, d_counts, comm=None) -> "SearchTicket":
        """One batch IN FLIGHT (lynse_hip_ivf_search_submit_f32_device): <= 256 queries, torch tensors on the index's device; `wait()`
        of the ticket makes d_rows / d_dists / d_counts final — the results of `search_device`.  Up to LYNSE_HIP_CONTEXTS - 1 batches
        overlap on the device.  `comm`: the communicator handle of a row-sharded index (a collective then)."""
        _sync_producer(d_queries)
        t = C.c_void_p()
        check(lib.lynse_hip_ivf_search_submit_f32_device(self._h, comm, C.c_void_p(d_queries.data_ptr()), d_queries.shape[0], int(k), int(nprobe),
                                                         C.c_void_p(d_rows.data_ptr()), C.c_void_p(d_dists.data_ptr()),
                                                         C.c_void_p(d_counts.data_ptr()), C.byref(t)))
        return SearchTicket(t, (d_queries, d_rows, d_dists, d_counts), ivf=True)

    def ticket_stats(self) -> dict:
        """Tickets of this index so far: enqueued without a host synchronisation / answered inside submit / re-answered inside wait."""
        out = np.zeros(3, np.uint64)
        check(lib.lynse_hip_ivf_ticket_stats(self._h, _ptr(out)))
        return {"in_flight": int(out[0]), "inside_submit": int(out[1]), "redone_in_wait": int(out[2])}

    def set_fused_search(self, on: bool = True) -> None:
        """on=False forces the staged pipeline for few-query searches too (tests, A/B); results are identical."""
        check(lib.lynse_hip_ivf_set_fused_search(self._h, 1 if on else 0))

    def search_device(self, d_queries, k: int, nprobe: int, d_rows, d_dists, d_counts) -> None:
        """Queries and outputs are torch tensors on the index's device (rows i64[nq,k] holding u64 bits, dists f32[nq,k],
        counts i32[nq])."""
        _sync_producer(d_queries)
        check(lib.lynse_hip_ivf_search_f32_device(self._h, C.c_void_p(d_queries.data_ptr()), d_queries.shape[0], int(k), int(nprobe),
                                                  C.c_void_p(d_rows.data_ptr()), C.c_void_p(d_dists.data_ptr()),
                                                  C.c_void_p(d_counts.data_ptr())))

    @staticmethod
    def load(data, centroids, assignments, metric: str = "ip", device: Optional[int] = None,
             ivfflat_routing: bool = False, thresholds=None) -> "IvfFlatIndex":
        """Assemble from given centroids + assignments (parity tests feed the oracle's k-means output).
        Binary metrics also take the BinaryQuantizer thresholds (`data` = the raw rows)."""
        m = metric_from_str(metric)
        a = _f32(data, 2, "data")
        c = _f32(centroids, 2, "centroids")
        asg = np.ascontiguousarray(assignments, dtype=np.uint32)
        h = C.c_void_p()
        dev = default_device() if device is None else int(device)
        if m >= 3:
            if thresholds is None:
                raise ValueError("binary metrics need the BinaryQuantizer thresholds")
            t = _f32(thresholds, 1, "thresholds")
            if t.size != a.shape[1]:
                raise ValueError("thresholds dimension mismatch")
            check(lib.lynse_hip_ivf_load_binary(_ptr(a), a.shape[0], a.shape[1], _ptr(c), c.shape[0], _ptr(asg), m, _ptr(t),
                                                dev, C.byref(h)))
            return IvfFlatIndex(h, a.shape[1])
        check(lib.lynse_hip_ivf_load(_ptr(a), a.shape[0], a.shape[1], _ptr(c), c.shape[0], _ptr(asg), m, dev, C.byref(h)))
        idx = IvfFlatIndex(h, a.shape[1])
        if ivfflat_routing:
            check(lib.lynse_hip_ivf_set_routing(h, 1))
        return idx

    def __len__(self) -> int:
        return int(lib.lynse_hip_ivf_len(self._h))

    @property
    def dim(self) -> int:
        return self._dim

    @property
    def n_partitions(self) -> int:
        return int(lib.lynse_hip_ivf_nlist(self._h))

    def insert(self, data) -> None:
        """`IVFIndex::insert` (ivf.rs:392-441): assign the new rows to the existing centroids, append them (no retraining)."""
        a = _f32(data, 2, "data")
        if a.shape[1] != self._dim:
            raise ValueError(f"dimension mismatch: expected {self._dim}, got {a.shape[1]}")
        check(lib.lynse_hip_ivf_insert_f32(self._h, _ptr(a), a.shape[0]))

    def delete(self, rows) -> None:
        """`IVFIndex::delete` (ivf.rs:350-390): drop the listed rows; the rest are renumbered in order and reassigned."""
        r = np.ascontiguousarray(np.asarray(rows).reshape(-1), dtype=np.uint64)
        check(lib.lynse_hip_ivf_delete_rows(self._h, _ptr(r) if r.size else None, r.size))

    def assign(self, data) -> np.ndarray:
        a = _f32(data, 2, "data")
        out = np.zeros(a.shape[0], np.uint32)
        check(lib.lynse_hip_ivf_assign_f32(self._h, _ptr(a), a.shape[0], _ptr(out)))
        return out

    def export(self):
        n, nl = len(self), self.n_partitions
        cen = np.empty((nl, self._dim), np.float32)
        asg = np.empty(n, np.uint32)
        off = np.empty(nl + 1, np.uint64)
        orig = np.empty(n, np.uint32)
        check(lib.lynse_hip_ivf_export(self._h, _ptr(cen), _ptr(asg), _ptr(off), _ptr(orig)))
        return cen, asg, off, orig

    def thresholds(self):
        """Binary index: (BinaryQuantizer thresholds f32[dim], already_binary)."""
        t = np.empty(self._dim, np.float32)
        ab = C.c_int(0)
        check(lib.lynse_hip_ivf_thresholds(self._h, _ptr(t), C.byref(ab)))
        return t, bool(ab.value)

    def search_batch_arrays(self, queries, k: int, nprobe: int):
        q = _f32(queries, 2, "queries")
        if q.shape[1] != self._dim:
            raise ValueError(f"query dimension mismatch: expected {self._dim}, got {q.shape[1]}")
        nq, k = q.shape[0], int(k)
        rows = np.empty((nq, max(k, 1)), np.uint64)
        dists = np.empty((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        check(lib.lynse_hip_ivf_search_f32(self._h, _ptr(q), nq, k, int(nprobe), _ptr(rows), _ptr(dists), _ptr(counts)))
        return rows[:, :k], dists[:, :k], counts

    def search_filtered_batch_arrays(self, queries, k: int, nprobe: int, subset_rows):
        """`IVFIndex::search` with `SearchParams.subset` (ivf.rs:251-265), one subset for the batch."""
        q = _f32(queries, 2, "queries")
        if q.shape[1] != self._dim:
            raise ValueError(f"query dimension mismatch: expected {self._dim}, got {q.shape[1]}")
        sub = np.ascontiguousarray(np.asarray(subset_rows).reshape(-1), dtype=np.uint64)
        nq, k = q.shape[0], int(k)
        rows = np.empty((nq, max(k, 1)), np.uint64)
        dists = np.empty((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        check(lib.lynse_hip_ivf_search_filtered_f32(self._h, _ptr(q), nq, k, int(nprobe), _ptr(sub) if sub.size else None, sub.size,
                                                    _ptr(rows), _ptr(dists), _ptr(counts)))
        return rows[:, :k], dists[:, :k], counts

    def search_metric_batch_arrays(self, queries, k: int, nprobe: int, metric):
        """`IvfFlatMmap::search(query, k, nprobe, metric)` (ivf_flat_mmap.rs:225-305): the metric of the CALL drives the
        centroid routing, the scoring and the sort direction — the partitions are metric-agnostic."""
        m = metric if isinstance(metric, int) else metric_from_str(metric)
        q = _f32(queries, 2, "queries")
        if q.shape[1] != self._dim:
            raise ValueError(f"query dimension mismatch: expected {self._dim}, got {q.shape[1]}")
        nq, k = q.shape[0], int(k)
        rows = np.empty((nq, max(k, 1)), np.uint64)
        dists = np.empty((nq, max(k, 1)), np.float32)
        counts = np.zeros(nq, np.uint32)
        check(lib.lynse_hip_ivf_search_metric_f32(self._h, _ptr(q), nq, k, int(nprobe), m, _ptr(rows), _ptr(dists), _ptr(counts)))
        return rows[:, :k], dists[:, :k], counts

    def search(self, query, k: int = 10, nprobe: int = 10, metric: str = "ip"):
        """PyIvfFlatIndex.search (src/python/mod.rs:2130-2155): `metric` is the metric of this search."""
        m = metric_from_str(metric)  # ValueError on unknown names, like the reference
        q = _f32(query, 1, "query")
        if q.size != self._dim:
            raise ValueError(f"query dimension mismatch: expected {self._dim}, got {q.size}")
        rows, dists, counts = self.search_metric_batch_arrays(q.reshape(1, -1), k, nprobe, m)
        c = int(counts[0])
        return rows[0, :c].astype(np.uint32), dists[0, :c].copy()


def py_compute_distance(a, b, metric: str) -> float:
    """src/python/mod.rs:2161-2185."""
    m = metric_from_str(metric)
    a = _f32(a, 1, "a")
    b = _f32(b, 1, "b")
    if a.size != b.size:
        raise ValueError("Vector dimensions must match")
    out = C.c_float(0.0)
    check(lib.lynse_hip_compute_distance(_ptr(a), _ptr(b), a.size, m, default_device(), C.byref(out)))
    return float(out.value)


def py_top_k_search(query, candidates, metric: str, k: int):
    """src/python/mod.rs:2189-2223 -> (indices u32[], distances f32[])."""
    m = metric_from_str(metric)
    q = _f32(query, 1, "query")
    c = _f32(candidates, 2, "candidates")
    if q.size != c.shape[1]:
        raise ValueError("Query dimension must match candidate dimension")
    k = int(k)
    idx = np.empty(max(k, 1), np.uint32)
    dist = np.empty(max(k, 1), np.float32)
    cnt = C.c_uint32(0)
    check(lib.lynse_hip_top_k_search(_ptr(q), _ptr(c), c.shape[0], c.shape[1], k, m, default_device(),
                                     _ptr(idx), _ptr(dist), C.byref(cnt)))
    return idx[:cnt.value].copy(), dist[:cnt.value].copy()


def merge_topk(ids, dists, counts, k: int, metric) -> tuple:
    """VectorStore::merge_results (vector_store.rs:953-970) over per-shard blocks [n_lists, stride]."""
    m = metric if isinstance(metric, int) else metric_from_str(metric)
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    dists = np.ascontiguousarray(dists, dtype=np.float32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n_lists, stride = ids.shape
    out_i = np.empty(max(k, 1), np.uint64)
    out_d = np.empty(max(k, 1), np.float32)
    cnt = C.c_uint32(0)
    check(lib.lynse_hip_merge_topk(_ptr(ids), _ptr(dists), _ptr(counts), n_lists, stride, int(k), m,
                                   _ptr(out_i), _ptr(out_d), C.byref(cnt)))
    return out_i[:cnt.value].copy(), out_d[:cnt.value].copy()


# ---------------------------------------------------------------------------------------------
# Minimal engine glue: only what `benchmarks/flat_search_bench.py:43-96` and the reference's
# search tests drive.  Storage/WAL/fields/filters are out of scope.
# ---------------------------------------------------------------------------------------------
class SearchResult:
    """src/python/mod.rs:1876-1921 / engine.rs:6895-6902."""

    def __init__(self, ids: np.ndarray, distances: np.ndarray, index_mode: str, dimension: int, k: int):
        self._ids = np.asarray(ids, dtype=np.int64)
        self._d = np.asarray(distances, dtype=np.float32)
        self._mode, self._dim, self._k = index_mode, dimension, k

    def ids(self) -> np.ndarray:
        return self._ids

    def distances(self) -> np.ndarray:
        return self._d

    def fields(self) -> list:
        return []

    def index_mode(self) -> str:
        return self._mode

    def to_tuple(self):
        return self._ids, self._d, []

    def __len__(self) -> int:
        return int(self._ids.size)

    def __repr__(self) -> str:
        return f"SearchResult(n={len(self)}, k={self._k}, dim={self._dim}, index={self._mode})"


PENDING_INGEST_FLUSH_ROWS = 10_000              # src/engine.rs:93
PENDING_INGEST_FLUSH_BYTES = 32 * 1024 * 1024   # src/engine.rs:94


class BitSet:
    """Row subset in the reference's layout (src/storage/bitset.rs): u64 words, bit r of word r // 64 = row r."""

    def __init__(self, n_rows: int, words: Optional[np.ndarray] = None):
        self.n_rows = int(n_rows)
        nw = (self.n_rows + 63) // 64
        self.words = np.zeros(nw, np.uint64) if words is None else np.ascontiguousarray(words, dtype=np.uint64)
        if self.words.size != nw:
            raise ValueError("BitSet words do not match the row count")

    @staticmethod
    def from_rows(rows, n_rows: int) -> "BitSet":
        b = BitSet(n_rows)
        r = np.unique(np.asarray(rows, dtype=np.uint64).reshape(-1))
        r = r[r < np.uint64(n_rows)]
        np.bitwise_or.at(b.words, (r >> np.uint64(6)).astype(np.int64), np.uint64(1) << (r & np.uint64(63)))
        return b

    def count(self) -> int:
        return int(np.unpackbits(self.words.view(np.uint8)).sum())

    def contains(self, row: int) -> bool:
        return 0 <= row < self.n_rows and bool((int(self.words[row >> 6]) >> (row & 63)) & 1)

    def to_vec(self) -> np.ndarray:
        bits = np.unpackbits(self.words.view(np.uint8), bitorder="little")
        return np.nonzero(bits)[0].astype(np.uint64)


class Collection:
    """The search-path subset of `lynse._core.Collection` (engine.rs Collection): buffered ingest, commit, index build
    and `search / batch_search` = `search_with_precomputed_filter` (src/engine.rs:4718-4833): k inflated by the
    tombstone count, the index / flat / subset-filtered search over the flushed rows on the GPU, the pending (not yet
    flushed) rows scored with `top_k_search`, `merge_row_results`, row -> user id, `filter_tombstoned_limit`.

    The reference resolves `where_expr` to a row BitSet through its field store (out of scope); the
    precomputed filter itself is in scope and is passed here as `subset=` (a `BitSet` or an array of row indices)."""

    def __init__(self, name: str, dim: int, device: Optional[int] = None):
        self._name, self._dim = name, int(dim)
        self._device = device
        self._flat = FlatIndex(None, dim, device)
        self._id_arrays: list = []     # row -> user id (engine.rs:3071-3073)
        self._index_mode = "FLAT-IP"   # resolve_metric default IP (engine.rs:5529-5534)
        self._metric = _lib.METRIC_IP
        self._ivf: Optional[IvfFlatIndex] = None
        self._ivf_rows = 0              # rows the IVF index was built over
        self._ivf_params: dict = {}
        self._ivf_nprobe = 32           # IndexBuildOptions default (src/index/mod.rs:498-655)
        self._pending_vecs: list = []   # PendingIngestBuffer (engine.rs:125, :190-245)
        self._pending_ids: list = []
        self._pending_rows = 0
        self._tombstone: set = set()    # user ids (engine.rs:3182-3194)

    def name(self) -> str:
        return self._name

    # -- ingest -------------------------------------------------------------------------------
    def add_items(self, vectors, ids: Sequence[int], fields=None) -> None:
        """Buffered like the reference (engine.rs:3886-3900): rows wait in the pending buffer until it holds
        PENDING_INGEST_FLUSH_ROWS rows / PENDING_INGEST_FLUSH_BYTES bytes, or until commit(); searches see them through
        `pending_search`."""
        a = _f32(vectors, 2, "vectors")
        if a.shape[1] != self._dim:
            raise RuntimeError(f"Dimension mismatch: expected {self._dim}, got {a.shape[1]}")
        if len(ids) != a.shape[0]:
            raise RuntimeError("ids length must match the number of vectors")
        if fields is not None:
            raise NotImplementedError("field metadata is outside the FLAT/IVF hot path (SURVEY.md §2 #20)")
        if a.shape[0] == 0:
            return
        self._pending_vecs.append(np.array(a, dtype=np.float32, copy=True))
        self._pending_ids.append(np.asarray(ids, dtype=np.int64).copy())
        self._pending_rows += a.shape[0]
        if self._pending_rows >= PENDING_INGEST_FLUSH_ROWS or self._pending_rows * self._dim * 4 >= PENDING_INGEST_FLUSH_BYTES:
            self._flush_pending()

    def pending_len(self) -> int:
        return self._pending_rows

    def _flush_pending(self) -> None:  # Collection::flush_pending_ingest (engine.rs:3573-3590)
        for a, i in zip(self._pending_vecs, self._pending_ids):
            self._flat.write(a)
            self._id_arrays.append(i)
        self._pending_vecs, self._pending_ids, self._pending_rows = [], [], 0

    def commit(self) -> None:
        self._flush_pending()
        self._flat.finalize()

    def shape(self):
        return (len(self._flat) + self._pending_rows, self._dim)

    def _id_map(self) -> np.ndarray:
        if len(self._id_arrays) != 1:
            self._id_arrays = [np.concatenate(self._id_arrays) if self._id_arrays else np.zeros(0, np.int64)]
        return self._id_arrays[0]

    # -- soft deletes (engine.rs:3182-3284) -----------------------------------------------------
    def delete_items(self, ids: Iterable[int]) -> None:
        self._tombstone.update(int(i) for i in ids)

    def restore_items(self, ids: Iterable[int]) -> None:
        self._tombstone.difference_update(int(i) for i in ids)

    def list_deleted_ids(self) -> list:
        return sorted(self._tombstone)

    # -- index --------------------------------------------------------------------------------
    def build_index(self, index_type: str, params: Optional[dict] = None) -> None:
        """Collection::build_index_with_build_options (engine.rs:4515-4655): flushes the pending rows (:4521); FLAT-*
        keeps no index object (engine.rs:4559-4567; `FLAT-*-SQ8` switches the flat scan to the SQ8 two-pass mode,
        flat_mmap.rs:891-905); IVF-* trains a k-means IVFIndex (engine.rs:4616-4627), `IVF-{HAMMING,JACCARD}-BINARY`
        the binary-quantised one (src/index/mod.rs:376-385)."""
        mode = str(index_type).upper()
        try:
            metric = metric_from_index_mode(mode)
        except NotImplementedError:
            raise
        except ValueError as e:
            raise RuntimeError(str(e))
        params = dict(params or {})
        self._flush_pending()
        if mode.startswith("FLAT"):
            if any(t in mode.split("-") for t in ("PQ", "RABITQ", "POLARVEC")):
                raise NotImplementedError("PQ / RaBitQ / PolarVec flat modes are outside this path (SURVEY.md §2)")
            self._ivf = None
        elif mode.startswith("IVF"):
            if any(t in mode.split("-") for t in ("SQ8", "PQ")):
                raise NotImplementedError("quantized IVF variants are outside this path")
            self._ivf_params = {"n_clusters": int(params.get("n_clusters", 256))}
            self._ivf_nprobe = int(params.get("nprobe", 32))
            self._index_mode, self._metric = mode, metric
            self._build_ivf()
        else:
            raise NotImplementedError(f"index type {index_type} is outside the FLAT/IVF hot path")
        self._index_mode, self._metric = mode, metric

    def _build_ivf(self) -> None:
        n = len(self._flat)
        data = self._flat.read_rows(0, n)
        nlist = min(self._ivf_params["n_clusters"], max(n, 1))
        # the metric id goes through as it is: binary metrics build the IVF-*-BINARY mode, float metrics an IVFIndex
        # trained with its routing metric (ivf.rs:163-170)
        self._ivf = IvfFlatIndex.build(None, data, self._dim, nlist, 20, int(self._metric), device=self._device, l2_partitions=False)
        self._ivf_rows = n

    def _use_sq8(self) -> bool:  # Collection::resolve_use_sq8 (engine.rs:4684-4689)
        return "SQ8" in self._index_mode.upper()

    # -- search -------------------------------------------------------------------------------
    def _subset_rows(self, subset) -> Optional[np.ndarray]:
        if subset is None:
            return None
        if isinstance(subset, BitSet):
            return subset.to_vec()
        return np.unique(np.asarray(subset, dtype=np.uint64).reshape(-1))

    def _pending_search(self, query: np.ndarray, k: int, subset_rows: Optional[np.ndarray]):
        """Collection::pending_search (engine.rs:3310-3361): the un-flushed rows, scored with `top_k_search`."""
        if k == 0 or self._pending_rows == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.float32)
        data = np.concatenate(self._pending_vecs) if len(self._pending_vecs) > 1 else self._pending_vecs[0]
        row_offsets = np.arange(len(self._flat), len(self._flat) + data.shape[0], dtype=np.uint64)
        if subset_rows is not None:
            keep = np.isin(row_offsets, subset_rows)
            data, row_offsets = np.ascontiguousarray(data[keep]), row_offsets[keep]
        if row_offsets.size == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.float32)
        idx, dists = py_top_k_search(query, data, _METRIC_NAMES[self._metric], k)
        return row_offsets[idx.astype(np.int64)], dists

    def _user_ids(self, rows: np.ndarray) -> np.ndarray:
        """row_to_user_id (engine.rs:3071-3073) over flushed and pending rows."""
        rows = rows.astype(np.int64)
        n_flat = len(self._flat)
        if self._pending_rows == 0:
            return self._id_map()[rows]
        ids = np.concatenate([self._id_map()] + self._pending_ids)
        assert ids.size == n_flat + self._pending_rows
        return ids[rows]

    def _base_search(self, q: np.ndarray, search_k: int, nprobe: int, subset_rows: Optional[np.ndarray]):
        """The device part of search_with_precomputed_filter for a batch sharing one subset -> rows, dists, counts."""
        nq = q.shape[0]
        if len(self._flat) == 0 or search_k == 0:
            return np.zeros((nq, 0), np.uint64), np.zeros((nq, 0), np.float32), np.zeros(nq, np.uint32)
        if self._ivf is not None:
            if self._ivf_rows != len(self._flat):
                # rows were committed after the build: Collection::flush hands them to idx.insert (engine.rs:3642-3645, :3858) —
                # assigned to the existing centroids, no retraining
                n_new = len(self._flat) - self._ivf_rows
                self._ivf.insert(self._flat.read_rows(self._ivf_rows, n_new))
                self._ivf_rows = len(self._flat)
            np_ = self._ivf_nprobe if not nprobe else int(nprobe)  # nprobe == 0 -> the build default (engine.rs:4743-4746)
            if subset_rows is not None:
                if subset_rows.size == 0:
                    return np.zeros((nq, 0), np.uint64), np.zeros((nq, 0), np.float32), np.zeros(nq, np.uint32)
                return self._ivf.search_filtered_batch_arrays(q, search_k, np_, subset_rows)
            return self._ivf.search_batch_arrays(q, search_k, np_)
        if subset_rows is not None:  # brute_force_search_filtered (engine.rs:5541-5566): always the exact filtered scan
            if subset_rows.size == 0:
                return np.zeros((nq, 0), np.uint64), np.zeros((nq, 0), np.float32), np.zeros(nq, np.uint32)
            return self._flat.search_filtered_batch_arrays(q, search_k, self._metric, subset_rows)
        if self._use_sq8() and self._metric in (_lib.METRIC_IP, _lib.METRIC_L2, _lib.METRIC_COSINE):
            return self._flat.search_sq8_batch_arrays(q, search_k, self._metric)
        return self._flat.search_batch_arrays(q, search_k, self._metric)

    def search(self, vector, k: Optional[int] = None, where_expr: Optional[str] = None,
               nprobe: Optional[int] = None, approx: Optional[bool] = None, eps: Optional[float] = None,
               subset=None) -> SearchResult:
        res = self.batch_search(np.asarray(vector, dtype=np.float32).reshape(1, -1), k, where_expr, nprobe, subset=subset)
        return res[0]

    def search_profile(self, vector, k: Optional[int] = None, where_expr: Optional[str] = None, nprobe: Optional[int] = None,
                       approx: Optional[bool] = None, eps: Optional[float] = None, subset=None) -> dict:
        """`Collection.search_profile` (src/python/mod.rs:1240-1271 over Collection::search_with_profile, src/engine.rs:5005-5054): the
        search + a `QueryProfile` (engine.rs:6906-6919) with the reference's field names — query_kind, vector_field, index_path
        ("ann_index" / "flat_mmap_filtered" / "flat_mmap", engine.rs:5163-5177), total_vectors, filter_expression, filter_matches,
        scanned_vectors (= filter_matches or total_vectors, as estimate_scanned_vectors does, :5179-5193), result_count, filter_us, search_us,
        rerank_us, total_us.  `device` is this build's addition: what `lynse_hip_flat_profile_get` / `lynse_hip_ivf_profile_get` measured for the
        search with HIP events on the search stream (pipeline_us, scan_us, scan_launches, rows and bytes the scan launches streamed,
        rescored_candidates, fallback_queries).  The precomputed filter is `subset=` (the field store that resolves `where_expr` is out of scope)."""
        import time

        started = time.perf_counter()
        filter_us, filter_matches = 0, None
        if where_expr:
            raise NotImplementedError("`where_expr` needs the field store (out of scope, SURVEY.md §2); pass the resolved row filter as subset=")
        if subset is not None:
            t0 = time.perf_counter()
            filter_matches = int(self._subset_rows(subset).size)
            filter_us = int((time.perf_counter() - t0) * 1e6)
        target = self._ivf if self._ivf is not None else self._flat
        target.profile_enable(True)
        target.profile_get(reset=True)
        t0 = time.perf_counter()
        try:
            res = self.search(vector, k, None, nprobe, approx, eps, subset=subset)
        finally:
            dev = target.profile_get(reset=True)
            target.profile_enable(False)
        search_us = int((time.perf_counter() - t0) * 1e6)
        total = int(self.shape()[0])
        profile = {"query_kind": "vector", "vector_field": "default",
                   "index_path": "ann_index" if self._ivf is not None else ("flat_mmap_filtered" if subset is not None else "flat_mmap"),
                   "total_vectors": total, "filter_expression": None, "filter_matches": filter_matches,
                   "scanned_vectors": filter_matches if filter_matches is not None else total, "result_count": len(res),
                   "filter_us": filter_us, "search_us": search_us, "rerank_us": 0, "total_us": int((time.perf_counter() - started) * 1e6),
                   "device": {"pipeline_us": float(dev["total_us"]), "scan_us": float(dev["scan_us"]), "scan_launches": int(dev["scan_launches"]),
                              "scan_rows": int(dev["scan_rows"]), "scan_bytes": int(dev["scan_bytes"]), "rescored_candidates": int(dev["pool_entries"]),
                              "fallback_queries": int(dev["fallback_queries"]), "plan": int(dev["last_plan"])}}
        return {"items": {"k": res._k, "ids": [int(x) for x in res.ids()], "scores": [float(x) for x in res.distances()], "index": res.index_mode()},
                "profile": profile}

    def batch_search(self, vectors, k: Optional[int] = None, where_expr: Optional[str] = None,
                     nprobe: Optional[int] = None, subset=None) -> list:
        """Collection::batch_search (engine.rs:5352-5498): one shared filter, every query through
        `search_with_precomputed_filter`; the flushed rows of the whole batch are scanned in ONE pass on the GPU."""
        if where_expr:
            raise NotImplementedError("`where_expr` needs the field store (out of scope, SURVEY.md §2); pass the resolved "
                                      "row filter as subset=BitSet | row indices (search_with_precomputed_filter)")
        k = 10 if k is None else int(k)
        q = _f32(vectors, 2, "vectors")
        if q.shape[1] != self._dim:  # engine.rs:4707-4712 -> wrapped as RuntimeError (src/python/mod.rs:1190)
            raise RuntimeError(f"Dimension mismatch: expected {self._dim}, got {q.shape[1]}")
        from .shard_node import filter_tombstoned_limit, merge_row_results

        subset_rows = self._subset_rows(subset)
        tomb = np.fromiter(self._tombstone, dtype=np.uint64, count=len(self._tombstone))
        search_k = k if tomb.size == 0 else k + int(tomb.size)   # engine.rs:4735-4747
        rows, dists, counts = self._base_search(q, search_k, int(nprobe or 0), subset_rows)
        out = []
        for i in range(q.shape[0]):
            c = int(counts[i])
            r_i, d_i = rows[i, :c].astype(np.uint64), dists[i, :c]
            if self._pending_rows:
                p_r, p_d = self._pending_search(q[i], search_k, subset_rows)
                r_i, d_i = merge_row_results(r_i, d_i, p_r, p_d, search_k, self._metric)   # engine.rs:4800-4813
            ids = self._user_ids(r_i).astype(np.uint64)
            ids, d_i = filter_tombstoned_limit(ids, d_i, tomb, k)                              # engine.rs:4819-4820
            out.append(SearchResult(ids.astype(np.int64), np.asarray(d_i, np.float32), self._index_mode, self._dim, k))
        return out


_METRIC_NAMES = {_lib.METRIC_IP: "ip", _lib.METRIC_L2: "l2", _lib.METRIC_COSINE: "cosine", _lib.METRIC_HAMMING: "hamming",
                 _lib.METRIC_JACCARD: "jaccard", _lib.METRIC_DICE: "dice", _lib.METRIC_TANIMOTO: "tanimoto"}


class DatabaseManager:
    """src/python/mod.rs:2235-2418 — in-memory registry (persistence is out of scope)."""

    def __init__(self, root: str):
        self.root = root
        self._dbs: dict = {}

    def create_database(self, name: str) -> None:
        self._dbs.setdefault(name, {})

    def require_collection(self, db: str, coll: str, dim: int) -> None:
        self._dbs.setdefault(db, {})
        if coll not in self._dbs[db]:
            self._dbs[db][coll] = Collection(coll, dim)

    def get_collection(self, db: str, coll: str, dim: int) -> Collection:
        self.require_collection(db, coll, dim)
        c = self._dbs[db][coll]
        if c._dim != dim:
            raise RuntimeError(f"Dimension mismatch: expected {c._dim}, got {dim}")
        return c
