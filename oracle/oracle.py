"""ctypes/numpy front-end of oracle/lynse_oracle.c — TEST INFRASTRUCTURE ONLY.

The C file restates the reference's algorithm (each function cites the
reference file:line); this module only marshals numpy arrays.  ``build()``
compiles the two shared objects with the committed Makefile.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

IP, L2, COS, HAMMING, JACCARD, DICE, TANIMOTO = range(7)
IPFORM_AUTO, IPFORM_SINGLE, IPFORM_BATCH8 = 0, 1, 2
METRIC_NAMES = {IP: "ip", L2: "l2", COS: "cosine", HAMMING: "hamming", JACCARD: "jaccard",
                DICE: "dice", TANIMOTO: "tanimoto"}

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)
_sz = C.c_size_t


def build(force: bool = False) -> None:
    """Compile liblynse_oracle{,_portable}.so (gcc; seconds)."""
    targets = [_HERE / "liblynse_oracle.so", _HERE / "liblynse_oracle_portable.so"]
    src_m = max((_HERE / n).stat().st_mtime for n in ("lynse_oracle.c", "lynse_oracle.h", "Makefile"))
    if not force and all(t.exists() and t.stat().st_mtime >= src_m for t in targets):
        return
    subprocess.run(["make", "-C", str(_HERE), "-B"], check=True, capture_output=True)


def _sig(lib, name, res, *args):
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = list(args)
    return fn


class Oracle:
    """One loaded oracle library (``portable=True`` → lane-emulation build)."""

    def __init__(self, portable: bool = False):
        build()
        name = "liblynse_oracle_portable.so" if portable else "liblynse_oracle.so"
        self.lib = lib = C.CDLL(str(_HERE / name))
        s = lambda n, r, *a: _sig(lib, n, r, *a)  # noqa: E731
        s("lo_has_avx2_fma", C.c_int)
        s("lo_metric_is_ascending", C.c_int, C.c_int)
        s("lo_metric_is_binary", C.c_int, C.c_int)
        s("lo_metric_from_str", C.c_int, C.c_char_p)
        s("lo_metric_from_index_mode", C.c_int, C.c_char_p)
        for n in ("lo_ip_single", "lo_ip_batch8_row", "lo_l2_single", "lo_cos_single", "lo_ip_scalar",
                  "lo_l2_scalar", "lo_cos_scalar", "lo_hamming_f32", "lo_jaccard_f32", "lo_dice_f32"):
            s(n, C.c_float, _f32p, _f32p, _sz)
        s("lo_ip_batch8", None, *([_f32p] * 9), _sz, _f32p)
        s("lo_pack_binary_f32", None, _f32p, _sz, _u64p)
        for n in ("lo_packed_hamming", "lo_packed_jaccard", "lo_packed_dice"):
            s(n, C.c_float, _u64p, _u64p, _sz)
        s("lo_compute_distance", C.c_float, _f32p, _f32p, _sz, C.c_int)
        s("lo_top_k_search", _sz, _f32p, _f32p, _sz, _sz, _sz, C.c_int, _u32p, _f32p)
        for n in ("lo_flat_search", "lo_flat_search_mt"):
            s(n, _sz, _f32p, _f32p, _sz, _sz, _sz, C.c_int, C.c_int, _u32p, _f32p)
        for n in ("lo_packed_binary_search", "lo_packed_binary_search_mt"):
            s(n, _sz, _u64p, _u64p, _sz, _sz, _sz, C.c_int, C.c_int, _u32p, _f32p)
        s("lo_pool_start", C.c_int, C.c_int)
        s("lo_pool_stop", None)
        s("lo_fill_uniform_mt", C.c_int, _f32p, _sz, _sz, C.c_uint64)
        s("lo_canonical_topk", _sz, _f32p, _f32p, _sz, _sz, _sz, C.c_int, C.c_int, _u32p, _f32p)
        s("lo_canonical_topk_packed", _sz, _u64p, _u64p, _sz, _sz, _sz, C.c_int, _u32p, _f32p)
        _u8p = C.POINTER(C.c_uint8)
        s("lo_sq8_fit", None, _f32p, _sz, _sz, _f32p, _f32p)
        s("lo_sq8_quantize", None, _f32p, _sz, _sz, _f32p, _f32p, _u8p)
        s("lo_sq8_search_canonical", _sz, _f32p, _f32p, _u8p, _f32p, _f32p, _sz, _sz, _sz, C.c_int, _u32p, _f32p)
        s("lo_distance_f16", C.c_float, _f32p, _f32p, _sz, C.c_int)
        s("lo_round_f16", None, _f32p, _sz, _f32p)
        s("lo_canonical_topk_f16", _sz, _f32p, _f32p, _sz, _sz, _sz, C.c_int, _u32p, _f32p)
        s("lo_flat_search_filtered", _sz, _f32p, _f32p, _sz, _sz, _sz, C.c_int, _u64p, _sz, C.c_int, _u32p, _f32p)
        s("lo_packed_search_filtered", _sz, _u64p, _u64p, _sz, _sz, _sz, C.c_int, _u64p, _sz, _u32p, _f32p)
        s("lo_canonical_topk_filtered", _sz, _f32p, _f32p, _u64p, _u64p, _sz, _sz, _sz, _sz, C.c_int, _u64p, _sz, _u32p, _f32p)
        s("lo_all_distances", None, _f32p, _f32p, _sz, _sz, C.c_int, C.c_int, _f32p)
        s("lo_merge_results", _sz, _u64p, _f32p, _sz, _sz, C.c_int, _u64p, _f32p)
        s("lo_kmeans_train", _sz, _f32p, _sz, _sz, _sz, _sz, C.c_int, _f32p, _u32p)
        s("lo_kmeans_train_sharded", _sz, _f32p, _sz, _sz, _sz, _sz, C.c_int, _sz, _f32p, _u32p)
        s("lo_kmeans_assign", None, _f32p, _sz, _sz, _f32p, _sz, C.c_int, _u32p)
        s("lo_fastrng_stream", None, C.c_uint64, _sz, _f64p)
        s("lo_ivf_search", _sz, _f32p, _f32p, _u64p, _sz, _sz, _sz, _f32p, _sz, _u64p, _u32p, _sz,
          _sz, C.c_int, _u64p, _f32p, _u32p)
        s("lo_ivf_search_filtered", _sz, _f32p, _f32p, _u64p, _sz, _sz, _sz, _f32p, _sz, _u64p, _u32p, _sz,
          _sz, C.c_int, _u64p, _sz, _u64p, _f32p)
        s("lo_binary_fit", C.c_int, _f32p, _sz, _sz, _f32p)
        s("lo_binary_quantize", None, _f32p, _sz, _sz, _f32p, _f32p)
        s("lo_ivf_flat_layout", None, _u32p, _sz, _sz, _u64p, _u32p)
        s("lo_ivf_routing_dims", _sz, _f32p, _sz, _sz, _u32p)
        s("lo_ivf_flat_probe", _sz, _f32p, _f32p, _sz, _sz, _sz, C.c_int, _u32p, _sz, _u32p)
        s("lo_ivf_flat_search", _sz, _f32p, _f32p, _sz, _sz, _f32p, _sz, _u64p, _u32p, _u32p, _sz,
          _sz, _sz, C.c_int, _u32p, _f32p)

    # ---- helpers
    @staticmethod
    def _f(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return a, a.ctypes.data_as(_f32p)

    @staticmethod
    def _u64(a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return a, a.ctypes.data_as(_u64p)

    @staticmethod
    def _u32(a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        return a, a.ctypes.data_as(_u32p)

    # ---- metrics
    def metric_from_str(self, s: str) -> int:
        return self.lib.lo_metric_from_str(s.encode())

    def metric_from_index_mode(self, s: str) -> int:
        return self.lib.lo_metric_from_index_mode(s.encode())

    def is_ascending(self, m: int) -> bool:
        return bool(self.lib.lo_metric_is_ascending(m))

    def is_binary(self, m: int) -> bool:
        return bool(self.lib.lo_metric_is_binary(m))

    # ---- distances
    def _pair(self, name, a, b):
        a, pa = self._f(a)
        b, pb = self._f(b)
        assert a.shape == b.shape and a.ndim == 1
        return float(getattr(self.lib, name)(pa, pb, a.size))

    def ip_single(self, a, b): return self._pair("lo_ip_single", a, b)
    def ip_batch8_row(self, a, b): return self._pair("lo_ip_batch8_row", a, b)
    def l2_single(self, a, b): return self._pair("lo_l2_single", a, b)
    def cos_single(self, a, b): return self._pair("lo_cos_single", a, b)
    def ip_scalar(self, a, b): return self._pair("lo_ip_scalar", a, b)
    def l2_scalar(self, a, b): return self._pair("lo_l2_scalar", a, b)
    def cos_scalar(self, a, b): return self._pair("lo_cos_scalar", a, b)

    def ip_batch8(self, q, rows8):
        q, pq = self._f(q)
        rows8 = np.ascontiguousarray(rows8, dtype=np.float32)
        assert rows8.shape == (8, q.size)
        out = np.zeros(8, np.float32)
        ptrs = [rows8[i].ctypes.data_as(_f32p) for i in range(8)]
        self.lib.lo_ip_batch8(pq, *ptrs, q.size, out.ctypes.data_as(_f32p))
        return out

    def compute_distance(self, a, b, metric: int) -> float:
        a, pa = self._f(a)
        b, pb = self._f(b)
        assert a.shape == b.shape
        return float(self.lib.lo_compute_distance(pa, pb, a.size, metric))

    def pack_binary(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        one = rows.ndim == 1
        r2 = rows.reshape(1, -1) if one else rows
        n, dim = r2.shape
        w = (dim + 63) // 64
        out = np.zeros((n, w), np.uint64)
        for i in range(n):
            self.lib.lo_pack_binary_f32(r2[i].ctypes.data_as(_f32p), dim, out[i].ctypes.data_as(_u64p))
        return out[0] if one else out

    def packed_distance(self, a, b, metric: int) -> float:
        a, pa = self._u64(a)
        b, pb = self._u64(b)
        fn = {HAMMING: "lo_packed_hamming", JACCARD: "lo_packed_jaccard", TANIMOTO: "lo_packed_jaccard",
              DICE: "lo_packed_dice"}[metric]
        return float(getattr(self.lib, fn)(pa, pb, a.size))

    # ---- top-k
    def _topk_call(self, fn, k, *args):
        idx = np.zeros(max(k, 1), np.uint32)
        dist = np.zeros(max(k, 1), np.float32)
        cnt = fn(*args, idx.ctypes.data_as(_u32p), dist.ctypes.data_as(_f32p))
        return idx[:cnt].copy(), dist[:cnt].copy()

    def top_k_search(self, query, cands, k, metric):
        q, pq = self._f(query)
        c, pc = self._f(cands)
        n, dim = (c.shape if c.ndim == 2 else (0, q.size))
        return self._topk_call(self.lib.lo_top_k_search, k, pq, pc, dim, n, k, metric)

    def flat_search(self, query, cands, k, metric, n_threads=8, mt=False):
        q, pq = self._f(query)
        c, pc = self._f(cands)
        n, dim = (c.shape if c.ndim == 2 else (0, q.size))
        fn = self.lib.lo_flat_search_mt if mt else self.lib.lo_flat_search
        return self._topk_call(fn, k, pq, pc, dim, n, k, metric, n_threads)

    # -- persistent pool of the timed CPU baseline (rayon's global pool) ---------------------------
    def pool_start(self, n_threads: int) -> int:
        return int(self.lib.lo_pool_start(int(n_threads)))

    def pool_stop(self) -> None:
        self.lib.lo_pool_stop()

    def fill_uniform_mt(self, rows: int, dim: int, seed: int) -> np.ndarray:
        """uniform [0,1) f32 rows, every page first-touched by the pool worker that will scan it."""
        a = np.empty((rows, dim), np.float32)
        if self.lib.lo_fill_uniform_mt(a.ctypes.data_as(_f32p), rows, dim, seed) != 0:
            raise RuntimeError("lo_fill_uniform_mt needs a running pool (pool_start)")
        return a

    def packed_binary_search(self, query_words, rows_words, k, metric, n_threads=8, mt=False):
        q, pq = self._u64(query_words)
        r, pr = self._u64(rows_words)
        n, w = r.shape
        fn = self.lib.lo_packed_binary_search_mt if mt else self.lib.lo_packed_binary_search
        return self._topk_call(fn, k, pq, pr, w, n, k, metric, n_threads)

    def canonical_topk(self, query, cands, k, metric, ip_form=IPFORM_AUTO):
        q, pq = self._f(query)
        c, pc = self._f(cands)
        n, dim = (c.shape if c.ndim == 2 else (0, q.size))
        return self._topk_call(self.lib.lo_canonical_topk, k, pq, pc, dim, n, k, metric, ip_form)

    def canonical_topk_packed(self, query_words, rows_words, k, metric):
        q, pq = self._u64(query_words)
        r, pr = self._u64(rows_words)
        n, w = r.shape
        return self._topk_call(self.lib.lo_canonical_topk_packed, k, pq, pr, w, n, k, metric)

    def sq8_fit(self, data):
        """SQ8Data::from_f32_parallel -> (mins f32[dim], scales f32[dim], codes u8[n, dim])."""
        d, pd = self._f(data)
        n, dim = d.shape
        mins, scales = np.zeros(dim, np.float32), np.zeros(dim, np.float32)
        self.lib.lo_sq8_fit(pd, n, dim, mins.ctypes.data_as(_f32p), scales.ctypes.data_as(_f32p))
        codes = np.zeros((n, dim), np.uint8)
        self.lib.lo_sq8_quantize(pd, n, dim, mins.ctypes.data_as(_f32p), scales.ctypes.data_as(_f32p),
                                 codes.ctypes.data_as(C.POINTER(C.c_uint8)))
        return mins, scales, codes

    def sq8_search(self, query, data, mins, scales, codes, k, metric):
        """sq8_two_pass_search, canonical tie handling (see lo_sq8_search_canonical)."""
        q, pq = self._f(query)
        d, pd = self._f(data)
        n, dim = d.shape
        c = np.ascontiguousarray(codes, np.uint8)
        return self._topk_call(self.lib.lo_sq8_search_canonical, k, pq, pd, c.ctypes.data_as(C.POINTER(C.c_uint8)),
                               mins.ctypes.data_as(_f32p), scales.ctypes.data_as(_f32p), dim, n, k, metric)

    def round_f16(self, a):
        """f32 -> f16 -> f32 (RNE): what VectorDtype::F16 storage keeps of a row."""
        x, px = self._f(np.asarray(a, np.float32).reshape(-1))
        out = np.zeros_like(x)
        self.lib.lo_round_f16(px, x.size, out.ctypes.data_as(_f32p))
        return out.reshape(np.asarray(a).shape)

    def distance_f16(self, query, cand_decoded, metric):
        q, pq = self._f(query)
        c, pc = self._f(cand_decoded)
        return float(self.lib.lo_distance_f16(pq, pc, q.size, metric))

    def canonical_topk_f16(self, query, cands_decoded, k, metric):
        """Search over f16-stored rows: sequential-sum kernels (simd.rs:805-846), canonical (distance, row) order."""
        q, pq = self._f(query)
        c, pc = self._f(cands_decoded)
        n, dim = c.shape
        return self._topk_call(self.lib.lo_canonical_topk_f16, k, pq, pc, dim, n, k, metric)

    def flat_search_filtered(self, query, cands, k, metric, subset, n_threads=8):
        """FlatMmap::search_filtered on f32 rows, the reference's policy (subset order / chunk order)."""
        q, pq = self._f(query)
        c, pc = self._f(cands)
        n, dim = c.shape
        sub, ps = self._u64(np.asarray(subset, np.uint64).reshape(-1))
        return self._topk_call(self.lib.lo_flat_search_filtered, k, pq, pc, dim, n, k, metric, ps, sub.size, n_threads)

    def packed_search_filtered(self, query_words, rows_words, k, metric, subset):
        q, pq = self._u64(query_words)
        r, pr = self._u64(rows_words)
        n, w = r.shape
        sub, ps = self._u64(np.asarray(subset, np.uint64).reshape(-1))
        return self._topk_call(self.lib.lo_packed_search_filtered, k, pq, pr, w, n, k, metric, ps, sub.size)

    def canonical_topk_filtered(self, query, cands, k, metric, subset, packed_query=None, packed_rows=None):
        """The subset as a set of valid rows, exact single-row-kernel distances, (distance, row) order."""
        sub, ps = self._u64(np.asarray(subset, np.uint64).reshape(-1))
        if packed_rows is not None:
            pqw, ppq = self._u64(packed_query)
            pr, ppr = self._u64(packed_rows)
            n, w = pr.shape
            return self._topk_call(self.lib.lo_canonical_topk_filtered, k, None, None, ppq, ppr, w, 0, n, k, metric, ps, sub.size)
        q, pq = self._f(query)
        c, pc = self._f(cands)
        n, dim = c.shape
        return self._topk_call(self.lib.lo_canonical_topk_filtered, k, pq, pc, None, None, 0, dim, n, k, metric, ps, sub.size)

    def all_distances(self, query, cands, metric, ip_form=IPFORM_AUTO):
        q, pq = self._f(query)
        c, pc = self._f(cands)
        n, dim = c.shape
        out = np.zeros(n, np.float32)
        self.lib.lo_all_distances(pq, pc, dim, n, metric, ip_form, out.ctypes.data_as(_f32p))
        return out

    def merge_results(self, ids, dists, k, metric):
        ids, pi = self._u64(ids)
        d, pd = self._f(dists)
        oi = np.zeros(max(k, 1), np.uint64)
        od = np.zeros(max(k, 1), np.float32)
        cnt = self.lib.lo_merge_results(pi, pd, ids.size, k, metric, oi.ctypes.data_as(_u64p),
                                        od.ctypes.data_as(_f32p))
        return oi[:cnt].copy(), od[:cnt].copy()

    # ---- k-means / IVF
    def fastrng_stream(self, seed, count):
        out = np.zeros(count, np.float64)
        self.lib.lo_fastrng_stream(seed, count, out.ctypes.data_as(_f64p))
        return out

    def kmeans_train(self, data, requested, max_iter, metric):
        d, pd = self._f(data)
        n, dim = d.shape
        cen = np.zeros((requested, dim), np.float32)
        asg = np.zeros(n, np.uint32)
        k = self.lib.lo_kmeans_train(pd, n, dim, requested, max_iter, metric,
                                     cen.ctypes.data_as(_f32p), asg.ctypes.data_as(_u32p))
        return cen[:k].copy(), asg

    def kmeans_train_sharded(self, data, requested, max_iter, metric, world):
        d, pd_ = self._f(data)
        n, dim = d.shape
        k = min(requested, n)
        cen = np.zeros((k, dim), np.float32)
        asg = np.zeros(n, np.uint32)
        got = self.lib.lo_kmeans_train_sharded(pd_, n, dim, requested, max_iter, metric, world, cen.ctypes.data_as(_f32p), asg.ctypes.data_as(_u32p))
        return cen[:got], asg

    def kmeans_assign(self, data, centroids, metric):
        d, pd = self._f(data)
        c, pc = self._f(centroids)
        asg = np.zeros(d.shape[0], np.uint32)
        self.lib.lo_kmeans_assign(pd, d.shape[0], d.shape[1], pc, c.shape[0], metric,
                                  asg.ctypes.data_as(_u32p))
        return asg

    @staticmethod
    def lists_from_assignments(assignments, nlist):
        """kmeans.rs:317-345 as CSR: rows ascending inside each list."""
        a = np.asarray(assignments, dtype=np.int64)
        order = np.argsort(a, kind="stable").astype(np.uint32)
        counts = np.bincount(a, minlength=nlist).astype(np.uint64)
        offsets = np.zeros(nlist + 1, np.uint64)
        offsets[1:] = np.cumsum(counts)
        return offsets, order

    def ivf_search(self, query, data, centroids, list_offsets, list_rows, nprobe, k, metric,
                   packed=None):
        q, pq = self._f(query)
        d, pd = self._f(data)
        c, pc = self._f(centroids)
        lo, plo = self._u64(list_offsets)
        lr, plr = self._u32(list_rows)
        n, dim = d.shape
        if packed is not None:
            pk, ppk = self._u64(packed)
            words = pk.shape[1]
        else:
            ppk, words = None, 0
        ids = np.zeros(max(k, 1), np.uint64)
        dist = np.zeros(max(k, 1), np.float32)
        probed = np.zeros(max(nprobe, 1), np.uint32)
        cnt = self.lib.lo_ivf_search(pq, pd, ppk, words, dim, n, pc, c.shape[0], plo, plr, nprobe, k,
                                     metric, ids.ctypes.data_as(_u64p), dist.ctypes.data_as(_f32p),
                                     probed.ctypes.data_as(_u32p))
        return ids[:cnt].copy(), dist[:cnt].copy(), probed[:min(nprobe, c.shape[0])].copy()

    def ivf_search_filtered(self, query, data, centroids, list_offsets, list_rows, nprobe, k, metric, subset,
                            packed=None):
        """IVFIndex::search with SearchParams.subset (ivf.rs:251-265)."""
        q, pq = self._f(query)
        d, pd = self._f(data)
        c, pc = self._f(centroids)
        lo, plo = self._u64(list_offsets)
        lr, plr = self._u32(list_rows)
        sub, ps = self._u64(np.asarray(subset, np.uint64).reshape(-1))
        n, dim = d.shape
        if packed is not None:
            pk, ppk = self._u64(packed)
            words = pk.shape[1]
        else:
            ppk, words = None, 0
        ids = np.zeros(max(k, 1), np.uint64)
        dist = np.zeros(max(k, 1), np.float32)
        cnt = self.lib.lo_ivf_search_filtered(pq, pd, ppk, words, dim, n, pc, c.shape[0], plo, plr, nprobe, k, metric,
                                              ps, sub.size, ids.ctypes.data_as(_u64p), dist.ctypes.data_as(_f32p))
        return ids[:cnt].copy(), dist[:cnt].copy()

    def binary_fit(self, data):
        """BinaryQuantizer::fit -> (already_binary, thresholds[dim])."""
        d, pd = self._f(data)
        thr = np.zeros(d.shape[1], np.float32)
        ab = self.lib.lo_binary_fit(pd, d.shape[0], d.shape[1], thr.ctypes.data_as(_f32p))
        return bool(ab), thr

    def binary_quantize(self, data, thresholds):
        d, pd = self._f(np.atleast_2d(data))
        t, pt = self._f(thresholds)
        out = np.zeros_like(d)
        self.lib.lo_binary_quantize(pd, d.shape[0], d.shape[1], pt, out.ctypes.data_as(_f32p))
        return out

    def ivf_flat_layout(self, assignments, nlist):
        a, pa = self._u32(assignments)
        offsets = np.zeros(nlist + 1, np.uint64)
        orig = np.zeros(a.size, np.uint32)
        self.lib.lo_ivf_flat_layout(pa, a.size, nlist, offsets.ctypes.data_as(_u64p),
                                    orig.ctypes.data_as(_u32p))
        return offsets, orig

    def ivf_routing_dims(self, centroids):
        c, pc = self._f(centroids)
        out = np.zeros(16, np.uint32)
        cnt = self.lib.lo_ivf_routing_dims(pc, c.shape[1], c.shape[0], out.ctypes.data_as(_u32p))
        return out[:cnt].copy()

    def ivf_flat_probe(self, query, centroids, nprobe, metric, routing_dims=None):
        q, pq = self._f(query)
        c, pc = self._f(centroids)
        rd = np.zeros(0, np.uint32) if routing_dims is None else np.ascontiguousarray(routing_dims, np.uint32)
        out = np.zeros(c.shape[0], np.uint32)
        cnt = self.lib.lo_ivf_flat_probe(pq, pc, c.shape[1], c.shape[0], nprobe, metric,
                                         rd.ctypes.data_as(_u32p), rd.size, out.ctypes.data_as(_u32p))
        return out[:cnt].copy()

    def ivf_flat_search(self, query, slab_data, centroids, offsets, original_ids, nprobe, k, metric,
                        routing_dims=None):
        q, pq = self._f(query)
        d, pd = self._f(slab_data)
        c, pc = self._f(centroids)
        o, po = self._u64(offsets)
        oi, poi = self._u32(original_ids)
        rd = np.zeros(0, np.uint32) if routing_dims is None else np.ascontiguousarray(routing_dims, np.uint32)
        ids = np.zeros(max(k, 1), np.uint32)
        dist = np.zeros(max(k, 1), np.float32)
        cnt = self.lib.lo_ivf_flat_search(pq, pd, d.shape[1], d.shape[0], pc, c.shape[0], po, poi,
                                          rd.ctypes.data_as(_u32p), rd.size, nprobe, k, metric,
                                          ids.ctypes.data_as(_u32p), dist.ctypes.data_as(_f32p))
        return ids[:cnt].copy(), dist[:cnt].copy()


_default = None


def get(portable: bool = False) -> Oracle:
    global _default
    if portable:
        return Oracle(portable=True)
    if _default is None:
        _default = Oracle()
    return _default


def host_threads() -> int:
    return os.cpu_count() or 1
