"""Shard-node glue around a search — thin ctypes wrappers over the host-only entry points of
`liblynse_hip.so` (csrc/shard_host.inc): tombstone filtering, flushed + pending row merge, and the binary result
block a shard returns to the coordinator.  Pending (un-flushed) rows are scored with `py_top_k_search`
(Collection::pending_search, src/engine.rs:3310-3361)."""
from __future__ import annotations

import ctypes as C
import json
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import check, lib


def _u64(a):
    return np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.uint64)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a.size else None


def filter_tombstoned_limit(ids, dists, tombstones, limit: int) -> Tuple[np.ndarray, np.ndarray]:
    """Collection::filter_tombstoned_limit (src/engine.rs:3286-3308)."""
    i, d, t = _u64(ids), _f32(dists), _u64(tombstones)
    if i.size != d.size:
        raise ValueError("ids and distances differ in length")
    cap = max(min(i.size, int(limit)), 1)
    oi, od, n = np.empty(cap, np.uint64), np.empty(cap, np.float32), C.c_uint64(0)
    check(lib.lynse_hip_filter_tombstoned_limit(_p(i), _p(d), i.size, _p(t), t.size, int(limit), _p(oi), _p(od), C.byref(n)))
    return oi[:n.value].copy(), od[:n.value].copy()


def merge_row_results(left_ids, left_dists, right_ids, right_dists, limit: int, metric) -> Tuple[np.ndarray, np.ndarray]:
    """Collection::merge_row_results (src/engine.rs:3363-3418)."""
    from .core import metric_from_str

    m = metric if isinstance(metric, int) else metric_from_str(metric)
    li, ld, ri, rd = _u64(left_ids), _f32(left_dists), _u64(right_ids), _f32(right_dists)
    cap = max(li.size + ri.size, 1)
    oi, od, n = np.empty(cap, np.uint64), np.empty(cap, np.float32), C.c_uint64(0)
    check(lib.lynse_hip_merge_row_results(_p(li), _p(ld), li.size, _p(ri), _p(rd), ri.size, int(limit), m, _p(oi), _p(od), C.byref(n)))
    return oi[:n.value].copy(), od[:n.value].copy()


def encode_search_result(ids, dists, fields: Optional[Sequence[dict]] = None) -> bytes:
    """encode_search_result_binary (src/rpc.rs:1156-1177).  `fields` is serialised like serde_json::to_vec (compact)."""
    i, d = _u64(ids), _f32(dists)
    if i.size != d.size:
        raise ValueError("ids and distances differ in length")
    fj = np.frombuffer(json.dumps(list(fields), separators=(",", ":")).encode(), np.uint8) if fields else np.zeros(0, np.uint8)
    need = C.c_uint64(0)
    buf = np.empty(4 + i.size * 12 + 4 + fj.size, np.uint8)
    check(lib.lynse_hip_encode_search_result(_p(i), _p(d), i.size, _p(fj), fj.size, _p(buf), buf.size, C.byref(need)))
    return buf[:need.value].tobytes()


def decode_search_result(frame: bytes, offset: int = 0):
    """decode_search_result_binary (src/cluster.rs:404-435) -> (ids, distances, fields, next_offset)."""
    b = np.frombuffer(frame, np.uint8)
    n, fo, fl, nxt = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
    cap = max((len(frame) - offset) // 12, 1)
    oi, od = np.empty(cap, np.uint64), np.empty(cap, np.float32)
    check(lib.lynse_hip_decode_search_result(_p(b) if b.size else None, b.size, int(offset), _p(oi), _p(od), cap, C.byref(n), C.byref(fo),
                                             C.byref(fl), C.byref(nxt)))
    fields = json.loads(frame[fo.value:fo.value + fl.value]) if fl.value else []
    return oi[:n.value].copy(), od[:n.value].copy(), fields, int(nxt.value)


def encode_batch(results: Sequence[Tuple[np.ndarray, np.ndarray]]) -> bytes:
    """handle_batch_search frame (src/rpc.rs:619-659): [u32 n_results] then one block per query."""
    out = [np.array([len(results)], "<u4").tobytes()]
    out += [encode_search_result(i, d) for i, d in results]
    return b"".join(out)


def decode_batch(frame: bytes) -> List[Tuple[np.ndarray, np.ndarray]]:
    """src/cluster.rs:140-156."""
    n = int(np.frombuffer(frame[:4], "<u4")[0])
    off, res = 4, []
    for _ in range(n):
        i, d, _f, off = decode_search_result(frame, off)
        res.append((i, d))
    return res
