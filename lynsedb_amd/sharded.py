"""Row-sharded FLAT search across the GPUs of one node — one process per GPU.

Replaces the reference's only parallelism/communication strategy, the TCP scatter-gather of
`src/cluster.rs` (fan-out to every shard :173-217, `merge_search_blocks` :327-393) and the Python
`ClusterCoordinator._merge_pairs` (python/lynse/cluster.py:535-556), with:

    per-rank scan of the local shard (liblynse_hip.so)  ->  ONE RCCL all-gather over xGMI of the
    fixed-size per-rank result block [rows u64 | dists f32 | counts u32] (B*k*12 + B*4 bytes)
    ->  k-way merge on the device in the canonical (distance, row ascending) order.

Global row g lives on rank g % world (local row l <-> global l*world + rank), which keeps local row
order monotone in the global id — the tie-break order is therefore identical to a single shard.
`torch.distributed` is used for the process group only (backend "nccl" = RCCL on ROCm, "gloo" in the
CPU tests); no tensor math of the search path runs in torch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, lib
from .core import FlatIndex


def shard_of_row(global_row: int, world: int) -> int:
    return global_row % world


BUCKET_COUNT = 4096  # python/lynse/cluster.py:1273 (ShardMap default)


def hash_u64(value: str) -> int:
    """`_hash_u64` of the reference's cluster router (python/lynse/cluster.py:156-158): blake2b, 8-byte digest read as a little-endian u64."""
    import hashlib

    return int.from_bytes(hashlib.blake2b(value.encode("utf-8"), digest_size=8).digest(), "little", signed=False)


def bucket_of_id(database: str, collection: str, item_id, bucket_count: int = BUCKET_COUNT) -> int:
    """The reference's partition rule for an item: blake2b64("{db}/{coll}/{id}") % bucket_count (cluster.py:1364-1370)."""
    return hash_u64(f"{database}/{collection}/{item_id}") % bucket_count


def shard_of_id(database: str, collection: str, item_id, world: int, bucket_count: int = BUCKET_COUNT) -> int:
    """bucket -> shard group: `bucket_to_group[b] = groups[b % n_groups]` (cluster.py:1273) — here group g = rank g."""
    return bucket_of_id(database, collection, item_id, bucket_count) % world


class NativeComm:
    """RCCL communicator behind the C-ABI (include/lynse_hip.h, multi-GPU section).  `dist` (torch.distributed) is only the
    launcher: it carries rank 0's 128-byte unique id to the other ranks."""

    def __init__(self, dist, rank: int, world: int, device: int):
        import os

        import torch

        self._c = C.c_void_p()
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")  # ONE RCCL per process: torch's copy
        # step 1 (local, may fail): load RCCL; rank 0 draws the unique id.  step 2 (collective, always runs): broadcast
        # [ok flag | 128-byte id] so that a failure on rank 0 cannot leave the other ranks waiting.  step 3: every rank creates
        # its communicator (ncclCommInitRank is itself a collective).
        msg = np.zeros(129, np.uint8)
        err = None
        try:
            check(lib.lynse_hip_comm_load_rccl(rccl.encode() if os.path.exists(rccl) else None))
            if rank == 0:
                check(lib.lynse_hip_comm_unique_id(msg[1:].ctypes.data_as(C.c_void_p)))
            msg[0] = 1
        except Exception as e:  # noqa: BLE001
            err = e
        if world > 1:
            backend = dist.get_backend()
            t = torch.from_numpy(msg.copy())
            t = t.to(torch.device("cuda", device)) if backend == "nccl" else t
            dist.broadcast(t, src=0)                      # rank 0's flag + id
            got = t.cpu().numpy()
            ok = torch.tensor([1 if (err is None and int(got[0]) == 1) else 0], dtype=torch.int32)
            ok = ok.to(torch.device("cuda", device)) if backend == "nccl" else ok
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)     # every rank loaded RCCL and rank 0 has an id
            if int(ok.item()) == 0:
                raise RuntimeError(f"RCCL bootstrap failed on some rank ({err!r})")
            uid = np.ascontiguousarray(got[1:])
        else:
            if err is not None:
                raise err
            uid = np.ascontiguousarray(msg[1:])
        check(lib.lynse_hip_comm_create(uid.ctypes.data_as(C.c_void_p), rank, world, device, C.byref(self._c)))
        self.rank, self.world, self.device = rank, world, device

    def ranks_seen(self) -> int:
        n = C.c_int(0)
        check(lib.lynse_hip_comm_ranks_seen(self._c, C.byref(n)))
        return n.value

    @property
    def handle(self):
        return self._c

    def close(self) -> None:
        c, self._c = getattr(self, "_c", None), None
        if c and lib is not None:  # (at interpreter shutdown the module globals may already be gone: the process exit frees the rest)
            lib.lynse_hip_comm_destroy(c)

    def __del__(self):
        self.close()


def block_layout(nq: int, k: int):
    """Byte layout of one rank's result block (the RCCL message; cf. the 12 B/candidate TCP block of
    src/rpc.rs:1156-1177)."""
    rows_off = 0
    dists_off = rows_off + nq * k * 8
    counts_off = dists_off + nq * k * 4
    total = counts_off + nq * 4
    total = (total + 15) // 16 * 16
    return rows_off, dists_off, counts_off, total


class ShardOutputs:
    def __init__(self, nq: int, k: int, world: int, device):
        import torch

        self.nq, self.k, self.world = nq, k, world
        self.rows_off, self.dists_off, self.counts_off, self.block_bytes = block_layout(nq, k)
        self.local = torch.zeros(self.block_bytes, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros(self.block_bytes * world, dtype=torch.uint8, device=device) if world > 1 else self.local
        self.rows = torch.zeros((nq, k), dtype=torch.int64, device=device)     # u64 bits
        self.dists = torch.zeros((nq, k), dtype=torch.float32, device=device)
        self.counts = torch.zeros(nq, dtype=torch.int32, device=device)        # u32 bits

    def local_ptrs(self):
        base = self.local.data_ptr()
        return base + self.rows_off, base + self.dists_off, base + self.counts_off


class ShardedFlat:
    def __init__(self, dim: int, rank: int = 0, world: int = 1, device: Optional[int] = None, group=None):
        self.dim, self.rank, self.world = dim, rank, world
        self.dist = group  # the torch.distributed module (or None when world == 1)
        self.index = FlatIndex(None, dim, device)
        self.index.set_row_map(world, rank)
        self.comm: Optional[NativeComm] = None   # the exchange behind the C-ABI (enable_native_comm)
        self.comm_error: Optional[str] = None
        self.ranks_seen: Optional[int] = None

    def enable_native_comm(self) -> bool:
        """Create the RCCL communicator inside the library (scan -> ncclAllGather -> merge on one stream, no torch in the
        data path).  Returns False — and keeps the torch.distributed exchange — when RCCL cannot be set up; the reason is
        kept in `comm_error`.  Collective: every rank calls it."""
        try:
            comm = NativeComm(self.dist, self.rank, self.world, self.index_device())
            seen = comm.ranks_seen()   # collective self-check (all-reduce of 1 per rank)
            self.ranks_seen = seen
            if seen != self.world:
                raise RuntimeError(f"communicator self-check saw {seen} of {self.world} ranks")
            self.comm = comm
        except Exception as e:  # noqa: BLE001  (device errors, missing RCCL)
            self.comm, self.comm_error = None, f"{type(e).__name__}: {e}"
        if self.world > 1 and self.dist is not None:  # agree: the exchange is a collective, all ranks take the same path
            import torch

            ok = torch.tensor([1 if self.comm is not None else 0], dtype=torch.int32)
            if self.dist.get_backend() == "nccl":
                ok = ok.to(torch.device("cuda", self.index_device()))
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                self.comm = None
        return self.comm is not None

    # -- data ------------------------------------------------------------------------------------
    def add_global_rows(self, data: np.ndarray, first_global_row: int = 0) -> None:
        """Append this rank's share of `data` (global rows first_global_row ...)."""
        first = (self.rank - first_global_row) % self.world
        mine = np.ascontiguousarray(data[first::self.world])
        if mine.shape[0]:
            self.index.write(mine)

    def add_items_hashed(self, data: np.ndarray, ids, database: str, collection: str) -> np.ndarray:
        """The reference's placement rule instead of row % world: item `id` lives on the shard of its bucket,
        blake2b64("{db}/{coll}/{id}") % 4096 -> bucket % world (python/lynse/cluster.py:156-158, :1273, :1364-1370).
        Appends this rank's items and returns their ids in local row order (the caller's local-row -> id map; the
        exchange then carries ids, not rows: set the row map to identity and translate after the merge)."""
        ids = np.asarray(ids)
        mine = np.fromiter((shard_of_id(database, collection, int(i), self.world) == self.rank for i in ids), bool, len(ids))
        if mine.any():
            self.index.write(np.ascontiguousarray(np.asarray(data, np.float32)[mine]))
        return ids[mine]

    def alloc_outputs(self, nq: int, k: int) -> ShardOutputs:
        import torch

        return ShardOutputs(nq, k, self.world, torch.device("cuda", self.index_device()))

    def index_device(self) -> int:
        return int(lib.lynse_hip_flat_device(self.index.handle))

    # -- search ----------------------------------------------------------------------------------
    def search_device(self, d_queries, k: int, metric: int, out: ShardOutputs) -> None:
        """Whole-collection search; results (identical on every rank) land in out.rows/dists/counts."""
        import torch

        nq = d_queries.shape[0]
        torch.cuda.current_stream().synchronize()  # inputs must be complete: the library uses its own stream
        if self.world == 1:  # single shard: results are already globally merged and ordered
            check(lib.lynse_hip_flat_search_f32_device(
                self.index.handle, C.c_void_p(d_queries.data_ptr()), nq, k, metric, C.c_void_p(out.rows.data_ptr()),
                C.c_void_p(out.dists.data_ptr()), C.c_void_p(out.counts.data_ptr()), None))
            return
        if self.comm is not None:  # scan -> ncclAllGather -> k_merge inside the library, one stream
            check(lib.lynse_hip_flat_search_sharded_f32_device(
                self.index.handle, self.comm.handle, C.c_void_p(d_queries.data_ptr()), nq, k, metric,
                C.c_void_p(out.rows.data_ptr()), C.c_void_p(out.dists.data_ptr()), C.c_void_p(out.counts.data_ptr())))
            return
        pr, pd, pc = out.local_ptrs()
        check(lib.lynse_hip_flat_search_f32_device(self.index.handle, C.c_void_p(d_queries.data_ptr()), nq, k,
                                                   metric, C.c_void_p(pr), C.c_void_p(pd), C.c_void_p(pc), None))
        self.dist.all_gather_into_tensor(out.gathered, out.local)
        stream = torch.cuda.current_stream().cuda_stream
        check(lib.lynse_hip_merge_topk_device(C.c_void_p(out.gathered.data_ptr()), out.block_bytes, out.rows_off,
                                              out.dists_off, out.counts_off, self.world, nq, k, metric,
                                              C.c_void_p(out.rows.data_ptr()), C.c_void_p(out.dists.data_ptr()),
                                              C.c_void_p(out.counts.data_ptr()), C.c_void_p(stream)))

    def search_packed_device(self, d_qwords, k: int, metric: int, out: ShardOutputs) -> None:
        """`search_device` for the packed-binary metrics: queries are packed u64 words (an int64 torch tensor [nq, words])."""
        import torch

        nq = d_qwords.shape[0]
        torch.cuda.current_stream().synchronize()
        if self.world == 1:
            self.index.search_packed_device(d_qwords, k, metric, out.rows, out.dists, out.counts)
            return
        if self.comm is not None:
            check(lib.lynse_hip_flat_search_sharded_packed_u64_device(
                self.index.handle, self.comm.handle, C.c_void_p(d_qwords.data_ptr()), nq, k, metric,
                C.c_void_p(out.rows.data_ptr()), C.c_void_p(out.dists.data_ptr()), C.c_void_p(out.counts.data_ptr())))
            return
        pr, pd, pc = out.local_ptrs()
        check(lib.lynse_hip_flat_search_packed_u64_device(self.index.handle, C.c_void_p(d_qwords.data_ptr()), nq, k, metric,
                                                          C.c_void_p(pr), C.c_void_p(pd), C.c_void_p(pc), None))
        self.dist.all_gather_into_tensor(out.gathered, out.local)
        stream = torch.cuda.current_stream().cuda_stream
        check(lib.lynse_hip_merge_topk_device(C.c_void_p(out.gathered.data_ptr()), out.block_bytes, out.rows_off,
                                              out.dists_off, out.counts_off, self.world, nq, k, metric,
                                              C.c_void_p(out.rows.data_ptr()), C.c_void_p(out.dists.data_ptr()),
                                              C.c_void_p(out.counts.data_ptr()), C.c_void_p(stream)))

    def search_submit(self, d_queries, k: int, metric: int, out: ShardOutputs):
        """One batch IN FLIGHT (lynse_hip_flat_search_submit_*): scan -> ncclAllGather -> merge enqueued on a search context of
        the shard; returns a ticket whose wait() makes out.rows / dists / counts final.  A collective with world > 1 (every
        rank submits and waits for the same sequence).  Needs the native communicator (or world == 1); without it the
        batch is answered here and the ticket is already complete."""
        class _Done:
            def wait(self):
                return None

        if metric >= _lib.METRIC_HAMMING and d_queries.is_floating_point():
            self.search_device(d_queries, k, metric, out)   # float queries of a binary metric: packed inside the blocking call
            return _Done()
        if self.world == 1 or self.comm is not None:  # (a 1-rank communicator still runs the exchange half: status word, merge)
            return self.index.search_submit(d_queries, k, metric, out.rows, out.dists, out.counts,
                                            comm=self.comm.handle if self.comm is not None else None)
        if d_queries.is_floating_point():
            self.search_device(d_queries, k, metric, out)
        else:
            self.search_packed_device(d_queries, k, metric, out)
        return _Done()

    def search(self, queries: np.ndarray, k: int, metric: int):
        """Host-array convenience wrapper around search_device."""
        import torch

        dev = torch.device("cuda", self.index_device())
        q = torch.as_tensor(np.ascontiguousarray(queries, dtype=np.float32), device=dev)
        out = self.alloc_outputs(q.shape[0], k)
        self.search_device(q, k, metric, out)
        torch.cuda.synchronize()
        return (out.rows.cpu().numpy().view(np.uint64), out.dists.cpu().numpy(),
                out.counts.cpu().numpy().view(np.uint32))

    # -- host-array variant of the exchange step (also the path the gloo CPU tests exercise) -------
    @staticmethod
    def pack_block(rows: np.ndarray, dists: np.ndarray, counts: np.ndarray) -> np.ndarray:
        nq, k = rows.shape
        ro, do, co, total = block_layout(nq, k)
        buf = np.zeros(total, np.uint8)
        buf[ro:ro + nq * k * 8] = np.ascontiguousarray(rows, np.uint64).view(np.uint8).ravel()
        buf[do:do + nq * k * 4] = np.ascontiguousarray(dists, np.float32).view(np.uint8).ravel()
        buf[co:co + nq * 4] = np.ascontiguousarray(counts, np.uint32).view(np.uint8).ravel()
        return buf

    @staticmethod
    def unpack_blocks(gathered: np.ndarray, world: int, nq: int, k: int):
        ro, do, co, total = block_layout(nq, k)
        g = gathered.reshape(world, total)
        rows = np.stack([g[r, ro:ro + nq * k * 8].view(np.uint64).reshape(nq, k) for r in range(world)])
        dists = np.stack([g[r, do:do + nq * k * 4].view(np.float32).reshape(nq, k) for r in range(world)])
        counts = np.stack([g[r, co:co + nq * 4].view(np.uint32) for r in range(world)])
        return rows, dists, counts

    @staticmethod
    def allgather_merge_host(dist, world: int, rows: np.ndarray, dists: np.ndarray, counts: np.ndarray, k: int, metric: int):
        """all-gather the per-rank result blocks over `dist` (any backend) and merge on the host with
        lynse_hip_merge_topk — the same message layout and order as the device path."""
        import torch

        from .core import merge_topk

        nq = rows.shape[0]
        local = torch.from_numpy(ShardedFlat.pack_block(rows, dists, counts))
        if world > 1:
            parts = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(parts, local)
            gathered = torch.cat(parts).numpy()
        else:
            gathered = local.numpy()
        g_rows, g_dists, g_counts = ShardedFlat.unpack_blocks(gathered, world, nq, k)
        out_r = np.full((nq, k), np.iinfo(np.uint64).max, np.uint64)
        out_d = np.zeros((nq, k), np.float32)
        out_c = np.zeros(nq, np.uint32)
        for q in range(nq):
            i, d = merge_topk(g_rows[:, q, :], g_dists[:, q, :], g_counts[:, q], k, metric)
            out_r[q, :len(i)], out_d[q, :len(i)], out_c[q] = i, d, len(i)
        return out_r, out_d, out_c

    # -- verification against an independent torch fp32 computation (bench.py, outside timing) -----
    def verify_against_torch(self, d_queries, k: int, metric: int, out: ShardOutputs, nverify: int = 16) -> dict:
        import torch

        if metric >= _lib.METRIC_HAMMING:
            return {"skipped": "binary metric"}
        n_local = len(self.index)
        dev = d_queries.device
        q = d_queries[:nverify]
        asc = metric != _lib.METRIC_IP
        best_s = torch.full((nverify, k), float("inf") if asc else float("-inf"), device=dev)
        best_i = torch.full((nverify, k), -1, dtype=torch.int64, device=dev)
        chunk = 500_000
        buf = torch.empty((chunk, self.dim), dtype=torch.float32, device=dev)
        qn = (q * q).sum(1, keepdim=True)
        for r0 in range(0, n_local, chunk):
            nr = min(chunk, n_local - r0)
            # rows come back from the library's own HBM copy (device-to-device) on the LIBRARY's stream:
            # torch work still reading `buf` from the previous chunk must be finished first
            torch.cuda.synchronize()
            check(lib.lynse_hip_flat_copy_rows_device(self.index.handle, r0, nr, C.c_void_p(buf.data_ptr())))
            v = buf[:nr]
            s = q @ v.T
            if metric == _lib.METRIC_L2:
                s = qn + (v * v).sum(1)[None, :] - 2.0 * s
            elif metric == _lib.METRIC_COSINE:
                s = 1.0 - s / (qn.sqrt() * (v * v).sum(1).sqrt()[None, :]).clamp_min(1e-30)
            gid = (torch.arange(r0, r0 + nr, device=dev, dtype=torch.int64) * self.world + self.rank)[None, :].expand(nverify, -1)
            cs = torch.cat([best_s, s], 1)
            ci = torch.cat([best_i, gid], 1)
            ts, ti = torch.topk(cs, k, dim=1, largest=not asc)
            best_s, best_i = ts, torch.gather(ci, 1, ti)
        if self.world > 1:
            gs = [torch.empty_like(best_s) for _ in range(self.world)]
            gi = [torch.empty_like(best_i) for _ in range(self.world)]
            self.dist.all_gather(gs, best_s)
            self.dist.all_gather(gi, best_i)
            cs, ci = torch.cat(gs, 1), torch.cat(gi, 1)
            ts, ti = torch.topk(cs, k, dim=1, largest=not asc)
            best_s, best_i = ts, torch.gather(ci, 1, ti)
        got_i = out.rows[:nverify]
        got_d = out.dists[:nverify]
        strict = 0
        tolerant = 0
        kth = best_s[:, -1:]
        scale = best_s.abs().max().clamp_min(1e-30)
        tol = 1e-5 * scale
        for r in range(nverify):
            ref = set(best_i[r].tolist())
            strict += sum(1 for x in got_i[r].tolist() if x in ref)
        ok = (got_d >= kth - tol) if not asc else (got_d <= kth + tol)
        tolerant = int(ok.sum().item())
        rel = ((got_d - best_s).abs() / best_s.abs().clamp_min(1e-6)).max().item()
        return {"queries": nverify, "recall_at_k": round(strict / (nverify * k), 6),
                "recall_at_k_tolerant": round(tolerant / (nverify * k), 6),
                "max_rel_score_diff_vs_torch_fp32": float("%.3g" % rel), "reference": "torch fp32 matmul top-k (GPU)"}


class ShardedIvf:
    """IVF-Flat over a row-sharded collection (BASELINE config 4): every rank holds the rows g % world == rank of EVERY
    inverted list (same centroids everywhere), routes the batch identically, scans its part of the probed lists, and the
    per-rank (distance, row) candidates are exchanged and merged exactly like ShardedFlat's.  The union over the ranks
    of "rows of the probed lists" is the single-index candidate set, so results equal an unsharded IVF index built
    from the same centroids + assignments.  The shards run with the empty-probe fallback of ivf.rs:258-265 switched off
    (`lynse_hip_ivf_set_routing(h, 2)`): a shard whose part of the probed lists is empty contributes nothing."""

    def __init__(self, dim: int, rank: int = 0, world: int = 1, device: Optional[int] = None, group=None):
        self.dim, self.rank, self.world, self.device = dim, rank, world, device
        self.dist = group
        self.index = None
        self.comm: Optional[NativeComm] = None   # RCCL behind the C-ABI (enable_native_comm); None = torch.distributed's all-gather

    def enable_native_comm(self) -> bool:
        """The exchange inside the library (lynse_hip_ivf_search_sharded_f32_device: local scan -> ncclAllGather -> k_merge on
        one stream).  Collective; False (and the torch.distributed path stays) if RCCL cannot be set up on some rank."""
        try:
            self.comm = NativeComm(self.dist, self.rank, self.world, self.device if self.device is not None else 0)
            return True
        except Exception:  # noqa: BLE001
            self.comm = None
            return False

    def train(self, local_rows, n_global: int, nlist: int, max_iter: int = 20, metric: str = "ip", ivfflat_routing: bool = False,
              reduce=None):
        """All-reduced k-means over the whole collection (`lynse_hip_ivf_kmeans_sharded`): every rank passes ITS rows (global row
        g = local row g // world on rank g % world; a numpy array, or a torch tensor on the rank's device) and gets the SAME centroids and
        the assignments of its rows — kmeans_train on the union (kmeans.rs:74-139) with the centroid sums formed per rank and added over
        the ranks.  The reduction is the library's RCCL communicator when `enable_native_comm` succeeded, else `torch.distributed`
        (`self.dist`: gloo or nccl), else the callable `reduce(host_array)` (tests).  A collective.  IvfFlat indexes train L2 cells."""
        import ctypes as C_

        from .core import _ptr, metric_from_str

        on_device = hasattr(local_rows, "data_ptr")
        if on_device:
            n_local, ptr = int(local_rows.shape[0]), C_.c_void_p(local_rows.data_ptr() if local_rows.shape[0] else 0)
            device = local_rows.device.index or 0
        else:
            local_rows = np.ascontiguousarray(local_rows, np.float32)
            n_local, ptr = int(local_rows.shape[0]), _ptr(local_rows) if local_rows.shape[0] else None
            device = self.device if self.device is not None else 0
        m = metric_from_str("l2" if ivfflat_routing else metric)
        k = min(int(nlist), int(n_global))
        cen = np.zeros((k, self.dim), np.float32)
        asg = np.zeros(max(n_local, 1), np.uint32)
        got = C_.c_uint32(0)

        host_reduce = self._host_reduce(reduce, device)
        cb = _lib.REDUCE_FN(host_reduce)
        comm = self.comm.handle if (self.comm is not None and reduce is None) else None
        if self.world > 1 and comm is None and self.dist is None and reduce is None:
            raise ValueError("ShardedIvf.train over more than one rank needs a reduction: enable_native_comm(), a torch.distributed group, or reduce=")
        check(lib.lynse_hip_ivf_kmeans_sharded(ptr, n_local, 1 if on_device else 0, int(n_global), self.rank, self.world, self.dim, int(nlist),
                                               int(max_iter), m, device, comm, C_.cast(cb, C_.c_void_p), None, _ptr(cen), _ptr(asg), C_.byref(got)))
        return cen[:got.value].copy(), asg[:n_local].copy()

    def build_device(self, d_local_rows, n_global: int, nlist: int, max_iter: int = 20, metric: str = "ip", ivfflat_routing: bool = False,
                     reduce=None) -> None:
        """`train` + `load_local_device` behind ONE C-ABI call (`lynse_hip_ivf_build_sharded_device`): this rank's rows are a torch tensor
        on its device; afterwards `search_device / search_submit / search` answer over the whole collection.  A collective."""
        import ctypes as C_

        from .core import IvfFlatIndex, _sync_producer, metric_from_str

        _sync_producer(d_local_rows)
        device = d_local_rows.device.index or 0
        host_reduce = self._host_reduce(reduce, device)
        cb = _lib.REDUCE_FN(host_reduce)
        comm = self.comm.handle if (self.comm is not None and reduce is None) else None
        if self.world > 1 and comm is None and self.dist is None and reduce is None:
            raise ValueError("ShardedIvf.build_device over more than one rank needs a reduction: enable_native_comm(), a torch.distributed group, or reduce=")
        h = C_.c_void_p()
        check(lib.lynse_hip_ivf_build_sharded_device(C_.c_void_p(d_local_rows.data_ptr()), int(d_local_rows.shape[0]), int(n_global), self.rank, self.world,
                                                     self.dim, int(nlist), int(max_iter), metric_from_str(metric), 1 if ivfflat_routing else 0, device,
                                                     comm, C_.cast(cb, C_.c_void_p), None, C_.byref(h)))
        self.index = IvfFlatIndex(h, self.dim)
        self.metric = metric

    def _host_reduce(self, reduce, device: int):
        """The `lynse_hip_reduce_fn` of this shard's launcher: sums a host buffer over all ranks in place (dtype 0: f32 in any order — the
        init sample, one owner per row —, 1: u32, 2: f32 in RANK order, ((p0 + p1) + p2) + ...: the centroid sums of a Lloyd iteration,
        whose bits `lo_kmeans_train_sharded` defines for every world size; an all-reduce associates as it likes from three ranks on, so
        this one is an all-gather + a sequential add).  `reduce(arr)` (tests) must add in rank order itself."""
        import ctypes as C_

        def host_reduce(_ctx, buf, count, dtype):
            try:
                arr = np.ctypeslib.as_array(C_.cast(buf, C_.POINTER(C_.c_uint32 if dtype == 1 else C_.c_float)), shape=(int(count),))
                if reduce is not None:
                    reduce(arr)
                elif self.world > 1 and self.dist is None:
                    return 1   # no reduction backend for more than one rank: every rank would train its own centroids, silently
                elif self.dist is not None and self.world > 1:
                    import torch

                    t = torch.from_numpy(arr)                      # (shares the buffer)
                    if dtype == 2:
                        on_gpu = self.dist.get_backend() == "nccl"
                        src = t.to(torch.device("cuda", device)) if on_gpu else t
                        parts = [torch.empty_like(src) for _ in range(self.world)]
                        self.dist.all_gather(parts, src)
                        total = parts[0].clone()
                        for r in range(1, self.world):
                            total += parts[r]                       # one IEEE f32 add per element and rank, in rank order
                        t.copy_(total.cpu() if on_gpu else total)
                        return 0
                    if dtype == 1:
                        t = t.view(torch.int32)                    # counts stay far below 2^31
                    if self.dist.get_backend() == "nccl":
                        g = t.to(torch.device("cuda", device))
                        self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM)
                        t.copy_(g.cpu())
                    else:
                        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
                return 0
            except Exception:  # noqa: BLE001  (the library turns a non-zero return into an error)
                return 1

        return host_reduce

    @staticmethod
    def assign(rows: np.ndarray, centroids: np.ndarray, metric: str, device: Optional[int] = None) -> np.ndarray:
        """kmeans::assign_metric (kmeans.rs:237-264) on the device: nearest centroid per row = a FLAT k=1 search of the
        rows against the centroid matrix with the single-row kernels (first smaller rank wins = canonical order)."""
        cen = FlatIndex(None, centroids.shape[1], device)
        cen.write(np.ascontiguousarray(centroids, np.float32))
        check(lib.lynse_hip_flat_set_ip_form(cen.handle, 1))
        out = np.empty(rows.shape[0], np.uint32)
        for r0 in range(0, rows.shape[0], 65536):
            ids, _, _ = cen.search_batch_arrays(rows[r0:r0 + 65536], 1, metric)
            out[r0:r0 + ids.shape[0]] = ids[:, 0].astype(np.uint32)
        return out

    def load_global(self, data: np.ndarray, centroids: np.ndarray, assignments: np.ndarray, metric: str,
                    ivfflat_routing: bool = False, first_global_row: int = 0) -> None:
        """Keep this rank's share of `data` (global rows first_global_row...) with the given global centroids/assignments."""
        first = (self.rank - first_global_row) % self.world
        self.load_local(np.ascontiguousarray(data[first::self.world]), centroids, np.ascontiguousarray(assignments[first::self.world]),
                        metric, ivfflat_routing)

    def load_local(self, local_rows: np.ndarray, centroids: np.ndarray, local_assignments: Optional[np.ndarray], metric: str,
                   ivfflat_routing: bool = False) -> None:
        from .core import IvfFlatIndex

        if local_assignments is None:
            local_assignments = self.assign(local_rows, centroids, "l2" if ivfflat_routing else metric, self.device)
        self.index = IvfFlatIndex.load(local_rows, centroids, local_assignments, metric, device=self.device, ivfflat_routing=ivfflat_routing)
        check(lib.lynse_hip_ivf_set_row_map(self.index._h, self.world, self.rank))
        if self.world > 1:
            check(lib.lynse_hip_ivf_set_routing(self.index._h, 2))
        self.metric = metric

    def load_local_device(self, d_local_rows, centroids: np.ndarray, local_assignments: np.ndarray, metric: str,
                          ivfflat_routing: bool = False) -> None:
        """`load_local` with this rank's rows already in HBM (a torch tensor on the rank's device): nothing of the shard is
        staged through host memory."""
        from .core import IvfFlatIndex

        self.index = IvfFlatIndex.load_device(d_local_rows, centroids, local_assignments, metric, ivfflat_routing=ivfflat_routing)
        check(lib.lynse_hip_ivf_set_row_map(self.index._h, self.world, self.rank))
        if self.world > 1:
            check(lib.lynse_hip_ivf_set_routing(self.index._h, 2))
        self.metric = metric

    def search_device(self, d_queries, k: int, nprobe: int, out: "ShardOutputs") -> None:
        """Whole-collection answer with queries, per-rank blocks and results in HBM: local scan of the probed lists straight
        into the rank's result block, one all-gather, device k-way merge — no numpy on the data path."""
        import torch

        from .core import metric_from_str

        nq = d_queries.shape[0]
        m = metric_from_str(self.metric)
        if self.comm is not None:   # scan -> ncclAllGather -> k_merge behind the C-ABI, no host round trip in between
            torch.cuda.current_stream().synchronize()
            check(lib.lynse_hip_ivf_search_sharded_f32_device(self.index._h, self.comm.handle, C.c_void_p(d_queries.data_ptr()), nq, int(k),
                                                              int(nprobe), C.c_void_p(out.rows.data_ptr()), C.c_void_p(out.dists.data_ptr()),
                                                              C.c_void_p(out.counts.data_ptr())))
            return
        if self.world == 1:
            self.index.search_device(d_queries, k, nprobe, out.rows, out.dists, out.counts)
            return
        pr, pd, pc = out.local_ptrs()
        torch.cuda.current_stream().synchronize()
        check(lib.lynse_hip_ivf_search_f32_device(self.index._h, C.c_void_p(d_queries.data_ptr()), nq, k, int(nprobe),
                                                  C.c_void_p(pr), C.c_void_p(pd), C.c_void_p(pc)))
        self.dist.all_gather_into_tensor(out.gathered, out.local)
        check(lib.lynse_hip_merge_topk_device(C.c_void_p(out.gathered.data_ptr()), out.block_bytes, out.rows_off, out.dists_off,
                                              out.counts_off, self.world, nq, k, m, C.c_void_p(out.rows.data_ptr()),
                                              C.c_void_p(out.dists.data_ptr()), C.c_void_p(out.counts.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def search_submit(self, d_queries, k: int, nprobe: int, out: "ShardOutputs"):
        """One batch IN FLIGHT (lynse_hip_ivf_search_submit_f32_device): local list scans -> ncclAllGather -> merge enqueued on a search
        context of the shard; the ticket's wait() makes out.rows / dists / counts final.  A collective with world > 1.  Needs the
        native communicator (or world == 1); without it the batch is answered here and the ticket is already complete."""
        if self.world == 1 or self.comm is not None:
            return self.index.search_submit(d_queries, k, nprobe, out.rows, out.dists, out.counts,
                                            comm=self.comm.handle if self.comm is not None else None)

        class _Done:
            def wait(self):
                return None

        self.search_device(d_queries, k, nprobe, out)
        return _Done()

    def search_local(self, queries: np.ndarray, k: int, nprobe: int):
        """This rank's candidates: global row ids (local row l -> l * world + rank), canonical order."""
        return self.index.search_batch_arrays(queries, k, nprobe)

    def search(self, queries: np.ndarray, k: int, nprobe: int):
        """Whole-collection answer on every rank: local scan, one all-gather of the result blocks, k-way merge."""
        from .core import metric_from_str

        rows, dists, counts = self.search_local(queries, k, nprobe)
        if self.world == 1:
            return rows, dists, counts
        nq = rows.shape[0]
        m = metric_from_str(self.metric)
        rows_p = np.full((nq, k), np.iinfo(np.uint64).max, np.uint64)
        dists_p = np.zeros((nq, k), np.float32)
        rows_p[:, :rows.shape[1]], dists_p[:, :dists.shape[1]] = rows, dists
        backend = self.dist.get_backend() if hasattr(self.dist, "get_backend") else "gloo"
        if backend == "nccl":  # RCCL moves device memory: stage the 12 B/candidate block in HBM, merge on the device
            import torch

            dev = torch.device("cuda", self.device if self.device is not None else 0)
            out = ShardOutputs(nq, k, self.world, dev)
            out.local.copy_(torch.from_numpy(ShardedFlat.pack_block(rows_p, dists_p, counts)))
            self.dist.all_gather_into_tensor(out.gathered, out.local)
            check(lib.lynse_hip_merge_topk_device(C.c_void_p(out.gathered.data_ptr()), out.block_bytes, out.rows_off, out.dists_off,
                                                  out.counts_off, self.world, nq, k, m, C.c_void_p(out.rows.data_ptr()),
                                                  C.c_void_p(out.dists.data_ptr()), C.c_void_p(out.counts.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            return (out.rows.cpu().numpy().view(np.uint64), out.dists.cpu().numpy(), out.counts.cpu().numpy().view(np.uint32))
        return ShardedFlat.allgather_merge_host(self.dist, self.world, rows_p, dists_p, counts, k, m)
