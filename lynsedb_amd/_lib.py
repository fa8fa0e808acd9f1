"""ctypes binding of liblynse_hip.so (the C ABI in include/lynse_hip.h).

There is NO CPU fallback: if the shared library is missing this module raises ImportError, and every
compute call fails with a device error when no HIP device is present.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C lynsedb_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

# PyTorch-ROCm bundles its own HIP runtime.  If liblynse_hip.so pulls in the system libamdhip64 first,
# a later `import torch` in the same process finds "No HIP GPUs": load torch's runtime first so both
# share one HIP runtime (torch is used by this package only for device memory / process groups).
try:  # pragma: no cover - depends on the environment
    import torch as _torch  # noqa: F401
except ImportError:  # pure C-ABI use without torch is fine
    _torch = None

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "liblynse_hip.so"

OK = 0
ERR_INVALID_ARGUMENT, ERR_DIMENSION_MISMATCH, ERR_UNKNOWN_METRIC, ERR_NOT_FINALIZED = 1, 2, 3, 4
ERR_OUT_OF_MEMORY, ERR_DEVICE, ERR_INTERNAL, ERR_INDEX_NOT_BUILT, ERR_UNSUPPORTED, ERR_TIMEOUT = 5, 6, 7, 8, 9, 10

METRIC_IP, METRIC_L2, METRIC_COSINE, METRIC_HAMMING, METRIC_JACCARD, METRIC_DICE, METRIC_TANIMOTO = range(7)
IPFORM_AUTO, IPFORM_SINGLE, IPFORM_BATCH8 = 0, 1, 2


class Profile(C.Structure):
    _fields_ = [("searches", C.c_uint64), ("scan_launches", C.c_uint64), ("scan_us", C.c_double),
                ("scan_rows", C.c_uint64), ("scan_bytes", C.c_uint64), ("total_us", C.c_double),
                ("fallback_queries", C.c_uint64), ("pool_entries", C.c_uint64), ("last_plan", C.c_uint64)]


_f32p, _u32p, _u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/lynse_hip.h
SIGNATURES = {
    "lynse_hip_abi_version": (C.c_int, []),
    "lynse_hip_last_error": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "lynse_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "lynse_hip_metric_from_str": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "lynse_hip_metric_from_index_mode": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "lynse_hip_metric_is_ascending": (C.c_int, [C.c_int]),
    "lynse_hip_metric_is_binary": (C.c_int, [C.c_int]),
    "lynse_hip_flat_create": (C.c_int, [C.c_uint32, C.c_int, C.POINTER(_vp)]),
    "lynse_hip_flat_destroy": (C.c_int, [_vp]),
    "lynse_hip_flat_reserve": (C.c_int, [_vp, C.c_uint64]),
    "lynse_hip_flat_append_f32": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lynse_hip_flat_append_f32_device": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lynse_hip_flat_append_packed_u64": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lynse_hip_flat_append_packed_u64_device": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lynse_hip_flat_finalize": (C.c_int, [_vp]),
    "lynse_hip_flat_set_row_map": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
    "lynse_hip_flat_set_ip_form": (C.c_int, [_vp, C.c_int]),
    "lynse_hip_flat_len": (C.c_uint64, [_vp]),
    "lynse_hip_flat_dim": (C.c_uint32, [_vp]),
    "lynse_hip_flat_device": (C.c_int, [_vp]),
    "lynse_hip_flat_read_rows": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "lynse_hip_flat_copy_rows_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "lynse_hip_flat_read_packed": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "lynse_hip_flat_search_f32": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp]),
    "lynse_hip_flat_search_sq8_f32": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp]),
    "lynse_hip_flat_sq8_params": (C.c_int, [_vp, _vp, _vp]),
    "lynse_hip_flat_set_dtype": (C.c_int, [_vp, C.c_int]),
    "lynse_hip_flat_append_f16_bits": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lynse_hip_flat_search_filtered_f32": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, C.c_uint64, _vp, _vp, _vp]),
    "lynse_hip_flat_search_filtered_bitset_f32": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, C.c_uint64, _vp, _vp, _vp]),
    "lynse_hip_flat_search_f32_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp, _vp]),
    "lynse_hip_flat_search_packed_u64": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp]),
    "lynse_hip_flat_search_packed_u64_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp, _vp]),
    "lynse_hip_flat_profile_enable": (C.c_int, [_vp, C.c_int]),
    "lynse_hip_flat_profile_get": (C.c_int, [_vp, C.POINTER(Profile), C.c_int]),
    "lynse_hip_flat_coarse_state": (C.c_int, [_vp, C.POINTER(C.c_int), _u64p]),
    "lynse_hip_flat_bpm_rows": (C.c_uint64, [_vp]),
    "lynse_hip_flat_prepare": (C.c_int, [_vp, C.c_int, C.c_uint64]),
    "lynse_hip_flat_hbm_bytes": (C.c_uint64, [_vp]),
    "lynse_hip_flat_set_fused_search": (C.c_int, [_vp, C.c_int]),
    "lynse_hip_flat_set_plan": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "lynse_hip_compute_distance": (C.c_int, [_vp, _vp, C.c_uint32, C.c_int, C.c_int, _f32p]),
    "lynse_hip_top_k_search": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, _vp, _vp, _u32p]),
    "lynse_hip_pack_binary_f32": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_int, _vp]),
    "lynse_hip_merge_topk": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _vp, _vp, _u32p]),
    "lynse_hip_merge_topk_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                              C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp, _vp]),
    "lynse_hip_ivf_build": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lynse_hip_ivf_load": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, C.c_uint32, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lynse_hip_ivf_load_binary": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, C.c_uint32, _vp, C.c_int, _vp, C.c_int, C.POINTER(_vp)]),
    "lynse_hip_ivf_thresholds": (C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    "lynse_hip_ivf_destroy": (C.c_int, [_vp]),
    "lynse_hip_ivf_len": (C.c_uint64, [_vp]),
    "lynse_hip_ivf_nlist": (C.c_uint32, [_vp]),
    "lynse_hip_ivf_export": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "lynse_hip_ivf_set_row_map": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
    "lynse_hip_ivf_set_routing": (C.c_int, [_vp, C.c_int]),
    "lynse_hip_ivf_set_fused_search": (C.c_int, [_vp, C.c_int]),
    "lynse_hip_ivf_search_f32": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "lynse_hip_ivf_build_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lynse_hip_ivf_load_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, C.c_uint32, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lynse_hip_ivf_search_f32_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "lynse_hip_ivf_search_metric_f32": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _vp, _vp, _vp]),
    "lynse_hip_ivf_search_filtered_f32": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp, _vp, _vp]),
    "lynse_hip_ivf_profile_enable": (C.c_int, [_vp, C.c_int]),
    "lynse_hip_filter_tombstoned_limit": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint64, _vp, _vp, C.POINTER(C.c_uint64)]),
    "lynse_hip_merge_row_results": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_uint64, C.c_int, _vp, _vp, C.POINTER(C.c_uint64)]),
    "lynse_hip_encode_search_result": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "lynse_hip_decode_search_result": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, _vp, C.c_uint32, C.POINTER(C.c_uint32),
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "lynse_hip_ivf_profile_get": (C.c_int, [_vp, C.POINTER(Profile), C.c_int]),
    "lynse_hip_ivf_insert_f32": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lynse_hip_ivf_search_sharded_f32_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "lynse_hip_ivf_delete_rows": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lynse_hip_ivf_assign_f32": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "lynse_hip_comm_load_rccl": (C.c_int, [C.c_char_p]),
    "lynse_hip_comm_unique_id": (C.c_int, [_vp]),
    "lynse_hip_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lynse_hip_comm_destroy": (C.c_int, [_vp]),
    "lynse_hip_comm_rank": (C.c_int, [_vp]),
    "lynse_hip_comm_world": (C.c_int, [_vp]),
    "lynse_hip_comm_ranks_seen": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "lynse_hip_flat_search_sharded_f32_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp]),
    "lynse_hip_flat_search_sharded_packed_u64_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp]),
    "lynse_hip_flat_search_submit_f32_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp, C.POINTER(_vp)]),
    "lynse_hip_flat_search_submit_packed_u64_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp, C.POINTER(_vp)]),
    "lynse_hip_flat_search_wait": (C.c_int, [_vp]),
    "lynse_hip_set_wait_timeout_ms": (C.c_int, [C.c_uint32]),
    "lynse_hip_ivf_search_submit_f32_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "lynse_hip_ivf_search_wait": (C.c_int, [_vp]),
    "lynse_hip_ivf_ticket_stats": (C.c_int, [_vp, _vp]),
    "lynse_hip_ivf_kmeans_sharded": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                              _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint32)]),
    "lynse_hip_ivf_build_sharded_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                                    _vp, _vp, _vp, C.POINTER(_vp)]),
    "lynse_hip_flat_coarse_scores": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, _vp, _vp, C.POINTER(C.c_int)]),
}

REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int)   # lynse_hip_reduce_fn

if not LIB_PATH.exists():
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built and lynsedb_amd has no CPU "
        "fallback. Run `make -C lynsedb_amd/csrc` (hipcc --offload-arch=gfx950).")

lib = C.CDLL(str(LIB_PATH))
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here means the .so is stale vs include/lynse_hip.h
    _fn.restype = _res
    _fn.argtypes = _args

if lib.lynse_hip_abi_version() != 1:
    raise ImportError("liblynse_hip.so ABI version mismatch; rebuild it")


class LynseHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class LynseUnsupportedError(LynseHipError, NotImplementedError):
    """LYNSE_ERR_UNSUPPORTED: valid in the reference, outside what this library implements (e.g. the metric 'l1')."""


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib.lynse_hip_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(code: int) -> None:
    """Map status codes to the exception classes the reference raises (src/error.rs:56-73)."""
    if code == OK:
        return
    msg = last_error() or f"lynse_hip error {code}"
    if code in (ERR_INVALID_ARGUMENT, ERR_DIMENSION_MISMATCH, ERR_UNKNOWN_METRIC):
        raise ValueError(msg)
    if code == ERR_OUT_OF_MEMORY:
        raise MemoryError(msg)
    if code == ERR_UNSUPPORTED:
        raise LynseUnsupportedError(code, msg)
    if code == ERR_TIMEOUT:
        raise TimeoutError(msg)
    raise LynseHipError(code, msg)


def device_count() -> int:
    n = C.c_int(0)
    rc = lib.lynse_hip_device_count(C.byref(n))
    return n.value if rc == OK else 0
