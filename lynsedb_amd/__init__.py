"""lynsedb_amd — MI355X-native FLAT / IVF-Flat search behind the `lynse._core` surface.

Importing this package loads liblynse_hip.so (hand-written HIP kernels for gfx950).  There is no
CPU fallback: the import fails loudly if the extension has not been built.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP extension is missing)
from .core import (BitSet, Collection, DatabaseManager, FlatIndex, IvfFlatIndex, SearchResult,  # noqa: F401
                   default_device, visible_devices, merge_topk, metric_from_index_mode, metric_from_str,
                   py_compute_distance, py_top_k_search)

__all__ = ["BitSet", "Collection", "DatabaseManager", "FlatIndex", "IvfFlatIndex", "SearchResult", "default_device", "visible_devices",
           "merge_topk", "metric_from_index_mode", "metric_from_str", "py_compute_distance", "py_top_k_search"]
__version__ = "0.1.0"
