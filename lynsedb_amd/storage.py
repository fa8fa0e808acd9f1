"""On-disk formats either side of the search path: what the reference writes, this
module reads into HBM shards — and writes back in the same bytes.

* segmented vector store — `vector_manifest.json` + raw little-endian row-major segment files, legacy
  single `vectors.bin` (src/storage/vector_store.rs:24-66, :160-215, :370-445);
* `id_map.bin` — one LE u64 user id per row; rows past its end map to themselves
  (src/engine.rs:2588-2617, :3071-3074);
* `<data>.ivf_meta.bin` of IvfFlatMmap — header (dim, n, n_partitions as LE u64), centroids f32,
  partition offsets u64, original ids u32, next to the slab-ordered data file
  (src/storage/ivf_flat_mmap.rs:448-530, :115-130).

Host logic only: parsing, validation and layout.  Searching needs the HIP library (no CPU fallback).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from pathlib import Path, PurePosixPath
from typing import List, Optional, Sequence, Tuple

import numpy as np

VECTOR_MANIFEST_FILE = "vector_manifest.json"      # vector_store.rs:24
VECTOR_MANIFEST_VERSION = 1                        # :25
SEGMENT_DIR = "vector_segments"                    # :26
DEFAULT_ID_MAP_FILE = "id_map.bin"                 # :27
UPDATE_JOURNAL_FILE = "vector_updates.wal"         # :28
DEFAULT_SEGMENT_TARGET_BYTES = 256 * 1024 * 1024   # :32


class StorageError(IOError):
    """LynseError::Storage."""


@dataclass
class SegmentEntry:  # vector_store.rs:36-40
    file: str
    rows: int


@dataclass
class VectorManifest:  # vector_store.rs:42-48
    version: int = VECTOR_MANIFEST_VERSION
    generation: int = 0
    id_map_file: str = DEFAULT_ID_MAP_FILE
    segments: List[SegmentEntry] = field(default_factory=list)

    def to_json(self) -> str:
        return json.dumps({"version": self.version, "generation": self.generation, "id_map_file": self.id_map_file,
                           "segments": [{"file": s.file, "rows": s.rows} for s in self.segments]}, indent=2)


def dtype_width(dtype: str) -> int:
    """VectorDtype::parse + byte_width (src/storage/dtype.rs:12-29)."""
    d = dtype.strip().lower()
    if d in ("f32", "float32", "float"):
        return 4
    if d in ("f16", "float16", "half", "fp16"):
        return 2
    raise ValueError(f"unsupported vector dtype '{dtype}'; expected float32/f32 or float16/f16")


def validate_manifest_path(value: str, label: str) -> None:
    """vector_store.rs:68-81: a non-empty relative path made of normal components only."""
    p = PurePosixPath(value)
    bad = (not value) or p.is_absolute() or value.startswith("\\") or any(c in ("..", ".") for c in value.replace("\\", "/").split("/")) \
        or any(c == "" for c in value.replace("\\", "/").split("/"))
    if bad:
        raise StorageError(f"vector manifest {label} must be a safe relative path: {value!r}")


def load_manifest(collection_path, dim: int, dtype: str = "f32") -> VectorManifest:
    """VectorStore::new (vector_store.rs:160-215): parse or synthesise (legacy `vectors.bin`) the manifest, validate the
    paths, and take every segment's row count from its FILE LENGTH (a partial trailing row is ignored)."""
    root = Path(collection_path)
    row_width = dim * dtype_width(dtype)
    if (root / UPDATE_JOURNAL_FILE).exists():
        # VectorStore::has_pending_updates (vector_store.rs:682-687): in-place row updates that were journalled but not yet
        # applied to the segment files.  Replaying the journal is the storage engine's job (out of scope); serving the
        # segments as they are would silently return stale rows.
        raise StorageError(f"{root / UPDATE_JOURNAL_FILE} exists: the collection has pending row updates that were not applied to "
                           "its segment files; open it with LynseDB once (the journal is replayed on open) before loading it here")
    mpath = root / VECTOR_MANIFEST_FILE
    if mpath.exists():
        try:
            raw = json.loads(mpath.read_bytes())
            m = VectorManifest(int(raw["version"]), int(raw["generation"]), str(raw["id_map_file"]),
                               [SegmentEntry(str(s["file"]), int(s["rows"])) for s in raw["segments"]])
        except (ValueError, KeyError, TypeError) as e:
            raise ValueError(f"vector manifest: {e}") from e  # LynseError::Serialization
        if m.version > VECTOR_MANIFEST_VERSION:
            raise StorageError(f"vector manifest version {m.version} is newer than supported version {VECTOR_MANIFEST_VERSION}")
    else:
        legacy = root / "vectors.bin"
        nbytes = legacy.stat().st_size if legacy.exists() else 0
        rows = 0 if row_width == 0 else nbytes // row_width
        m = VectorManifest(segments=[SegmentEntry("vectors.bin", rows)] if rows else [])
    validate_manifest_path(m.id_map_file, "ID-map path")
    seen = set()
    for s in m.segments:
        validate_manifest_path(s.file, "segment path")
        if s.file in seen:
            raise StorageError(f"vector manifest contains duplicate segment path {s.file!r}")
        seen.add(s.file)
    for s in m.segments:
        sp = root / s.file
        if not sp.exists():
            raise StorageError(f"vector manifest segment {sp} is unavailable: not found")
        s.rows = 0 if row_width == 0 else sp.stat().st_size // row_width
    return m


def load_id_map(path) -> np.ndarray:
    """engine.rs:2588-2603: LE u64 per row, partial trailing bytes ignored, missing file = empty."""
    p = Path(path)
    if not p.exists():
        return np.zeros(0, np.uint64)
    raw = p.read_bytes()
    return np.frombuffer(raw[:len(raw) // 8 * 8], dtype="<u8").astype(np.uint64)


def rows_to_user_ids(rows: np.ndarray, id_map: np.ndarray) -> np.ndarray:
    """row_to_user_id (engine.rs:3071-3074): id_map[row], or the row itself past the end of the map."""
    r = np.asarray(rows, np.uint64)
    out = r.copy()
    inside = r < id_map.size
    out[inside] = id_map[r[inside].astype(np.int64)]
    return out


def read_segments(collection_path, dim: int, dtype: str = "f32"):
    """Yield (first_global_row, rows[n, dim]) per segment in manifest order (global row = concatenation order,
    vector_store.rs:1016-1037): f32 values, or the u16 words of an F16 store."""
    root = Path(collection_path)
    m = load_manifest(root, dim, dtype)
    f16 = dtype_width(dtype) == 2  # F16 segments: the rows come back as their u16 words (FlatIndex.write_f16_bits)
    base = 0
    for s in m.segments:
        if s.rows:
            a = np.fromfile(root / s.file, dtype="<u2" if f16 else "<f4", count=s.rows * dim).reshape(s.rows, dim)
            yield base, a
        base += s.rows


def open_flat_collection(collection_path, dim: int, dtype: str = "f32", device: Optional[int] = None):
    """Load a collection directory written by the reference into one HBM shard.
    -> (FlatIndex, id_map u64[...], manifest).  Search rows are global rows; `rows_to_user_ids` maps them."""
    from .core import FlatIndex

    m = load_manifest(collection_path, dim, dtype)
    f16 = dtype_width(dtype) == 2
    idx = FlatIndex(None, dim, device, dtype="f16" if f16 else "f32")
    total = sum(s.rows for s in m.segments)
    if total:
        idx.reserve(total)
    for _, rows in read_segments(collection_path, dim, dtype):
        if f16:
            idx.write_f16_bits(rows)
        else:
            idx.write(rows)
    id_map = load_id_map(Path(collection_path) / m.id_map_file)
    return idx, id_map, m


def write_flat_collection(collection_path, batches: Sequence[np.ndarray], ids: Optional[np.ndarray] = None,
                          segment_target_bytes: int = DEFAULT_SEGMENT_TARGET_BYTES, dtype: str = "f32") -> VectorManifest:
    """Append `batches` the way VectorStore::write does (append_encoded_bytes, vector_store.rs:379-445): the first
    segment is `vectors.bin`; a batch that does not fit the current segment's target size opens
    `vector_segments/seg-{generation+1:020}-{index:06}.bin`; the manifest file appears with the second segment."""
    root = Path(collection_path)
    root.mkdir(parents=True, exist_ok=True)
    mpath = root / VECTOR_MANIFEST_FILE
    batches = list(batches)
    m = VectorManifest()
    if batches and (mpath.exists() or (root / "vectors.bin").exists()):
        # an existing collection is APPENDED to, like VectorStore::write: start from its manifest (row counts from the file
        # lengths), never truncate its segments
        m = load_manifest(root, int(np.asarray(batches[0]).shape[1]), dtype)
    elif mpath.exists() or (root / "vectors.bin").exists():
        return load_manifest(root, 0, dtype)
    for b in batches:
        # encode_f32_slice_as_le_bytes (src/storage/dtype.rs): F16 rounds to nearest even
        a = np.ascontiguousarray(b, dtype="<f4") if dtype_width(dtype) == 4 else np.ascontiguousarray(np.asarray(b, np.float32).astype("<f2"))
        row_width = a.shape[1] * dtype_width(dtype)
        target = max(segment_target_bytes, row_width)
        data = a.tobytes()
        fits = bool(m.segments) and m.segments[-1].rows * row_width + len(data) <= target
        if not m.segments or not fits:
            name = "vectors.bin" if (not m.segments and not mpath.exists()) else \
                f"{SEGMENT_DIR}/seg-{m.generation + 1:020d}-{len(m.segments):06d}.bin"
            p = root / name
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_bytes(data)
            m.generation += 1
            m.segments.append(SegmentEntry(name, a.shape[0]))
            if len(m.segments) > 1 or mpath.exists():
                tmp = mpath.with_suffix(".tmp")
                tmp.write_text(m.to_json())
                os.replace(tmp, mpath)
        else:
            with open(root / m.segments[-1].file, "ab") as f:
                f.write(data)
            m.segments[-1].rows += a.shape[0]
    if ids is not None:  # append_id_map_path (engine.rs:3058-3067): ids of the appended rows go to the end of the map
        with open(root / m.id_map_file, "ab") as f:
            f.write(np.ascontiguousarray(ids, dtype="<u8").tobytes())
    return m


# ------------------------------------------------------------------------------ IvfFlatMmap files
@dataclass
class IvfMeta:  # ivf_flat_mmap.rs:22-39
    dim: int
    n_vectors: int
    n_partitions: int
    centroids: np.ndarray          # f32 [n_partitions, dim]
    partition_offsets: np.ndarray  # u64 [n_partitions + 1]
    original_ids: np.ndarray       # u32 [n_vectors]: slab position -> original row


def ivf_meta_path(data_path) -> Path:
    """`data_path.with_extension("ivf_meta.bin")` (ivf_flat_mmap.rs:133)."""
    p = Path(data_path)
    return p.with_suffix(".ivf_meta.bin") if p.suffix else p.with_name(p.name + ".ivf_meta.bin")


def save_ivf_meta(path, meta: IvfMeta) -> None:  # ivf_flat_mmap.rs:448-482
    with open(path, "wb") as f:
        f.write(np.array([meta.dim, meta.n_vectors, meta.n_partitions], "<u8").tobytes())
        f.write(np.ascontiguousarray(meta.centroids, "<f4").tobytes())
        f.write(np.ascontiguousarray(meta.partition_offsets, "<u8").tobytes())
        f.write(np.ascontiguousarray(meta.original_ids, "<u4").tobytes())


def load_ivf_meta(path) -> IvfMeta:  # ivf_flat_mmap.rs:484-530
    raw = Path(path).read_bytes()
    if len(raw) < 24:
        raise IOError("failed to fill whole buffer")  # read_exact on a short file
    dim, n, k = (int(x) for x in np.frombuffer(raw[:24], "<u8"))
    need = 24 + k * dim * 4 + (k + 1) * 8 + n * 4
    if len(raw) < need:
        raise IOError("failed to fill whole buffer")
    o = 24
    cen = np.frombuffer(raw[o:o + k * dim * 4], "<f4").reshape(k, dim).astype(np.float32)
    o += k * dim * 4
    off = np.frombuffer(raw[o:o + (k + 1) * 8], "<u8").astype(np.uint64)
    o += (k + 1) * 8
    ids = np.frombuffer(raw[o:o + n * 4], "<u4").astype(np.uint32)
    return IvfMeta(dim, n, k, cen, off, ids)


def ivf_assignments_from_meta(meta: IvfMeta) -> np.ndarray:
    """assignment of every ORIGINAL row, from the slab layout (inverse of ivf_flat_mmap.rs:105-130)."""
    part_of_pos = np.repeat(np.arange(meta.n_partitions, dtype=np.uint32), np.diff(meta.partition_offsets.astype(np.int64)))
    asg = np.zeros(meta.n_vectors, np.uint32)
    asg[meta.original_ids.astype(np.int64)] = part_of_pos
    return asg


def open_ivf_flat(data_path, metric: str = "ip", device: Optional[int] = None):
    """IvfFlatMmap::open (ivf_flat_mmap.rs:161-223): the slab-ordered data file + its `.ivf_meta.bin`.
    -> IvfFlatIndex with the IvfFlat routing semantics; search results are ORIGINAL row ids."""
    from .core import IvfFlatIndex

    meta = load_ivf_meta(ivf_meta_path(data_path))
    slab = np.fromfile(data_path, dtype="<f4", count=meta.n_vectors * meta.dim)
    if slab.size != meta.n_vectors * meta.dim:
        raise IOError("IVF data file is shorter than its metadata")
    slab = slab.reshape(meta.n_vectors, meta.dim)
    original = np.empty_like(slab)
    original[meta.original_ids.astype(np.int64)] = slab  # back to original row order; load() rebuilds the same slabs
    return IvfFlatIndex.load(original, meta.centroids, ivf_assignments_from_meta(meta), metric, device=device, ivfflat_routing=True)


def save_ivf_flat(data_path, index, data: np.ndarray) -> IvfMeta:
    """Write what IvfFlatMmap::build writes (ivf_flat_mmap.rs:105-150) for a built index over `data` (original order)."""
    cen, asg, off, orig = index.export()
    slab = np.ascontiguousarray(np.asarray(data, np.float32)[orig.astype(np.int64)], dtype="<f4")
    Path(data_path).write_bytes(slab.tobytes())
    meta = IvfMeta(cen.shape[1], slab.shape[0], cen.shape[0], cen, off, orig)
    save_ivf_meta(ivf_meta_path(data_path), meta)
    return meta
