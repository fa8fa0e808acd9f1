/*
 * lynse_hip.h — C ABI of liblynse_hip.so: the MI355X (gfx950) implementation of LynseDB's
 * FLAT / IVF-Flat search hot path (src/distance + the FLAT/IVF search in src/index, src/storage).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each entry point
 * names the reference interface it replaces (file:line relative to BirchKwok/lynsedb).  The
 * reference-side binding a maintainer would add (Rust `extern "C"` + the PyO3 glue that already
 * exists) is shown in INTEGRATION.md.
 *
 * Conventions (mirroring the reference boundary):
 *  - inputs are borrowed for the duration of the call; outputs are CALLER-allocated, fixed size
 *    (nq*k), short results are padded (row ~0, distance +inf / -inf: the worst of the metric) and the true length is
 *    written to out_counts[q];
 *  - empty store or k == 0 -> counts 0, not an error (flat_mmap.rs:832-835); k > N clamps (:836);
 *  - rows returned are ROW INDICES (u32 per segment in the reference, u64 here after the row map),
 *    best-first; ties are ordered by row ascending — the order VectorStore::merge_results imposes
 *    (vector_store.rs:953-970);
 *  - distances: IP raw dot (descending), L2 squared, cosine distance, Hamming/Jaccard/Dice as f32;
 *  - every function returns a status code (LynseError variants, src/error.rs:5-52); the message is
 *    available per thread from lynse_hip_last_error(); nothing throws or aborts across the ABI;
 *  - NON-FINITE VALUES.  The reference compares distances with partial_cmp(..).unwrap_or(Equal) (flat_mmap.rs:2141-2149,
 *    :2170-2176): a NaN distance is "equal" to everything, is never admitted once a top-k array is full and stays wherever the first
 *    fill put it — the result depends on the row order and the thread count.  This boundary pins ONE order instead: a NaN score is
 *    reported as the WORST value of the metric (+inf for the distances, -inf for IP) and ranks behind every better score, ties by
 *    ascending row like any other tie; +inf / -inf scores are ordinary values of the order.  A query with a NaN element (cosine: or
 *    an infinite one) scores NaN against every row: an unfiltered FLAT search answers it with rows 0 .. min(k, N) - 1 at the worst
 *    value — the rows the reference's first fill keeps.  Rows may hold NaN / +-inf elements (the certified int8 pass switches itself
 *    off for such a shard).  Pinned by tests/test_gpu_flat_parity.py::test_nan_and_infinite_rows_and_queries.
 *  - a handle may be searched from several threads: unfiltered searches run under a SHARED lock, each on one of the handle's
 *    search contexts (stream + workspace, up to LYNSE_HIP_CONTEXTS = 8 at a time, further callers wait for a free one) — the
 *    Arc<RwLock<Collection>>::read of the reference (src/python/mod.rs:950, :1187); append / finalize / lazy builds of derived copies
 *    and the gathered-rows strategy of a subset filter take the lock EXCLUSIVE, like `&mut self` in FlatMmap::write, and are
 *    refused while tickets of lynse_hip_flat_search_submit_* are outstanding.
 *  - there is NO CPU fallback: without a usable HIP device every compute entry returns
 *    LYNSE_ERR_DEVICE.
 */
#ifndef LYNSE_HIP_H
#define LYNSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LYNSE_HIP_ABI_VERSION 1

/* Status codes — src/error.rs:5-52 (DimensionMismatch / InvalidArgument / IndexNotBuilt / Io ...). */
enum {
    LYNSE_OK = 0,
    LYNSE_ERR_INVALID_ARGUMENT = 1,
    LYNSE_ERR_DIMENSION_MISMATCH = 2,
    LYNSE_ERR_UNKNOWN_METRIC = 3,
    LYNSE_ERR_NOT_FINALIZED = 4,
    LYNSE_ERR_OUT_OF_MEMORY = 5,
    LYNSE_ERR_DEVICE = 6,
    LYNSE_ERR_INTERNAL = 7,
    LYNSE_ERR_INDEX_NOT_BUILT = 8,
    LYNSE_ERR_UNSUPPORTED = 9,
    LYNSE_ERR_TIMEOUT = 10 /* lynse_hip_flat_search_wait: the batch did not finish within the wait timeout (a peer rank is gone) */
};

/* Metric ids — DistanceMetric (src/distance/mod.rs:19-36), in-scope subset. */
enum {
    LYNSE_METRIC_IP = 0,
    LYNSE_METRIC_L2 = 1,
    LYNSE_METRIC_COSINE = 2,
    LYNSE_METRIC_HAMMING = 3,
    LYNSE_METRIC_JACCARD = 4,
    LYNSE_METRIC_DICE = 5,
    LYNSE_METRIC_TANIMOTO = 6
};

/* IP accumulation form of the exact rescoring pass. */
enum { LYNSE_IPFORM_AUTO = 0, LYNSE_IPFORM_SINGLE = 1, LYNSE_IPFORM_BATCH8 = 2,
       LYNSE_IPFORM_F16SEQ = 3 /* f16 storage: sequential sums of simd.rs:805-846 (set by the dtype, not by set_ip_form) */ };

/* Storage dtype of a float shard (src/storage/dtype.rs:6-29). */
enum { LYNSE_DTYPE_F32 = 0, LYNSE_DTYPE_F16 = 1 };

typedef struct lynse_hip_flat lynse_hip_flat; /* one HBM-resident FLAT shard: replaces FlatMmap */
typedef struct lynse_hip_ivf lynse_hip_ivf;   /* IVF-Flat over cluster slabs: replaces IVFIndex / IvfFlatMmap */

/* Per-search profile (the reference's QueryProfile, src/engine.rs:6906-6919, plus kernel timings
 * measured with HIP events on the launch stream).  Times in microseconds. */
typedef struct lynse_hip_profile {
    uint64_t searches;        /* profiled search calls accumulated */
    uint64_t scan_launches;   /* launches of the dominant scan kernel */
    double scan_us;           /* summed duration of those launches */
    uint64_t scan_rows;       /* rows scanned by those launches */
    uint64_t scan_bytes;      /* algorithmic bytes of those launches (rows * row bytes) */
    double total_us;          /* whole pipeline, first launch -> last launch */
    uint64_t fallback_queries;/* queries re-run on the exhaustive safe plan */
    uint64_t pool_entries;    /* candidates rescored exactly (sum over queries) */
    uint64_t last_plan;       /* plan of the last profiled float chunk: bit 0 sampled stage plan, bit 1 threshold-only
                                 (lane-max) sample stage, bit 2 certified int8 coarse pass, bit 3 segmented emission,
                                 bit 4 <= 32-query kernel, bit 5 fused single-launch search (k_small_search), bit 6 the search STARTED
                                 on the certified int8 pass (bit 2 tells what the LAST run used: an overflow retries on the f16
                                 pass), bit 7 the sample stage ran INSIDE the launch of the first threshold stage (one scan launch less than
                                 stages), bits 8..15 number of scan stages, bits 16..23 wave tiling
                                 (0x24 = <2,4,4,2>, 0x42 = <4,2,2,4>, 0x14 = <1,4,1,1>, 0x81 = the query-stationary k_scan_qs), bit 24 the self-tightening
                                 single-launch scan, bit 25 the sample stage ran on the query-stationary tiling — lets a test pin the kernel
                                 instantiation a benchmark configuration runs */
} lynse_hip_profile;

/* ---- library ---- */
int lynse_hip_abi_version(void);
/* Copies the calling thread's last error message (NUL-terminated) and returns its length. */
size_t lynse_hip_last_error(char *buf, size_t cap);
int lynse_hip_device_count(int *out_count);
/* DistanceMetric::from_str (distance/mod.rs:39-63) / from_index_mode (:67-107). */
int lynse_hip_metric_from_str(const char *name, int *out_metric);
int lynse_hip_metric_from_index_mode(const char *mode, int *out_metric);
int lynse_hip_metric_is_ascending(int metric); /* distance/mod.rs:111-116 */
int lynse_hip_metric_is_binary(int metric);    /* distance/mod.rs:161-166 */

/* ---- FLAT shard: FlatMmap (src/storage/flat_mmap.rs:89-109, :187-221) ---- */

/* FlatMmap::open(path, dim, F32) -> an empty row-major f32 store on `device`. */
int lynse_hip_flat_create(uint32_t dim, int device, lynse_hip_flat **out);
int lynse_hip_flat_destroy(lynse_hip_flat *h);
int lynse_hip_flat_reserve(lynse_hip_flat *h, uint64_t rows);
/* FlatMmap::write / append (flat_mmap.rs:223-330): rows are n*dim little-endian f32, row-major —
 * exactly a reference segment file's bytes. */
int lynse_hip_flat_append_f32(lynse_hip_flat *h, const float *rows, uint64_t n);
int lynse_hip_flat_append_f32_device(lynse_hip_flat *h, const float *d_rows, uint64_t n);
/* Pre-packed one-bit rows, ceil(dim/64) u64 words per row, bit i of word i/64 LSB-first — the
 * BinaryData layout (flat_mmap.rs:145-160, simd.rs:750-757).  A store holds f32 rows or packed
 * rows, not both appended. */
int lynse_hip_flat_append_packed_u64(lynse_hip_flat *h, const uint64_t *words, uint64_t n);
int lynse_hip_flat_append_packed_u64_device(lynse_hip_flat *h, const uint64_t *d_words, uint64_t n);
/* Row statistics for newly appended rows (norms, scale).  Idempotent. */
/* VectorDtype::F16 storage (FlatMmap with dtype F16, flat_mmap.rs:187-221, search :905-908): appended f32 rows are
 * rounded through f16 (RNE) like `encode_f32_slice_as_le_bytes`, `append_f16_bits` takes the file's u16 words as they
 * are, and every distance follows the reference's f16 kernels (sequential f32 sums, simd.rs:805-846).  Must be called
 * on an empty shard.  ip / l2 / cosine (+ the packed-binary metrics, which only see value > 0.5). */
int lynse_hip_flat_set_dtype(lynse_hip_flat *h, int dtype);
int lynse_hip_flat_append_f16_bits(lynse_hip_flat *h, const uint16_t *rows, uint64_t n);
int lynse_hip_flat_finalize(lynse_hip_flat *h);
/* Returned row = local_row * stride + offset (multi-GPU shards; default 1, 0). */
int lynse_hip_flat_set_row_map(lynse_hip_flat *h, uint64_t stride, uint64_t offset);
int lynse_hip_flat_set_ip_form(lynse_hip_flat *h, int ip_form);
uint64_t lynse_hip_flat_len(const lynse_hip_flat *h); /* FlatMmap::len */
uint32_t lynse_hip_flat_dim(const lynse_hip_flat *h); /* FlatMmap::dim */
int lynse_hip_flat_device(const lynse_hip_flat *h);
/* Copy rows [first, first+n) back to host f32 (FlatMmap::as_slice). */
int lynse_hip_flat_read_rows(const lynse_hip_flat *h, uint64_t first, uint64_t n, float *out);
/* Same, into a device buffer of this handle's GPU (n*dim f32, dense). */
int lynse_hip_flat_copy_rows_device(const lynse_hip_flat *h, uint64_t first, uint64_t n, float *d_out);
/* ensure_binary (flat_mmap.rs:388-401) contents: packed words of rows [first, first+n). */
int lynse_hip_flat_read_packed(lynse_hip_flat *h, uint64_t first, uint64_t n, uint64_t *out_words);

/* FlatMmap::search (flat_mmap.rs:824-923) for a batch of queries (Collection::batch_search,
 * engine.rs:5352-5498).  queries: nq*dim f32.  Binary metrics threshold the f32 query at 0.5
 * (pack_binary_query, flat_mmap.rs:1292-1296).  out_rows/out_dists: nq*k; out_counts: nq. */
int lynse_hip_flat_search_f32(lynse_hip_flat *h, const float *queries, uint64_t nq, uint32_t k,
                              int metric, uint64_t *out_rows, float *out_dists,
                              uint32_t *out_counts);
/* FlatMmap::search_filtered (src/storage/flat_mmap.rs:491-815; VectorStore::search_filtered
 * vector_store.rs:1006-1040; Collection brute_force_search_filtered engine.rs:5541-5566): top-k among the rows
 * listed in `subset_rows` (local row indices of this shard, host memory; ids >= len are skipped like the
 * reference skips them, duplicates count once).  k is clamped to min(k, n_subset, len) (:501).  One subset serves
 * the whole batch.  Binary metrics pack the f32 queries (pack_binary_query).  The subset becomes a row bitmask on
 * the device — the reference's own bitset strategy (:672-679) — applied in the scan epilogue; distances use the
 * single-row kernels as the reference does on this path (:553-560), results are in canonical (distance, row)
 * order. */
int lynse_hip_flat_search_filtered_f32(lynse_hip_flat *h, const float *queries, uint64_t nq, uint32_t k,
                                       int metric, const uint64_t *subset_rows, uint64_t n_subset,
                                       uint64_t *out_rows, float *out_dists, uint32_t *out_counts);
/* Same with the subset as the reference's BitSet words (src/storage/bitset.rs:15-24: bit r of word r/64, LSB first;
 * SearchParams.subset, engine.rs:5541-5566).  Bits at or beyond len are ignored. */
int lynse_hip_flat_search_filtered_bitset_f32(lynse_hip_flat *h, const float *queries, uint64_t nq,
                                              uint32_t k, int metric, const uint64_t *bitset_words,
                                              uint64_t n_words, uint64_t *out_rows, float *out_dists,
                                              uint32_t *out_counts);
/* FLAT-*-SQ8 (`use_sq8`, src/storage/flat_mmap.rs:891-905, sq8_two_pass_search :5868-5926): pass 1 ranks ALL rows by the
 * integer score of their per-dimension u8 codes (u32 dot for ip, u32 squared L2 for l2 and cosine; SQ8Data :5676-5750,
 * built lazily like ensure_sq8 and rebuilt after appends) and keeps n_cand = max(20 k, 200) rows (cosine max(100 k, 500));
 * pass 2 rescores those exactly with the single-row f32 kernels and returns the best k.  Approximate by design: a true
 * neighbour outside the n_cand best codes is lost, exactly as in the reference.  Ties at the n_cand cut and between equal
 * exact distances are broken by row id (the reference leaves them to heap / unstable-sort order).  Other metrics take the
 * ordinary exact path, as `use_sq8` does. */
int lynse_hip_flat_search_sq8_f32(lynse_hip_flat *h, const float *queries, uint64_t nq, uint32_t k,
                                  int metric, uint64_t *out_rows, float *out_dists,
                                  uint32_t *out_counts);
/* The SQ8 quantiser state (per-dimension minimum and scale = 255 / range, 0 for a constant dimension). */
int lynse_hip_flat_sq8_params(lynse_hip_flat *h, float *mins, float *scales);
/* Same with every buffer already resident in this handle's device memory; enqueued on `stream`
 * (a hipStream_t, NULL = the handle's own non-blocking stream) and synchronised before returning.
 * Device inputs of every *_device entry must be COMPLETE when the call is made (synchronise the
 * stream that produced them): the handle's stream does not order against other streams. */
int lynse_hip_flat_search_f32_device(lynse_hip_flat *h, const float *d_queries, uint64_t nq,
                                     uint32_t k, int metric, uint64_t *d_out_rows,
                                     float *d_out_dists, uint32_t *d_out_counts, void *stream);
/* packed_binary_search (flat_mmap.rs:1345-1409) with pre-packed queries (nq * words u64). */
int lynse_hip_flat_search_packed_u64(lynse_hip_flat *h, const uint64_t *query_words, uint64_t nq,
                                     uint32_t k, int metric, uint64_t *out_rows, float *out_dists,
                                     uint32_t *out_counts);
int lynse_hip_flat_search_packed_u64_device(lynse_hip_flat *h, const uint64_t *d_query_words,
                                            uint64_t nq, uint32_t k, int metric,
                                            uint64_t *d_out_rows, float *d_out_dists,
                                            uint32_t *d_out_counts, void *stream);

/* Profiling: when enabled, searches bracket the scan kernel with HIP events on its stream.  on = n > 1 times every n-th
 * search only (an event recorded between two kernels costs a few microseconds of the stream's time). */
int lynse_hip_flat_profile_enable(lynse_hip_flat *h, int on);
int lynse_hip_flat_profile_get(lynse_hip_flat *h, lynse_hip_profile *out, int reset);
/* Coarse-pass state of a shard (no reference counterpart: the reference has one exact kernel, flat_mmap.rs:4845-4982;
 * here a certified low-precision pass runs in front of the exact rescoring and can be switched off by the data):
 * strikes = candidate overflows of the certified int8 pass so far (3 switch it off for the handle; -1 = off because the
 * rows are not finite), sq8_rows = rows covered by the SQ8 codes currently built (0 = none). */
int lynse_hip_flat_coarse_state(lynse_hip_flat *h, int *out_strikes, uint64_t *out_sq8_rows);
/* Rows covered by the +-1 byte copy of the packed rows that feeds batched Hamming searches (>= 96 queries) to the int8 MFMA
 * (packed_binary_search for a batch, flat_mmap.rs:1345-1409); 0 = not built (diagnostics / tests). */
uint64_t lynse_hip_flat_bpm_rows(const lynse_hip_flat *h);
/* Builds the derived copies a batch of nq queries of `metric` will read NOW instead of inside the first such search
 * (the reference's ensure_binary / ensure_sq8 are lazy as well, flat_mmap.rs:375-401): statistics + f16 shadow, SQ8 codes of
 * the certified int8 pass, packed words, the +-1 byte copy of batched Hamming.  lynse_hip_flat_hbm_bytes: HBM held by the
 * shard, source rows + derived copies. */
int lynse_hip_flat_prepare(lynse_hip_flat *h, int metric, uint64_t nq);
uint64_t lynse_hip_flat_hbm_bytes(const lynse_hip_flat *h);
/* Tuning knobs (defaults are fine): first-stage rows and growth factor of the contiguous stage plan (the fallback of the
 * default sampled plan), candidate capacity per query (power of two in [256, 16384], default 16384; k <= cap / 4 when the
 * shard holds more than cap rows). */
/* Few queries (<= 4) over a small shard (<= 64 MB of rows) with k <= 64 are answered by ONE fused launch
 * (k_small_search: exact scores straight from the f32 rows, per-wave top-k in registers, last-workgroup merge) instead
 * of the staged pipeline — the single-query latency path (flat_search_bench.py: 100k x 128, k = 10).  Results are
 * identical; on = 0 forces the staged pipeline (tests, A/B). */
int lynse_hip_flat_set_fused_search(lynse_hip_flat *h, int on);
int lynse_hip_flat_set_plan(lynse_hip_flat *h, uint32_t stage0_rows, uint32_t growth, uint32_t cap);

/* ---- stand-alone functions (src/python/mod.rs:2161-2223) ---- */
/* py_compute_distance -> distance::compute_distance_f32 (distance/mod.rs:193-213). */
int lynse_hip_compute_distance(const float *a, const float *b, uint32_t dim, int metric, int device,
                               float *out);
/* py_top_k_search -> distance::top_k_search (distance/mod.rs:373-422): u32 indices. */
int lynse_hip_top_k_search(const float *query, const float *candidates, uint64_t n, uint32_t dim,
                           uint32_t k, int metric, int device, uint32_t *out_idx, float *out_dist,
                           uint32_t *out_count);
/* pack_binary_f32 (simd.rs:760-763) for n rows on the device. */
int lynse_hip_pack_binary_f32(const float *rows, uint64_t n, uint32_t dim, int device,
                              uint64_t *out_words);
/* VectorStore::merge_results (vector_store.rs:953-970) / cluster::merge_search_blocks
 * (cluster.rs:327-393): host k-way merge of per-shard candidates, canonical (dist, id) order.
 * ids/dists: n_lists blocks of `stride` entries, counts[i] valid in block i. */
int lynse_hip_merge_topk(const uint64_t *ids, const float *dists, const uint32_t *counts,
                         uint32_t n_lists, uint32_t stride, uint32_t k, int metric,
                         uint64_t *out_ids, float *out_dists, uint32_t *out_count);

/* Device-side variant used after the RCCL all-gather of per-rank result blocks.  `blocks` holds
 * n_lists blocks of `block_bytes`; inside a block: rows u64[nq*k] at rows_off, dists f32[nq*k] at
 * dists_off, counts u32[nq] at counts_off (the message shape of rpc.rs:1156-1177, fixed size).
 * Output: merged rows u64[nq*k], dists f32[nq*k], counts u32[nq] in device memory.  Enqueued on
 * `stream` (hipStream_t; NULL = default stream) without synchronising. */
int lynse_hip_merge_topk_device(const void *d_blocks, uint64_t block_bytes, uint64_t rows_off,
                                uint64_t dists_off, uint64_t counts_off, uint32_t n_lists,
                                uint64_t nq, uint32_t k, int metric, uint64_t *d_out_rows,
                                float *d_out_dists, uint32_t *d_out_counts, void *stream);

/* ---- IVF-Flat: IVFIndex (src/index/ivf.rs) / IvfFlatMmap (src/storage/ivf_flat_mmap.rs) ---- */

/* Build from host rows with device k-means (kmeans.rs:74-139 semantics: FastRng(42) init,
 * Lloyd <= max_iter, empty-cluster reseed, final re-assign).  routing follows ivf.rs:81-87
 * (`l2_partitions` = 1 forces L2 Voronoi cells like IvfFlatMmap::build, ivf_flat_mmap.rs:98).
 * A binary metric (hamming / jaccard / dice / tanimoto) builds the IVF-*-BINARY mode
 * (src/index/mod.rs:376-385, ivf.rs:147-175): BinaryQuantizer::fit on the device (quantizer/mod.rs:321-356:
 * 0.5 for {0,1} corpora, else per-dimension median with midrange fallback), rows stored as their {0,1}
 * codes + packed u64 words, k-means and routing with L2 on the codes, packed popcount list scans. */
int lynse_hip_ivf_build(const float *rows, uint64_t n, uint32_t dim, uint32_t nlist,
                        uint32_t max_iter, int metric, int l2_partitions, int device,
                        lynse_hip_ivf **out);
/* Load given centroids (nlist*dim) + assignments (n): parity tests feed the oracle's. */
int lynse_hip_ivf_load(const float *rows, uint64_t n, uint32_t dim, const float *centroids,
                       uint32_t nlist, const uint32_t *assignments, int metric, int device,
                       lynse_hip_ivf **out);
/* Binary-metric load: `rows` are the RAW rows, binarised on the device with `thresholds` (dim floats,
 * BinaryQuantizer state) before the slabs are laid out. */
int lynse_hip_ivf_load_binary(const float *rows, uint64_t n, uint32_t dim, const float *centroids,
                              uint32_t nlist, const uint32_t *assignments, int metric,
                              const float *thresholds, int device, lynse_hip_ivf **out);
/* Fitted BinaryQuantizer state of a binary index: thresholds[dim], already_binary flag. */
int lynse_hip_ivf_thresholds(const lynse_hip_ivf *h, float *thresholds, int *already_binary);
/* IVFIndex::insert (src/index/ivf.rs:392-441): `rows` (n x dim f32; a binary index pushes them through its quantizer) are
 * assigned to the EXISTING centroids with the routing metric (every centroid in ascending order, strictly better wins) and
 * appended behind the rows already indexed (new row ids old_len .. old_len + n - 1); no retraining.
 * IVFIndex::delete (:350-390): the listed rows go, the rest keep their order under consecutive row ids and are all
 * reassigned to the existing centroids (kmeans::assign_metric).  lynse_hip_ivf_assign_f32 is the assignment rule alone. */
int lynse_hip_ivf_insert_f32(lynse_hip_ivf *h, const float *rows, uint64_t n);
int lynse_hip_ivf_delete_rows(lynse_hip_ivf *h, const uint64_t *row_ids, uint64_t n_ids);
int lynse_hip_ivf_assign_f32(lynse_hip_ivf *h, const float *rows, uint64_t n, uint32_t *out_assignments);
int lynse_hip_ivf_destroy(lynse_hip_ivf *h);
uint64_t lynse_hip_ivf_len(const lynse_hip_ivf *h);
uint32_t lynse_hip_ivf_nlist(const lynse_hip_ivf *h);
/* Trained state back to the host: centroids nlist*dim, assignments n, slab offsets nlist+1,
 * original row id per slab position (ivf_flat_mmap.rs:22-39).  NULL pointers are skipped. */
int lynse_hip_ivf_export(const lynse_hip_ivf *h, float *centroids, uint32_t *assignments,
                         uint64_t *offsets, uint32_t *original_ids);
int lynse_hip_ivf_set_row_map(lynse_hip_ivf *h, uint64_t stride, uint64_t offset);
/* Probe selection semantics: 0 = IVFIndex (rank every centroid, ivf.rs:227-249, empty-probe fallback
 * :258-265); 1 = IvfFlatMmap (IP 16-dim shortlist heuristic, ivf_flat_mmap.rs:381-421).  build() sets
 * it from `l2_partitions`, load() defaults to 0.  2 = keep the current routing but never fall back to the whole corpus
 * when the probed lists are empty: one ROW SHARD of a larger index must not answer from rows outside the probed lists
 * just because its own part of them is empty. */
int lynse_hip_ivf_set_routing(lynse_hip_ivf *h, int ivfflat_routing);
/* One to four queries (k <= 64, nprobe <= 64 < nlist, float metrics, exact centroid ranking, no subset) are answered by TWO
 * fused launches with no host round trip in between: k_small_search ranks the centroids, then scans the probed lists
 * straight from that ranking in device memory, every row scored exactly from the f32 slab — the reference's usual IVF
 * call is one query at a time (ivf.rs:181-348).  Same results as the staged path; on = 0 forces the staged path. */
int lynse_hip_ivf_set_fused_search(lynse_hip_ivf *h, int on);
/* IVFIndex::search (ivf.rs:181-348): rank all centroids with the routing metric, scan the nprobe
 * nearest lists, exact top-k of the probed rows.  nprobe == 0 -> 1 (ivf.rs:192-196). */
int lynse_hip_ivf_search_f32(lynse_hip_ivf *h, const float *queries, uint64_t nq, uint32_t k,
                             uint32_t nprobe, uint64_t *out_rows, float *out_dists,
                             uint32_t *out_counts);
/* Device-resident twins (float metrics): rows / queries / outputs already in HBM.  IvfFlatMmap::build reads its rows from an
 * mmapped store (src/storage/ivf_flat_mmap.rs:56-159); here k-means, the slab reordering (a gather inside HBM) and the
 * search never stage row data or candidates through host memory — one GPU's share of BASELINE config 4 is 19 GB.
 * Centroids and assignments (lynse_hip_ivf_load_device) are host arrays, as in lynse_hip_ivf_load. */
int lynse_hip_ivf_build_device(const float *d_rows, uint64_t n, uint32_t dim, uint32_t nlist,
                               uint32_t max_iter, int metric, int l2_partitions, int device,
                               lynse_hip_ivf **out);
int lynse_hip_ivf_load_device(const float *d_rows, uint64_t n, uint32_t dim, const float *centroids,
                              uint32_t nlist, const uint32_t *assignments, int metric, int device,
                              lynse_hip_ivf **out);
int lynse_hip_ivf_search_f32_device(lynse_hip_ivf *h, const float *d_queries, uint64_t nq, uint32_t k,
                                    uint32_t nprobe, uint64_t *d_out_rows, float *d_out_dists,
                                    uint32_t *d_out_counts);
/* IvfFlatMmap::search(query, k, nprobe, metric) (src/storage/ivf_flat_mmap.rs:225-305; PyIvfFlatIndex.search,
 * src/python/mod.rs:2130-2155): the metric of the CALL drives centroid routing, scoring and the sort direction (the
 * partitions are metric-agnostic).  Float indexes take ip / l2 / cosine; a binary index only its build metric. */
int lynse_hip_ivf_search_metric_f32(lynse_hip_ivf *h, const float *queries, uint64_t nq, uint32_t k,
                                    uint32_t nprobe, int metric, uint64_t *out_rows, float *out_dists,
                                    uint32_t *out_counts);
/* IVFIndex::search with SearchParams.subset (ivf.rs:251-265): the rows of the probed lists are intersected with
 * `subset_rows` (original row ids, host memory); a query whose probed lists hold no subset row is answered from the
 * whole corpus restricted to the subset, as the reference does.  One subset per batch. */
int lynse_hip_ivf_search_filtered_f32(lynse_hip_ivf *h, const float *queries, uint64_t nq, uint32_t k,
                                      uint32_t nprobe, const uint64_t *subset_rows, uint64_t n_subset,
                                      uint64_t *out_rows, float *out_dists, uint32_t *out_counts);
int lynse_hip_ivf_profile_enable(lynse_hip_ivf *h, int on);
int lynse_hip_ivf_profile_get(lynse_hip_ivf *h, lynse_hip_profile *out, int reset);

/* ---- multi-GPU exchange: one process per GPU, rows sharded, RCCL all-gather over xGMI ----
 *
 * Stands where the reference's TCP scatter / gather stands: fan-out of the batch to every shard (src/cluster.rs:101-123,
 * :173-217), the result block a shard returns (src/rpc.rs:1156-1177), the coordinator's merge (src/cluster.rs:327-393).
 * RCCL is bound at run time (dlopen; no link-time dependency): lynse_hip_comm_load_rccl(path) names the copy to use — a
 * host process that already carries one (PyTorch) passes its path so the process keeps ONE RCCL — else librccl.so.1 of
 * the loader path.  Bootstrap: rank 0 calls lynse_hip_comm_unique_id, the launcher hands the 128 bytes to every rank
 * (torch.distributed, MPI, a file), every rank calls lynse_hip_comm_create. */
typedef struct lynse_hip_comm lynse_hip_comm;
int lynse_hip_comm_load_rccl(const char *path /* NULL = search */);
int lynse_hip_comm_unique_id(uint8_t *id128);
int lynse_hip_comm_create(const uint8_t *id128, int rank, int world, int device, lynse_hip_comm **out);
/* A BLOCKING collective of a communicator that timed out (LYNSE_ERR_TIMEOUT: a rank is gone) leaves its all-gather / merge enqueued: the
 * communicator is marked failed — every later sharded call or submit on it returns LYNSE_ERR_DEVICE at once — and the output buffers of the
 * call that timed out must stay allocated until lynse_hip_comm_destroy. */
int lynse_hip_comm_destroy(lynse_hip_comm *c);
int lynse_hip_comm_rank(const lynse_hip_comm *c);
int lynse_hip_comm_world(const lynse_hip_comm *c);
/* Self-check: all-reduce(sum) of one word per rank; *out must equal the world size on every rank. */
int lynse_hip_comm_ranks_seen(lynse_hip_comm *c, int *out);
/* Whole-collection search of a row-sharded collection (a COLLECTIVE: every rank calls it with the same nq / k / metric):
 * scan of this rank's shard -> ncclAllGather of the fixed-size per-rank blocks [rows u64 | dists f32 | counts u32]
 * (nq*k*12 + nq*4 bytes) -> device k-way merge in the canonical (distance, global row) order, stream-ordered on the
 * shard's stream with no host synchronisation between the three.  Rows are global through the shard's row map
 * (lynse_hip_flat_set_row_map).  Outputs are device memory, identical on every rank, complete on return. */
int lynse_hip_flat_search_sharded_f32_device(lynse_hip_flat *h, lynse_hip_comm *c, const float *d_queries,
                                             uint64_t nq, uint32_t k, int metric, uint64_t *d_out_rows,
                                             float *d_out_dists, uint32_t *d_out_counts);
int lynse_hip_flat_search_sharded_packed_u64_device(lynse_hip_flat *h, lynse_hip_comm *c,
                                                    const uint64_t *d_query_words, uint64_t nq, uint32_t k,
                                                    int metric, uint64_t *d_out_rows, float *d_out_dists,
                                                    uint32_t *d_out_counts);
/* Row-sharded IVF search (BASELINE config 4): the local part of the probed lists -> ncclAllGather of the result blocks ->
 * device merge in the canonical (distance, global row) order, on the shard's stream.  Every rank holds its rows of every
 * list under the same centroids (lynse_hip_ivf_load_device with the shared centroids, lynse_hip_ivf_set_row_map for global
 * row ids, lynse_hip_ivf_set_routing(h, 2): no all-lists-empty fallback on a shard).  Stands where the reference fans an
 * index search out to its shard nodes and merges (src/cluster.rs:101-123, :173-217, :327-393).  A collective. */
int lynse_hip_ivf_search_sharded_f32_device(lynse_hip_ivf *h, lynse_hip_comm *c, const float *d_queries, uint64_t nq,
                                            uint32_t k, uint32_t nprobe, uint64_t *d_out_rows, float *d_out_dists,
                                            uint32_t *d_out_counts);
/* Row-sharded IVF TRAINING: k-means over the whole collection (src/index/kmeans.rs:74-139) with every rank holding only its rows
 * (global row g = local row g / world on rank g % world).  Same FastRng sample + farthest-first init, assignment, last-maximum /
 * empty-cluster rules and stop test as training on the union; the centroid sums are formed per rank (sequential over the rank's
 * members, kmeans.rs:273-286) and added over the ranks IN RANK ORDER, ((p0 + p1) + p2) + ... with f32 adds — the result is defined
 * bit for bit at every world size (an all-reduce associates as it likes from three ranks on).  Per Lloyd iteration: nlist * dim
 * floats per rank + nlist + 1 integer words.  Every rank gets the same centroids (out_centroids: nlist x dim, *out_k
 * lists trained) and the assignments of its rows; load the shard with lynse_hip_ivf_load / _load_device afterwards.  The
 * reduction: the library's communicator `c` (on device buffers: ncclAllGather of the per-rank sums + a rank-ordered add kernel +
 * one integer ncclAllReduce), or — c == NULL — the launcher's callback, which sums a HOST buffer of `count` elements over all
 * ranks in place (dtype 0: f32 in any order — only ever one non-zero contribution per element —, 1: u32, 2: f32 in RANK order as
 * above; 0 = success).  rows_on_device: rows_local is device memory.  A collective.  (No reference counterpart: the reference's
 * cluster mode trains one index per shard node.) */
typedef int (*lynse_hip_reduce_fn)(void *ctx, void *buf, uint64_t count, int dtype);
int lynse_hip_ivf_kmeans_sharded(const float *rows_local, uint64_t n_local, int rows_on_device, uint64_t n_global,
                                 uint32_t rank, uint32_t world, uint32_t dim, uint32_t nlist, uint32_t max_iter, int metric,
                                 int device, lynse_hip_comm *c, lynse_hip_reduce_fn reduce, void *reduce_ctx,
                                 float *out_centroids, uint32_t *out_assignments, uint32_t *out_k);

/* IVFIndex::build (src/index/ivf.rs:132-179) for ONE RANK of a row-sharded collection, in one call: lynse_hip_ivf_kmeans_sharded over the
 * rank's device-resident rows (global row g = local row g / world on rank g % world), then the rank's slab store under the shared
 * centroids with global row ids (lynse_hip_ivf_set_row_map(h, world, rank)) and the shard routing rule (no all-lists-empty fallback
 * when world > 1) — the index lynse_hip_ivf_search_sharded_f32_device and the IVF tickets search.  A collective; every rank needs at
 * least one row.  ivfflat_routing != 0: IvfFlatMmap semantics (L2 cells, src/storage/ivf_flat_mmap.rs:56-159). */
int lynse_hip_ivf_build_sharded_device(const float *d_rows_local, uint64_t n_local, uint64_t n_global, uint32_t rank,
                                       uint32_t world, uint32_t dim, uint32_t nlist, uint32_t max_iter, int metric,
                                       int ivfflat_routing, int device, lynse_hip_comm *c, lynse_hip_reduce_fn reduce,
                                       void *reduce_ctx, lynse_hip_ivf **out);

/* ---- searches in flight: submit / wait ----
 *
 * The reference serves concurrent readers (Arc<RwLock<Collection>> with inner.read() on the search path,
 * src/python/mod.rs:950, :1187; RPC / HTTP workers call search concurrently, src/rpc.rs:588-590): a batch does not wait
 * for the previous one to be answered.  submit enqueues a whole batch of <= 256 queries on one of the handle's search
 * contexts (stream + workspace; LYNSE_HIP_CONTEXTS of them) — the staged pipeline and, with a communicator, the
 * ncclAllGather of the result blocks and the device merge — and returns a ticket; wait blocks until that batch is final in
 * the caller's DEVICE arrays (identical to the blocking entry points: an overflowed plan is re-run on the next plan level
 * inside wait, on every rank of a sharded collection) and frees the ticket.  `c` = NULL: this shard only.  Queries and
 * outputs must stay valid until wait returns.  Tickets are waited for by the thread that submitted them, in any order;
 * append / finalize and the filtered searches need every ticket waited for first.  With a communicator submit is a
 * COLLECTIVE: every rank submits and waits for the same sequence.  A search that cannot be pipelined (more than 256
 * queries, k beyond one pass, the fused few-query search, derived data still to build) is answered inside submit. */
typedef struct lynse_hip_ticket lynse_hip_ticket;
int lynse_hip_flat_search_submit_f32_device(lynse_hip_flat *h, lynse_hip_comm *c, const float *d_queries, uint64_t nq,
                                            uint32_t k, int metric, uint64_t *d_out_rows, float *d_out_dists,
                                            uint32_t *d_out_counts, lynse_hip_ticket **out);
int lynse_hip_flat_search_submit_packed_u64_device(lynse_hip_flat *h, lynse_hip_comm *c,
                                                   const uint64_t *d_query_words, uint64_t nq, uint32_t k, int metric,
                                                   uint64_t *d_out_rows, float *d_out_dists, uint32_t *d_out_counts,
                                                   lynse_hip_ticket **out);
int lynse_hip_flat_search_wait(lynse_hip_ticket *t);
/* Upper bound of one lynse_hip_flat_search_wait in milliseconds, process-wide.  0 (default): tickets of ONE shard wait without a
 * limit, tickets that end in a collective (submitted with a communicator) give up after 30 s — a rank that died inside the
 * exchange would otherwise hang its peers forever (the reference's scatter-gather times its shard calls out the same way,
 * src/cluster.rs:173-217).  On expiry wait returns LYNSE_ERR_TIMEOUT; the ticket is freed, its context is NOT reused (the device
 * may still be writing into it).  LYNSE_HIP_WAIT_TIMEOUT_MS sets the initial value. */
int lynse_hip_set_wait_timeout_ms(uint32_t ms);

/* IVF searches in flight (IVFIndex::search, src/index/ivf.rs:181-348, under the same concurrent readers): submit ENQUEUES a batch of
 * <= 256 queries of a float IVF index on one of the slab store's search contexts 1 .. LYNSE_HIP_CONTEXTS-1 (context 0 stays with
 * the blocking searches, which keep working next to tickets) — query images, the centroid ranking, device-side grouping, the list
 * scans, exact rescoring and, with a communicator, the ncclAllGather of the result blocks + the device merge — with NO host
 * synchronisation in between; wait blocks until the batch is final in the caller's DEVICE arrays and frees the ticket.  Results
 * are those of lynse_hip_ivf_search_f32_device / lynse_hip_ivf_search_sharded_f32_device: whatever the blocking path would have
 * checked on the host (candidate overflow of the scans or of the centroid ranking, the all-lists-empty fallback of ivf.rs:258-265)
 * travels as a status word and is re-answered by the blocking ladder inside wait, on every rank of a sharded index.  The index
 * metric is used.  A shape the device-side grouping does not take (nprobe >= nlist, more than 16384 (query, list) pairs, more
 * than 8192 lists) is answered synchronously inside submit (a sharded ticket still exchanges in flight); binary indexes and k
 * beyond the staged pipeline return LYNSE_ERR_UNSUPPORTED.  insert / delete are refused while tickets are outstanding.  With a
 * communicator submit is a COLLECTIVE: every rank submits and waits for the same sequence; one communicator serves the tickets of
 * ONE handle at a time.  lynse_hip_set_wait_timeout_ms applies. */
typedef struct lynse_hip_ivf_ticket lynse_hip_ivf_ticket;
int lynse_hip_ivf_search_submit_f32_device(lynse_hip_ivf *h, lynse_hip_comm *c, const float *d_queries, uint64_t nq, uint32_t k,
                                           uint32_t nprobe, uint64_t *d_out_rows, float *d_out_dists, uint32_t *d_out_counts,
                                           lynse_hip_ivf_ticket **out);
int lynse_hip_ivf_search_wait(lynse_hip_ivf_ticket *t);
/* out[0..2]: tickets of this index whose local part was enqueued without a host synchronisation / answered inside submit /
 * re-answered inside wait (monitoring; that a batch ran in flight is not visible in its results). */
int lynse_hip_ivf_ticket_stats(lynse_hip_ivf *h, uint64_t *out);

/* Diagnostics of the certified coarse pass (no reference counterpart: the reference scans f32 rows).  For a shard of at most `cap`
 * (16,384) rows and 1..256 host queries: out_scores[q][row] = the COARSE score of (row, query) exactly as the scan kernels compute
 * it (one emit-all stage of the real pipeline), in the metric's own space (IP: score; L2 / cosine: distance); out_bound[q] = the
 * bound E the pipeline certified for |coarse - reference-order f32 score| (the margin it keeps is 2E).  coarse = 0: the f16 shadow,
 * 1: the certified int8 pass; *out_form (may be NULL): bit 0 int8, bit 1 augmented-L2 codes, bit 2 plain-code L2 (exact f32 row
 * norms), bit 3 unit-row cosine codes.  tests/test_gpu_certificate.py holds the bound against constructed worst cases. */
int lynse_hip_flat_coarse_scores(lynse_hip_flat *h, const float *queries, uint64_t nq, int metric, int coarse,
                                 float *out_scores, float *out_bound, int *out_form);

/* ---- shard-node glue around a search (host only, no device work) ---- */

/* Collection::filter_tombstoned_limit (src/engine.rs:3286-3308): drop the tombstoned ids, keep the order, at most
 * `limit` pairs.  Outputs hold min(n, limit) entries. */
int lynse_hip_filter_tombstoned_limit(const uint64_t *ids, const float *dists, uint64_t n,
                                      const uint64_t *tombstones, uint64_t n_tombstones, uint64_t limit,
                                      uint64_t *out_ids, float *out_dists, uint64_t *out_n);
/* Collection::merge_row_results (src/engine.rs:3363-3418): flushed + pending rows.  Duplicate ids keep the better
 * distance; order (distance in metric order, id); truncated to `limit` — except that an empty side returns the other
 * side untouched, as the reference does.  Outputs hold n_left + n_right entries. */
int lynse_hip_merge_row_results(const uint64_t *left_ids, const float *left_dists, uint64_t n_left,
                                const uint64_t *right_ids, const float *right_dists, uint64_t n_right,
                                uint64_t limit, int metric, uint64_t *out_ids, float *out_dists,
                                uint64_t *out_n);
/* encode_search_result_binary (src/rpc.rs:1156-1177): [u32 n][n x u64 id][n x f32 distance][u32 fields_len][fields
 * JSON], little endian — the block a shard returns to the coordinator.  *out_len = bytes needed / written. */
int lynse_hip_encode_search_result(const uint64_t *ids, const float *dists, uint32_t n,
                                   const uint8_t *fields_json, uint32_t fields_len, uint8_t *buf,
                                   uint64_t cap, uint64_t *out_len);
/* decode_search_result_binary (src/cluster.rs:404-435) at `offset` of a frame that may hold several blocks. */
int lynse_hip_decode_search_result(const uint8_t *buf, uint64_t len, uint64_t offset, uint64_t *out_ids,
                                   float *out_dists, uint32_t cap_n, uint32_t *out_n,
                                   uint64_t *fields_offset, uint32_t *fields_len, uint64_t *next_offset);

#ifdef __cplusplus
}
#endif
#endif /* LYNSE_HIP_H */
