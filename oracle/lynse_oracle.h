/*
 * lynse_oracle.h — CPU restatement of the LynseDB FLAT / IVF-Flat search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and there only as the checker / timed CPU baseline.
 *
 * Every function cites the reference file:line (relative to the upstream
 * BirchKwok/lynsedb tree) whose arithmetic and ordering it restates.  The
 * reference is Rust and cannot be built here (no cargo/rustc); parity is
 * pinned against the known-answer tests the reference's own test-suite holds
 * (tests/test_oracle_kat.py).
 */
#ifndef LYNSE_ORACLE_H
#define LYNSE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Metric ids (src/distance/mod.rs:19-36, in-scope subset). */
enum {
    LO_IP = 0,       /* InnerProduct: descending */
    LO_L2 = 1,       /* L2Squared */
    LO_COS = 2,      /* Cosine distance */
    LO_HAMMING = 3,
    LO_JACCARD = 4,
    LO_DICE = 5,
    LO_TANIMOTO = 6  /* == Jaccard arithmetic (distance/mod.rs:207) */
};

/* IP accumulation form used by a row. */
enum {
    LO_IPFORM_AUTO = 0,   /* n < 4096 -> single, else batch8 (flat_mmap.rs:4852, :2191) */
    LO_IPFORM_SINGLE = 1, /* simd.rs:1343-1396, two 8-lane accumulators */
    LO_IPFORM_BATCH8 = 2  /* simd.rs:1452-1525, one 8-lane accumulator */
};

int lo_metric_is_ascending(int metric);   /* distance/mod.rs:111-116 */
int lo_metric_is_binary(int metric);      /* distance/mod.rs:161-166 */
int lo_metric_from_str(const char *s);    /* distance/mod.rs:39-63 (in-scope aliases); -1 if unknown */
int lo_metric_from_index_mode(const char *s); /* distance/mod.rs:67-107; -1 if unknown */
int lo_has_avx2_fma(void);                /* 1 when compiled with the AVX2+FMA intrinsic path */

/* ---- distance kernels (src/distance/simd.rs) ---- */
float lo_ip_single(const float *a, const float *b, size_t n);        /* :1343-1396 */
void lo_ip_batch8(const float *q, const float *v0, const float *v1, const float *v2,
                  const float *v3, const float *v4, const float *v5, const float *v6,
                  const float *v7, size_t n, float out[8]);          /* :1452-1525 */
float lo_ip_batch8_row(const float *q, const float *v, size_t n);    /* one row of the batch8 form */
float lo_l2_single(const float *a, const float *b, size_t n);        /* :1529-1581 */
float lo_cos_single(const float *a, const float *b, size_t n);       /* :1585-1636 */
float lo_ip_scalar(const float *a, const float *b, size_t n);        /* :1223-1250 (f64 accumulate) */
float lo_l2_scalar(const float *a, const float *b, size_t n);        /* :1296-1317 */
float lo_cos_scalar(const float *a, const float *b, size_t n);       /* :1319-1337 */
float lo_hamming_f32(const float *a, const float *b, size_t n);      /* :175-187 */
float lo_jaccard_f32(const float *a, const float *b, size_t n);      /* :190-209 */
float lo_dice_f32(const float *a, const float *b, size_t n);         /* :717-736 */
void lo_pack_binary_f32(const float *src, size_t dim, uint64_t *words); /* :750-763, threshold 0.5 */
float lo_packed_hamming(const uint64_t *a, const uint64_t *b, size_t words); /* :766-771 */
float lo_packed_jaccard(const uint64_t *a, const uint64_t *b, size_t words); /* :774-786 */
float lo_packed_dice(const uint64_t *a, const uint64_t *b, size_t words);    /* :789-801 */
float lo_compute_distance(const float *a, const float *b, size_t n, int metric); /* distance/mod.rs:193-213 */

/* ---- top-k (src/distance/mod.rs, src/storage/flat_mmap.rs) ---- */

/* distance::top_k_search (distance/mod.rs:373-422): all distances -> quickselect
 * (median-of-3 Lomuto, :304-352) -> sort of the k survivors.  Returns count. */
size_t lo_top_k_search(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                       int metric, uint32_t *out_idx, float *out_dist);

/* FlatMmap::search exact f32 path (flat_mmap.rs:824-923 -> :1173-1230 ->
 * :4845-5044, :2132-2256, :5183-5214) with the reference's chunking policy for
 * `n_threads` rayon threads (chunks are evaluated in order; result is
 * independent of real parallelism).  Binary metrics go through the packed
 * path (:839-845 -> :1345-1409).  Returns count = min(k, n). */
size_t lo_flat_search(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                      int metric, int n_threads, uint32_t *out_idx, float *out_dist);

/* Same scan executed by `n_threads` worker threads pulling chunks (the rayon
 * par_chunks schedule); used as the timed CPU baseline. */
size_t lo_flat_search_mt(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                         int metric, int n_threads, uint32_t *out_idx, float *out_dist);

/* packed_binary_search (flat_mmap.rs:1345-1409) over pre-packed rows. */
size_t lo_packed_binary_search(const uint64_t *query, const uint64_t *rows, size_t words, size_t n,
                               size_t k, int metric, int n_threads, uint32_t *out_idx,
                               float *out_dist);
size_t lo_packed_binary_search_mt(const uint64_t *query, const uint64_t *rows, size_t words,
                                  size_t n, size_t k, int metric, int n_threads,
                                  uint32_t *out_idx, float *out_dist);

/* Persistent worker pool for the TIMED CPU baseline (bench.py cpu_baseline): the reference scans on rayon's global pool
 * (created once; RAYON_NUM_THREADS), not on threads spawned per query.  While a pool of exactly `n_threads` workers
 * runs, lo_flat_search_mt / lo_packed_binary_search_mt(.., n_threads, ..) use it (chunk ci -> worker ci % n); results
 * are unchanged.  lo_fill_uniform_mt first-touches an untouched buffer with the same chunking (NUMA-local pages). */
int lo_pool_start(int n_threads);
void lo_pool_stop(void);
int lo_fill_uniform_mt(float *dst, size_t n, size_t dim, uint64_t seed);

/* Canonical exact top-k: every row scored with the reference kernel
 * (ip_form selects the IP accumulation form), total order (distance in metric
 * order, then row ascending) — the order VectorStore::merge_results imposes
 * (vector_store.rs:953-970).  This is what the HIP path must reproduce. */
size_t lo_canonical_topk(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                         int metric, int ip_form, uint32_t *out_idx, float *out_dist);
size_t lo_canonical_topk_packed(const uint64_t *query, const uint64_t *rows, size_t words,
                                size_t n, size_t k, int metric, uint32_t *out_idx,
                                float *out_dist);
/* All n distances with the reference kernels (row-wise), for property tests. */
void lo_all_distances(const float *query, const float *cands, size_t dim, size_t n, int metric,
                      int ip_form, float *out);

/* VectorStore::merge_results (vector_store.rs:953-970) / engine merge
 * (engine.rs:3402-3414): sort (dist by metric order, id asc), truncate. */
/* SQ8 two-pass FLAT (flat_mmap.rs:5676-5926): fit / quantize / canonical two-pass search */
void lo_sq8_fit(const float *data, size_t n, size_t dim, float *mins, float *scales);
void lo_sq8_quantize(const float *data, size_t n, size_t dim, const float *mins, const float *scales, uint8_t *out);
size_t lo_sq8_search_canonical(const float *query, const float *cands, const uint8_t *codes, const float *mins,
                               const float *scales, size_t dim, size_t n, size_t k, int metric,
                               uint32_t *out_idx, float *out_dist);
/* f16 storage (VectorDtype::F16): sequential-sum kernels of simd.rs:805-846 on decoded rows */
float lo_distance_f16(const float *query, const float *cand, size_t dim, int metric);
void lo_round_f16(const float *in, size_t n, float *out);
void lo_f16_bits(const float *in, size_t n, uint16_t *out);
size_t lo_canonical_topk_f16(const float *query, const float *cands_decoded, size_t dim, size_t n, size_t k,
                             int metric, uint32_t *out_idx, float *out_dist);
/* FlatMmap::search_filtered (flat_mmap.rs:491-815) and its canonical (set, (distance,row) order) answer */
size_t lo_flat_search_filtered(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                               int metric, const uint64_t *subset, size_t m, int n_threads,
                               uint32_t *out_idx, float *out_dist);
size_t lo_packed_search_filtered(const uint64_t *query, const uint64_t *rows, size_t words, size_t n,
                                 size_t k, int metric, const uint64_t *subset, size_t m,
                                 uint32_t *out_idx, float *out_dist);
size_t lo_canonical_topk_filtered(const float *query, const float *cands, const uint64_t *packed_query,
                                  const uint64_t *packed_rows, size_t words, size_t dim, size_t n,
                                  size_t k, int metric, const uint64_t *subset, size_t m,
                                  uint32_t *out_idx, float *out_dist);
size_t lo_merge_results(const uint64_t *ids, const float *dists, size_t n, size_t k, int metric,
                        uint64_t *out_ids, float *out_dists);

/* ---- k-means + IVF (src/index/kmeans.rs, src/index/ivf.rs, src/storage/ivf_flat_mmap.rs) ---- */

/* kmeans::train_for_metric (kmeans.rs:74-139) incl. kmeans_pp_init_metric
 * (:141-196, FastRng seed 42 :21-48).  Centroid sums are accumulated in row
 * order (the n<8192 branch of :266-286; the rayon fold order of the large
 * branch is not deterministic in the reference either).  Returns n_centroids. */
size_t lo_kmeans_train(const float *data, size_t n, size_t dim, size_t requested, size_t max_iter,
                       int metric, float *centroids /* requested*dim */,
                       uint32_t *assignments /* n */);
/* kmeans_train over the union of `world` row shards with the centroid sums formed per shard and added in rank order (the
 * all-reduced training of a row-sharded IVF collection; see the .c file). */
size_t lo_kmeans_train_sharded(const float *data, size_t n, size_t dim, size_t requested, size_t max_iter,
                               int metric, size_t world, float *centroids, uint32_t *assignments);
void lo_kmeans_assign(const float *data, size_t n, size_t dim, const float *centroids,
                      size_t n_centroids, int metric, uint32_t *assignments); /* :237-264 */
/* FastRng stream (kmeans.rs:21-35) for KATs. */
void lo_fastrng_stream(uint64_t seed, size_t count, double *out);

/* IVFIndex::search (ivf.rs:181-348), QuantizerType::None / packed-binary:
 * lists are given as CSR (list_offsets[nlist+1], list_rows ascending per list,
 * kmeans.rs:317-345).  `packed` may be NULL for float metrics.  Final order is
 * canonical (distance, row id), which equals the reference for distinct
 * distances (its sort_unstable leaves ties unpinned).  Returns count. */
size_t lo_ivf_search_filtered(const float *query, const float *data, const uint64_t *packed, size_t words,
                              size_t dim, size_t n, const float *centroids, size_t nlist,
                              const uint64_t *list_offsets, const uint32_t *list_rows, size_t nprobe,
                              size_t k, int metric, const uint64_t *subset, size_t m, uint64_t *out_ids,
                              float *out_dist);
size_t lo_ivf_search(const float *query, const float *data, const uint64_t *packed, size_t words,
                     size_t dim, size_t n, const float *centroids, size_t nlist,
                     const uint64_t *list_offsets, const uint32_t *list_rows, size_t nprobe,
                     size_t k, int metric, uint64_t *out_ids, float *out_dist,
                     uint32_t *out_probed /* nprobe, may be NULL */);

/* BinaryQuantizer::fit (src/quantizer/mod.rs:321-357): returns 1 when the corpus is already {0,1}
 * (threshold 0.5 everywhere), else fills per-dimension thresholds: the median column[n/2], or the
 * midrange when the median equals the column min or max. */
int lo_binary_fit(const float *data, size_t n, size_t dim, float *thresholds /* dim */);
/* decode(encode(x)) (quantizer/mod.rs:359-393): out[i][d] = x[i][d] > thresholds[d] ? 1 : 0. */
void lo_binary_quantize(const float *data, size_t n, size_t dim, const float *thresholds, float *out);

/* IvfFlatMmap::build step 2-4 (ivf_flat_mmap.rs:105-130): slab layout from assignments. */
void lo_ivf_flat_layout(const uint32_t *assignments, size_t n, size_t nlist,
                        uint64_t *offsets /* nlist+1 */, uint32_t *original_ids /* n */);
/* select_routing_dims (ivf_flat_mmap.rs:316-348); returns count (0 or 16). */
size_t lo_ivf_routing_dims(const float *centroids, size_t dim, size_t nlist, uint32_t *out_dims);
/* find_nearest_centroids (ivf_flat_mmap.rs:381-444) — canonical tie order
 * (score, centroid id) in place of select_nth_unstable. Returns count. */
size_t lo_ivf_flat_probe(const float *query, const float *centroids, size_t dim, size_t nlist,
                         size_t nprobe, int metric, const uint32_t *routing_dims,
                         size_t n_routing, uint32_t *out_parts);
/* IvfFlatMmap::search (ivf_flat_mmap.rs:225-304) over the reordered slab data. */
size_t lo_ivf_flat_search(const float *query, const float *slab_data, size_t dim, size_t n,
                          const float *centroids, size_t nlist, const uint64_t *offsets,
                          const uint32_t *original_ids, const uint32_t *routing_dims,
                          size_t n_routing, size_t nprobe, size_t k, int metric,
                          uint32_t *out_ids, float *out_dist);

#ifdef __cplusplus
}
#endif
#endif
