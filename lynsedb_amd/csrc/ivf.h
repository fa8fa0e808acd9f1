// ivf.h — IVF-Flat kernels: centroid routing (heuristic mode; the exact mode is a FLAT search over the centroids),
// per-group query image gather, k-means helpers, BinaryQuantizer fit, tiled packed-binary list scan.
// The float slab scan itself is k_scan_h16 in work-list (TILED) mode (kernels.h).
#pragma once
#include "kernels.h"

namespace lynse {

// ------------------------------------------------------------------------------------------------
// k_ivf_route: one block per query.  Ranks centroids with the routing metric using the reference's
// single-row kernels (compute_distance_f32, distance/mod.rs:193-213) and a total (score, centroid id)
// order — the stable sort_by of IVFIndex::search (ivf.rs:227-241).
//   mode 0: rank ALL centroids exactly, take nprobe                       (ivf.rs:227-249)
//   mode 1: IvfFlatMmap IP routing heuristic (ivf_flat_mmap.rs:381-421): coarse score on the 16
//           highest-variance dims -> shortlist (strictly-greater replacement == (score desc, id asc)
//           top-S) -> exact score of the shortlist -> top nprobe.
// ------------------------------------------------------------------------------------------------
struct RouteArgs {
    const float* Qf;
    uint32_t D;
    const float* C;
    uint32_t ldc, nlist;
    int metric;
    uint32_t nprobe;
    int mode;
    const uint32_t* rdims;
    uint32_t nrd, shortlist;
    uint32_t* probes;  // nq x nprobe
};

template <int NT>
__global__ void __launch_bounds__(NT) k_ivf_route(RouteArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x, g = tid & 7;
    const uint32_t grp = tid >> 3;
    const float* qv = a.Qf + (size_t)q * a.D;
    const bool asc = metric_ascending(a.metric);
    const uint32_t np2 = next_pow2(a.nlist < 2 ? 2 : a.nlist);
    if (a.mode == 0) {
        const uint32_t rounds = (a.nlist + NT / 8 - 1) / (NT / 8);
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t c = r * (NT / 8) + grp;
            if (c < a.nlist) {
                const float s = exact_score(a.metric, LYNSE_IPFORM_SINGLE, qv, a.C + (size_t)c * a.ldc, a.D, g);
                if (g == 0) keys[c] = make_key(s, c, asc);
            }
        }
        for (uint32_t i = a.nlist + tid; i < np2; i += NT) keys[i] = KEY_SENTINEL;
        bitonic_sort_lds<NT>(keys, np2, tid);
    } else {
        for (uint32_t c = tid; c < np2; c += NT) {
            if (c < a.nlist) {
                const float* cen = a.C + (size_t)c * a.ldc;
                float sc = 0.0f;
                for (uint32_t r = 0; r < a.nrd; ++r) sc = __fadd_rn(sc, __fmul_rn(qv[a.rdims[r]], cen[a.rdims[r]]));
                keys[c] = make_key(sc, c, false);
            } else {
                keys[c] = KEY_SENTINEL;
            }
        }
        bitonic_sort_lds<NT>(keys, np2, tid);
        const uint32_t S = a.shortlist < a.nlist ? a.shortlist : a.nlist;
        const uint32_t rounds = (S + NT / 8 - 1) / (NT / 8);
        uint64_t mine[8];  // rounds <= 8 for S <= 96 and NT >= 128 (asserted on the host)
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t i = r * (NT / 8) + grp;
            mine[r] = KEY_SENTINEL;
            if (i < S) {
                const uint32_t c = key_row(keys[i]);
                const float s = exact_score(M_IP, LYNSE_IPFORM_SINGLE, qv, a.C + (size_t)c * a.ldc, a.D, g);
                mine[r] = make_key(s, c, false);
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < np2; i += NT) keys[i] = KEY_SENTINEL;
        __syncthreads();
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t i = r * (NT / 8) + grp;
            if (i < S && g == 0) keys[i] = mine[r];
        }
        bitonic_sort_lds<NT>(keys, np2, tid);
    }
    const uint32_t np = a.nprobe < a.nlist ? a.nprobe : a.nlist;
    for (uint32_t r = tid; r < a.nprobe; r += NT) a.probes[(size_t)q * a.nprobe + r] = r < np ? key_row(keys[r]) : 0xffffffffu;
}

// ------------------------------------------------------------------------------------------------
// k_ivf_gather_q: builds the f16 query image of every (list, query-group) from the per-batch image
// written by k_prep_queries (layout 1).  One thread per (pair, slab, 16-B slot).
// ------------------------------------------------------------------------------------------------
struct GatherPair {
    uint32_t q;        // query index in the batch
    uint32_t dst_off;  // group image offset in halves
    uint32_t n;        // local index inside the group (< 32)
};

__global__ void __launch_bounds__(256) k_ivf_gather_q(const _Float16* __restrict__ base, uint32_t qpad,
                                                      _Float16* __restrict__ gimg, const GatherPair* __restrict__ pairs,
                                                      uint32_t npairs, uint32_t nslab, uint32_t slots) {
    // slots = 16-B slots per query per slab: 4 (layout 1, swizzle (q>>2)&3) or 8 (layout 2, swizzle (q>>1)&7)
    const uint64_t total = (uint64_t)npairs * nslab * slots;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = (uint32_t)(i % slots);
        const uint32_t s = (uint32_t)((i / slots) % nslab);
        const uint32_t p = (uint32_t)((i / slots) / nslab);
        const GatherPair gp = pairs[p];
        const uint32_t sw_src = slots == 4 ? ((gp.q >> 2) & 3) : ((gp.q >> 1) & 7);
        const uint32_t sw_dst = slots == 4 ? ((gp.n >> 2) & 3) : ((gp.n >> 1) & 7);
        const u32x4* src = reinterpret_cast<const u32x4*>(base + (((size_t)s * qpad + gp.q) * slots + (l ^ sw_src)) * 8);
        u32x4* dst = reinterpret_cast<u32x4*>(gimg + gp.dst_off + (((size_t)s * 32 + gp.n) * slots + (l ^ sw_dst)) * 8);
        *dst = *src;
    }
}

// ------------------------------------------------------------------------------------------------
// Device-side grouping of a batch's (query, list) pairs (the host loop of ivf_search_chunk_staged, on the device): pairs
// grouped by list (a counting sort over the lists with LDS atomics: the order of the queries inside a list is whatever the
// atomics make it — results do not depend on it, every candidate goes through the canonical select), lists cut into groups of
// at most 32 queries (one 32-query image each), and the tile lists of the three position windows [0, a0), [a0, 16 a0),
// [16 a0, ..) of every probed list.  Everything the scan launches need stays in device memory: the tile counts are read by
// k_scan_h16 through ScanArgs::ntiles_dev, the group count by k_ivf_gather_groups / k_ivf_emit_tiles.  One workgroup: at most
// IVF_GROUP_MAX_PAIRS pairs and IVF_GROUP_MAX_LISTS lists (its LDS); anything larger keeps the host path.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t IVF_GROUP_MAX_PAIRS = 16384;
constexpr uint32_t IVF_GROUP_MAX_LISTS = 8192;

struct IvfGroup {
    uint32_t list, pair0, nq, qimg_off;   // qimg_off in halves (16-bit units), like IvfTile
};

struct IvfGroupArgs {
    const uint64_t* probes64;  // [nq][np]: the centroid ranking as row ids of the centroid store (exact routing) — or NULL
    const uint32_t* probes32;  // [nq][np]: k_ivf_route's output
    uint32_t nq, np, nlist;
    const uint64_t* offsets;   // [nlist + 1] slab offsets of the lists
    uint32_t gimg_halves;      // halves per group image
    uint32_t a0, tile_rows;
    uint32_t win_cap;          // tiles reserved per window in `tiles`
    uint32_t flag_empty;       // IVFIndex: a query whose probed lists are all empty scans EVERY list (ivf.rs:258-265): flagged, host path
    uint32_t* pair_q;          // out [pairs]: the query of pair slot i (slots grouped by list)
    IvfGroup* groups;          // out [groups]
    uint32_t* prank;           // scratch [pairs]: rank of pair i inside its list (global memory: the LDS holds the per-list arrays)
    uint32_t* tbase;           // out [3][gmax]: first tile of group g in window w
    uint32_t gmax;
    IvfTile* tiles;            // out [3][win_cap] (k_ivf_emit_tiles)
    uint32_t* hdr;             // out: [0] groups, [1..3] tiles of window 0..2, [4] host path needed, [5] pairs, [6] rows of all tiles (profiling)
};

// inclusive scan of v[0, n) in place by one workgroup of NT threads; tot: NT / 64 words of LDS
template <int NT>
__device__ inline void ivf_block_scan(uint32_t* v, uint32_t n, uint32_t* tot) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ch = (n + NT - 1) / NT;
    const uint32_t b = tid * ch < n ? tid * ch : n, e = b + ch < n ? b + ch : n;
    uint32_t acc = 0;
    for (uint32_t i = b; i < e; ++i) { acc += v[i]; v[i] = acc; }
    uint32_t inc = acc;   // wave-level inclusive scan of the thread totals
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64);
        if ((int)lane >= off) inc += t;
    }
    if (lane == 63) tot[wave] = inc;
    __syncthreads();
    uint32_t wpre = 0;
    for (uint32_t x = 0; x < wave; ++x) wpre += tot[x];
    const uint32_t pre = wpre + inc - acc;
    for (uint32_t i = b; i < e; ++i) v[i] += pre;
    __syncthreads();
}

template <int NT>
__global__ void __launch_bounds__(NT) k_ivf_group(IvfGroupArgs a) {
    extern __shared__ __attribute__((aligned(16))) char ivf_smem[];
    const uint32_t P = a.nq * a.np, NL = a.nlist;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(ivf_smem);   // [NL] pairs per list
    uint32_t* pinc = cnt + NL;                                // [NL] inclusive scan of cnt; later tiles per group (<= P entries)
    uint32_t* ginc = pinc + (NL > P ? NL : P);                // [NL] inclusive scan of the groups per list
    uint32_t* prank = a.prank;                                // [P]  rank of pair i inside its list, ~0 = no pair
    uint32_t* tot = ginc + NL;                                // [NT / 64]
    uint32_t* qcnt = tot + NT / 64;                           // [256] valid pairs per query
    __shared__ uint32_t s_flag;
    const uint32_t tid = threadIdx.x;
    for (uint32_t c = tid; c < NL; c += NT) cnt[c] = 0;
    if (tid < 256) qcnt[tid] = 0;
    if (tid == 0) s_flag = 0;
    __syncthreads();
    for (uint32_t i = tid; i < P; i += NT) {
        const uint64_t c64 = a.probes64 ? a.probes64[i] : (uint64_t)a.probes32[i];
        uint32_t r = 0xffffffffu;
        if (c64 < NL && a.offsets[c64 + 1] > a.offsets[c64]) {
            r = atomicAdd(&cnt[(uint32_t)c64], 1u);
            atomicAdd(&qcnt[i / a.np], 1u);
        }
        prank[i] = r;
    }
    __syncthreads();
    if (a.flag_empty && tid < a.nq && qcnt[tid] == 0) s_flag = 1;
    for (uint32_t c = tid; c < NL; c += NT) { pinc[c] = cnt[c]; ginc[c] = (cnt[c] + 31) / 32; }
    __syncthreads();
    ivf_block_scan<NT>(pinc, NL, tot);
    ivf_block_scan<NT>(ginc, NL, tot);
    const uint32_t nv = NL ? pinc[NL - 1] : 0u, G = NL ? ginc[NL - 1] : 0u;
    for (uint32_t i = tid; i < P; i += NT) {
        const uint32_t r = prank[i];
        if (r != 0xffffffffu) {
            const uint32_t c = (uint32_t)(a.probes64 ? a.probes64[i] : (uint64_t)a.probes32[i]);
            a.pair_q[pinc[c] - cnt[c] + r] = i / a.np;
        }
    }
    for (uint32_t c = tid; c < NL; c += NT) {
        const uint32_t m = cnt[c];
        if (m) {
            const uint32_t p0 = pinc[c] - m, gb = ginc[c] - (m + 31) / 32;
            for (uint32_t j = 0; j * 32 < m; ++j)
                a.groups[gb + j] = {c, p0 + 32 * j, m - 32 * j < 32u ? m - 32 * j : 32u, (gb + j) * a.gimg_halves};
        }
    }
    __syncthreads();
    // first tile of every group in each of the three position windows (groups in list order inside a window)
    uint32_t* tcnt = pinc;   // [G <= P]
    for (int w = 0; w < 3; ++w) {
        const uint64_t wlo = w == 0 ? 0ull : (w == 1 ? (uint64_t)a.a0 : (uint64_t)a.a0 * 16);
        const uint64_t whi = w == 0 ? (uint64_t)a.a0 : (w == 1 ? (uint64_t)a.a0 * 16 : ~0ull);
        for (uint32_t g = tid; g < G; g += NT) {
            const uint32_t c = a.groups[g].list;
            const uint64_t len = a.offsets[c + 1] - a.offsets[c];
            const uint64_t lo = wlo < len ? wlo : len, hi = whi < len ? whi : len;
            tcnt[g] = (uint32_t)((hi - lo + a.tile_rows - 1) / a.tile_rows);
        }
        __syncthreads();
        ivf_block_scan<NT>(tcnt, G, tot);
        for (uint32_t g = tid; g < G; g += NT) a.tbase[(size_t)w * a.gmax + g] = g ? tcnt[g - 1] : 0u;
        if (tid == 0) {
            const uint32_t total = G ? tcnt[G - 1] : 0u;
            a.hdr[1 + w] = total < a.win_cap ? total : a.win_cap;
            if (total > a.win_cap) s_flag = 1;   // (the host's bound makes this unreachable; never scan a truncated list silently)
        }
        __syncthreads();
    }
    if (tid == 0) { a.hdr[0] = G; a.hdr[4] = s_flag; a.hdr[5] = nv; a.hdr[6] = 0u; }
}

// the tile records of every (window, group): one thread each
__global__ void __launch_bounds__(256) k_ivf_emit_tiles(IvfGroupArgs a) {
    const uint32_t G = a.hdr[0];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)G * 3; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = (uint32_t)(i / G), g = (uint32_t)(i % G);
        const uint64_t wlo = w == 0 ? 0ull : (w == 1 ? (uint64_t)a.a0 : (uint64_t)a.a0 * 16);
        const uint64_t whi = w == 0 ? (uint64_t)a.a0 : (w == 1 ? (uint64_t)a.a0 * 16 : ~0ull);
        const IvfGroup gr = a.groups[g];
        const uint64_t b0 = a.offsets[gr.list], len = a.offsets[gr.list + 1] - b0;
        const uint64_t lo = wlo < len ? wlo : len, hi = whi < len ? whi : len;
        uint32_t t = a.tbase[(size_t)w * a.gmax + g];
        if (hi > lo) atomicAdd(&a.hdr[6], (uint32_t)(hi - lo));
        for (uint64_t r = lo; r < hi; r += a.tile_rows, ++t)
            if (t < a.win_cap)
                a.tiles[(size_t)w * a.win_cap + t] = {(uint32_t)(b0 + r), (uint32_t)(hi - r < a.tile_rows ? hi - r : a.tile_rows), gr.qimg_off, gr.pair0, gr.nq};
    }
}

// The 32-query image of every group from the batch image (both in the 8-slot line layout of k_scan_h16: f16 layout 2 and the
// int8 image); slots past the group's queries are zeroed (no memset of the group images).  One thread per 16-B slot.
__global__ void __launch_bounds__(256) k_ivf_gather_groups(const _Float16* __restrict__ base, uint32_t qpad, _Float16* __restrict__ gimg,
                                                           const IvfGroup* __restrict__ groups, const uint32_t* __restrict__ pair_q,
                                                           const uint32_t* __restrict__ hdr, uint32_t nslab) {
    const uint64_t total = (uint64_t)hdr[0] * 32 * nslab * 8;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = (uint32_t)(i & 7);
        const uint32_t s = (uint32_t)((i >> 3) % nslab);
        const uint32_t n = (uint32_t)(((i >> 3) / nslab) & 31);
        const uint32_t g = (uint32_t)(((i >> 3) / nslab) >> 5);
        const IvfGroup gr = groups[g];
        u32x4 v = {0u, 0u, 0u, 0u};
        if (n < gr.nq) {
            const uint32_t q = pair_q[gr.pair0 + n];
            v = *reinterpret_cast<const u32x4*>(base + (((size_t)s * qpad + q) * 8 + (l ^ ((q >> 1) & 7))) * 8);
        }
        *reinterpret_cast<u32x4*>(gimg + gr.qimg_off + (((size_t)s * 32 + n) * 8 + (l ^ ((n >> 1) & 7))) * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// k-means helpers (kmeans.rs).  Assignment is a FLAT k=1 search of the rows against the centroid
// matrix (same canonical arithmetic and first-smaller-wins tie rule, kmeans.rs:237-264); the update
// sums each cluster's rows in ascending row order with one thread per dimension — the SEQUENTIAL
// order of accumulate_centroid_sums (kmeans.rs:273-286), so centroids are bit-identical to the
// reference's n < 8192 branch.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_kmeans_sums(const float* __restrict__ V, uint32_t ld, uint32_t D,
                                                     const uint32_t* __restrict__ members,   // rows grouped by cluster, ascending
                                                     const uint64_t* __restrict__ offsets,   // nlist + 1
                                                     float* __restrict__ sums, uint32_t ldc) {
    const uint32_t c = blockIdx.x;
    const uint64_t b = offsets[c], e = offsets[c + 1];
    for (uint32_t d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.0f;
        for (uint64_t j = b; j < e; ++j) s = __fadd_rn(s, V[(size_t)members[j] * ld + d]);
        sums[(size_t)c * ldc + d] = s;
    }
}

// Row-sharded training (lynse_hip_ivf_kmeans_sharded through the communicator): the per-rank centroid sums arrive as one
// all-gathered array [world][count]; they are added in RANK order — ((p0 + p1) + p2) + ... with f32 adds, the order of the oracle's
// lo_kmeans_train_sharded — so the centroids are defined bit for bit at every world size (a ring / tree all-reduce associates
// differently per chunk from world 3 on).
__global__ void __launch_bounds__(256) k_sum_rank_order(const float* __restrict__ parts, uint32_t world, uint64_t count,
                                                        float* __restrict__ out) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) {
        float s = parts[j];
        for (uint32_t r = 1; r < world; ++r) s = __fadd_rn(s, parts[(size_t)r * count + j]);
        out[j] = s;
    }
}

// ... and the words that ride along as ONE integer all-reduce: the rank's member count of every list (from the offsets the sums
// kernel reads) + "an assignment changed on this rank" (the stop test of the whole collection, kmeans.rs:127-129)
__global__ void __launch_bounds__(256) k_counts_from_offsets(const uint64_t* __restrict__ offsets, uint32_t k, uint32_t changed,
                                                             uint32_t* __restrict__ out /* k + 1 */) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < k) out[c] = (uint32_t)(offsets[c + 1] - offsets[c]);
    else if (c == k) out[k] = changed;
}

// distance of every sample row to one point + running min-rank (kmeans.rs:170-182)
__global__ void __launch_bounds__(256) k_kmeans_minrank(const float* __restrict__ S, uint32_t ld, uint32_t D, uint32_t n,
                                                        const float* __restrict__ point, int metric,
                                                        float* __restrict__ min_rank) {
    const int g = threadIdx.x & 7;
    const uint32_t row = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (row >= n) return;  // whole 8-lane groups exit together
    const float raw = exact_score(metric, LYNSE_IPFORM_SINGLE, S + (size_t)row * ld, point, D, g);
    const float rank = metric_ascending(metric) ? raw : -raw;
    if (g == 0 && rank < min_rank[row]) min_rank[row] = rank;
}

// ------------------------------------------------------------------------------------------------
// BinaryQuantizer (src/quantizer/mod.rs:286-413) for IVF-*-BINARY.
//   k_bq_colstats : per column min / max and "all values are 0 or 1" flag
//   k_bq_median   : column[n/2] of the ascending-sorted column by 4-pass radix select on the
//                   order-preserving u32 image of the floats (one workgroup per column)
//   k_bq_binarize : out = value > threshold[d] ? 1 : 0   (decode(encode(x)), :359-393)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bq_colstats(const float* __restrict__ V, uint32_t ld, uint32_t D, uint64_t n,
                                                     float* __restrict__ cmin, float* __restrict__ cmax,
                                                     uint32_t* __restrict__ not_binary) {
    __shared__ float smin[256], smax[256];
    __shared__ uint32_t snb;
    const uint32_t d = blockIdx.x;
    if (threadIdx.x == 0) snb = 0;
    __syncthreads();
    float mn = LY_INF, mx = -LY_INF;
    uint32_t nb = 0;
    for (uint64_t i = threadIdx.x; i < n; i += 256) {
        const float v = V[i * ld + d];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        nb |= !(v == 0.0f || v == 1.0f);
    }
    smin[threadIdx.x] = mn;
    smax[threadIdx.x] = mx;
    if (nb) atomicOr(&snb, 1u);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cmin[d] = smin[0];
        cmax[d] = smax[0];
        if (snb) atomicOr(not_binary, 1u);
    }
}

__global__ void __launch_bounds__(256) k_bq_median(const float* __restrict__ V, uint32_t ld, uint32_t D, uint64_t n,
                                                   float* __restrict__ med) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_mask;
    __shared__ uint64_t s_rank;
    const uint32_t d = blockIdx.x;
    if (threadIdx.x == 0) { s_prefix = 0; s_mask = 0; s_rank = n / 2; }
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, mask = s_mask;
        for (uint64_t i = threadIdx.x; i < n; i += 256) {
            float v = V[i * ld + d];
            v = v + 0.0f;  // -0.0 and +0.0 compare equal in the reference's partial_cmp
            const uint32_t u = f32_to_ord(v);
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t rank = s_rank, cum = 0;
            uint32_t b = 0;
            for (; b < 256; ++b) {
                if (cum + hist[b] > rank) break;
                cum += hist[b];
            }
            s_rank = rank - cum;
            s_prefix = prefix | (b << shift);
            s_mask = mask | (255u << shift);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) med[d] = ord_to_f32(s_prefix);
}

__global__ void __launch_bounds__(256) k_bq_binarize(const float* __restrict__ V, uint32_t ld_in, uint32_t D, uint64_t n,
                                                     const float* __restrict__ thr, float* __restrict__ out, uint32_t ld_out) {
    const uint64_t total = n * D;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / D;
        const uint32_t d = (uint32_t)(i % D);
        out[r * ld_out + d] = V[r * ld_in + d] > thr[d] ? 1.0f : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan_binary_tiled: packed-binary slab scan for IVF (work-list mode of k_scan_binary).  One
// workgroup walks tiles; per tile the <=32 packed queries of the group and their thresholds are staged
// in LDS; eight lanes own one row.  Non-strict threshold (slab order is not id order).
// ------------------------------------------------------------------------------------------------
struct BinTileArgs {
    const uint64_t* P;
    uint32_t W;
    const uint64_t* QW;  // batch packed queries, nq x W
    const IvfTile* tiles;
    uint32_t ntiles;
    const uint32_t* pair_q;
    const uint32_t* orig;  // slab position -> original row: the keys carry ORIGINAL rows (no rescoring needs the position)
    const uint32_t* mask;  // subset filter by slab position (nullptr = none)
    const float* thr;
    uint64_t* cand;
    uint32_t* count;
    uint32_t cap;
};

template <int KIND>
__global__ void __launch_bounds__(256) k_scan_binary_tiled(BinTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* qw = reinterpret_cast<uint64_t*>(smem);               // 32 * W
    float* thr_l = reinterpret_cast<float*>(qw + (size_t)32 * a.W);   // 32
    uint32_t* qid_l = reinterpret_cast<uint32_t*>(thr_l + 32);        // 32
    const int tid = threadIdx.x, g = tid & 7;
    const uint32_t nchunks = (a.W + 15) / 16;
    for (uint32_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const IvfTile td = a.tiles[t];
        __syncthreads();
        for (uint32_t i = tid; i < td.nq; i += 256) {
            const uint32_t q = a.pair_q[td.pair0 + i];
            qid_l[i] = q;
            thr_l[i] = a.thr[q];
        }
        __syncthreads();
        for (uint32_t i = tid; i < td.nq * a.W; i += 256) qw[i] = a.QW[(size_t)qid_l[i / a.W] * a.W + (i % a.W)];
        __syncthreads();
        for (uint32_t rb = 0; rb < td.nrows; rb += 32) {
            const uint32_t lr = rb + (tid >> 3);
            const bool valid = lr < td.nrows;
            const uint32_t row = td.row0 + lr;
            uint64_t rw[2 * BIN_MAX_CHUNKS];
#pragma unroll
            for (int c = 0; c < BIN_MAX_CHUNKS; ++c) {
                const uint32_t w0 = 2 * (g + 8 * c);
                rw[2 * c] = (valid && (uint32_t)c < nchunks && w0 < a.W) ? a.P[(size_t)row * a.W + w0] : 0ull;
                rw[2 * c + 1] = (valid && (uint32_t)c < nchunks && w0 + 1 < a.W) ? a.P[(size_t)row * a.W + w0 + 1] : 0ull;
            }
            uint32_t popr = 0;
            if (KIND == 2) {
#pragma unroll
                for (int c = 0; c < 2 * BIN_MAX_CHUNKS; ++c) popr += __popcll(rw[c]);
            }
            for (uint32_t q = 0; q < td.nq; ++q) {
                uint32_t c0 = 0, c1 = 0;
#pragma unroll
                for (int c = 0; c < BIN_MAX_CHUNKS; ++c) {
                    const uint32_t w0 = 2 * (g + 8 * c);
                    if ((uint32_t)c < nchunks) {
                        const uint64_t x0 = w0 < a.W ? qw[(size_t)q * a.W + w0] : 0ull;
                        const uint64_t x1 = w0 + 1 < a.W ? qw[(size_t)q * a.W + w0 + 1] : 0ull;
                        if (KIND == 0) {
                            c0 += __popcll(x0 ^ rw[2 * c]) + __popcll(x1 ^ rw[2 * c + 1]);
                        } else if (KIND == 1) {
                            c0 += __popcll(x0 & rw[2 * c]) + __popcll(x1 & rw[2 * c + 1]);
                            c1 += __popcll(x0 | rw[2 * c]) + __popcll(x1 | rw[2 * c + 1]);
                        } else {
                            c0 += __popcll(x0 & rw[2 * c]) + __popcll(x1 & rw[2 * c + 1]);
                            c1 += __popcll(x0) + __popcll(x1);
                        }
                    }
                }
                if (KIND == 2) c1 += popr;
                c0 += __shfl_xor(c0, 1, 8); c0 += __shfl_xor(c0, 2, 8); c0 += __shfl_xor(c0, 4, 8);
                if (KIND != 0) { c1 += __shfl_xor(c1, 1, 8); c1 += __shfl_xor(c1, 2, 8); c1 += __shfl_xor(c1, 4, 8); }
                if (g == 0 && valid) {
                    float dist;
                    if (KIND == 0) dist = (float)c0;
                    else if (KIND == 1) dist = c1 == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)c0, (float)c1));
                    else dist = c1 == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)(2u * c0), (float)c1));
                    if (dist <= thr_l[q] && (!a.mask || ((a.mask[row >> 5] >> (row & 31)) & 1u))) {
                        const uint32_t qid = qid_l[q];
                        const uint32_t slot = atomicAdd(&a.count[qid], 1u);
                        if (slot < a.cap) a.cand[(size_t)qid * a.cap + slot] = make_key(dist, a.orig[row], true);
                    }
                }
            }
        }
    }
}


// Rows gathered by index inside HBM (device-resident IVF build / load: slab reordering of IvfFlatMmap::build,
// ivf_flat_mmap.rs:115-130, and the k-means init sample): dst[i] = src[ids[i]], `width` floats of a row, pad columns zero.
__global__ void __launch_bounds__(256) k_gather_rows_f32(const float* __restrict__ src, uint32_t src_ld, const uint32_t* __restrict__ ids,
                                                         uint64_t m, float* __restrict__ dst, uint32_t dst_ld, uint32_t width) {
    const uint32_t per_row = (dst_ld + 3) / 4;  // float4 pieces of a destination row
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m * per_row; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / per_row;
        const uint32_t c0 = (uint32_t)(i % per_row) * 4;
        const float* s = src + (size_t)ids[r] * src_ld;
        float* d = dst + (size_t)r * dst_ld;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < dst_ld) d[c0 + e] = (c0 + e < width) ? s[c0 + e] : 0.0f;
    }
}


// dst[pos[i]] = src[i] (IVFIndex::insert on the device: the new rows go to their slab positions of the re-assembled store)
__global__ void __launch_bounds__(256) k_scatter_rows_f32(const float* __restrict__ src, uint32_t src_ld, const uint32_t* __restrict__ pos,
                                                          uint64_t m, float* __restrict__ dst, uint32_t dst_ld, uint32_t width) {
    const uint32_t per_row = (dst_ld + 3) / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m * per_row; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / per_row;
        const uint32_t c0 = (uint32_t)(i % per_row) * 4;
        const float* s = src + (size_t)r * src_ld;
        float* d = dst + (size_t)pos[r] * dst_ld;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < dst_ld) d[c0 + e] = (c0 + e < width) ? s[c0 + e] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// k_ivf_score_positions — the exact scores of a list of slab positions against ONE query, as (score, ORIGINAL row) keys.
// IVF search with k beyond the candidate capacity of the staged pipeline (k > cap / 4: IVFIndex::search accepts any k,
// ivf.rs:304-310, the server caps at MAX_TOP_K = 10,000): every row of the probed lists is scored with the single-row
// kernels (compute_distance_f32, ivf.rs:294-298) and the host takes the top k of the keys.
// ------------------------------------------------------------------------------------------------
struct IvfScoreArgs {
    const float* V;
    uint32_t ld, D;
    const float* q;
    const uint32_t* pos;
    const uint32_t* orig;
    uint32_t n;
    int metric;
    uint64_t* keys;
};

__global__ void __launch_bounds__(256) k_ivf_score_positions(IvfScoreArgs a) {
    const int g = threadIdx.x & 7;
    const bool asc = metric_ascending(a.metric);
    const uint32_t bound = (a.n + 31u) / 32u * 32u;   // whole waves run the same trip count (exact_score shuffles inside its 8 lanes)
    for (uint32_t i = blockIdx.x * 32u + (threadIdx.x >> 3); i < bound; i += gridDim.x * 32u) {
        const uint32_t p = a.pos[i < a.n ? i : a.n - 1];
        const float s = exact_score<32>(a.metric, LYNSE_IPFORM_SINGLE, a.q, a.V + (size_t)p * a.ld, a.D, g);
        if (g == 0 && i < a.n) a.keys[i] = make_key(s, a.orig[p], asc);
    }
}


// Packed-binary twin of k_ivf_score_positions: the distance of every listed slab position to ONE packed query (Hamming /
// Jaccard = Tanimoto / Dice, as k_scan_binary_wide), 8 lanes per row striding over the words — any width.
struct IvfScorePackedArgs {
    const uint64_t* P;
    uint32_t W;
    const uint64_t* qw;
    const uint32_t* pos;
    const uint32_t* orig;
    uint32_t n;
    int kind;   // 0 hamming, 1 jaccard / tanimoto, 2 dice
    uint64_t* keys;
};

__global__ void __launch_bounds__(256) k_ivf_score_positions_packed(IvfScorePackedArgs a) {
    const int g = threadIdx.x & 7;
    const uint32_t bound = (a.n + 31u) / 32u * 32u;
    for (uint32_t i = blockIdx.x * 32u + (threadIdx.x >> 3); i < bound; i += gridDim.x * 32u) {
        const uint32_t p = a.pos[i < a.n ? i : a.n - 1];
        const uint64_t* rp = a.P + (size_t)p * a.W;
        uint32_t c0 = 0, c1 = 0;
        for (uint32_t w = g; w < a.W; w += 8) {
            const uint64_t x = a.qw[w], r = rp[w];
            if (a.kind == 0) {
                c0 += __popcll(x ^ r);
            } else if (a.kind == 1) {
                c0 += __popcll(x & r);
                c1 += __popcll(x | r);
            } else {
                c0 += __popcll(x & r);
                c1 += __popcll(x) + __popcll(r);
            }
        }
        c0 += __shfl_xor(c0, 1, 8); c0 += __shfl_xor(c0, 2, 8); c0 += __shfl_xor(c0, 4, 8);
        c1 += __shfl_xor(c1, 1, 8); c1 += __shfl_xor(c1, 2, 8); c1 += __shfl_xor(c1, 4, 8);
        if (g == 0 && i < a.n) {
            float dist;
            if (a.kind == 0) dist = (float)c0;
            else if (a.kind == 1) dist = c1 == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)c0, (float)c1));
            else dist = c1 == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)(2u * c0), (float)c1));
            a.keys[i] = make_key(dist, a.orig[p], true);
        }
    }
}

}  // namespace lynse

// searches in flight: OR a device flag (the all-lists-empty flag of k_ivf_group) into the status word of the ticket
__global__ void k_or_word(uint32_t* dst, const uint32_t* src) {
    if (*src) atomicOr(dst, 2u);
}
