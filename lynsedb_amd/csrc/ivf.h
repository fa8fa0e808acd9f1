// ivf.h — IVF-Flat kernels: centroid routing, per-group query image gather, k-means helpers.
// The slab scan itself is k_scan_glds in work-list (TILED) mode (kernels.h).
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
                                                      uint32_t npairs, uint32_t nslab) {
    const uint64_t total = (uint64_t)npairs * nslab * 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = (uint32_t)(i & 3);
        const uint32_t s = (uint32_t)((i >> 2) % nslab);
        const uint32_t p = (uint32_t)((i >> 2) / nslab);
        const GatherPair gp = pairs[p];
        const u32x4* src = reinterpret_cast<const u32x4*>(base + (((size_t)s * qpad + gp.q) * 4 + (l ^ ((gp.q >> 2) & 3))) * 8);
        u32x4* dst = reinterpret_cast<u32x4*>(gimg + gp.dst_off + (((size_t)s * 32 + gp.n) * 4 + (l ^ ((gp.n >> 2) & 3))) * 8);
        *dst = *src;
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

}  // namespace lynse
