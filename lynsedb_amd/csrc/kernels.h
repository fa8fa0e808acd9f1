// kernels.h — hand-written HIP kernels for CDNA4 (gfx950, wave64) of the FLAT scan pipeline.
//
// Pipeline for a batch of queries against one HBM-resident shard:
//   k_prep_queries        per-query scale / norms / certified error margin, f16 query image
//   k_scan_h16            THE HOT KERNEL (default): the f16 shadow rows streamed once from HBM by LDS-DMA, Q·Vᵀ on
//                         MFMA 32x32x16 f16, threshold-filter epilogue appending (score,row) keys per query
//   k_scan_glds / _f16    its predecessors over the f32 rows (LDS-DMA ring / register-staged), kept as A/B references
//   k_scan_binary_rows    packed-binary rows: lane-per-row xor/and + popcount with scalar query words (exact ints);
//   k_scan_binary         the 8-lanes-per-row predecessor
//   k_select              per query: radix-select the k-th best candidate key in LDS, new threshold, prune
//   k_final               exact rescoring of the survivors in the REFERENCE's accumulation order
//                         (bit-exact with src/distance/simd.rs AVX2 kernels), final sort, output
//   k_row_stats / k_rows_to_f16 / k_pack_bits   one-time per appended row range
//   filtered search: k_mask_build, k_bits_*, k_gather_rows16 / k_gather_norms;  multi-GPU: k_merge
#pragma once

#include "common.h"
#include <utility>

namespace lynse {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LY_INF __builtin_huge_valf()

// ------------------------------------------------------------------------------------------------
// Exact scoring in the reference's accumulation order.  Eight lanes emulate the eight AVX lanes:
// lane g accumulates elements 8*i+g with fused multiply-add, the horizontal sum follows
// lo128+hi128 -> +movehdup -> +movehl, and the D%8 tail is separate multiply + add (Rust never
// contracts).  Result is identical on all 8 lanes and bit-identical to the oracle / AVX2 reference.
//   IP single : simd.rs:1343-1396   IP batch8 : simd.rs:1452-1525
//   L2        : simd.rs:1529-1581   cosine    : simd.rs:1585-1636
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hsum8(float a) {
    float t = __fadd_rn(a, __shfl_xor(a, 4, 8));  // lo128 + hi128
    t = __fadd_rn(t, __shfl_xor(t, 1, 8));        // + movehdup
    t = __fadd_rn(t, __shfl_xor(t, 2, 8));        // + movehl
    return t;
}

// f16 storage (VectorDtype::F16): simd::inner_product_f16 / l2_squared_f16 / cosine_distance_f16 (simd.rs:805-846) are
// plain sequential f32 sums over the decoded row, separate multiply and add.  One lane does the chain, the 8-lane
// group gets the result.
// An F16 shard keeps its rows as f16 bits only (no f32 decode in HBM): `v` points at the row's HALVES (callers hand the f16
// row matrix in as float* with a pitch of ld16 / 2 floats — LYNSE_IPFORM_F16SEQ implies it); the decode f16 -> f32 is exact.
__device__ __forceinline__ float exact_score_f16seq(int metric, const float* __restrict__ q, const float* __restrict__ v_as_f32,
                                                    uint32_t D, int g) {
    // The reference's f16 kernels (simd.rs:805-846) add the per-element terms ONE BY ONE in element order into a single f32
    // accumulator.  The terms themselves are independent: the 8 lanes of a candidate's group load 8 elements each per step (one
    // 16-B load of halves, two of floats), form their terms, and every lane then adds the 64 terms of the step in element order
    // (width-8 shuffles) — the same additions in the same order, with coalesced loads instead of one 2-byte load per addition
    // (which made this rescoring 0.6-0.8 ms of a 10M-row batch on an F16 shard).  Widths that are no multiple of 8, or rows /
    // queries that are not 16-B aligned, keep the one-lane loop.
    const _Float16* __restrict__ v = reinterpret_cast<const _Float16*>(v_as_f32);
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const bool fast = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(v)) % 16 == 0);
    if (__builtin_amdgcn_readfirstlane(fast ? 1 : 0) && __all(fast)) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;   // IP / L2: s0; cosine: dot, |q|^2, |v|^2
        for (uint32_t b0 = 0; b0 < D; b0 += 64) {
            const uint32_t e0 = b0 + 8 * (uint32_t)g;
            float t0[8], t1[8], t2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { t0[e] = 0.0f; t1[e] = 0.0f; t2[e] = 0.0f; }
            if (e0 < D) {
                const f32x4 qa = *reinterpret_cast<const f32x4*>(q + e0), qb = *reinterpret_cast<const f32x4*>(q + e0 + 4);
                const h8 vv = *reinterpret_cast<const h8*>(v + e0);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = e < 4 ? qa[e] : qb[e - 4], c = (float)vv[e];
                    if (metric == M_IP) t0[e] = __fmul_rn(a, c);
                    else if (metric == M_L2) { const float d = __fsub_rn(a, c); t0[e] = __fmul_rn(d, d); }
                    else { t0[e] = __fmul_rn(a, c); t1[e] = __fmul_rn(a, a); t2[e] = __fmul_rn(c, c); }
                }
            }
            const uint32_t left = D - b0, ng = left >= 64 ? 8u : left / 8;   // groups of 8 elements in this step (D % 8 == 0)
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) {
                if ((uint32_t)gg < ng) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s0 = __fadd_rn(s0, __shfl(t0[e], gg, 8));
                        if (metric == M_COS) { s1 = __fadd_rn(s1, __shfl(t1[e], gg, 8)); s2 = __fadd_rn(s2, __shfl(t2[e], gg, 8)); }
                    }
                }
            }
        }
        if (metric != M_COS) return s0;
        if (s1 == 0.0f || s2 == 0.0f) return 1.0f;
        return __fsub_rn(1.0f, __fdiv_rn(s0, __fmul_rn(__builtin_sqrtf(s1), __builtin_sqrtf(s2))));
    }
    float r = 0.0f;
    if (g == 0) {
        if (metric == M_IP) {
            float sum = 0.0f;
            for (uint32_t i = 0; i < D; ++i) sum = __fadd_rn(sum, __fmul_rn(q[i], (float)v[i]));
            r = sum;
        } else if (metric == M_L2) {
            float sum = 0.0f;
            for (uint32_t i = 0; i < D; ++i) {
                const float d = __fsub_rn(q[i], (float)v[i]);
                sum = __fadd_rn(sum, __fmul_rn(d, d));
            }
            r = sum;
        } else {
            float dot = 0.0f, nq = 0.0f, nc = 0.0f;
            for (uint32_t i = 0; i < D; ++i) {
                const float a = q[i], c = (float)v[i];
                dot = __fadd_rn(dot, __fmul_rn(a, c));
                nq = __fadd_rn(nq, __fmul_rn(a, a));
                nc = __fadd_rn(nc, __fmul_rn(c, c));
            }
            if (nq == 0.0f || nc == 0.0f) r = 1.0f;
            else r = __fsub_rn(1.0f, __fdiv_rn(dot, __fmul_rn(__builtin_sqrtf(nq), __builtin_sqrtf(nc))));
        }
    }
    return __shfl(r, 0, 8);
}

// UB: 8-element steps whose loads are issued together (the FMA chain keeps the reference's order either way).  One candidate
// is a chain of D / 8 dependent FMAs per lane; with the loads of 8 steps in flight a 768-d row is 12 dependent global round
// trips (~1 us each: the exact rescoring of a handful of rows was half of k_select's 26 us) — the latency-bound callers
// (rescore_keys, k_rescore_pool) ask for 32.
template <int UB = 8>
__device__ __forceinline__ float exact_score(int metric, int ip_form, const float* __restrict__ q,
                                             const float* __restrict__ v, uint32_t D, int g) {
    if (ip_form == LYNSE_IPFORM_F16SEQ) return exact_score_f16seq(metric, q, v, D, g);
    const uint32_t chunks = D / 8, rem = D % 8, base = chunks * 8;
    if (metric == M_IP && ip_form == LYNSE_IPFORM_BATCH8) {
        // loads are issued eight steps at a time (the FMA chain keeps the reference's order): one candidate is a chain
        // of D/8 dependent FMAs, and waiting out a global-load round trip per step made k_final latency-bound
        float acc = 0.0f;
        uint32_t i = 0;
        for (; i + UB <= chunks; i += UB) {
            float a[UB], b[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) { a[u] = q[(i + u) * 8 + g]; b[u] = v[(i + u) * 8 + g]; }
#pragma unroll
            for (int u = 0; u < UB; ++u) acc = __fmaf_rn(a[u], b[u], acc);
        }
        if constexpr (UB > 8) {
            for (; i + 8 <= chunks; i += 8) {
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { a[u] = q[(i + u) * 8 + g]; b[u] = v[(i + u) * 8 + g]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __fmaf_rn(a[u], b[u], acc);
            }
        }
        for (; i < chunks; ++i) acc = __fmaf_rn(q[i * 8 + g], v[i * 8 + g], acc);
        float sum = hsum8(acc);
        for (uint32_t i = 0; i < rem; ++i) sum = __fadd_rn(sum, __fmul_rn(q[base + i], v[base + i]));
        return sum;
    }
    if (metric == M_IP || metric == M_L2) {
        const bool l2 = metric == M_L2;
        const uint32_t dbl = chunks / 2, single = chunks % 2;
        float acc0 = 0.0f, acc1 = 0.0f;
        uint32_t i = 0;
        for (; i + UB / 2 <= dbl; i += UB / 2) {
            float a[UB], b[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) { a[u] = q[i * 16 + u * 8 + g]; b[u] = v[i * 16 + u * 8 + g]; }
#pragma unroll
            for (int u = 0; u < UB / 2; ++u) {
                if (l2) {
                    const float d0 = __fsub_rn(a[2 * u], b[2 * u]), d1 = __fsub_rn(a[2 * u + 1], b[2 * u + 1]);
                    acc0 = __fmaf_rn(d0, d0, acc0);
                    acc1 = __fmaf_rn(d1, d1, acc1);
                } else {
                    acc0 = __fmaf_rn(a[2 * u], b[2 * u], acc0);
                    acc1 = __fmaf_rn(a[2 * u + 1], b[2 * u + 1], acc1);
                }
            }
        }
        if constexpr (UB > 8) {
            for (; i + 4 <= dbl; i += 4) {
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { a[u] = q[i * 16 + u * 8 + g]; b[u] = v[i * 16 + u * 8 + g]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (l2) {
                        const float d0 = __fsub_rn(a[2 * u], b[2 * u]), d1 = __fsub_rn(a[2 * u + 1], b[2 * u + 1]);
                        acc0 = __fmaf_rn(d0, d0, acc0);
                        acc1 = __fmaf_rn(d1, d1, acc1);
                    } else {
                        acc0 = __fmaf_rn(a[2 * u], b[2 * u], acc0);
                        acc1 = __fmaf_rn(a[2 * u + 1], b[2 * u + 1], acc1);
                    }
                }
            }
        }
        for (; i < dbl; ++i) {
            float a0 = q[i * 16 + g], b0 = v[i * 16 + g];
            float a1 = q[i * 16 + 8 + g], b1 = v[i * 16 + 8 + g];
            if (l2) {
                float d0 = __fsub_rn(a0, b0), d1 = __fsub_rn(a1, b1);
                acc0 = __fmaf_rn(d0, d0, acc0);
                acc1 = __fmaf_rn(d1, d1, acc1);
            } else {
                acc0 = __fmaf_rn(a0, b0, acc0);
                acc1 = __fmaf_rn(a1, b1, acc1);
            }
        }
        if (single) {
            float a0 = q[dbl * 16 + g], b0 = v[dbl * 16 + g];
            if (l2) {
                float d0 = __fsub_rn(a0, b0);
                acc0 = __fmaf_rn(d0, d0, acc0);
            } else {
                acc0 = __fmaf_rn(a0, b0, acc0);
            }
        }
        float sum = hsum8(__fadd_rn(acc0, acc1));
        for (uint32_t i = 0; i < rem; ++i) {
            if (l2) {
                float d = __fsub_rn(q[base + i], v[base + i]);
                sum = __fadd_rn(sum, __fmul_rn(d, d));
            } else {
                sum = __fadd_rn(sum, __fmul_rn(q[base + i], v[base + i]));
            }
        }
        return sum;
    }
    // cosine distance
    float d = 0.0f, x = 0.0f, y = 0.0f;
    uint32_t i = 0;
    for (; i + UB <= chunks; i += UB) {
        float a8[UB], b8[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) { a8[u] = q[(i + u) * 8 + g]; b8[u] = v[(i + u) * 8 + g]; }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            d = __fmaf_rn(a8[u], b8[u], d);
            x = __fmaf_rn(a8[u], a8[u], x);
            y = __fmaf_rn(b8[u], b8[u], y);
        }
    }
    if constexpr (UB > 8) {
        for (; i + 8 <= chunks; i += 8) {
            float a8[8], b8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a8[u] = q[(i + u) * 8 + g]; b8[u] = v[(i + u) * 8 + g]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d = __fmaf_rn(a8[u], b8[u], d);
                x = __fmaf_rn(a8[u], a8[u], x);
                y = __fmaf_rn(b8[u], b8[u], y);
            }
        }
    }
    for (; i < chunks; ++i) {
        float a = q[i * 8 + g], b = v[i * 8 + g];
        d = __fmaf_rn(a, b, d);
        x = __fmaf_rn(a, a, x);
        y = __fmaf_rn(b, b, y);
    }
    float dot = hsum8(d), na = hsum8(x), nb = hsum8(y);
    for (uint32_t i = 0; i < rem; ++i) {
        float a = q[base + i], b = v[base + i];
        dot = __fadd_rn(dot, __fmul_rn(a, b));
        na = __fadd_rn(na, __fmul_rn(a, a));
        nb = __fadd_rn(nb, __fmul_rn(b, b));
    }
    float denom = __builtin_sqrtf(__fmul_rn(na, nb));  // IEEE sqrt (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is the native approximation
    if (denom < 1e-30f) return 1.0f;
    return __fsub_rn(1.0f, __fdiv_rn(dot, denom));
}

// ------------------------------------------------------------------------------------------------
// k_row_stats: per-row squared norm + reciprocal norm for rows [row0,row1), and collection-wide
// statistics (max |v|, max / min-nonzero squared norm, count of degenerate tiny-norm rows) that
// parameterise the certified f16 error margin.  One wave per row.
// stats[0]=bits(max|v|) stats[1]=bits(max n2) stats[2]=bits(min nonzero n2) stats[3]=#rows with 0<n2<1e-30
// stats[4]=bit 0: some element is not an integer (NaN counts), bit 1: some element is negative — 0 for SIFT-like u8 / count data:
// k_prep_queries' exactness rule
// ------------------------------------------------------------------------------------------------
template <typename T>  // float rows, or the f16 bits of an F16 shard
__global__ void __launch_bounds__(256) k_row_stats(const T* __restrict__ V, uint32_t ld, uint32_t D,
                                                   uint32_t row0, uint32_t row1, float* __restrict__ vn2,
                                                   float* __restrict__ vrinv, uint32_t* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    float amax = 0.0f, n2max = 0.0f, n2min = LY_INF;
    uint32_t ndegen = 0, nonint = 0;
    for (uint32_t row = row0 + wave; row < row1; row += nwaves) {
        const T* v = V + (size_t)row * ld;
        float s = 0.0f;
        for (uint32_t i = lane; i < D; i += 64) {
            float x = (float)v[i];
            s = __fmaf_rn(x, x, s);
            amax = fmaxf(amax, fabsf(x));
            nonint |= ((x != truncf(x)) ? 1u : 0u) | ((x < 0.0f) ? 2u : 0u);
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            vn2[row] = s;
            vrinv[row] = s > 0.0f ? 1.0f / sqrtf(s) : 0.0f;
        }
        n2max = fmaxf(n2max, s);
        if (s > 0.0f) n2min = fminf(n2min, s);
        if (s > 0.0f && s < 1e-30f) ndegen += 1;
    }
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) {  // one set of atomics per wave (per-row atomics on one address serialise)
        atomicMax(&stats[0], __float_as_uint(amax));
        atomicMax(&stats[1], __float_as_uint(n2max));
        if (n2min < LY_INF) atomicMin(&stats[2], __float_as_uint(n2min));
        if (ndegen) atomicAdd(&stats[3], ndegen);
    }
    {
        const uint32_t bits = (__ballot((nonint & 1u) != 0u) != 0ull ? 1u : 0u) | (__ballot((nonint & 2u) != 0u) != 0ull ? 2u : 0u);
        if (bits && lane == 0) atomicOr(&stats[4], bits);
    }
}

// ------------------------------------------------------------------------------------------------
// k_copy_rows: dst[r][0..width) = src[r][0..width) with independent pitches (elements); columns
// [width, dst_pitch) of dst are zero-filled when zero_pad is set.  Used for the padded row layout
// (ld = round_up(dim,4)) — hipMemcpy2D proved unreliable on this stack for multi-GB buffers.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_copy_rows(float* __restrict__ dst, uint32_t dst_pitch,
                                                   const float* __restrict__ src, uint32_t src_pitch,
                                                   uint32_t width, uint64_t nrows, int zero_pad) {
    const uint32_t cols = zero_pad ? dst_pitch : width;
    const uint64_t total = nrows * cols;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / cols;
        const uint32_t c = (uint32_t)(i % cols);
        dst[r * dst_pitch + c] = c < width ? src[r * src_pitch + c] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// k_pack_bits: bit i of word i/64 = (value > 0.5), LSB first — pack_binary_row_f32
// (flat_mmap.rs:1284-1290, simd.rs:750-757).  One wave per row; wave64 __ballot IS one u64 word.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_pack_bits(const T* __restrict__ V, uint32_t ld, uint32_t D,
                                                   uint32_t nrows, uint64_t* __restrict__ out, uint32_t W) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t row = wave; row < nrows; row += nwaves) {
        const T* v = V + (size_t)row * ld;
        for (uint32_t w = 0; w < W; ++w) {
            uint32_t i = w * 64 + lane;
            bool bit = (i < D) && ((float)v[i] > 0.5f);
            uint64_t m = __ballot(bit);
            if (lane == 0) out[(size_t)row * W + w] = m;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_prep_queries: one block per query.
//  - scale sq = 2^(13-ilogb(max|q|)) so the f16 image uses the full normal range;
//  - f16 query image Q16[slab][q][72] (k >= D zero) — exactly the LDS image of a Q slab;
//  - qinv = 1/(sq*sv) (power of two), |q|^2, 1/|q|;
//  - marg2 = 2E where E bounds |coarse f16/MFMA score - reference-order f32 score|;
//  - thr = worst, count = 0.
// ------------------------------------------------------------------------------------------------
struct PrepArgs {
    const float* Q;      // nq x D f32
    uint32_t D, nq, qpad, nslab;
    int layout;          // 0: [slab64][q][72] (k_scan_f16)   1: [slab32][q][4 slots ^ ((q>>2)&3)][8] (k_scan_glds)   2: [slab64][q][8 slots ^ ((q>>1)&7)][8] (k_scan_h16)
    int metric;
    float sv;            // row scale (power of two)
    float vmax, vmin;    // max / min-nonzero row norm
    int cos_degenerate;  // rows with 0 < |v|^2 < 1e-30 exist
    // Exactness rule (round 5): every element of the shard is an INTEGER of magnitude <= amax_v (k_row_stats, stats[4] / stats[0]).  If
    // the query is integer-valued too, both sides are exact in f16 (|x| <= 2048) and D aq av, D aq^2, D av^2 and D dmax^2 are all below
    // 2^24 (aq, av = max |q_i|, max |v_i|; dmax = the largest |q_i - v_i| possible: max(aq, av) when both sides are non-negative, else
    // aq + av), then every product and every partial sum of the coarse pass AND of the reference's f32 kernels (simd.rs:1343-1581: IP,
    // squared L2 as sum (q_i - v_i)^2) is an integer below 2^24 — both compute the TRUE value, whatever their summation order: E = 0,
    // the thresholds are exact, nothing but real ties of the k-th score is kept beside the k best (SIFT / GIST-style u8 collections:
    // BASELINE config 3).
    int rows_integer, rows_nonneg;
    float amax_v;
    _Float16* Q16;
    float *qinv, *qn2, *qrinv, *marg2, *thr;
    uint32_t* count;
    uint32_t* overflow;
};

__global__ void __launch_bounds__(256) k_prep_queries(PrepArgs a) {
    __shared__ float red[3][4];
    __shared__ float s_sq;
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (q >= a.nq) {   // (uniform) a pad column of the query tile (the grid covers qpad queries): a ZERO image — finite scores for the MFMAs; it used to be a
                       // hipMemsetAsync of the whole image in front of this kernel: a 5-us launch of its own in every batch (round 6)
        const uint32_t per = a.layout == 0 ? (uint32_t)SCAN_LDK : (a.layout == 2 ? 64u : 32u);   // halves of one (slab, query) line
        for (uint32_t i = tid; i < a.nslab * per; i += 256) a.Q16[((size_t)(i / per) * a.qpad + q) * per + (i % per)] = (_Float16)0.0f;
        return;
    }
    const float* qv = a.Q + (size_t)q * a.D;
    float amax = 0.0f, s2 = 0.0f, s1 = 0.0f;
    bool nonint = false, qneg = false;
    for (uint32_t i = tid; i < a.D; i += 256) {
        float x = qv[i];
        amax = fmaxf(amax, fabsf(x));
        s2 = __fmaf_rn(x, x, s2);
        s1 += fabsf(x);
        nonint = nonint || x != truncf(x);
        qneg = qneg || x < 0.0f;
    }
    for (int o = 32; o > 0; o >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        s2 += __shfl_xor(s2, o, 64);
        s1 += __shfl_xor(s1, o, 64);
    }
    __shared__ uint32_t s_nonint[4];
    const uint32_t wave_bits = (__ballot(nonint) != 0ull ? 1u : 0u) | (__ballot(qneg) != 0ull ? 2u : 0u);
    if (lane == 0) { red[0][wave] = amax; red[1][wave] = s2; red[2][wave] = s1; s_nonint[wave] = wave_bits; }
    __syncthreads();
    if (tid == 0) {
        amax = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        const uint32_t q_bits = s_nonint[0] | s_nonint[1] | s_nonint[2] | s_nonint[3];
        const bool q_integer = (q_bits & 1u) == 0u, q_nonneg = (q_bits & 2u) == 0u;
        s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        s1 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        int e = (amax > 0.0f && amax < LY_INF) ? ilogbf(amax) : 13;
        if (e < -100) e = -100;
        if (e > 100) e = 100;
        const float sq = ldexpf(1.0f, 13 - e);
        s_sq = sq;
        const float qn = sqrtf(s2);
        const float Df = (float)a.D;
        const float u = 4.8828125e-4f;                 // 2^-11, f16 unit roundoff
        const float c1 = 2.0f * u + u * u;
        const float gam = 10.0f * Df * 5.9604645e-8f;  // MFMA f32 accumulate + reference f32 order
        const float eta_v = ldexpf(1.0f, -25) / a.sv;  // f16 subnormal floor, original units
        const float eta_q = ldexpf(1.0f, -25) / sq;
        const float e_ip = (c1 + gam) * qn * a.vmax + eta_v * (1.0f + u) * s1 +
                           eta_q * (1.0f + u) * sqrtf(Df) * a.vmax + Df * eta_q * eta_v;
        float E;
        if (a.metric == M_IP) {
            E = 1.02f * e_ip;
        } else if (a.metric == M_L2) {
            E = 1.02f * (2.0f * e_ip + 4.0f * (Df + 8.0f) * 5.9604645e-8f * (s2 + a.vmax * a.vmax));
        } else {
            if (s2 < 1e-30f || a.cos_degenerate) {
                E = 4.0f;  // covers the whole [0,2] range: forces the exhaustive exact path
            } else {
                const float vmin = a.vmin > 0.0f ? a.vmin : 1.0f;
                E = 1.02f * (c1 + gam + sqrtf(Df) * (1.0f + u) * (eta_v / vmin + eta_q / qn) +
                             Df * eta_q * eta_v / (qn * vmin) + 4.0f * (Df + 8.0f) * 5.9604645e-8f);
            }
        }
        if (!(E == E) || E > 3.0e38f) E = 3.0e38f;
        {   // the exactness rule (PrepArgs::rows_integer): integer rows x integer query, everything below 2^24
            const float aq = amax, av = a.amax_v, dmax = (a.rows_nonneg && q_nonneg) ? fmaxf(aq, av) : aq + av;
            const float big = fmaxf(fmaxf(aq * av, dmax * dmax), fmaxf(aq * aq, av * av));
            if (a.rows_integer && q_integer && a.metric != M_COS && a.sv == 1.0f && aq <= 2048.0f && av <= 2048.0f &&
                (float)a.D * big * 1.0001f < 16777216.0f)
                E = 0.0f;
        }
        a.qinv[q] = 1.0f / (sq * a.sv);
        a.qn2[q] = s2;
        a.qrinv[q] = s2 > 0.0f ? 1.0f / qn : 0.0f;
        a.marg2[q] = 2.0f * E;
        a.thr[q] = metric_ascending(a.metric) ? LY_INF : -LY_INF;
        a.count[q] = 0u;
        a.overflow[q] = 0u;
    }
    __syncthreads();
    const float sq = s_sq;
    if (a.layout == 0) {
        const uint32_t total = a.nslab * SCAN_BK;
        for (uint32_t i = tid; i < total; i += 256) {
            const uint32_t s = i / SCAN_BK, k = i % SCAN_BK;
            const float x = i < a.D ? qv[i] * sq : 0.0f;
            a.Q16[((size_t)s * a.qpad + q) * SCAN_LDK + k] = (_Float16)x;
        }
    } else if (a.layout == 2) {  // k_scan_h16: [slab64][q][8 slots ^ ((q>>1)&7)][8]
        const uint32_t total = a.nslab * 64;
        for (uint32_t i = tid; i < total; i += 256) {
            const uint32_t s = i / 64, k = i % 64;
            const uint32_t l = k >> 3, e = k & 7;
            const uint32_t p = l ^ ((q >> 1) & 7);
            const float x = i < a.D ? qv[i] * sq : 0.0f;
            a.Q16[(((size_t)s * a.qpad + q) * 8 + p) * 8 + e] = (_Float16)x;
        }
    } else {
        const uint32_t total = a.nslab * 32;
        for (uint32_t i = tid; i < total; i += 256) {
            const uint32_t s = i / 32, k = i % 32;
            const uint32_t l = k >> 3, e = k & 7;          // logical 16-B slot, element
            const uint32_t p = l ^ ((q >> 2) & 3);          // physical slot (bank-conflict swizzle)
            const float x = i < a.D ? qv[i] * sq : 0.0f;
            a.Q16[(((size_t)s * a.qpad + q) * 4 + p) * 8 + e] = (_Float16)x;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan_f16 — the hot kernel.
//
// Tile: BR=128 rows x BQ queries (256: 8 waves as 4(q) x 2(r), each 64x64; 32: 4 waves as 1 x 4,
// each 32x32).  Persistent blocks walk row tiles; for each K slab of 64:
//   HBM  -> VGPR : 16 consecutive lanes read one row's 256 contiguous bytes (float4 each)
//   VGPR -> LDS  : scale (power of two), cvt f32->f16 (RNE), ds_write_b64, row stride 144 B
//   L2   -> LDS  : the pre-built f16 query slab image, linear 16-B copies
//   LDS  -> MFMA : ds_read_b128 fragments (stride 144 B is conflict-free), v_mfma_f32_32x32x16_f16,
//                  A = rows (M), B = queries (N) so that each lane owns ONE query column per block
// LDS is double buffered with register prefetch of the next slab (also across tile boundaries), one
// barrier per slab.  Epilogue: score transform (IP / L2 via norms / cosine via reciprocal norms),
// compare with the per-query threshold, append passing (score,row) keys with a global atomic slot
// (stage 0: slot = row, no atomics).  Every row byte is read from HBM exactly once per batch.
// Algorithmic bytes per launch = rows * D * 4.
// ------------------------------------------------------------------------------------------------
struct ScanArgs {
    const float* V;
    uint32_t ld, D;
    const _Float16* V16;  // k_scan_h16: f16 shadow of the rows, (half)(v * sv), pitch ld16 halves (multiple of 8), pad columns zero
    uint32_t ld16;
    // k_scan_h16 sample stage: tile t starts at row0 + t * tile_stride (0 = contiguous tiles).  Later stages skip the
    // emission of the skip_tiles sample tiles (rows [t * skip_stride, +BR), t < skip_tiles) — they are already candidates.
    uint32_t tile_stride, skip_stride, skip_tiles;
    // filtered search (FlatMmap::search_filtered): bit r of mask = row r is in the subset (32-bit words; stage
    // boundaries are multiples of 32).  Rows outside the subset are never emitted; the emit-all stage writes
    // KEY_SENTINEL into their slots and k_select drops those.  nullptr = unfiltered.
    const uint32_t* mask;
    const uint32_t* row_ids;  // gathered filtered search: tile row m is original row row_ids[m] (keys carry the original row)
    uint32_t row0, row1;  // stage rows [row0,row1)
    const _Float16* Q16;  // [nslab][qpad][72]
    uint32_t qpad, nq, nslab, ntiles;
    const float *qinv, *qn2, *qrinv, *thr;
    const float *vn2, *vrinv;
    float sv;
    float vmax2;  // max squared row norm of the store: rounding slack of the L2 pre-filter (0 = unknown -> pre-filter off)
    // Segmented emission (threshold stages of the FLAT scan, not the IVF work-list mode): every (workgroup, row-wave) owns
    // a private segment of `seg` key slots per query in candB — slots are handed out from a per-lane register counter, no
    // atomic and no wait in the epilogue; at the end each wave stores its per-query counts (u8) to segcnt[q][segment] and
    // k_select gathers.  A segment that would overflow falls back to the shared region of `cand` (returning atomic on
    // count[q]) for that block.  seg == 0: every key goes through the atomic path.
    uint64_t* candB;     // [q][nseg][seg]
    uint8_t* segcnt;     // [q][nseg]
    uint32_t seg, nseg;
    uint64_t* cand;
    uint32_t* count;
    uint32_t cap;
    int emit_all;
    int dense;        // host-side dispatch only: this threshold stage expects many survivors per block -> the DENSE epilogue
    int debug_flags;  // timing experiments only: 1 = linear (blocked-layout-like) row addressing, 2 = no emission
    // work-list mode (IVF slabs): tile t covers rows [tiles[t].row0, +nrows) for the query group whose
    // f16 image starts at Q16 + qimg_off halves; local query n of the group is query pair_q[pair0+n]
    const struct IvfTile* tiles;
    const uint32_t* pair_q;
    const uint32_t* ntiles_dev;   // work-list mode: the number of tiles lives in device memory (k_ivf_group built the list); NULL: ntiles
    unsigned long long* dbg;  // debug_flags & 64: per-block phase cycle sums [block][wave][4]: wait, barrier, issue, compute
    // Fused sample stage (k_scan_h16<..., FS = 1>): workgroup b first scores sample tile b (rows [b * fs_stride, +BR) of the
    // WHOLE shard, fs_rows rows), publishes its lane-max keys, and the grid agrees on the first thresholds inside the
    // launch (two grid-wide hand-overs around a per-query select by workgroup q) before the ordinary tiles of the stage run
    // under them — the sample launch and its k_select are gone (fused_sample_threshold, below).
    uint32_t fs_stride, fs_rows;
    uint32_t* gsync;          // [0] arrivals 1, [1] arrivals 2, [2] abort (zeroed by the prep kernel of the batch)
    const float* Qf;          // f32 queries (exact rescoring of the k best sample rows)
    const float* marg2;
    float* thr_out;           // = thr (written by the select of the fused sample)
    uint32_t k;
    int ip_form, metric;
    // Self-tightening thresholds of the query-stationary scan (k_scan_qs<.., STS>, scan_qs.h): per query dyn_ks running maxima
    // of the coarse integer dot product over DISJOINT row partitions (partition of a row = its tile index % dyn_ks) and their
    // minimum dyn_thr — at least dyn_ks distinct rows reach it, so with dyn_ks = k every row below dyn_thr - dyn_marg (the
    // certified margin 2E in dot units) is out of the top-k whatever the scan has seen so far.  Seeded by k_i8c_prep_queries.
    int* dyn_thr;          // [nq]
    int* dyn_slot;         // [nq][32]
    const int* dyn_marg;   // [nq]
    uint32_t dyn_ks;
    uint32_t dyn_warm;     // tiles per workgroup scanned first WITHOUT emission (they only feed the maxima) and again at the end
    uint32_t dyn_pitch;    // tiles per workgroup (workgroup b owns tiles [b * dyn_pitch, +dyn_pitch)); coprime to dyn_ks
};

struct IvfTile {
    uint32_t row0, nrows;
    uint32_t qimg_off;  // in halves
    uint32_t pair0, nq;
};

#ifdef LYNSE_EXPERIMENTS  // the register-staged predecessor (A/B reference only: make EXPERIMENTS=1)
template <int WQ, int WR, int TQ, int TR, int METRIC, int PD>
__global__ void __launch_bounds__(WQ * WR * 64, 2) k_scan_f16(ScanArgs a) {
    constexpr int NT = WQ * WR * 64;
    constexpr int BQ = WQ * TQ * 32;
    constexpr int BR = WR * TR * 32;
    static_assert(BR == SCAN_BR, "tile rows");
    static_assert(PD == 1 || PD == 2, "prefetch depth");
    constexpr int ROWS_PER_PASS = NT / 16;
    constexpr int PV = BR / ROWS_PER_PASS;             // float4 loads per thread per slab
    constexpr int QCHUNKS = BQ * (SCAN_LDK * 2 / 16);  // 16-B chunks of one Q slab image
    constexpr int PQ = (QCHUNKS + NT - 1) / NT;
    constexpr int VBUF = BR * SCAN_LDK;  // halves per V buffer
    constexpr int QBUF = BQ * SCAN_LDK;
    constexpr bool ASC = METRIC != M_IP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* Vl = reinterpret_cast<_Float16*>(smem);             // [2][VBUF]
    _Float16* Ql = reinterpret_cast<_Float16*>(smem) + 2 * VBUF;  // [2][QBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wq = wave % WQ, wr = wave / WQ;
    const int vrow = tid >> 4, vk = (tid & 15) * 4;

    if (blockIdx.x >= a.ntiles) return;

    // Register staging: vA / vB hold the f32 row pieces of the next one (PD=1) or two (PD=2) K slabs
    // while they are in flight from HBM; qreg holds the next f16 query slab (L2-resident).
    f32x4 vA[PV], vB[PV];
    u32x4 qreg[PQ];

    // Branch-free loads (a divergent branch around a load makes hipcc drain vmcnt(0) and kills the
    // prefetch pipeline): rows past the stage end are clamped to the last row (their scores are
    // masked in the epilogue), columns past ld are clamped here and zeroed in store_v.
    const uint32_t last_row = a.row1 - 1;
    auto load_v = [&](f32x4(&v)[PV], uint32_t t, uint32_t s) {
        const uint32_t rbase = a.row0 + t * BR;
        uint32_t k = s * SCAN_BK + vk;
        k = k < a.ld ? k : a.ld - 4;
#pragma unroll
        for (int p = 0; p < PV; ++p) {
            uint32_t row = rbase + p * ROWS_PER_PASS + vrow;
            row = row < last_row ? row : last_row;
            v[p] = *reinterpret_cast<const f32x4*>(a.V + (size_t)row * a.ld + k);
        }
    };
    auto load_q = [&](uint32_t s) {
        const u32x4* qsrc = reinterpret_cast<const u32x4*>(a.Q16 + (size_t)s * a.qpad * SCAN_LDK);
#pragma unroll
        for (int j = 0; j < PQ; ++j) {
            const int c = tid + j * NT;
            qreg[j] = qsrc[c < QCHUNKS ? c : QCHUNKS - 1];  // unconditional: keeps qreg in VGPRs
        }
    };
    auto store_v = [&](const f32x4(&v)[PV], int buf, uint32_t s) {
        _Float16* vb = Vl + buf * VBUF;
        const float scale = (s * SCAN_BK + vk < a.ld) ? a.sv : 0.0f;  // zero the K padding of the last slab
        const bool kin = s * SCAN_BK + vk < a.ld;
#pragma unroll
        for (int p = 0; p < PV; ++p) {
            half4 h;
            h[0] = kin ? (_Float16)(v[p][0] * scale) : (_Float16)0.0f;
            h[1] = kin ? (_Float16)(v[p][1] * scale) : (_Float16)0.0f;
            h[2] = kin ? (_Float16)(v[p][2] * scale) : (_Float16)0.0f;
            h[3] = kin ? (_Float16)(v[p][3] * scale) : (_Float16)0.0f;
            *reinterpret_cast<half4*>(vb + (p * ROWS_PER_PASS + vrow) * SCAN_LDK + vk) = h;
        }
    };
    auto store_q = [&](int buf) {
        u32x4* qb = reinterpret_cast<u32x4*>(Ql + buf * QBUF);
#pragma unroll
        for (int j = 0; j < PQ; ++j) {
            const int c = tid + j * NT;
            if (c < QCHUNKS) qb[c] = qreg[j];
        }
    };

    f32x16 acc[TR][TQ];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frag_k = (lane >> 5) * 8;
    const int a_row = wr * (TR * 32) + (lane & 31);
    const int b_row = wq * (TQ * 32) + (lane & 31);

    auto compute = [&](int buf) {
        const _Float16* vb = Vl + buf * VBUF;
        const _Float16* qb = Ql + buf * QBUF;
#pragma unroll
        for (int kk = 0; kk < SCAN_BK / 16; ++kk) {
            half8 af[TR], bf[TQ];
#pragma unroll
            for (int i = 0; i < TR; ++i)
                af[i] = *reinterpret_cast<const half8*>(vb + (a_row + i * 32) * SCAN_LDK + kk * 16 + frag_k);
#pragma unroll
            for (int j = 0; j < TQ; ++j)
                bf[j] = *reinterpret_cast<const half8*>(qb + (b_row + j * 32) * SCAN_LDK + kk * 16 + frag_k);
#pragma unroll
            for (int i = 0; i < TR; ++i)
#pragma unroll
                for (int j = 0; j < TQ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    // epilogue for one finished tile: C[row m][query n]; a lane owns column n = lane&31 of each block
    auto epilogue = [&](uint32_t tile) {
        const uint32_t rbase = a.row0 + tile * BR;
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const uint32_t n = wq * (TQ * 32) + j * 32 + (lane & 31);
            const bool qok = n < a.nq;
            const float qinv = qok ? a.qinv[n] : 0.0f;
            const float thr = qok ? a.thr[n] : 0.0f;
            float qextra = 0.0f;
            if (METRIC == M_L2) qextra = qok ? a.qn2[n] : 0.0f;
            if (METRIC == M_COS) qextra = qok ? a.qrinv[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < TR; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t m = rbase + wr * (TR * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const bool rok = m < a.row1;
                    float sc = acc[i][j][r] * qinv;
                    if (METRIC == M_L2) sc = (rok ? a.vn2[m] : 0.0f) - 2.0f * sc + qextra;
                    if (METRIC == M_COS) sc = 1.0f - sc * (rok ? a.vrinv[m] : 0.0f) * qextra;
                    acc[i][j][r] = 0.0f;
                    const bool pass = ASC ? (sc <= thr) : (sc >= thr);
                    if (qok && rok && (a.emit_all || pass)) {
                        const uint32_t slot = a.emit_all ? (m - a.row0) : atomicAdd(&a.count[n], 1u);
                        if (slot < a.cap) a.cand[(size_t)n * a.cap + slot] = make_key(sc, m, ASC);
                    }
                }
            }
        }
    };

    // Flat sequence of (tile, slab) steps of this persistent block; the pipeline runs across tile
    // boundaries.  cur = step being computed, n1 / n2 = the next two steps.
    uint32_t cur_t = blockIdx.x, cur_s = 0;
    uint32_t n1_t = cur_t, n1_s = 1;
    if (n1_s == a.nslab) { n1_s = 0; n1_t += gridDim.x; }
    uint32_t n2_t = n1_t, n2_s = n1_s + 1;
    if (n2_s == a.nslab) { n2_s = 0; n2_t += gridDim.x; }

    load_v(vA, cur_t, cur_s);
    load_q(cur_s);
    store_v(vA, 0, cur_s);
    store_q(0);
    if (PD == 2 && n1_t < a.ntiles) load_v(vB, n1_t, n1_s);
    __syncthreads();
    int buf = 0;

    // X: registers whose slab goes to LDS at the end of this iteration; Y: registers free to receive
    // the slab two steps ahead (PD=2).  With PD=1, X is loaded and stored within the iteration.
    auto iteration = [&](f32x4(&X)[PV], f32x4(&Y)[PV]) {
        const bool v1 = n1_t < a.ntiles, v2 = n2_t < a.ntiles;
        if (v1) load_q(n1_s);
        if (PD == 2) {
            if (v2) load_v(Y, n2_t, n2_s);
        } else {
            if (v1) load_v(X, n1_t, n1_s);
        }
        compute(buf);
        if (v1) {
            store_v(X, buf ^ 1, n1_s);
            store_q(buf ^ 1);
        }
        __syncthreads();
        buf ^= 1;
        if (cur_s == a.nslab - 1) epilogue(cur_t);
        cur_t = n1_t; cur_s = n1_s;
        n1_t = n2_t; n1_s = n2_s;
        n2_s += 1;
        if (n2_s == a.nslab) { n2_s = 0; n2_t += gridDim.x; }
    };
    while (true) {
        if (PD == 2) iteration(vB, vA); else iteration(vA, vA);
        if (cur_t >= a.ntiles) break;
        iteration(vA, vB);
        if (cur_t >= a.ntiles) break;
    }
}

#endif  // LYNSE_EXPERIMENTS
// ------------------------------------------------------------------------------------------------
// k_scan_glds — the hot kernel, LDS-DMA edition (default).
//
// Same tile / MFMA / epilogue structure as k_scan_f16, but every operand byte travels HBM/L2 -> LDS
// with `global_load_lds_dwordx4` (no VGPR staging), through a 4-stage LDS ring of K=32 slabs, so up
// to three slabs (48 KB of rows + 48 KB of query image per CU) are in flight while one is computed:
//   per slab and wave: VPW + QPW LDS-DMA instructions (1 KiB each) -> s_waitcnt vmcnt((NS-2)*OPS)
//   -> ONE raw s_barrier -> issue slab g+3 into the stage that was computed last -> MFMA on slab g.
// Rows sit in LDS as f32 (the DMA cannot convert): the A fragment is two ds_read_b128 + cvt to f16 in
// registers.  Bank conflicts are removed by an XOR swizzle of the 16-B slot index applied on the
// per-lane SOURCE address (the DMA destination is lane-linear): rows use slot ^ ((row>>1)&7), the
// query image is stored pre-swizzled in global memory (slot ^ ((q>>2)&3)) by k_prep_queries.
// Out-of-range rows / K columns are clamped to valid addresses: clamped rows are masked in the
// epilogue, clamped columns meet zeros in the query image.
// ------------------------------------------------------------------------------------------------
constexpr int GL_BK = 32;   // K elements per slab
constexpr int GL_NS = 4;    // ring stages

template <int AUX>
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, AUX);
}

#ifdef LYNSE_EXPERIMENTS  // the f32-row LDS-DMA predecessor of k_scan_h16 (A/B reference only: make EXPERIMENTS=1)
// NS ring stages; NT_HINT = 2 marks the row stream non-temporal (each row byte is read once per
// batch by exactly one CU), the query image keeps the default policy (re-read by every CU from L2).
// TILED = work-list mode for IVF slabs (tile descriptors + per-group query images).
template <int WQ, int WR, int TQ, int TR, int METRIC, bool SCALE, int NS, int NT_HINT, bool TILED = false>
__global__ void __launch_bounds__(WQ * WR * 64, (WQ * WR >= 8) ? (WQ * WR / 4) : 2) k_scan_glds(ScanArgs a) {
    constexpr int NW = WQ * WR;
    constexpr int BQ = WQ * TQ * 32;
    constexpr int BR = WR * TR * 32;
    static_assert(NS >= 3, "ring depth");
    constexpr int V_BYTES = BR * GL_BK * 4;   // f32 rows per stage
    constexpr int Q_BYTES = BQ * GL_BK * 2;   // f16 query slab
    constexpr int STAGE = V_BYTES + Q_BYTES;
    constexpr int V_INSTR = V_BYTES / 1024;
    constexpr int Q_INSTR = Q_BYTES / 1024;
    static_assert(V_INSTR % NW == 0, "row DMA split");
    constexpr int VPW = V_INSTR / NW;
    constexpr int QPW = (Q_INSTR + NW - 1) / NW;
    constexpr int OPS = VPW + QPW;  // LDS-DMA instructions per wave per slab
    constexpr bool ASC = METRIC != M_IP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave % WQ, wr = wave / WQ;

    if (blockIdx.x >= a.ntiles) return;
    const uint32_t my_tiles = (a.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const uint32_t G = my_tiles * a.nslab;  // slab steps of this persistent block
    const bool ragged_k = (a.ld % GL_BK) != 0;  // last slab reaches past ld: clamp columns (they meet zeros in the query image)

    // ---- DMA issue stream: position (tile, slab), ring stage, per-lane source pointers of the tile.
    // Addresses advance incrementally (one 64-bit add per DMA); everything else is recomputed only at
    // tile changes — the per-slab SALU/VALU overhead of the first version cost more than the MFMAs.
    uint32_t v_rowoff[VPW], v_col[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        const uint32_t r = (wave * VPW + j) * 8 + (lane >> 3);  // row inside the tile
        v_rowoff[j] = r;
        v_col[j] = ((lane & 7) ^ ((r >> 1) & 7)) * 4;            // physical 16-B slot -> logical f32 column
    }
    const float* v_src[VPW];  // row pointers (incl. swizzled column) of the tile being issued
    const char* q_src[QPW];
    uint32_t q_dst[QPW];
#pragma unroll
    for (int j = 0; j < QPW; ++j) q_dst[j] = V_BYTES + ((wave * QPW + j) % Q_INSTR) * 1024;
    uint32_t is_tile = blockIdx.x, is_slab = 0, is_stage = 0, is_count = 0, is_tileseq = 0;
    uint32_t is_qslab = a.qpad * (GL_BK * 2);
    uint32_t is_rbase = 0, is_last = 0;
    // Per-row norms of a tile (L2: |v|^2, cosine: 1/|v|) travel by LDS-DMA too: one 1-KiB DMA per tile
    // into a small ring after the stages — an ordinary load in the epilogue would drain the pipeline.
    constexpr bool NORMS_LDS = !TILED && METRIC != M_IP;
    constexpr int NORM_RING = NS * STAGE;
    const float* norm_src = METRIC == M_L2 ? a.vn2 : a.vrinv;

    auto issue_enter_tile = [&]() {
        const char* qbase = reinterpret_cast<const char*>(a.Q16);
        if (TILED) {
            const IvfTile td = a.tiles[is_tile];  // uniform -> scalar loads
            is_rbase = td.row0;
            is_last = td.row0 + td.nrows - 1;
            qbase += (size_t)td.qimg_off * 2;
            is_qslab = BQ * (GL_BK * 2);
        } else {
            is_rbase = a.row0 + is_tile * BR;
            is_last = a.row1 - 1;
        }
        if (NORMS_LDS) {
            if (wave == 0)  // extra VM ops only make the counted waits more conservative (in-order retirement)
                glds16<0>(norm_src + is_rbase + lane * 4, smem + NORM_RING + (is_tileseq % NS) * 1024);
            ++is_tileseq;
        }
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            uint32_t row = is_rbase + v_rowoff[j];
            row = row < is_last ? row : is_last;  // clamped rows are masked in the epilogue
            v_src[j] = a.V + (size_t)row * a.ld + v_col[j];
        }
#pragma unroll
        for (int j = 0; j < QPW; ++j) q_src[j] = qbase + ((wave * QPW + j) % Q_INSTR) * 1024 + lane * 16;
    };
    // One DMA piece (1 KiB per wave) of the slab at the issue position: pieces 0..VPW-1 are row pieces,
    // VPW..OPS-1 query-image pieces.  Pieces are interleaved with the MFMA groups of the compute phase:
    // a global_load_lds stalls ~100 cycles at issue when all waves fire at once (measured 500-1000
    // cycles per slab right after the barrier) — behind MFMAs that stall is hidden.
    auto issue_piece = [&](int p) {
        char* stage = smem + is_stage * STAGE;
        if (p < VPW) {
            const uint32_t koff = is_slab * GL_BK;
            const float* src = v_src[p] + koff;
            if (ragged_k) {
                uint32_t col = koff + v_col[p];
                col = col < a.ld ? col : a.ld - 4;
                src = v_src[p] - v_col[p] + col;
            }
            glds16<NT_HINT>(src, stage + (wave * VPW + p) * 1024);
        } else {
            const int j = p - VPW;
            glds16<0>(q_src[j] + is_slab * is_qslab, stage + q_dst[j]);
        }
    };
    auto issue_advance = [&]() {
        // past the end the last real step is re-issued (keeps the per-wave DMA count uniform)
        is_stage = is_stage + 1 == NS ? 0 : is_stage + 1;
        if (++is_count < G) {
            if (++is_slab == a.nslab) {
                is_slab = 0;
                is_tile += gridDim.x;
                issue_enter_tile();
            }
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int p = 0; p < OPS; ++p) issue_piece(p);
        issue_advance();
    };

    f32x16 acc[TR][TQ];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment geometry
    const int l32 = lane & 31, hi = lane >> 5;
    const int a_swz = (l32 >> 1) & 7;   // (row>>1)&7 — tile-row offsets are multiples of 32
    const int b_swz = (l32 >> 2) & 3;   // (q>>2)&3
    const int a_base = (wr * (TR * 32) + l32) * (GL_BK * 4);
    const int b_base = V_BYTES + (wq * (TQ * 32) + l32) * (GL_BK * 2);

    // per-query constants of this lane's columns (loaded once: no ordinary loads inside the ring loop)
    float c_qinv[TQ], c_thr[TQ], c_extra[TQ];
    bool c_ok[TQ];
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
        const uint32_t n = wq * (TQ * 32) + j * 32 + l32;
        c_ok[j] = !TILED && n < a.nq;
        c_qinv[j] = c_ok[j] ? a.qinv[n] : 0.0f;
        c_thr[j] = c_ok[j] ? a.thr[n] : 0.0f;
        if (a.debug_flags & 2) c_thr[j] = ASC ? -LY_INF : LY_INF;
        c_extra[j] = 0.0f;
        if (METRIC == M_L2) c_extra[j] = c_ok[j] ? a.qn2[n] : 0.0f;
        if (METRIC == M_COS) c_extra[j] = c_ok[j] ? a.qrinv[n] : 0.0f;
    }
    // Make the compiler wait for these ordinary loads HERE: a first use inside the ring loop would
    // cost an s_waitcnt vmcnt(0) per tile, draining the in-flight LDS-DMA slabs.
#pragma unroll
    for (int j = 0; j < TQ; ++j) asm volatile("" : "+v"(c_qinv[j]), "+v"(c_thr[j]), "+v"(c_extra[j]));

    // Ring protocol: wait until at most (NS-2)*OPS of this wave's DMAs are outstanding (slab g landed)
    // -> ONE raw barrier (everyone's pieces landed, everyone finished slab g-1) -> refill the stage
    // that was computed last with slab g+NS-1 -> compute slab g.
    issue_enter_tile();
#pragma unroll
    for (int g0 = 0; g0 < NS - 1; ++g0) issue();

    uint32_t s_in_tile = 0, tile = blockIdx.x, c_stage = 0, c_tileseq = 0;
    const bool timing = (a.debug_flags & 64) && a.dbg;
    unsigned long long t_wait = 0, t_bar = 0, t_issue = 0, t_comp = 0, tp = timing ? __builtin_amdgcn_s_memtime() : 0;
    for (uint32_t g = 0; g < G; ++g) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * OPS) : "memory");
        if (timing) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_wait += t - tp; tp = t; }
        __builtin_amdgcn_s_barrier();
        if (timing) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_bar += t - tp; tp = t; }
        if (a.debug_flags & 4) issue();  // DMA-only experiment: no compute phase to interleave with

        const char* st = smem + c_stage * STAGE;
        c_stage = c_stage + 1 == NS ? 0 : c_stage + 1;
        if (!(a.debug_flags & 4)) {  // EXPERIMENT flag 4: DMA stream only, no LDS reads / MFMA
#pragma unroll
            for (int kk = 0; kk < GL_BK / 16; ++kk) {
                if ((a.debug_flags & 16) && kk == 1) break;  // EXPERIMENT: half the LDS reads + MFMAs
                half8 bf[TQ];
                const int lb = (kk * 2 + hi) ^ b_swz;
#pragma unroll
                for (int j = 0; j < TQ; ++j)
                    bf[j] = *reinterpret_cast<const half8*>(st + b_base + j * 32 * (GL_BK * 2) + lb * 16);
                const int la = (kk * 4 + hi * 2) ^ a_swz;  // physical slot of the first 16 B
#pragma unroll
                for (int i = 0; i < TR; ++i) {
                    const char* rp = st + a_base + i * 32 * (GL_BK * 4);
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(rp + la * 16);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(rp + (la ^ 1) * 16);
                    half8 h;
                    if (SCALE) {
                        h[0] = (_Float16)(x0[0] * a.sv); h[1] = (_Float16)(x0[1] * a.sv);
                        h[2] = (_Float16)(x0[2] * a.sv); h[3] = (_Float16)(x0[3] * a.sv);
                        h[4] = (_Float16)(x1[0] * a.sv); h[5] = (_Float16)(x1[1] * a.sv);
                        h[6] = (_Float16)(x1[2] * a.sv); h[7] = (_Float16)(x1[3] * a.sv);
                    } else {
                        h[0] = (_Float16)x0[0]; h[1] = (_Float16)x0[1]; h[2] = (_Float16)x0[2]; h[3] = (_Float16)x0[3];
                        h[4] = (_Float16)x1[0]; h[5] = (_Float16)x1[1]; h[6] = (_Float16)x1[2]; h[7] = (_Float16)x1[3];
                    }
                    if (a.debug_flags & 32) {  // EXPERIMENT: LDS reads + cvt only, no MFMA
                        asm volatile("" ::"v"(h));
#pragma unroll
                        for (int j = 0; j < TQ; ++j) asm volatile("" ::"v"(bf[j]));
                    } else {
#pragma unroll
                        for (int j = 0; j < TQ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, bf[j], acc[i][j], 0, 0, 0);
                    }
                    // refill pieces scheduled behind this MFMA group (slot kk*TR+i of NSLOT)
                    {
                        constexpr int NSLOT = (GL_BK / 16) * TR;
                        const int slot = kk * TR + i;
                        if (a.debug_flags & 128) {  // EXPERIMENT: all pieces behind the first MFMA group
                            if (slot == 0) {
#pragma unroll
                                for (int p = 0; p < OPS; ++p) issue_piece(p);
                            }
                        } else if (a.debug_flags & 256) {  // EXPERIMENT: all pieces behind the last MFMA group
                            if (slot == NSLOT - 1) {
#pragma unroll
                                for (int p = 0; p < OPS; ++p) issue_piece(p);
                            }
                        } else {
#pragma unroll
                            for (int p = 0; p < OPS; ++p)
                                if (p % NSLOT == slot) issue_piece(p);
                        }
                    }
                }
            }
            issue_advance();
        }

        if (++s_in_tile == a.nslab) {
            // ---- epilogue of a finished tile: score transform, threshold filter, candidate append.
            // Two passes: (1) build a per-lane bit mask of passing rows per query column, (2) ONE
            // returning atomic per (lane, column) reserves the slots, then the keys are written.
            uint32_t rbase = a.row0 + tile * BR;
            uint32_t row_end = a.row1;
            IvfTile td{};
            if (TILED) {
                td = a.tiles[tile];
                rbase = td.row0;
                row_end = td.row0 + td.nrows;
            }
            const float* nrm = reinterpret_cast<const float*>(smem + NORM_RING + (c_tileseq % NS) * 1024);
            ++c_tileseq;
            auto score = [&](int i, int j, int r, uint32_t m, bool rok) -> float {
                float sc = acc[i][j][r] * c_qinv[j];
                if (METRIC != M_IP) {
                    float nv;
                    if (NORMS_LDS) nv = nrm[m - rbase];
                    else nv = rok ? (METRIC == M_L2 ? a.vn2[m] : a.vrinv[m]) : 0.0f;
                    if (METRIC == M_L2) sc = nv - 2.0f * sc + c_extra[j];
                    else sc = 1.0f - sc * nv * c_extra[j];
                }
                return sc;
            };
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                uint32_t n = wq * (TQ * 32) + j * 32 + l32;
                if (TILED) {  // group-local query -> global query id and its per-query constants
                    c_ok[j] = n < td.nq;
                    n = c_ok[j] ? a.pair_q[td.pair0 + n] : 0u;
                    c_qinv[j] = c_ok[j] ? a.qinv[n] : 0.0f;
                    c_thr[j] = c_ok[j] ? a.thr[n] : 0.0f;
                    if (METRIC == M_L2) c_extra[j] = c_ok[j] ? a.qn2[n] : 0.0f;
                    if (METRIC == M_COS) c_extra[j] = c_ok[j] ? a.qrinv[n] : 0.0f;
                }
                if (a.emit_all && !TILED) {  // stage 0: slot = row, no atomics
#pragma unroll
                    for (int i = 0; i < TR; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t m = rbase + wr * (TR * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            const bool rok = m < row_end;
                            const float sc = score(i, j, r, m, rok);
                            if (c_ok[j] && rok && (m - a.row0) < a.cap)
                                a.cand[(size_t)n * a.cap + (m - a.row0)] = make_key(sc, m, ASC);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < TR; ++i) {  // one 32x32 block at a time keeps only 16 scores live
                        uint32_t msk = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t m = rbase + wr * (TR * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            const bool rok = m < row_end;
                            const float sc = score(i, j, r, m, rok);
                            const bool pass = ASC ? (sc <= c_thr[j]) : (sc >= c_thr[j]);
                            if (c_ok[j] && rok && pass) msk |= 1u << r;
                        }
                        if (msk) {
                            const uint32_t base = atomicAdd(&a.count[n], (uint32_t)__popc(msk));
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if ((msk >> r) & 1u) {
                                    const uint32_t m = rbase + wr * (TR * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                                    const uint32_t slot = base + (uint32_t)__popc(msk & ((1u << r) - 1u));
                                    if (slot < a.cap) a.cand[(size_t)n * a.cap + slot] = make_key(score(i, j, r, m, true), m, ASC);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            }
            s_in_tile = 0;
            tile += gridDim.x;
        }
        if (timing) {
            asm volatile("" ::"v"(acc[0][0][0]));
            const unsigned long long t = __builtin_amdgcn_s_memtime(); t_comp += t - tp; tp = t;
        }
    }
    if (timing && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * NW + wave) * 4;
        o[0] = t_wait; o[1] = t_bar; o[2] = t_issue; o[3] = t_comp;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the clamped tail DMAs before the LDS is released
}
#endif  // LYNSE_EXPERIMENTS

// f16 storage: rows [r0,r1) <- f16::to_f32(f16::from_f32(row)) in place (encode_f32_slice_as_le_bytes F16, RNE)
// F16 shards: f32 rows in -> f16 bits (RNE, what an F16 segment file keeps: src/storage/dtype.rs, flat_mmap.rs:187-221),
// pitch ld16 halves with zero pad columns; and the exact decode back for the row readers.
__global__ void __launch_bounds__(256) k_f32_to_f16_rows(_Float16* __restrict__ dst, uint32_t ld16, const float* __restrict__ src,
                                                         uint32_t src_pitch, uint32_t width, uint64_t nrows) {
    const uint64_t total = nrows * ld16;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld16;
        const uint32_t c = (uint32_t)(i % ld16);
        dst[i] = c < width ? (_Float16)src[r * src_pitch + c] : (_Float16)0.0f;
    }
}
__global__ void __launch_bounds__(256) k_f16_to_f32_rows(float* __restrict__ dst, uint32_t dst_pitch, const _Float16* __restrict__ src,
                                                         uint32_t ld16, uint32_t width, uint64_t nrows) {
    const uint64_t total = nrows * width;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / width;
        const uint32_t c = (uint32_t)(i % width);
        dst[r * dst_pitch + c] = (float)src[r * ld16 + c];
    }
}

__global__ void __launch_bounds__(256) k_round_rows_f16(float* __restrict__ V, uint32_t ld, uint32_t D, uint64_t r0, uint64_t r1) {
    const uint64_t total = (r1 - r0) * D;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        float* p = V + (r0 + i / D) * ld + (i % D);
        *p = (float)(_Float16)*p;
    }
}

// ------------------------------------------------------------------------------------------------
// k_rows_to_f16: the f16 SHADOW of the f32 rows that k_scan_h16 streams: out[r][c] = (half)(v[r][c] * sv),
// RNE — bit for bit the conversion k_scan_glds does in flight, done once per appended row at finalize.
// One thread per 8 output halves (16 B); columns [D, ld16) are zero.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_rows_to_f16(const T* __restrict__ V, uint32_t ld, uint32_t D, uint64_t r0,
                                                     uint64_t r1, float sv, _Float16* __restrict__ out, uint32_t ld16) {
    const uint32_t cpr = ld16 / 8;  // 16-B chunks per row
    const uint64_t total = (r1 - r0) * cpr;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = r0 + i / cpr;
        const uint32_t c0 = (uint32_t)(i % cpr) * 8;
        const T* src = V + r * ld + c0;
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (c0 + e < D) ? (_Float16)((float)src[e] * sv) : (_Float16)0.0f;
        *reinterpret_cast<half8*>(out + r * ld16 + c0) = h;
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan_h16 — the hot kernel (default): k_scan_glds over the f16 shadow rows.
//
// The coarse pass only ever used the rows rounded to f16 (the certified margin E is built on exactly
// that rounding), so the scan streams a resident f16 copy instead of converting f32 in flight: HALF
// the HBM bytes per row, half the LDS bytes per MFMA operand, no cvt instructions.  Exactness is
// untouched: survivors are rescored from the f32 rows in the reference's accumulation order.
// Geometry: K slab = 64 halves = 128 B per row AND per query, so rows and query image share one LDS
// format: 8 16-B slots per line, physical slot = logical ^ ((line>>1)&7) (applied on the DMA source
// address for rows, pre-applied in the image for queries).  A and B fragments are single
// ds_read_b128s.  Ring protocol as k_scan_glds, NS >= 2 stages.
// ------------------------------------------------------------------------------------------------

// ---- fused sample stage of k_scan_h16 (FS = 1) ----------------------------------------------------
__device__ __forceinline__ uint64_t fs_readlane_u64(uint64_t v, int src_uniform) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src_uniform), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src_uniform);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t fs_shfl_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Grid-wide hand-over of the fused sample stage: every workgroup is resident (grid <= CUs, one workgroup per CU by LDS),
// thread 0 arrives on a counter the prep kernel zeroed (agent-scope release in front: MI355X_MICROARCH.md, Guideline 16)
// and polls it relaxed; `flag` is a word of the LDS ring stage this workgroup has just consumed (free until the next slab
// step issues into it).  A launch that cannot get all its workgroups resident (another fully occupying launch holding
// CUs) never completes the count: the poll gives up after ~0.1 s, raises the abort word, and every workgroup leaves — the
// select behind the launch turns the abort into the overflow flag of every query and the host re-runs the batch on the
// ordinary plan.  Returns false on abort.
__device__ __forceinline__ bool fs_grid_sync(uint32_t* ctr, uint32_t* abort_word, uint32_t target, volatile uint32_t* flag, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores (and its LDS-DMA prefetch) have completed
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        uint32_t ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            if (__builtin_amdgcn_s_memtime() - t0 > (1ull << 28)) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *flag = ok;
    }
    __syncthreads();
    const uint32_t ok = *flag;
    __syncthreads();
    return ok != 0;
}

// The workgroup (NT threads) derives the first threshold of query q from the nkeys lane-max keys the sample tiles left in
// keys_g (best-first u64 keys, KEY_SENTINEL = no row).  k_select's threshold-only rule with two changes that keep it valid:
// the k "best" rows are the k smallest of a two-level "two smallest per lane" reduction (the k-th of them is >= the true k-th
// smallest, so >= k rows are at least as good as it: a valid tau, looser only when three of the true best k share a lane's
// stripe), and tau_x is the WORST exact score of exactly those k rows (k rows with an exact score >= tau_x exist).  cut =
// the tighter of tau -/+ 2E and tau_x -/+ E, as k_select's cut_of.  Fewer than k real keys: the threshold stays open.
// scr_a / scr_b: two free 32 KiB ring stages of the caller's LDS.  Every step is one memory round trip for the whole
// workgroup: the keys (one batch of loads per wave), then the k rows, staged in LDS with coalesced loads so that the
// reference-order FMA chains of exact_score run out of LDS (a chain straight from HBM is D / 64 dependent round trips —
// 70-100 us per query in the first version of this function).
template <int NT>
__device__ __forceinline__ void fused_sample_threshold(const uint64_t* __restrict__ keys_g, uint32_t nkeys, uint32_t k, float marg2, int metric,
                                                       int ip_form, const float* __restrict__ qv, const float* __restrict__ V, uint32_t ld,
                                                       uint32_t D, float* __restrict__ thr_q, char* scr_a, char* scr_b, int tid) {
    constexpr int NWV = NT / 64;
    const int lane = tid & 63, wave = tid >> 6;
    const bool asc = metric_ascending(metric);
    uint64_t* cand2 = reinterpret_cast<uint64_t*>(scr_b);                 // [NWV][128] per-wave candidates
    uint32_t* sel_rows = reinterpret_cast<uint32_t*>(scr_b + 8192);       // [k] rows to rescore
    float* sel_score = reinterpret_cast<float*>(scr_b + 8192 + 256);      // [k] their exact scores
    volatile float* s_tau = reinterpret_cast<volatile float*>(scr_b + 8192 + 512);  // [0] tau, [1] 1.0 = k keys found
    float* q_lds = reinterpret_cast<float*>(scr_b + 9216);                // [D] the query (<= 22 KiB)
    float* rows_lds = reinterpret_cast<float*>(scr_a);                    // [chunk][D] rows being rescored
    auto top2 = [&](uint64_t key, uint64_t& b0, uint64_t& b1) {
        const bool lt0 = key < b0, lt1 = key < b1;
        b1 = lt0 ? b0 : (lt1 ? key : b1);
        b0 = lt0 ? key : b0;
    };
    // ---- level 1: every wave reduces its share of the keys to two per lane
    {
        uint64_t b0 = KEY_SENTINEL, b1 = KEY_SENTINEL;
        const uint32_t per_wave = (nkeys + NWV - 1) / NWV, w0 = wave * per_wave, w1 = (w0 + per_wave < nkeys) ? w0 + per_wave : nkeys;
        for (uint32_t i0 = w0; i0 < w1; i0 += 64 * 8) {
            uint64_t kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + u * 64 + lane;
                kv[u] = i < w1 ? keys_g[i] : KEY_SENTINEL;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) top2(kv[u], b0, b1);
        }
        cand2[wave * 128 + lane] = b0;
        cand2[wave * 128 + 64 + lane] = b1;
    }
    const bool q_in_lds = D * 4u <= 22528u;
    if (q_in_lds)
        for (uint32_t i = tid; i < D; i += NT) q_lds[i] = qv[i];
    if (tid == 0) { s_tau[0] = 0.0f; s_tau[1] = 0.0f; }
    __syncthreads();
    // ---- level 2 (one wave): two per lane again, ranks, the k smallest
    if (wave == 0) {
        uint64_t b0 = KEY_SENTINEL, b1 = KEY_SENTINEL;
#pragma unroll
        for (int u = 0; u < 2 * NWV; ++u) top2(cand2[u * 64 + lane], b0, b1);
        uint32_t r0 = 0, r1 = 0;  // ranks among the 128 candidates (real keys are unique: the row is part of the key)
        for (int l = 0; l < 64; ++l) {
            const uint64_t o0 = fs_readlane_u64(b0, l), o1 = fs_readlane_u64(b1, l);
            r0 += (o0 < b0 ? 1u : 0u) + (o1 < b0 ? 1u : 0u);
            r1 += (o0 < b1 ? 1u : 0u) + (o1 < b1 ? 1u : 0u);
        }
        const bool in0 = b0 != KEY_SENTINEL && r0 < k, in1 = b1 != KEY_SENTINEL && r1 < k;
        if (in0) sel_rows[r0] = key_row(b0);
        if (in1) sel_rows[r1] = key_row(b1);
        const uint64_t mk = __ballot((in0 && r0 == k - 1) || (in1 && r1 == k - 1));
        if (mk) {   // the key of rank k - 1 exists <=> at least k real keys
            const int src = __builtin_ctzll(mk);
            const uint64_t kk0 = fs_readlane_u64(b0, src), kk1 = fs_readlane_u64(b1, src);
            const uint32_t rr0 = __builtin_amdgcn_readlane(r0, src);
            if (lane == 0) { s_tau[0] = key_score(rr0 == k - 1 ? kk0 : kk1, asc); s_tau[1] = 1.0f; }
        }
    }
    __syncthreads();
    if (s_tau[1] == 0.0f) return;   // fewer than k sample keys: the threshold stays open (uniform: every thread reads the same word)
    // ---- exact scores of the k rows, in chunks that fit the row stage
    const uint32_t chunk_rows = (32768u / (D * 4u)) < 1u ? 1u : (32768u / (D * 4u));
    const int g = lane & 7;
    for (uint32_t c0 = 0; c0 < k; c0 += chunk_rows) {
        const uint32_t cn = (k - c0 < chunk_rows) ? k - c0 : chunk_rows;
        if (D * 4u <= 32768u) {
            const uint32_t total = cn * D;
            for (uint32_t i = tid; i < total; i += NT) {
                const uint32_t r = i / D, d = i - r * D;
                rows_lds[i] = V[(size_t)sel_rows[c0 + r] * ld + d];
            }
        }
        __syncthreads();
        for (uint32_t r = tid >> 3; r < cn; r += NT / 8) {   // (the 8 lanes of a group share r: uniform control flow inside exact_score)
            const float* vrow = (D * 4u <= 32768u) ? rows_lds + (size_t)r * D : V + (size_t)sel_rows[c0 + r] * ld;
            float sc = exact_score(metric, ip_form, q_in_lds ? q_lds : qv, vrow, D, g);
            if (sc != sc) sc = asc ? LY_INF : -LY_INF;   // NaN sorts last (make_key)
            if (g == 0) sel_score[c0 + r] = sc;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float worst = asc ? -LY_INF : LY_INF;
        for (uint32_t i = 0; i < k; ++i) worst = asc ? fmaxf(worst, sel_score[i]) : fminf(worst, sel_score[i]);
        const float tau = s_tau[0];
        const float c = asc ? tau + marg2 : tau - marg2;
        const float cx = asc ? worst + 0.5f * marg2 : worst - 0.5f * marg2;
        *thr_q = asc ? (cx < c ? cx : c) : (cx > c ? cx : c);
    }
    __syncthreads();
}

// ---- accumulators in AGPRs (one wave per SIMD: 4 waves x 4 x 4 blocks of 32 x 32, 256 accumulator registers per lane) ----
// hipcc keeps 4 of the 16 accumulator tuples in a[0:63] and cycles the rest through scratch around every group of four MFMAs
// when it allocates them itself (1152 B of scratch per lane inside the MFMA loop, ISA inspected); here the MFMAs are inline
// assembly on FIXED AGPR tuples a[16 t : 16 t + 15], the epilogue reads / clears them with v_accvgpr_read / _write, and the
// compiler never sees an accumulator value during the MFMA loop.
template <typename F, int... I>
__device__ __forceinline__ void ly_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void ly_static_for(F&& f) { ly_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int R>
__device__ __forceinline__ float ly_agpr_take() {   // read accumulator register R and clear it
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]\n\tv_accvgpr_write_b32 a[%1], 0" : "=v"(x) : "n"(R));
    return x;
}
template <int T, typename V>
__device__ __forceinline__ void ly_mfma_i8_agpr(V a, V b) {
    asm volatile("v_mfma_i32_32x32x32_i8 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(T * 16), "n"(T * 16 + 15));
}
__device__ __forceinline__ void ly_agpr_clear_all() {   // (the clobber list is what tells the compiler that the kernel uses a0..a255)
    ly_static_for<256>([&](auto rc) { asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(decltype(rc)::value)); });
    asm volatile("s_nop 4" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
}

// 32 x 32 x 64 FP4 x FP4 MFMA with unit block scales (E8M0 127 = 2^0): a / b carry 16 B per lane (32 nibbles), the upper half of
// the 8-dword operands is not read for FP4.  (Which K index a nibble position stands for does not matter here: rows and queries
// are packed the same way, and a dot product does not depend on the order of its terms.)
// `unit` = 0x7f7f7f7f in a VGPR the CALLER keeps live across its loop (made opaque there): as an immediate the compiler
// re-materialises it with a v_mov at the bottom of the loop body and every MFMA of the body sinks below that v_mov — past the
// sched_barriers, all 32 in one cluster behind all 24 fragment reads (ISA inspected: 236 B of scratch from the fragments alone).
template <typename V>
__device__ __forceinline__ f32x16 ly_mfma_fp4(V a, V b, f32x16 c, int unit) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    const i32x4 a4 = __builtin_bit_cast(i32x4, a), b4 = __builtin_bit_cast(i32x4, b);
    const i32x8 a8 = {a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0}, b8 = {b4[0], b4[1], b4[2], b4[3], 0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, unit, 0, unit);
#else
    (void)a8; (void)b8; (void)unit;
    return c;
#endif
}

constexpr int HK = 64;  // K elements per slab

// DBG (compile-time experiments, never launched by the product path): 1 no MFMA, 2 no LDS fragment reads,
// 4 no query-image DMA, 8 no row DMA, 16 no epilogue.  FILT compiles the subset-filter paths in (mask / row_ids of ScanArgs): they
// cost registers the unfiltered kernel does not have to spare (56 B/lane of scratch and 13 % of its speed when they
// were runtime branches).
#ifndef LYNSE_ZEROC
#define LYNSE_ZEROC 1   // (0: the DENSE float epilogue clears its accumulators, A/B build)
#endif
template <int WQ, int WR, int TQ, int TR, int METRIC, int NSV, int NSQ, int NT_HINT, bool TILED = false, bool RAG = true, int DBG = 0, bool FILT = false,
          int I8Q = 0, int EMIT = -1, int PLACE = 0, bool DENSE = false, bool PRIO = false, bool QCREG = true, int FS = 0>
__global__ void __launch_bounds__(WQ * WR * 64, (TQ * TR >= 16) ? 1 : ((WQ * WR >= 8) ? (WQ * WR / 4) : 2)) k_scan_h16(ScanArgs a) {
    // Two LDS rings: NSV stages of row slabs (HBM latency: deeper) and NSQ <= NSV stages of query-image
    // slabs (L2 latency) — 3 + 2 stages of 32 KiB fill the 160 KiB of a CU for the 256 x 256 tile.
    // I8 = SQ8 pass 1 (FLAT-*-SQ8): rows and query image are signed bytes (code - 128), a slab is 128 elements = the same
    // 128 B per line, the MFMA is v_mfma_i32_32x32x32_i8 (exact integers), the epilogue rebuilds the u32 score of the
    // reference's u8 kernels (flat_mmap.rs:5847-5863) from the i8 dot product and per-row / per-query sums.
    // I8Q = 2: CERTIFIED int8 coarse pass (FLAT-IP): rows are the same SQ8 codes, the query image is the symmetric int8
    // image of w = q / scale (k_i8c_prep_queries); coarse score = B_q + s_q * (int dot), with a certified bound on its
    // distance to the reference-order f32 score — half the HBM / LDS-DMA bytes and half the MFMA time of
    // the f16 shadow; survivors are rescored exactly from the f32 rows as always.
    // I8Q = 3 (F4): the operands are FP4 (E2M1) nibbles, two per byte — +1.0 / -1.0 / 0 are exact in it — on
    // v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales: batched Hamming as a +-1 GEMM (k_bits_to_fp4) at twice the int8 MFMA
    // rate and half the bytes.  A 128-B line holds 256 elements; the K loop, the rings and the epilogue are those of the
    // certified int8 pass (the f32 accumulators hold exact integers: converted where the integer epilogue reads them).
    // I8Q = 4 (I8L): squared L2 on the PLAIN SQ8 codes — the int8 MFMA of the certified pass feeding the FLOAT L2 epilogue of the
    // f16 shadow: the accumulators become (float)dot when a tile is done and from there the kernel is the f16 L2 kernel, with
    // qinv = s_q and qn2 = |q|^2 - 2 B_q, i.e. coarse distance = |v|^2 - 2 s_q dot + (|q|^2 - 2 B_q) with the exact f32 row norm
    // from the norm ring (k_i8c_prep_queries, l2n).  768 B per row instead of the 896 of the augmented codes, IP-sized margins.
    constexpr bool I8OPS = I8Q != 0;        // byte-addressed operands (SQ8 pass 1, the certified coarse pass, FP4 pairs, I8L)
    constexpr bool I8L = I8Q == 4;
    constexpr bool I8 = I8OPS && !I8L;      // ... with an integer-score epilogue
    static_assert(!I8L || (METRIC == M_L2 && !TILED), "int8 operands with the float epilogue: squared L2, FLAT");
    constexpr bool F4 = I8Q == 3;
    constexpr bool I8C = I8Q == 2 || F4;
    static_assert(!I8C || METRIC == M_IP, "certified int8 coarse pass: IP only");
    constexpr bool AG = I8Q == 2 && TQ * TR == 16;   // one wave per SIMD: accumulators in fixed AGPR tuples (ly_mfma_i8_agpr)
    constexpr bool ZEROC = DENSE && I8Q == 0 && !TILED && !FILT && LYNSE_ZEROC;   // (= DENSEF below) the float DENSE threshold stages: tiles start from src C = 0
#ifndef LYNSE_DEFER
#define LYNSE_DEFER 1
#endif
    // DEFER: the MFMAs of a slab's LAST k-step run at the start of the NEXT slab step, behind the barrier and the first fragment
    // reads of the new slab — every slab step used to open with all eight waves waiting for their first LDS reads (nothing left
    // to issue: both waves of a SIMD stand at the same point), now the deferred MFMA group covers that round trip.  The
    // fragments of the deferred k-step are in registers before the barrier, so "every wave has finished reading slab g" still
    // holds when the ring stages are refilled.  (256 x 256 int8 tilings; the tile epilogue follows the deferred group.)
    // Measured (MI355X, 10M x 768 x 256, builds alternated on one box): the 8-wave tilings LOSE 1.7 % with it (2.20 against 2.165
    // ms: their LDS port is as busy as the matrix pipe — 192 KB of fragment reads + 64 KB of DMA writes per slab step are 2048
    // cycles at 128 B / cycle, the MFMAs are 2048 too — so there is no idle round trip to cover, only added branches); the
    // one-wave-per-SIMD tiling (AG: a third fewer fragment reads, nobody else to issue while a wave waits) is what it is for.
    constexpr bool DEFER = LYNSE_DEFER && AG;
    static_assert(!AG || (!TILED && !FS && DBG == 0 && EMIT == 0), "AGPR accumulators: threshold stages of the FLAT int8 scan");
    constexpr int ES = I8OPS ? 1 : 2;       // element size in bytes
    constexpr int KS = 128 / ES;            // elements per slab
    constexpr int EPS = 16 / ES;            // elements per 16-B slot
    constexpr int NW = WQ * WR;
    constexpr int BQ = WQ * TQ * 32;
    constexpr int BR = WR * TR * 32;
    static_assert(NSQ >= 2 && NSV >= NSQ, "ring depths");
    constexpr int LINE = HK * 2;              // bytes per row / query per slab
    constexpr int V_BYTES = BR * LINE;
    constexpr int Q_BYTES = BQ * LINE;
    constexpr int V_INSTR = V_BYTES / 1024;
    constexpr int Q_INSTR = Q_BYTES / 1024;
    static_assert(V_INSTR % NW == 0, "row DMA split");
    constexpr int VPW = V_INSTR / NW;
    constexpr int QPW = (Q_INSTR + NW - 1) / NW;
    constexpr int OPS = VPW + QPW;  // LDS-DMA instructions per wave per slab step: QPW query pieces, then VPW row pieces
    constexpr int NSLOT = (HK / 16) * TR;  // MFMA groups per slab behind which the pieces are issued
    static_assert(NSV == NSQ || OPS <= NSLOT, "unequal ring depths need the query pieces issued before the row pieces");
    // step g needs rows(g) and queries(g): everything younger than queries(g) may still be in flight
    constexpr int WAIT_OPS = (NSV > NSQ ? VPW : 0) + (NSQ - 2) * OPS;
    constexpr int Q_RING = NSV * V_BYTES;
    constexpr int NORM_RING = Q_RING + NSQ * Q_BYTES;
    constexpr bool ASC = METRIC != M_IP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave % WQ, wr = wave / WQ;

    const uint32_t ntiles_ = (TILED && a.ntiles_dev) ? __builtin_amdgcn_readfirstlane(*a.ntiles_dev) : a.ntiles;
    if (blockIdx.x >= ntiles_) return;
    [[maybe_unused]] const unsigned long long t_kernel0 = FS ? __builtin_amdgcn_s_memtime() : 0ull;
    // blockIdx.y: which chunk of BQ queries this workgroup scores (one launch scores a.qpad / BQ chunks against the same rows:
    // the k-means assignment step searches thousands of rows against a few thousand centroids); 0 for ordinary searches
    const uint32_t qchunk = (TILED || I8OPS || FILT) ? 0u : blockIdx.y * (uint32_t)BQ;  // (the int8 passes always run one chunk)
    // FS (fused sample stage): the workgroup's FIRST tile is its sample tile (rows [blockIdx.x * fs_stride, +BR) of the whole
    // shard), then its ordinary tiles blockIdx.x, blockIdx.x + grid, ... of [row0, row1) — the tile counters start one grid
    // stride below blockIdx.x (modulo 2^32) so that the ordinary advance lands on blockIdx.x
    static_assert(!FS || (!TILED && !FILT && EMIT == 0), "fused sample stage: unfiltered threshold stages of the FLAT scan");
    const uint32_t my_tiles = (ntiles_ - blockIdx.x + gridDim.x - 1) / gridDim.x + (FS ? 1u : 0u);
    const uint32_t G = my_tiles * a.nslab;
    const uint32_t tstride = a.tile_stride ? a.tile_stride : (uint32_t)BR;
    const bool ragged_k = RAG && (a.ld16 % KS) != 0;  // last slab reaches past ld16: clamp columns (they meet zeros in the query image)

    // DMA addressing: every source address is a UNIFORM 64-bit base (tile / slab / piece: SGPRs) plus a 32-bit per-lane
    // byte offset (one VGPR per piece) — the saddr form of global_load_lds.  (64-bit per-lane pointers cost 16 VGPRs
    // more; the 256 x 256 IP tiling has none to spare.)
    uint32_t v_off[VPW];       // row pieces: ((row slot, clamped to the last valid row) * ld16 + swizzled column) * ES
    const char* v_base = nullptr;  // uniform: first row of the tile being issued
    const char* q_base = nullptr;  // uniform: query image of the tile being issued
    const uint32_t q_lane = lane * 16;
    uint32_t q_piece[QPW];  // uniform
#pragma unroll
    for (int j = 0; j < QPW; ++j) q_piece[j] = ((wave * QPW + j) % Q_INSTR) * 1024;
    auto v_swz_col = [&](int j) -> uint32_t {  // physical 16-B slot -> logical element column of this lane's piece
        const uint32_t r = (wave * VPW + j) * 8 + (lane >> 3);
        return ((lane & 7) ^ ((r >> 1) & 7)) * EPS;
    };
    // row stream position
    uint32_t vs_tile = FS ? blockIdx.x - gridDim.x : blockIdx.x, vs_slab = 0, vs_stage = 0, vs_count = 0, vs_tileseq = 0;
    // query stream position
    uint32_t qs_tile = blockIdx.x, qs_slab = 0, qs_stage = 0, qs_count = 0;
    uint32_t qs_qslab = a.qpad * LINE;
    constexpr bool NORMS_LDS = !TILED && !I8C && (METRIC != M_IP || I8) && (NORM_RING + (NSV + 1) * 1024 <= 160 * 1024);
    constexpr int NORM_SLOTS = NSV + 1;
    // Per-query constants of the epilogue (1/scale, the additive term, the threshold): constant for the whole launch, so each
    // wave keeps the TQ*32 queries of its column in QCN registers per constant (query q of the column lives in lane q % 64 of
    // register q / 64) and the epilogue fetches its lane's value with ds_bpermute.  Loading them from global memory in the
    // epilogue put every tile's epilogue behind the LDS-DMA still in flight (VM operations retire in order: the load of a
    // 4-byte constant waited for the next slab's 64 KiB) and behind one L2 round trip per column block.
    // (FILT && I8C: the masked certified-int8 scan — only the DENSE threshold epilogue and the emit-all sample are instantiated
    // for it, both with registers to spare)
    constexpr bool QC_REG = QCREG && !TILED && (!FILT || I8C) && !(RAG && WR >= 4);  // (the ragged <2,4,4,2> bodies have no registers to spare: 8 B of scratch with them)
    constexpr int QCN = (TQ * 32 + 63) / 64;
    static_assert(!FS || (QC_REG && I8C), "fused sample stage: the certified int8 pass with register-resident query constants");
    float qc_inv[QCN], qc_extra[QCN], qc_thr[QCN];
    // (re)loads the thresholds of the wave's query column; FS: called again once the grid has agreed on them
    auto load_qc_thr = [&](const float* thr_src, uint32_t nq_) {
#pragma unroll
        for (int t = 0; t < QCN; ++t) {
            const uint32_t nl = t * 64 + lane;
            const uint32_t n = qchunk + wq * (TQ * 32) + nl;
            const bool ok = nl < TQ * 32 && n < nq_;
            qc_thr[t] = ok ? thr_src[n] : 0.0f;
            if constexpr (I8C) {
                // INTEGER image of the threshold: the coarse score B_q + s_q * (float)dot is monotone non-decreasing in the
                // integer dot product (s_q >= 0; conversion, product and sum round monotonically), so "score >= thr" is
                // EXACTLY "dot >= T" with T = the smallest dot whose score passes — found once per launch by bisection over
                // |dot| <= 2^29 (128 * 127 * D).  The epilogue then compares accumulators as integers: no cvt / mul / add per
                // element.  T = 2^29 + 1: nothing passes (thr = NaN / +inf, or a query slot past nq).
                const float s_q = qc_inv[t], b_q = qc_extra[t], th = qc_thr[t];
                int lo = -(1 << 29), hi_ = 1 << 29;
#pragma unroll 1
                for (int it = 0; it < 31; ++it) {
                    const int mid = lo + ((hi_ - lo) >> 1);
                    const bool ge = (b_q + s_q * (float)mid) >= th;
                    hi_ = ge ? mid : hi_;
                    lo = ge ? lo : mid + 1;
                }
                qc_thr[t] = __int_as_float(ok ? lo : 0x7fffffff);
            }
        }
    };
    if (QC_REG) {
#pragma unroll
        for (int t = 0; t < QCN; ++t) {
            const uint32_t nl = t * 64 + lane;
            const uint32_t n = qchunk + wq * (TQ * 32) + nl;
            const bool ok = nl < TQ * 32 && n < a.nq;
            qc_inv[t] = ok ? a.qinv[n] : 0.0f;
            qc_extra[t] = 0.0f;
            if (METRIC == M_L2 || I8) qc_extra[t] = ok ? a.qn2[n] : 0.0f;
            if (METRIC == M_COS && !I8) qc_extra[t] = ok ? a.qrinv[n] : 0.0f;
            qc_thr[t] = 0.0f;
        }
        if (!FS) load_qc_thr(a.thr, a.nq);   // (FS: the thresholds do not exist yet; the sample tile's epilogue needs none)
    }
    const float* norm_src = (METRIC == M_L2 || I8) ? a.vn2 : a.vrinv;  // I8: a.vn2 carries the per-row int sums

    auto v_enter_tile = [&]() {
        uint32_t rbase, last;
        if (TILED) {
            const IvfTile td = a.tiles[vs_tile];
            rbase = td.row0;
            last = td.row0 + td.nrows - 1;
        } else {
            rbase = a.row0 + vs_tile * tstride;
            last = a.row1 - 1;
        }
        if (NORMS_LDS) {
            static_assert(BR <= 256, "one 1-KiB norm DMA covers a tile");
            if (wave == 0)  // extra VM ops only make the counted waits more conservative (in-order retirement)
                glds16<0>(norm_src + rbase + lane * 4, smem + NORM_RING + (vs_tileseq % NORM_SLOTS) * 1024);
            ++vs_tileseq;
        }
        v_base = reinterpret_cast<const char*>(a.V16) + (size_t)rbase * a.ld16 * ES;
        const uint32_t span = last - rbase;  // rows past `last` re-read it (masked in the epilogue)
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            uint32_t r = (wave * VPW + j) * 8 + (lane >> 3);  // row slot inside the tile
            r = r < span ? r : span;
            v_off[j] = (r * a.ld16 + v_swz_col(j)) * ES;
        }
    };
    auto q_enter_tile = [&]() {
        q_base = reinterpret_cast<const char*>(a.Q16) + (size_t)qchunk * LINE;  // (chunk's first query; the slab stride is a.qpad lines)
        if (TILED) {
            const IvfTile td = a.tiles[qs_tile];
            q_base += (size_t)td.qimg_off * 2;
            qs_qslab = BQ * LINE;
        }
    };
    // piece p of a slab step: 0..QPW-1 query-image pieces, QPW..OPS-1 row pieces
    auto issue_piece = [&](int p) {
        if (p < QPW) {
            // saddr form like the row pieces: uniform 64-bit base (SGPRs) + the lane's 32-bit byte offset.  (Left to itself the
            // compiler keeps the uniform slab counter in a VGPR and issues the 64-bit-per-lane address form: ~110 cycles of
            // issue per instruction instead of ~13, measured with s_memtime around the four pieces of a slab step.)
            const uint32_t q_uni = __builtin_amdgcn_readfirstlane(qs_slab * qs_qslab + q_piece[p]);
            uint32_t ql = q_lane;
            asm volatile("" : "+v"(ql));  // opaque: keeps `q_base + q_lane` from being hoisted into a 64-bit VGPR pair (which forces the VGPR-address form)
            if (!(DBG & 4)) glds16<0>(q_base + (size_t)q_uni + ql, smem + Q_RING + qs_stage * Q_BYTES + q_piece[p]);
        } else {
            if (DBG & 8) return;
            const int j = p - QPW;
            const uint32_t koff = vs_slab * KS;
            uint32_t off = v_off[j];
            if (ragged_k) {  // the last slab reaches past ld16: clamp the column (it meets zeros in the query image)
                const uint32_t c0 = v_swz_col(j);
                uint32_t col = koff + c0;
                col = col < a.ld16 ? col : a.ld16 - EPS;
                off = v_off[j] + (col - koff - c0) * ES;   // (wraps for clamped columns: 32-bit arithmetic, added below)
                glds16<NT_HINT>(v_base + (size_t)koff * ES + (size_t)(int64_t)(int32_t)(off - v_off[j]) + v_off[j], smem + vs_stage * V_BYTES + (wave * VPW + j) * 1024);
                return;
            }
            glds16<NT_HINT>(v_base + (size_t)koff * ES + off, smem + vs_stage * V_BYTES + (wave * VPW + j) * 1024);
        }
    };
    // past the end the last real step is re-issued (keeps the per-wave DMA count uniform)
    auto v_advance = [&]() {
        vs_stage = vs_stage + 1 == NSV ? 0 : vs_stage + 1;
        if (++vs_count < G) {
            if (++vs_slab == a.nslab) {
                vs_slab = 0;
                vs_tile += gridDim.x;
                v_enter_tile();
            }
        }
    };
    auto q_advance = [&]() {
        qs_stage = qs_stage + 1 == NSQ ? 0 : qs_stage + 1;
        if (++qs_count < G) {
            if (++qs_slab == a.nslab) {
                qs_slab = 0;
                if (TILED) {
                    qs_tile += gridDim.x;
                    q_enter_tile();
                }
            }
        }
    };

    uint32_t segpk = 0;  // segmented emission: this lane's TQ per-query slot counters, 8 bits each
    static_assert(TQ <= 4, "packed segment counters");
    // the integer dot product an accumulator holds (I8C epilogues compare integers): the i32 MFMA's bits, or the exact integer
    // value of the FP4 MFMA's f32 accumulator
    [[maybe_unused]] auto acc_int = [](float x) -> int { return F4 ? (int)x : __float_as_int(x); };
    [[maybe_unused]] int f4_unit = 0x7f7f7f7f;   // E8M0 scale bytes 127 = 2^0 (ly_mfma_fp4)
    if constexpr (F4) asm volatile("" : "+v"(f4_unit));
    f32x16 acc[TR][TQ];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    if constexpr (AG) ly_agpr_clear_all();   // (AG: acc[][] only carries one query column at a time through the epilogue)

    const int l32 = lane & 31, hi = lane >> 5;
    const int swz = (l32 >> 1) & 7;   // (line>>1)&7 — tile offsets are multiples of 32
    const int a_base = (wr * (TR * 32) + l32) * LINE;
    const int b_base = (wq * (TQ * 32) + l32) * LINE;

    // Epilogue-only kernel arguments are re-read from the kernarg segment through a laundered pointer when a tile's
    // epilogue starts: kept live across the MFMA loop they fill the SGPR file (106 of 106), push uniform loop state into
    // VGPRs and from there into scratch — whose reloads sit behind s_waitcnt vmcnt(0) and drain the DMA ring.
    typedef const __attribute__((address_space(4))) ScanArgs* EpiArgsPtr;
    auto epi_args = [&]() -> EpiArgsPtr {
        EpiArgsPtr p = (EpiArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return p;
    };
    // Per-query constants of the two-level epilogue filter, (re)loaded per tile in the epilogue (NOT kept in registers
    // across the MFMA loop: 12+ VGPRs the 256 x 256 tilings do not have): 1 / scale (I8C: s_q), |q|^2 or 1 / |q| (I8C: B_q),
    // and c_pre — the threshold of LEVEL 1.  Level 1 reduces the accumulators of a query column to ONE value that is
    // monotone in the score and compares it with the per-query threshold:
    //   IP (f16 and certified int8): the score expression itself is monotone non-decreasing in the accumulator
    //       (acc * qinv with qinv > 0; B_q + s_q * dot with s_q >= 0), so level 1 evaluates the EXACT expression on the
    //       column maximum — it rejects a block iff level 2 would reject every row of it;
    //   L2 / cosine: 2 q.v - |v|^2 resp. q.v / |v| against the threshold mapped into that space and LOOSENED by more than
    //       the rounding differences between this form and the exact expression: never rejects a row level 2 would accept.
    // Only column blocks with a hit run level 2 (the exact expression against the exact threshold + emission).
    // c_pre = -inf: always run level 2.
    float c_qinv[TQ], c_pre[TQ], c_extra[TQ], c_thr[TQ];
    bool c_ok[TQ];
    auto set_pre = [&](int j, float thr, float vmax2) {
        float pre = -LY_INF;
        if (!I8 && METRIC != M_IP && c_ok[j]) {
            const float qi = c_qinv[j];
            if (METRIC == M_L2) {
                const float qn2 = c_extra[j];
                pre = (qn2 - thr) - 2e-6f * (qn2 + fabsf(thr) + vmax2);
                if (!(vmax2 > 0.0f)) pre = -LY_INF;
            } else {
                const float den = qi * c_extra[j];                // qinv / |q|
                pre = ((1.0f - thr) - 1e-6f * (1.0f + fabsf(thr))) / den;
                pre = pre - fabsf(pre) * 2e-6f;
                if (!(den > 0.0f)) pre = -LY_INF;
            }
            if (!(fabsf(pre) < 3.0e38f)) pre = -LY_INF;           // inf / NaN (open threshold, overflow): no pre-filter
        }
        c_pre[j] = pre;
    };

    // Prologue: the issue order of the steady state (per step: queries(s+NSQ-1), then rows(s+NSV-1))
    if constexpr (FS != 0) {   // the sample tile of this workgroup (its arguments through the laundered pointer: no SGPRs live in the loop)
        static_assert(!FS || !NORMS_LDS, "fused sample stage: no norm ring");
        const EpiArgsPtr ea = epi_args();
        const uint32_t rbase = blockIdx.x * ea->fs_stride, span = ea->fs_rows - 1 - rbase;
        v_base = reinterpret_cast<const char*>(a.V16) + (size_t)rbase * a.ld16 * ES;
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            uint32_t r = (wave * VPW + j) * 8 + (lane >> 3);
            r = r < span ? r : span;
            v_off[j] = (r * a.ld16 + v_swz_col(j)) * ES;
        }
    } else {
        v_enter_tile();
    }
    q_enter_tile();
#pragma unroll
    for (int s0 = -(NSV - 1); s0 < 0; ++s0) {
        if (s0 + NSQ - 1 >= 0) {
#pragma unroll
            for (int p = 0; p < QPW; ++p) issue_piece(p);
            q_advance();
        }
#pragma unroll
        for (int p = QPW; p < OPS; ++p) issue_piece(p);
        v_advance();
    }

    // static priority for the second-dispatched half of an 8-wave workgroup (MI355X_MICROARCH.md, "Two waves per SIMD", item 4):
    // it is the arbitration loser on every slab step (s_memtime: 3500-3900 cycles per step against 2600-3500 for waves 0-3,
    // which then wait at the barrier)
    if (PRIO && NW == 8 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
    uint32_t s_in_tile = 0, tile = FS ? blockIdx.x - gridDim.x : blockIdx.x, cv_stage = 0, cq_stage = 0, c_tileseq = 0;
    bool c_first = FS != 0;   // FS: the tile being computed is the workgroup's sample tile
#ifdef LYNSE_EXPERIMENTS
    const bool timing = (a.debug_flags & 64) && a.dbg;
#else
    constexpr bool timing = false;  // (phase timing lives in the EXPERIMENTS build: its flag and counters cost SGPRs in the hot loop)
#endif
    [[maybe_unused]] unsigned long long t_wait = 0, t_bar = 0, t_comp = 0, t_iq = 0, t_iv = 0, tp = timing ? __builtin_amdgcn_s_memtime() : 0;  // (EXPERIMENTS build only)
    half8 af[2][TR], bf[2][TQ];          // (DEFER: buffer 1 carries the last k-step's fragments across the loop edge)
    const char *stv = smem, *stq = smem;
    bool pend_done = false;              // DEFER: the deferred MFMA group completes a tile
    bool tile_done = false;
    for (uint32_t g = 0; g < G + (DEFER ? 1u : 0u); ++g) {
        if (!DEFER || g < G) {
        if constexpr (DEFER) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the deferred k-step's fragments are in registers: this wave is done with the old slab
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_OPS) : "memory");
        if (timing) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_wait += t - tp; tp = t; }
        __builtin_amdgcn_s_barrier();
        if (timing) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_bar += t - tp; tp = t; }

        stv = smem + cv_stage * V_BYTES;
        stq = smem + Q_RING + cq_stage * Q_BYTES;
        cv_stage = cv_stage + 1 == NSV ? 0 : cv_stage + 1;
        cq_stage = cq_stage + 1 == NSQ ? 0 : cq_stage + 1;
        }
        // Software pipeline over the four 16-wide k-steps: the fragments of step kk+1 are read from LDS
        // while the MFMAs of step kk run (sched_barriers pin the order — left alone, the scheduler sinks
        // every ds_read next to its first use and each MFMA group then waits out an LDS round trip).
        auto load_frags = [&](int kk, int buf) {
            const int ls = ((kk * 2 + hi) ^ swz) * 16;
#pragma unroll
            for (int j = 0; j < TQ; ++j) bf[buf][j] = *reinterpret_cast<const half8*>(stq + b_base + j * 32 * LINE + ls);
#pragma unroll
            for (int i = 0; i < TR; ++i) af[buf][i] = *reinterpret_cast<const half8*>(stv + a_base + i * 32 * LINE + ls);
        };
        [[maybe_unused]] auto mfma_group = [&](int buf, int i) {
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                typedef int i32x16 __attribute__((ext_vector_type(16)));
                acc[i][j] = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(
                    __builtin_bit_cast(i32x4, af[buf][i]), __builtin_bit_cast(i32x4, bf[buf][j]), __builtin_bit_cast(i32x16, acc[i][j]), 0, 0, 0));
            }
        };
        [[maybe_unused]] auto slot_pieces = [&](int slot) {   // the refill pieces scheduled behind MFMA group `slot` of NSLOT
#pragma unroll
            for (int p = 0; p < OPS; ++p)
                if (p % NSLOT == slot) issue_piece(p);
        };
        // (AG: MFMAs on fixed AGPR tuples need compile-time block indices)
        [[maybe_unused]] auto group_at = [&](auto bufc, auto ic) {
            constexpr int buf = decltype(bufc)::value, i = decltype(ic)::value;
            if constexpr (AG) ly_static_for<TQ>([&](auto jc) { ly_mfma_i8_agpr<i * TQ + decltype(jc)::value>(af[buf][i], bf[buf][decltype(jc)::value]); });
            else mfma_group(buf, i);
        };
        if constexpr (DEFER) {
            // slots of a slab step: 0 .. TR-1 the deferred group (last k-step of the previous slab), then k-steps 0 .. 2 of this slab
            const bool td_prev = pend_done;
            if (g < G && !td_prev) load_frags(0, 0);   // (not across a tile epilogue: its registers are spoken for)
            __builtin_amdgcn_sched_barrier(0);
            ly_static_for<TR>([&](auto ic) {
                if (g > 0) group_at(std::integral_constant<int, 1>{}, ic);
                if (g < G) slot_pieces(decltype(ic)::value);
            });
            __builtin_amdgcn_sched_barrier(0);
            tile_done = td_prev;
        } else if constexpr (AG) {
            load_frags(0, 0);
            ly_static_for<HK / 16>([&](auto kkc) {
                constexpr int kk = decltype(kkc)::value, cur = kk & 1;
                if constexpr (kk + 1 < HK / 16) load_frags(kk + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                ly_static_for<TR>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    ly_static_for<TQ>([&](auto jc) { ly_mfma_i8_agpr<i * TQ + decltype(jc)::value>(af[cur][i], bf[cur][decltype(jc)::value]); });
                    ly_static_for<OPS>([&](auto pc) {   // refill pieces behind this MFMA group (slot kk*TR+i of NSLOT)
                        if constexpr (decltype(pc)::value % NSLOT == kk * TR + i) issue_piece(decltype(pc)::value);
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
        if (!(DBG & 2) || g == 0) load_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < HK / 16; ++kk) {
            const int cur = (DBG & 2) ? 0 : (kk & 1);
            if (kk + 1 < HK / 16 && !(DBG & 2)) load_frags(kk + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TR; ++i) {
                if (DBG & 1) {
                    asm volatile("" ::"v"(af[cur][i]));
#pragma unroll
                    for (int j = 0; j < TQ; ++j) asm volatile("" ::"v"(bf[cur][j]));
                } else {
#pragma unroll
                    for (int j = 0; j < TQ; ++j) {
                        if constexpr (F4) {
                            acc[i][j] = ly_mfma_fp4(af[cur][i], bf[cur][j], acc[i][j], f4_unit);
#if defined(__HIP_DEVICE_COMPILE__)
                            // pins the MFMA in front of the next sched_barrier: the scaled MFMA is in none of the instruction
                            // classes sched_barrier(0) fences, and all 32 of a slab step sank below the last barrier of the
                            // body — behind all 24 fragment reads, whose registers then went to scratch (ISA inspected)
                            asm volatile("" : "+v"(acc[i][j]));
#endif
                        } else if constexpr (I8OPS) {
                            typedef int i32x4 __attribute__((ext_vector_type(4)));
                            typedef int i32x16 __attribute__((ext_vector_type(16)));
                            acc[i][j] = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(
                                __builtin_bit_cast(i32x4, af[cur][i]), __builtin_bit_cast(i32x4, bf[cur][j]),
                                __builtin_bit_cast(i32x16, acc[i][j]), 0, 0, 0));
                        } else {
                            if constexpr (ZEROC) {
                                // the first k-step of a tile starts from the constant 0 (src C = 0): the DENSE epilogue leaves the 16 x TR x TQ
                                // accumulators as they are instead of clearing them with one v_mov each
                                const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                                if (kk == 0 && s_in_tile == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], bf[cur][j], z, 0, 0, 0);
                                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
                            } else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
                        }
                    }
                }
                {   // refill pieces scheduled behind this MFMA group (slot kk*TR+i of NSLOT)
                    const int slot = kk * TR + i;
#pragma unroll
                    for (int p = 0; p < OPS; ++p)
                        if (PLACE == 0 ? (p % NSLOT == slot)                                   // spread: one piece behind every MFMA group
                                       : PLACE == 1 ? (slot == 0)                              // all pieces behind the first MFMA group
                                                    : (slot == ((wave < NW / 2) ? 0 : NSLOT / 2))) {  // half the waves early, half mid-phase
                            if (timing) {  // (experiments) issue time of this DMA instruction: query-image vs row pieces
                                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                                issue_piece(p);
                                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                                if (p < QPW) t_iq += t1 - t0; else t_iv += t1 - t0;
                            } else {
                                issue_piece(p);
                            }
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if constexpr (!DEFER) {
            q_advance();
            v_advance();
            tile_done = ++s_in_tile == a.nslab;
        }
        if constexpr ((DBG & 16) != 0) {  // (experiments) no epilogue at all: keep the accumulators alive, restart the tile
          if (tile_done) {
#pragma unroll
            for (int i = 0; i < TR; ++i)
#pragma unroll
                for (int j = 0; j < TQ; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        asm volatile("" ::"v"(acc[i][j][r]));   // (a 16-wide vector operand is not a valid "v" operand in the host pass)
                        acc[i][j][r] = 0.0f;
                    }
                }
            s_in_tile = 0;
            tile += gridDim.x;
          }
        } else if (tile_done) {
            if constexpr (I8L) {   // integer dot products -> floats: the float L2 epilogue takes over
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int j = 0; j < TQ; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = (float)__float_as_int(acc[i][j][r]);
            }
            const EpiArgsPtr ea = epi_args();
            uint32_t rbase = ea->row0 + tile * tstride;
            uint32_t row_end = ea->row1;
            if (FS && c_first) { rbase = blockIdx.x * ea->fs_stride; row_end = ea->fs_rows; }
            IvfTile td{};
            if (TILED) {
                td = ea->tiles[tile];
                rbase = td.row0;
                row_end = td.row0 + td.nrows;
            } else if (ea->skip_stride && rbase % ea->skip_stride == 0 && rbase / ea->skip_stride < ea->skip_tiles) {
                row_end = rbase;  // a sample tile: its rows were emitted by the sample stage
            }
            const float* nrm = reinterpret_cast<const float*>(smem + NORM_RING + (c_tileseq % NORM_SLOTS) * 1024);
            ++c_tileseq;
            uint32_t mw_t[TR];  // subset filter: the tile's mask words, loaded once per tile (not once per query column: each load
#pragma unroll          // would pay the full memory latency behind the candidate stores)
            for (int i = 0; i < TR; ++i) mw_t[i] = (FILT && !TILED && ea->mask) ? ea->mask[(rbase + wr * (TR * 32) + i * 32) >> 5] : 0xffffffffu;
            const float e_vmax2 = (!I8 && METRIC == M_L2) ? ea->vmax2 : 0.0f;
            // EMIT >= 0: the emission mode is a compile-time constant (0 threshold stages, 1 emit-all, 2 lane-max sample) — the
            // 256 x 256 tilings compile ONE epilogue per kernel: with all three in one body the row-index terms of the
            // emit-all / lane-max branches are hoisted across the unrolled column loop and spill (600 B / lane).
            const int e_emit_all = TILED ? 0 : (FS ? (c_first ? 2 : 0) : (EMIT >= 0 ? EMIT : ea->emit_all));
            auto score = [&](int i, int j, int r, uint32_t m, bool rok) -> float {
                if constexpr (I8C) {
                    return c_extra[j] + c_qinv[j] * (float)acc_int(acc[i][j][r]);  // B_q + s_q * dot (separate mul / add)
                } else if constexpr (I8) {
                    // dot of the u8 codes = i8 dot + 128 (sum q' + sum r') + 16384 D; squared L2 = sum q'^2 + sum r'^2 - 2 dot
                    const float accv = acc[i][j][r];
                    const int dotp = __float_as_int(accv);
                    const int rowsum = NORMS_LDS ? __float_as_int(nrm[m - rbase]) : (rok ? __float_as_int(ea->vn2[m]) : 0);
                    const int qconst = __float_as_int(c_extra[j]);
                    const uint32_t u = METRIC == M_IP ? (uint32_t)(dotp + qconst + 128 * rowsum) : (uint32_t)(qconst + rowsum - 2 * dotp);
                    return (float)u;  // `dist_fn(..) as f32` (flat_mmap.rs:5956)
                }
                float sc = acc[i][j][r] * c_qinv[j];
                if (METRIC != M_IP) {
                    float nv;
                    if (NORMS_LDS) nv = nrm[m - rbase];
                    else nv = rok ? (METRIC == M_L2 ? ea->vn2[m] : ea->vrinv[m]) : 0.0f;
                    if (METRIC == M_L2) sc = nv - 2.0f * sc + c_extra[j];
                    else sc = 1.0f - sc * nv * c_extra[j];
                }
                return sc;
            };
            // DENSE float epilogue: the column loop below only prepares the per-query constants; the accumulators of ALL query
            // columns are scored afterwards, one norm load per group of four rows shared by the columns and issued one group ahead
            constexpr bool DENSEF = DENSE && !I8 && !TILED && !FILT;
            uint32_t n_j[TQ];
            if constexpr (AG) asm volatile("s_nop 15\n\ts_nop 15");   // (the last MFMAs' results before v_accvgpr_read: the hazard recognizer does not see inline assembly)
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                if constexpr (AG) {   // this query column's TR x 16 accumulators out of the AGPRs (cleared as they are read)
                    auto take = [&](auto jc) {
                        constexpr int jj = decltype(jc)::value;
                        ly_static_for<TR>([&](auto ic) {
                            ly_static_for<16>([&](auto rc) {
                                acc[decltype(ic)::value][jj][decltype(rc)::value] = ly_agpr_take<(decltype(ic)::value * TQ + jj) * 16 + decltype(rc)::value>();
                            });
                        });
                    };
                    switch (j) {
                    case 0: take(std::integral_constant<int, 0>{}); break;
                    case 1: take(std::integral_constant<int, (TQ > 1 ? 1 : 0)>{}); break;
                    case 2: take(std::integral_constant<int, (TQ > 2 ? 2 : 0)>{}); break;
                    default: take(std::integral_constant<int, (TQ > 3 ? 3 : 0)>{}); break;
                    }
                }
                uint32_t rb = __builtin_amdgcn_readfirstlane(rbase);  // (uniform) opaque per column block: keeps the row-index terms of the TR x 16 rows from being hoisted out of
                asm volatile("" : "+s"(rb));  // the unrolled column loop (all live at once: hundreds of bytes of scratch per lane)
                uint32_t n = qchunk + wq * (TQ * 32) + j * 32 + l32;  // this lane's query of column block j (TILED: through the group's pair list)
                if (TILED) {
                    c_ok[j] = n < td.nq;
                    n = c_ok[j] ? ea->pair_q[td.pair0 + n] : 0u;
                } else {
                    c_ok[j] = n < ea->nq;
                }
                if (QC_REG) {
                    const int src = ((j & 1) * 32 + l32) * 4;  // lane holding query j*32+l32 of the wave's column
                    c_qinv[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(qc_inv[j / 2])));
                    c_extra[j] = (METRIC != M_IP || I8) ? __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(qc_extra[j / 2]))) : 0.0f;
                    c_thr[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(qc_thr[j / 2])));
                } else {
                    c_qinv[j] = c_ok[j] ? ea->qinv[n] : 0.0f;
                    c_extra[j] = 0.0f;
                    if (METRIC == M_L2 || I8) c_extra[j] = c_ok[j] ? ea->qn2[n] : 0.0f;  // SQ8: the per-query integer constant, as bits; I8C: B_q
                    if (METRIC == M_COS && !I8) c_extra[j] = c_ok[j] ? ea->qrinv[n] : 0.0f;
                    c_thr[j] = c_ok[j] ? ea->thr[n] : 0.0f;
                }
#ifdef LYNSE_EXPERIMENTS
                if (ea->debug_flags & 2) c_thr[j] = (I8C && QC_REG) ? __int_as_float(0x7fffffff) : (ASC ? -LY_INF : LY_INF);
#endif
                set_pre(j, c_thr[j], e_vmax2);
                n_j[j] = n;
                if (DENSEF && !e_emit_all) continue;   // scored after this loop (all columns per norm load)
                if ((WR >= 4 || EMIT == 2) && (FS || EMIT != 0) && !FILT && e_emit_all == 2 && !TILED) {  // (runtime-EMIT bodies of the <4,2,2,4> tiling and the subset-filter variants leave it out: it would spill there)
                    // threshold-only sample stage: each lane keeps the best LM of its TR*16 rows for this query column
                    // (4 WR keys per tile and query) and writes only those.  k_select turns the k-th best of them into a
                    // valid threshold and keeps no candidate: the sample tiles are scanned again by the ordinary stages.
                    // (Emitting all 256 x 256 scores of a tile cost as much as computing them.)  The host uses this mode only when the best
                    // tile alone supplies k keys (k <= 4 WR), so even a shard sorted by score gets a threshold from its best tile.
                    constexpr int LM = 2;  // 4 WR keys per tile and query: 16 for the <.,4,.,.> tilings, 8 for <4,2,2,4>
                    float bs[LM];
                    uint32_t bm[LM];
#pragma unroll
                    for (int t = 0; t < LM; ++t) { bs[t] = ASC ? LY_INF : -LY_INF; bm[t] = 0xffffffffu; }
#pragma unroll
                    for (int i = 0; i < TR; ++i) {
                        const uint32_t mw = mw_t[i];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t bit = (r & 3) + 8 * (r >> 2) + 4 * hi;
                            uint32_t m = rb + wr * (TR * 32) + i * 32 + bit;
                            float sc = score(i, j, r, m, m < row_end);
                            const bool ok = m < row_end && ((mw >> bit) & 1u);
                            if (!ok) { sc = ASC ? LY_INF : -LY_INF; m = 0xffffffffu; }
#pragma unroll
                            for (int t = 0; t < LM; ++t) {  // insertion into the sorted best-LM list
                                const bool better = bm[t] == 0xffffffffu ? m != 0xffffffffu : (m != 0xffffffffu && (ASC ? sc < bs[t] : sc > bs[t]));
                                const float ts = bs[t];
                                const uint32_t tm = bm[t];
                                bs[t] = better ? sc : ts;
                                bm[t] = better ? m : tm;
                                sc = better ? ts : sc;
                                m = better ? tm : m;
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < LM; ++t) {
                        const uint64_t key = bm[t] == 0xffffffffu ? KEY_SENTINEL : make_key(bs[t], (FILT && ea->row_ids) ? ea->row_ids[bm[t]] : bm[t], ASC);
                        const uint32_t slot = ((FS ? blockIdx.x : tile) * (2 * WR) + 2 * wr + hi) * LM + t;
                        if (c_ok[j] && slot < ea->cap) ea->cand[(size_t)n * ea->cap + slot] = key;
                    }
                } else if (!FS && e_emit_all && !TILED) {
#pragma unroll
                    for (int i = 0; i < TR; ++i) {
                        const uint32_t mw = mw_t[i];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t bit = (r & 3) + 8 * (r >> 2) + 4 * hi;
                            const uint32_t m = rb + wr * (TR * 32) + i * 32 + bit;
                            const bool rok = m < row_end;
                            const float sc = score(i, j, r, m, rok);
                            const uint32_t slot = tile * BR + (m - rb);  // dense over the (possibly strided) tiles
                            if (c_ok[j] && rok && slot < ea->cap)
                                ea->cand[(size_t)n * ea->cap + slot] =
                                    ((mw >> bit) & 1u) ? make_key(sc, (FILT && !TILED && ea->row_ids) ? ea->row_ids[m] : m, ASC) : KEY_SENTINEL;
                        }
                    }
                } else if (DENSE && I8C && QC_REG && !TILED) {
                    // ---- DENSE threshold stage of the certified int8 pass: while the threshold is loose (the first stage behind
                    // the sample: the int8 margin keeps ~5x the rows an exact threshold would) most 32-query x 64-row blocks
                    // hold a survivor and the two-level filter pays level 2 on top of level 1 almost always.  One pass instead:
                    // one integer compare per accumulator against T and a wave-level branch; only elements some lane passes
                    // build the key.  Lane-private segments as below.
                    const uint32_t e_seg = ea->seg;
                    uint32_t cnt = (segpk >> (8 * j)) & 0xffu;
                    uint64_t* segdst = ea->candB + ((size_t)n * ea->nseg + ((blockIdx.x * WR + wr) * 2 + hi)) * e_seg;
                    const int T = __float_as_int(c_thr[j]);
#pragma unroll
                    for (int i = 0; i < TR; ++i) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            // one wave-level branch per FOUR accumulators (their maximum against T): a taken branch per element
                            // cost more than the compare it guards
                            const int v0 = acc_int(acc[i][j][4 * g4]), v1 = acc_int(acc[i][j][4 * g4 + 1]);
                            const int v2 = acc_int(acc[i][j][4 * g4 + 2]), v3 = acc_int(acc[i][j][4 * g4 + 3]);
                            const int m01 = v0 > v1 ? v0 : v1, m23 = v2 > v3 ? v2 : v3;
                            if (__builtin_expect((m01 > m23 ? m01 : m23) >= T, 0)) {   // (unlikely: the skip falls through)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int r = 4 * g4 + e;
                                    const int v = acc_int(acc[i][j][r]);
                                    if (v >= T) {
                                        const uint32_t bit = (r & 3) + 8 * (r >> 2) + 4 * hi;
                                        const uint32_t m = rb + wr * (TR * 32) + i * 32 + bit;
                                        if (m < row_end && (!FILT || ((mw_t[i] >> bit) & 1u))) {   // (FILT: rows outside the subset never become candidates)
                                            const uint64_t key = make_key(c_extra[j] + c_qinv[j] * (float)v, m, ASC);
                                            if (cnt < e_seg) {
                                                segdst[cnt] = key;
                                                ++cnt;
                                            } else {
                                                const uint32_t slot = atomicAdd(&ea->count[n], 1u);
                                                if (slot < ea->cap) ea->cand[(size_t)n * ea->cap + slot] = key;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                    segpk = (segpk & ~(0xffu << (8 * j))) | (cnt << (8 * j));
                } else {
                    // ---- level 1
                    float best = -LY_INF;
                    if constexpr (I8C) {  // integer accumulators against the integer threshold
                        int bi = -2147483647 - 1;
#pragma unroll
                        for (int i = 0; i < TR; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int v = acc_int(acc[i][j][r]);
                                bi = bi > v ? bi : v;
                            }
                        // the exact score expression on the column maximum (monotone: s_q >= 0) — or, with the integer image of
                        // the threshold, the column maximum itself
                        if constexpr (QC_REG) best = (bi >= __float_as_int(c_thr[j])) ? LY_INF : -LY_INF;
                        else best = c_extra[j] + c_qinv[j] * (float)bi;
                    }
                    if constexpr (!I8) {
#pragma unroll
                        for (int i = 0; i < TR; ++i) {
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                float nv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                                if constexpr (METRIC != M_IP) {
                                    const uint32_t off = wr * (TR * 32) + i * 32 + 8 * g4 + 4 * hi;  // rows off .. off+3 <-> r = 4 g4 .. 4 g4 + 3
                                    if (NORMS_LDS) {
                                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(nrm + off);
                                        nv[0] = t4[0]; nv[1] = t4[1]; nv[2] = t4[2]; nv[3] = t4[3];
                                    } else {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) {
                                            const uint32_t m = rbase + off + e;
                                            nv[e] = m < row_end ? (METRIC == M_L2 ? ea->vn2[m] : ea->vrinv[m]) : 0.0f;
                                        }
                                    }
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float pv = acc[i][j][4 * g4 + e];
                                    if (METRIC == M_L2) pv = __fmaf_rn(pv, 2.0f * c_qinv[j], -nv[e]);
                                    if (METRIC == M_COS) pv = pv * nv[e];
                                    best = fmaxf(best, pv);
                                }
                            }
                        }
                    }
                    if constexpr (!I8 && METRIC == M_IP) best = best * c_qinv[j];  // the exact expression on the column maximum (qinv > 0)
                    const bool hit = c_ok[j] && ((I8C && QC_REG) ? best > 0.0f : ((I8C || (!I8 && METRIC == M_IP)) ? best >= c_thr[j] : (I8 || best >= c_pre[j])));
                    if (__ballot(hit) != 0ull) {
                    // ---- level 2: the exact expression against the exact threshold; pass masks of the block's TR x 16 rows
                    const float e_thr = c_thr[j];
                    uint32_t mk[TR], tot = 0;
#pragma unroll
                    for (int i = 0; i < TR; ++i) {
                        uint32_t msk = 0;
                        const uint32_t mw = mw_t[i];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint32_t bit = (r & 3) + 8 * (r >> 2) + 4 * hi;
                            const uint32_t m = rb + wr * (TR * 32) + i * 32 + bit;
                            const bool rok = m < row_end && ((mw >> bit) & 1u);
                            bool pass;
                            if constexpr (I8C && QC_REG) {
                                pass = acc_int(acc[i][j][r]) >= __float_as_int(e_thr);  // integer image of the threshold
                            } else {
                                const float sc = score(i, j, r, m, rok);
                                pass = ASC ? (sc <= e_thr) : (sc >= e_thr);
                            }
                            if (FILT && TILED && ea->mask && c_ok[j] && rok && pass)  // IVF subset filter: mask by slab position, looked up
                                pass = (ea->mask[m >> 5] >> (m & 31)) & 1u;     // only for rows that beat the threshold
                            if (c_ok[j] && rok && pass) msk |= 1u << r;
                        }
                        mk[i] = msk;
                        tot += (uint32_t)__popc(msk);
                    }
                    // slots: the private segment of this (workgroup, row-wave) while it has room — the two half-waves share a
                    // query column, lanes 0..31 take the first slots — else ONE reservation in the shared region
                    uint64_t* dst = ea->cand + (size_t)n * ea->cap;
                    uint32_t slot = 0, limit = 0;
                    bool segmented = false;
                    const uint32_t e_seg = TILED ? 0u : ea->seg;
                    if (!TILED && e_seg) {
                        const uint32_t other = (uint32_t)__shfl_xor((int)tot, 32, 64);
                        const uint32_t c = (segpk >> (8 * j)) & 0xffu;
                        const uint32_t both = tot + other;
                        if (c + both <= e_seg) {
                            segmented = true;
                            dst = ea->candB + ((size_t)n * ea->nseg + (blockIdx.x * WR + wr)) * e_seg;
                            slot = c + (hi ? other : 0u);
                            limit = e_seg;
                            segpk += both << (8 * j);
                        }
                    }
                    if (!segmented && tot) {
                        slot = atomicAdd(&ea->count[n], tot);
                        limit = ea->cap;
                    }
                    if (tot) {
#pragma unroll
                        for (int i = 0; i < TR; ++i) {
                            const uint32_t msk = mk[i];
                            if (msk) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    if ((msk >> r) & 1u) {
                                        const uint32_t m = rbase + wr * (TR * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                                        if (slot < limit)
                                            dst[slot] = make_key(score(i, j, r, m, true), (FILT && !TILED && ea->row_ids) ? ea->row_ids[m] : m, ASC);
                                        ++slot;
                                    }
                                }
                            }
                        }
                    }
                    }
                }
#pragma unroll
                for (int i = 0; i < TR; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            }
            if constexpr (DENSEF) {
                if (!e_emit_all) {
                    // ---- DENSE threshold stage (many survivors per block: large k over few rows seen so far, e.g. k = 100 on 1M
                    // rows).  The two-level filter degenerates there — nearly every 32-query x TR*32-row block holds a survivor,
                    // so level 2 (exact expression + pass masks for all TR x 16 values) runs on top of level 1 almost always.
                    // One pass instead: per group of FOUR accumulators the maximum of the level-1 values against the LOOSENED
                    // threshold and ONE wave-level branch (a taken branch per element cost several times the fma + compare it
                    // guarded; NaN values drop out of the maximum as they fail the compare); only groups some lane passes run the
                    // exact expression, the exact threshold and the store.  The row norms of a group (one ds_read_b128 from the
                    // norm ring) serve every query column and are fetched one group ahead — loaded where they were used, each of
                    // the 64 groups of a tile waited out an LDS round trip (with the branches: 60 % of the C3 scan).
                    // Slots: every lane owns one private segment of `seg` slots (segments are per (workgroup, row-wave, wave half):
                    // no counts to agree on between the two half-waves of a query column); a full segment falls back to the shared
                    // region (returning atomic).
                    uint32_t rb = __builtin_amdgcn_readfirstlane(rbase);
                    asm volatile("" : "+s"(rb));
                    const uint32_t e_seg = ea->seg;
                    uint32_t cnt[TQ];
                    uint64_t* segdst[TQ];
                    float pre_e[TQ];
#pragma unroll
                    for (int j = 0; j < TQ; ++j) {
                        cnt[j] = (segpk >> (8 * j)) & 0xffu;
                        segdst[j] = ea->candB + ((size_t)n_j[j] * ea->nseg + ((blockIdx.x * WR + wr) * 2 + hi)) * e_seg;
                        pre_e[j] = c_ok[j] ? (METRIC == M_IP ? c_thr[j] : c_pre[j]) : LY_INF;  // (IP: the level-1 value is the score itself)
                    }
                    auto load_nv = [&](int grp) -> f32x4 {  // norms of rows off .. off+3 of group grp = (i, g4)
                        f32x4 t4 = {0.0f, 0.0f, 0.0f, 0.0f};
                        if constexpr (METRIC != M_IP) {
                            const uint32_t off = wr * (TR * 32) + (grp >> 2) * 32 + 8 * (grp & 3) + 4 * hi;
                            if (NORMS_LDS) {
                                t4 = *reinterpret_cast<const f32x4*>(nrm + off);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const uint32_t m = rb + off + e;
                                    t4[e] = m < row_end ? (METRIC == M_L2 ? ea->vn2[m] : ea->vrinv[m]) : 0.0f;
                                }
                            }
                        }
                        return t4;
                    };
                    f32x4 nxt = load_nv(0);
#pragma unroll
                    for (int grp = 0; grp < TR * 4; ++grp) {
                        const int i = grp >> 2, g4 = grp & 3;
                        const f32x4 nv = nxt;
                        if (grp + 1 < TR * 4) nxt = load_nv(grp + 1);
                        const uint32_t off = wr * (TR * 32) + i * 32 + 8 * g4 + 4 * hi;  // rows off .. off+3 <-> r = 4 g4 .. 4 g4 + 3
#pragma unroll
                        for (int j = 0; j < TQ; ++j) {
                            float pvs[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float pv = acc[i][j][4 * g4 + e];
                                if (METRIC == M_L2) pv = __fmaf_rn(pv, 2.0f * c_qinv[j], -nv[e]);
                                if (METRIC == M_COS) pv = pv * nv[e];
                                if (METRIC == M_IP) pv = pv * c_qinv[j];
                                pvs[e] = pv;
                            }
                            if (__builtin_expect(fmaxf(fmaxf(fmaxf(pvs[0], pvs[1]), pvs[2]), pvs[3]) >= pre_e[j], 0)) {   // (unlikely: the skip falls through)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (pvs[e] >= pre_e[j]) {
                                        const uint32_t m = rb + off + e;
                                        float sc = acc[i][j][4 * g4 + e] * c_qinv[j];
                                        if (METRIC == M_L2) sc = nv[e] - 2.0f * sc + c_extra[j];
                                        if (METRIC == M_COS) sc = 1.0f - sc * nv[e] * c_extra[j];
                                        const bool pass = m < row_end && (ASC ? (sc <= c_thr[j]) : (sc >= c_thr[j]));
                                        if (pass) {
                                            const uint64_t key = make_key(sc, m, ASC);
                                            if (cnt[j] < e_seg) {
                                                segdst[j][cnt[j]] = key;
                                                ++cnt[j];
                                            } else {
                                                const uint32_t slot = atomicAdd(&ea->count[n_j[j]], 1u);
                                                if (slot < ea->cap) ea->cand[(size_t)n_j[j] * ea->cap + slot] = key;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < TQ; ++j) segpk = (segpk & ~(0xffu << (8 * j))) | (cnt[j] << (8 * j));
                    if constexpr (!ZEROC) {
#pragma unroll
                    for (int i = 0; i < TR; ++i)
#pragma unroll
                        for (int j = 0; j < TQ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
                    }
                }
            }
            if constexpr (FS != 0) {
                if (c_first) {
                    // ---- the grid agrees on the first thresholds: every workgroup has published the lane-max keys of its sample
                    // tile; workgroup q derives the threshold of query q (one wave); everybody picks the thresholds up
                    c_first = false;
                    uint32_t* gs = ea->gsync;
                    // (debug_flags & 128: s_memtime stamps of this phase, [block][8] u64 behind the first 64 words of gsync)
                    unsigned long long* stamp = (ea->debug_flags & 128) ? reinterpret_cast<unsigned long long*>(gs + 64) + (size_t)blockIdx.x * 8 : nullptr;
                    if (stamp && tid == 0) { stamp[0] = t_kernel0; stamp[1] = __builtin_amdgcn_s_memtime(); }
                    volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(const_cast<char*>(stv));  // ring stage just consumed: free
                    bool ok = fs_grid_sync(gs, gs + 2, gridDim.x, flag, tid);
                    if (stamp && tid == 0) stamp[2] = __builtin_amdgcn_s_memtime();
                    if (ok) {
                        for (uint32_t q = blockIdx.x; q < ea->nq; q += gridDim.x)   // (uniform per workgroup)
                            fused_sample_threshold<NW * 64>(ea->cand + (size_t)q * ea->cap, gridDim.x * (2 * WR * 2), ea->k, ea->marg2[q], METRIC, ea->ip_form,
                                                            ea->Qf + (size_t)q * ea->D, ea->V, ea->ld, ea->D, ea->thr_out + q, const_cast<char*>(stv),
                                                            const_cast<char*>(stq), tid);
                        if (stamp && tid == 0) stamp[3] = __builtin_amdgcn_s_memtime();
                        ok = fs_grid_sync(gs + 1, gs + 2, gridDim.x, flag, tid);
                        if (stamp && tid == 0) stamp[4] = __builtin_amdgcn_s_memtime();
                    }
                    if (!ok) return;   // aborted (fs_grid_sync): nothing of this launch is used; every DMA has landed
                    load_qc_thr(ea->thr, ea->nq);
                    if (stamp && tid == 0) stamp[5] = __builtin_amdgcn_s_memtime();
                }
            }
            if constexpr (!DEFER) s_in_tile = 0;
            tile += gridDim.x;
        }
        if constexpr (DEFER) {
            if (g == G) break;
            if (tile_done) load_frags(0, 0);
            ly_static_for<HK / 16 - 1>([&](auto kkc) {
                constexpr int kk = decltype(kkc)::value, cur = kk & 1;
                load_frags(kk + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                ly_static_for<TR>([&](auto ic) {
                    group_at(std::integral_constant<int, cur>{}, ic);
                    slot_pieces(TR + kk * TR + decltype(ic)::value);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            q_advance();
            v_advance();
            pend_done = ++s_in_tile == a.nslab;
            if (pend_done) s_in_tile = 0;
        }
        if (timing) {
            asm volatile("" ::"v"(acc[0][0][0]));
            const unsigned long long t = __builtin_amdgcn_s_memtime(); t_comp += t - tp; tp = t;
        }
    }
#ifdef LYNSE_EXPERIMENTS
    if (timing && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * NW + wave) * 4;
        o[0] = t_wait; o[1] = t_bar; o[2] = (t_iq << 32) | (t_iv & 0xffffffffull); o[3] = t_comp;
    }
#endif
    if (!TILED) {
        const EpiArgsPtr ea = epi_args();
        if (FS && (ea->debug_flags & 128) && tid == 0) reinterpret_cast<unsigned long long*>(ea->gsync + 64)[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime();
        if (ea->seg && (hi == 0 || DENSE)) {  // DENSE: one segment per wave half (the lane's own), else one per row-wave
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                const uint32_t n = qchunk + wq * (TQ * 32) + j * 32 + l32;
                const uint32_t sgm = DENSE ? (blockIdx.x * WR + wr) * 2 + hi : blockIdx.x * WR + wr;
                if (n < ea->nq) ea->segcnt[(size_t)n * ea->nseg + sgm] = (uint8_t)((segpk >> (8 * j)) & 0xffu);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// SQ8 (FLAT-*-SQ8, flat_mmap.rs:5676-5750): per-dimension min / max over all rows, codes
// clamp(round((v - min) * scale), 0, 255) stored as code - 128 (signed byte for the i8 MFMA), per-row sums of the
// signed codes and of their squares, and the same for the queries.
// ------------------------------------------------------------------------------------------------
template <typename T>   // float rows, or the f16 bits of an F16 shard (decoded exactly)
__global__ void __launch_bounds__(256) k_sq8_minmax(const T* __restrict__ V, uint32_t ld, uint32_t D, uint64_t n,
                                                    uint32_t* __restrict__ omin, uint32_t* __restrict__ omax,
                                                    const float* __restrict__ row_scale = nullptr) {
    // row_scale: the coded value of element (r, d) is V[r][d] * row_scale[r] (one f32 multiply) — the unit-norm rows of the
    // cosine form of the certified int8 pass
    // thread = one dimension (coalesced across the row), block = a strip of rows; `<` / `>` updates like the reference
    // (NaN never replaces), merged with atomics on the order-preserving image
    const uint64_t rows_per_block = (n + gridDim.y - 1) / gridDim.y;
    const uint64_t r0 = (uint64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const uint32_t d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    float mn = LY_INF, mx = -LY_INF;
    for (uint64_t r = r0; r < r1; ++r) {
        const float v = row_scale ? __fmul_rn((float)V[r * ld + d], row_scale[r]) : (float)V[r * ld + d];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    atomicMin(&omin[d], f32_to_ord(mn));
    atomicMax(&omax[d], f32_to_ord(mx));
}

__global__ void __launch_bounds__(256) k_sq8_scales(const uint32_t* __restrict__ omin, const uint32_t* __restrict__ omax, uint32_t D,
                                                    float* __restrict__ mins, float* __restrict__ scales) {
    const uint32_t d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const float mn = ord_to_f32(omin[d]), mx = ord_to_f32(omax[d]);
    const float range = __fsub_rn(mx, mn);
    mins[d] = mn;
    scales[d] = range > 1e-30f ? __fdiv_rn(255.0f, range) : 0.0f;
}

// did merging the new rows' min / max into the per-dimension table move any entry?  (ordered-int images; flag |= 1)
__global__ void __launch_bounds__(256) k_mm_changed(const uint32_t* __restrict__ now, const uint32_t* __restrict__ before, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && now[i] != before[i]) atomicOr(flag, 1u);
}

// min / max of one column held as a vector (the squared row norms: column D of the augmented rows), same update rule
__global__ void __launch_bounds__(256) k_vec_minmax(const float* __restrict__ x, uint64_t n, uint32_t* __restrict__ omin, uint32_t* __restrict__ omax,
                                                    uint32_t ncopies) {
    float mn = LY_INF, mx = -LY_INF;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0)
        for (uint32_t j = 0; j < ncopies; ++j) { atomicMin(omin + j, f32_to_ord(mn)); atomicMax(omax + j, f32_to_ord(mx)); }
}

__device__ __forceinline__ int sq8_code(float v, float mn, float sc) {
    float q = roundf(__fmul_rn(__fsub_rn(v, mn), sc));  // f32::round: half away from zero
    if (!(q == q)) return 0;
    q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
    return (int)q;
}

// one wave per row (or query): codes - 128 into `out` (pitch ld8, pad columns 0), sum and sum of squares of the signed codes
template <typename T>
__global__ void __launch_bounds__(256) k_sq8_quantize(const T* __restrict__ V, uint32_t ld, uint32_t D, uint64_t n,
                                                      const float* __restrict__ mins, const float* __restrict__ scales,
                                                      int8_t* __restrict__ out, uint32_t ld8, int* __restrict__ sums,
                                                      int* __restrict__ sums2, uint32_t* __restrict__ stats,
                                                      const float* __restrict__ extra_col = nullptr, uint32_t n_extra = 0,
                                                      const float* __restrict__ row_scale = nullptr) {
    // extra_col != nullptr: columns D .. D + n_extra - 1 of the coded row all hold extra_col[row] (mins / scales have D + n_extra
    // entries) — the squared row norm of the L2 form of the certified int8 pass (k_i8c_prep_queries, aug); sums / sums2 may be
    // NULL then
    // stats[0] = max over rows of sum |code - 128| (the query-quantisation term of the certified int8 bound),
    // stats[1] = number of non-finite elements (the certified int8 coarse pass is only valid without them),
    // round 5 (the Cauchy-Schwarz forms of the two quantisation terms, k_i8c_prep_queries):
    // stats[2] = max over rows of sum (code - 128)^2 (an exact integer), stats[3] = float bits of an UPPER bound of max over rows of
    // sum eps_d^2, eps_d = (v_d - min_d) scale_d - code_d the residual of the row's own quantisation in real arithmetic: the f32
    // evaluation of (v - min) scale is off by <= 255 * 2^-22 < 6.1e-5 per element, so sum eps^2 <= sum e^2 + 2 * 6.1e-5 * sum |e| +
    // D * (6.1e-5)^2 for the computed residuals e; the f32 sums carry a relative 1e-5 on top.
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t a1max = 0, a2max = 0, nonfinite = 0;
    float e2max = 0.0f;
    for (uint64_t row = wave; row < n; row += nwaves) {
        int s1 = 0, s2 = 0, l1 = 0;
        float e2 = 0.0f, e1 = 0.0f;
        for (uint32_t d = lane; d < ld8; d += 64) {
            int c = 0;
            if (d < D || (extra_col && d < D + n_extra)) {
                const float v = d < D ? (row_scale ? __fmul_rn((float)V[row * ld + d], row_scale[row]) : (float)V[row * ld + d]) : extra_col[row];
                if (!(fabsf(v) < LY_INF)) nonfinite += 1;
                const int code = sq8_code(v, mins[d], scales[d]);
                c = code - 128;
                s1 += c;
                s2 += c * c;
                l1 += c < 0 ? -c : c;
                const float e = __fsub_rn(__fmul_rn(__fsub_rn(v, mins[d]), scales[d]), (float)code);   // (the value sq8_code rounded, minus the code)
                const float ea = fabsf(e);
                if (ea == ea) { e2 = __fmaf_rn(e, e, e2); e1 += ea; }
            }
            out[row * ld8 + d] = (int8_t)c;
        }
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); l1 += __shfl_xor(l1, o, 64);
            e2 += __shfl_xor(e2, o, 64); e1 += __shfl_xor(e1, o, 64);
        }
        if (lane == 0 && sums) { sums[row] = s1; sums2[row] = s2; }
        a1max = a1max > (uint32_t)l1 ? a1max : (uint32_t)l1;
        a2max = a2max > (uint32_t)s2 ? a2max : (uint32_t)s2;
        const float e2b = (e2 + 1.22e-4f * e1 + (float)(D + n_extra) * 3.8e-9f) * 1.00002f;
        e2max = e2b > e2max ? e2b : e2max;
    }
    for (int o = 32; o > 0; o >>= 1) nonfinite += __shfl_xor(nonfinite, o, 64);
    if (lane == 0 && stats) {
        atomicMax(&stats[0], a1max);
        if (nonfinite) atomicAdd(&stats[1], nonfinite);
        atomicMax(&stats[2], a2max);
        atomicMax(&stats[3], __float_as_uint(e2max));   // (non-negative floats order like their bit patterns)
    }
}

// query side of SQ8 pass 1: one block per query.  Image layout = k_scan_h16's ([slab of 128][q][8 slots ^ ((q>>1)&7)][16 B]),
// qconst = the per-query integer of the score (IP: 128 * sum q' + 16384 * D; L2: sum q'^2), thresholds open.
__global__ void __launch_bounds__(256) k_sq8_prep_queries(const float* __restrict__ Q, uint32_t D, uint32_t qpad, uint32_t nslab,
                                                          const float* __restrict__ mins, const float* __restrict__ scales,
                                                          int8_t* __restrict__ img, float* __restrict__ qconst, float* __restrict__ thr,
                                                          uint32_t* __restrict__ count, uint32_t* __restrict__ overflow, int ip) {
    __shared__ int red[2][4];
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int s1 = 0, s2 = 0;
    const uint32_t total = nslab * 128;
    for (uint32_t i = tid; i < total; i += 256) {
        int c = 0;
        if (i < D) {
            c = sq8_code(Q[(size_t)q * D + i], mins[i], scales[i]) - 128;
            s1 += c;
            s2 += c * c;
        }
        const uint32_t s = i / 128, k = i % 128, l = k >> 4, e = k & 15, p = l ^ ((q >> 1) & 7);
        img[(((size_t)s * qpad + q) * 8 + p) * 16 + e] = (int8_t)c;
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (tid == 0) {
        s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const int c = ip ? 128 * s1 + 16384 * (int)D : s2;
        qconst[q] = __builtin_bit_cast(float, c);
        thr[q] = ip ? -LY_INF : LY_INF;
        count[q] = 0u;
        overflow[q] = 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// k_i8c_prep_queries — query side of the CERTIFIED int8 coarse pass (FLAT-IP, k_scan_h16<.., I8Q = 2>).
//
// Rows are the SQ8 codes: v_d = min_d + (code_d + eps_d) / scale_d with |eps_d| <= 0.50004 (round + the two f32 roundings
// of (v - min) * scale; exact for a constant dimension, scale_d = 0, up to its 1e-30 range).  With w_d = q_d / scale_d
// (0 for constant dimensions), c'_d = code_d - 128 (what the shard stores) and the symmetric int8 image
// w_d = s_q (u_d + eta_d), u_d = rint(w_d / s_q) in [-127, 127], |eta_d| <= 0.5:
//     q . v = [sum q_d min_d + 128 sum w_d]  +  s_q sum u_d c'_d  +  s_q sum eta_d c'_d  +  sum w_d eps_d
//           =            B_q                 +  s_q * (int dot)   +  (|.| <= 0.5 s_q A1)  +  (|.| <= 0.50004 sum |w_d|)
// A1 = max over rows of sum |c'_d| (k_sq8_quantize).  The kernel evaluates B_q + s_q * dot in f32 (two roundings) and the
// reference's own f32 accumulation differs from the real dot product by <= gamma |q| |v|: both go into E as well.
// Everything per query is accumulated in f64 (products of f32 are exact there).  marg2 = 2 E like k_prep_queries.
// Image layout = k_sq8_prep_queries ([slab of 128][q][8 slots ^ ((q>>1)&7)][16 B]).
// ------------------------------------------------------------------------------------------------
struct I8cPrepArgs {
    const float* Q;
    uint32_t D, qpad, nslab;
    uint32_t nq;         // queries of the batch; blocks nq .. qpad - 1 of the grid write the ZERO image of a pad column (0 = the grid is the batch)
    const float *mins, *scales;
    uint32_t a1;         // max row L1 norm of the signed codes
    uint32_t a2sq;       // max row sum of squares of the signed codes (0: not collected — the L1 forms alone)
    float eps2;          // upper bound of the max row sum of squared quantisation residuals (0: not collected)
    float vmax;          // max row norm
    int8_t* img;
    float *sq, *bq, *marg2, *thr;  // s_q -> ScanArgs::qinv, B_q -> ScanArgs::qn2
    uint32_t *count, *overflow;
    uint32_t* gsync;     // hand-over words of the fused sample stage (ScanArgs::gsync): zeroed here, once per batch
    // aug = m > 0: squared L2 as an inner product of AUGMENTED vectors.  -|q - v|^2 = q'.v' - |q|^2 with q' = [2 q, -1/m x m]
    // and v' = [v, |v|^2 x m]: the rows are coded with their f32 squared norm in columns D .. D + m - 1 (m a power of two: the
    // m query entries -1/m sum to -1 exactly; ONE column would carry a weight w = -1/scale that dwarfs the data columns'
    // and with it the resolution s_q = max |w| / 127 of the whole query image — uniform 100-d data: margin 2.7 against
    // neighbour distances of ~10; spread over m columns the weight shrinks m-fold).  mins / scales: D + m entries, a1 over
    // D + m columns; the coarse score B' + s_q dot, B' = B(q', mins) - |q|^2, brackets the NEGATED reference-order distance —
    // "smaller distance" is "larger score", the IP scan / select run unchanged, the exact rescoring negates the true L2
    // (SelectArgs / FinalArgs::neg_metric1) and k_final negates it back.  E = the IP bound over (q', v') + the f32 error of
    // the stored norm and of the reference's own difference-form sum: 8 (D + 8) 2^-24 (|q|^2 + max |v|^2).
    int aug;
    // l2n = 1: squared L2 on the PLAIN codes (k_scan_h16<.., I8Q = 4>): the image and s_q are the IP ones; bq receives
    // |q|^2 - 2 B_q (the kernel's qn2), thr opens at +inf (ascending keys), E = 2 x the IP quantisation terms + the f32 roundings of
    // |v|^2 - 2 s_q dot + qn2, of the stored row norm and of the reference's own difference-form sum.
    int l2n;
    // cosine = 1: cosine distance as an inner product of UNIT vectors.  The rows are coded as fl(v_d * rinv[row]) (rinv = the
    // stored reciprocal f32 norm; a zero row codes as zeros and scores distance 1 like the reference's denom < 1e-30 rule), the
    // query enters as q / |q|; coarse score q^.v^ - 1 = -(coarse distance), exact scores are the NEGATED reference cosine
    // distances (neg_metric1).  E = the IP bound over the unit vectors + 9 (D + 8) 2^-24 for the f32 normalisation of the rows
    // and the reference's own dot / norm sums.
    int cosine;
    // Seeding of the self-tightening scan (ScanArgs::dyn_*; dyn_thr == nullptr: off): the integer dot products of this query's
    // image with seed_rows sample rows of the shard (spread over the tiles of every partition) give the first partition
    // maxima; dyn_marg = the margin 2E (+ the float evaluation error of B_q + s_q dot on both sides) in dot units, rounded up.
    const int8_t* codes;   // the rows the scan streams: [n_rows][ld8] signed SQ8 codes
    uint32_t ld8, n_rows, tile_rows, dyn_ks, seed_rows;
    int *dyn_thr, *dyn_slot, *dyn_marg;
};

__global__ void __launch_bounds__(256) k_i8c_prep_queries(I8cPrepArgs a) {
    __shared__ double red[5][4];
    __shared__ float s_sq;
    __shared__ int s_slot[32];
    extern __shared__ __attribute__((aligned(16))) int8_t s_u[];   // seeding only: the query's plain int8 image (nslab * 128 bytes)
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.nq && q >= a.nq) {   // (uniform) a pad column of the query tile: zero image (was a hipMemsetAsync launch in front of this kernel)
        for (uint32_t i = tid; i < a.nslab * 128; i += 256) a.img[((size_t)(i / 128) * a.qpad + q) * 128 + (i % 128)] = 0;
        return;
    }
    const float* qv = a.Q + (size_t)q * a.D;
    const uint32_t DA = a.D + (uint32_t)a.aug;     // dimensions of the coded vectors
    double qrn = 1.0;                               // cosine: 1 / |q| (0 for a zero query: every coarse score is -1 = distance 1)
    if (a.cosine) {
        __shared__ double rs[4];
        double t = 0.0;
        for (uint32_t i = tid; i < a.D; i += 256) { const double o = (double)qv[i]; t += o * o; }
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) rs[wave] = t;
        __syncthreads();
        t = (rs[0] + rs[1]) + (rs[2] + rs[3]);
        qrn = t > 0.0 ? 1.0 / sqrt(t) : 0.0;
        __syncthreads();
    }
    auto qel = [&](uint32_t i) -> double {          // element i of the (augmented / normalised) query
        if (a.cosine) return (double)qv[i] * qrn;
        if (!a.aug) return (double)qv[i];
        return i < a.D ? 2.0 * (double)qv[i] : -1.0 / (double)a.aug;
    };
    double wmax = 0.0, sw = 0.0, swabs = 0.0, sqm = 0.0, s2 = 0.0, sw2 = 0.0;
    for (uint32_t i = tid; i < DA; i += 256) {
        const double x = qel(i);
        const float sc = a.scales[i];
        const double w = sc > 0.0f ? x / (double)sc : 0.0;
        wmax = fmax(wmax, fabs(w));
        sw += w;
        swabs += fabs(w);
        sw2 += w * w;
        sqm += x * (double)a.mins[i];
        if (i < a.D) { const double o = (double)qv[i]; s2 += o * o; }   // |q|^2 of the ORIGINAL query
    }
    for (int o = 32; o > 0; o >>= 1) {
        wmax = fmax(wmax, __shfl_xor(wmax, o, 64));
        sw += __shfl_xor(sw, o, 64);
        swabs += __shfl_xor(swabs, o, 64);
        sqm += __shfl_xor(sqm, o, 64);
        s2 += __shfl_xor(s2, o, 64);
        sw2 += __shfl_xor(sw2, o, 64);
    }
    __shared__ double red2[2][4];
    if (lane == 0) { red[0][wave] = wmax; red[1][wave] = sw; red[2][wave] = swabs; red[3][wave] = sqm; red[4][wave] = s2; red2[0][wave] = sw2; }
    __syncthreads();
    if (tid == 0) {   // the query's scale first: the image (below) needs it, and the bound needs the image's residuals
        wmax = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
        float sq = 1.0f;
        if (wmax > 0.0 && wmax < 1.0e30) {
            sq = (float)(wmax / 127.0);
            if ((double)sq * 127.0 < wmax) sq = nextafterf(sq, LY_INF);  // |w_d| / s_q <= 127 for every d
            if (!(sq > 0.0f)) sq = 1.1754944e-38f;
        }
        s_sq = sq;
    }
    __syncthreads();
    // ---- the int8 image + sum eta_d^2 (eta_d = w_d / s_q - u_d: the residual of the query's own quantisation)
    const double inv = 1.0 / (double)s_sq;
    const uint32_t total = a.nslab * 128;
    double eta2 = 0.0;
    for (uint32_t i = tid; i < total; i += 256) {
        int u = 0;
        if (i < DA) {
            const float sc = a.scales[i];
            const double w = sc > 0.0f ? qel(i) / (double)sc : 0.0;
            const double t = w * inv;
            double r = rint(t);
            r = r < -127.0 ? -127.0 : (r > 127.0 ? 127.0 : r);
            u = (r == r) ? (int)r : 0;
            const double eta = t - (double)u;
            if (eta == eta) eta2 += eta * eta;
        }
        const uint32_t s = i / 128, k = i % 128, l = k >> 4, e = k & 15, p = l ^ ((q >> 1) & 7);
        a.img[(((size_t)s * a.qpad + q) * 8 + p) * 16 + e] = (int8_t)u;
        if (a.dyn_thr) s_u[i] = (int8_t)u;
    }
    for (int o = 32; o > 0; o >>= 1) eta2 += __shfl_xor(eta2, o, 64);
    if (lane == 0) red2[1][wave] = eta2;
    __syncthreads();
    if (tid == 0) {
        wmax = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
        sw = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        swabs = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        sqm = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
        s2 = (red[4][0] + red[4][1]) + (red[4][2] + red[4][3]);
        sw2 = (red2[0][0] + red2[0][1]) + (red2[0][2] + red2[0][3]);
        eta2 = (red2[1][0] + red2[1][1]) + (red2[1][2] + red2[1][3]);
        const float sq = s_sq;
        const double bq = sqm + 128.0 * sw - (a.aug ? s2 : 0.0) - (a.cosine ? 1.0 : 0.0);
        const float bqf = (float)bq;
        const double a1 = (double)a.a1;
        const double gam = 10.0 * (double)a.D * 5.9604645e-8;  // reference f32 accumulation order vs the real dot product
        const double ref_term = a.cosine ? 9.0 * ((double)a.D + 8.0) * 5.9604645e-8
                                : a.aug  ? 8.0 * ((double)a.D + 8.0) * 5.9604645e-8 * (s2 + (double)a.vmax * (double)a.vmax)
                                         : gam * sqrt(s2) * (double)a.vmax;
        // The two quantisation terms of q.v - coarse = s_q sum eta_d c'_d + sum w_d eps_d.  Hoelder (round 2): |.| <= 0.5 s_q A1 and
        // <= 0.50004 sum |w_d| — every residual at its worst, all of one sign.  Cauchy-Schwarz (round 5): |.| <= s_q ||eta|| ||c'|| and
        // <= ||w|| ||eps||, with ||eta|| of THIS query (computed above), ||w|| of this query, and the largest ||c'||, ||eps|| any row of
        // the shard has (k_sq8_quantize, stats[2..3]).  Both hold for every (row, query), so does their minimum; on data whose
        // residuals are not aligned (anything but the constructed worst case of tests/test_gpu_certificate.py) the second is ~1.5x
        // tighter: fewer keys emitted, kept and rescored at every stage.
        double tq = 0.5001 * (double)sq * a1, tr = 0.5001 * swabs;
        if (a.a2sq) { const double c2 = 1.0002 * (double)sq * sqrt(eta2) * sqrt((double)a.a2sq); tq = c2 < tq ? c2 : tq; }
        if (a.eps2 > 0.0f) { const double c2 = 1.0002 * sqrt(sw2) * sqrt((double)a.eps2); tr = c2 < tr ? c2 : tr; }
        double E = tr + tq + 2.5e-7 * (fabs(bq) + (a.cosine ? 1.0 : fabs(s2)) + 127.0 * (double)sq * a1) + ref_term;
        float bq_out = bqf;
        if (a.l2n) {
            const double vm2 = (double)a.vmax * (double)a.vmax;
            E = 2.0 * (tr + tq) + 5.0e-7 * (2.0 * fabs(bq) + s2 + vm2 + 2.0 * 127.0 * (double)sq * a1) +
                8.0 * ((double)a.D + 8.0) * 5.9604645e-8 * (s2 + vm2);
            bq_out = (float)(s2 - 2.0 * bq);
        }
        E *= 1.02;
        if (!(wmax < 1.0e30) || !(E == E) || E > 3.0e38) E = 3.0e38;
        a.sq[q] = sq;
        a.bq[q] = bq_out;
        a.marg2[q] = (float)(2.0 * E);
        a.thr[q] = a.l2n ? LY_INF : -LY_INF;
        a.count[q] = 0u;
        a.overflow[q] = 0u;
        if (q == 0 && a.gsync) { a.gsync[0] = 0u; a.gsync[1] = 0u; a.gsync[2] = 0u; }
        if (a.dyn_thr) {
            // rows that can still matter have coarse_f(dot) >= coarse_f(tau) - 2E; coarse_f = fl(B_q + fl(s_q fl(dot))) is within
            // delta of B_q + s_q dot, so s_q M > 2E + 2 delta makes every dot < tau - M fail that test
            const double delta = 2.4e-7 * (fabs(bq) + 127.0 * (double)sq * a1);
            double md = ceil((2.0 * E + 2.0 * delta) / (double)sq) + 1.0;
            if (!(md < 1073741824.0)) md = 1073741824.0;
            a.dyn_marg[q] = (int)md;
        }
    }
    __syncthreads();
    if (a.dyn_thr) {
        // ---- first partition maxima: sample i belongs to partition i % ks and sits in a tile of that residue class, the classes'
        // tiles visited with an even stride
        const uint32_t ks = a.dyn_ks, rt = a.tile_rows;
        const uint32_t ntiles = (a.n_rows + rt - 1) / rt;
        if (tid < 32) s_slot[tid] = -2147483647 - 1;
        __syncthreads();
        const uint32_t per = a.seed_rows / ks;                       // samples per partition
        for (uint32_t i = tid; i < per * ks; i += 256) {
            const uint32_t p = i % ks, m = i / ks;
            const uint32_t up = (ntiles > p) ? (ntiles - p + ks - 1) / ks : 0u;   // tiles of residue p
            if (!up) continue;
            const uint32_t u_idx = (uint32_t)(((uint64_t)m * up) / per);          // even spread over the class
            uint64_t row = (uint64_t)(p + ks * u_idx) * rt + (i * 7u) % rt;
            if (row >= a.n_rows) row = (uint64_t)(p + ks * u_idx) * rt;           // (a ragged last tile: its first row exists)
            const int8_t* src = a.codes + row * a.ld8;
            int dot = 0;
            const uint32_t nb = a.nslab * 128;
            for (uint32_t d = 0; d < nb && d < a.ld8; d += 16) {
                typedef int i32x4_t __attribute__((ext_vector_type(4)));
                const i32x4_t c = *reinterpret_cast<const i32x4_t*>(src + d);
                const i32x4_t w = *reinterpret_cast<const i32x4_t*>(s_u + d);
#if defined(__HIP_DEVICE_COMPILE__)
                dot = __builtin_amdgcn_sdot4(c[0], w[0], dot, false);
                dot = __builtin_amdgcn_sdot4(c[1], w[1], dot, false);
                dot = __builtin_amdgcn_sdot4(c[2], w[2], dot, false);
                dot = __builtin_amdgcn_sdot4(c[3], w[3], dot, false);
#else
                (void)c; (void)w;
#endif
            }
            atomicMax(&s_slot[p], dot);
        }
        __syncthreads();
        if (tid < 32) a.dyn_slot[(size_t)q * 32 + tid] = tid < (int)ks ? s_slot[tid] : 2147483647;
        if (tid == 0) {
            int mn = 2147483647;
            for (uint32_t j = 0; j < ks; ++j) mn = s_slot[j] < mn ? s_slot[j] : mn;
            a.dyn_thr[q] = mn;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Batched Hamming on the matrix pipe (packed_binary_search, flat_mmap.rs:1345-1409, for batches of queries).
// popcount(x ^ q) = (D - sum_d s_x[d] s_q[d]) / 2 with s = +1 for a set bit, -1 for a clear one (0 in the pad columns of
// both sides): a batch of queries against the rows is an int8 GEMM with EXACT integer results, and "smaller distance" is
// "larger dot product" — the certified-int8 IP scan (k_scan_h16<2,4,4,2,IP,I8Q=2>) runs it unchanged with s_q = 1, B_q = 0
// and a zero margin; k_select takes the strict cut of the binary metrics, k_final turns the dot product back into the
// distance (FinalArgs::ham_dim).  The lane-per-row popcount kernel (k_scan_binary_rows) needs 2 VALU operations per 32 bits
// and (row, query) pair and is VALU-bound from ~16 queries on (36.8k queries/s at 12.5M x 1024 bits x 256 queries).
//   k_bits_to_fp4      : packed words -> one FP4 nibble per bit (+1.0 / -1.0), pitch round_up(D, 256) / 2 bytes, pad columns 0;
//                        a resident copy like the f16 shadow / the SQ8 codes, built on the first batched Hamming search
//                        (round 3 first fed the int8 MFMA from +-1 BYTES: 8x the packed words and the int8 rate; the FP4
//                        form is 4x the words at twice the rate: v_mfma_scale_f32_32x32x64_f8f6f4, k_scan_h16<.., I8Q = 3>)
//   k_bpm_prep_queries : packed query words -> the +-1 query image in the scan kernel's slab layout, open thresholds
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bits_to_fp4(const uint64_t* __restrict__ P, uint32_t W, uint32_t D, uint64_t r0, uint64_t r1,
                                                     uint8_t* __restrict__ out, uint32_t ldb) {
    // one FP4 (E2M1) nibble per bit: 0x2 = +1.0 for a set bit, 0xA = -1.0 for a clear one, 0x0 = +0.0 in the pad columns;
    // column c sits in nibble c & 1 of byte c / 2 (pitch ldb bytes = round_up(D, 256) / 2: whole 128-B slabs)
    const uint32_t cpr = ldb / 16;   // 16-byte pieces (32 columns) per row
    const uint64_t total = (r1 - r0) * cpr;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = r0 + i / cpr;
        const uint32_t c0 = (uint32_t)(i % cpr) * 32;
        const uint32_t w = c0 >> 6;
        const uint32_t bits = w < W ? (uint32_t)(P[r * W + w] >> (c0 & 63)) : 0u;
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t col = c0 + j * 8 + b;
                const uint32_t nib = col < D ? (((bits >> (j * 8 + b)) & 1u) ? 0x2u : 0xAu) : 0u;
                word |= nib << (4 * b);
            }
            v[j] = word;
        }
        *reinterpret_cast<u32x4*>(out + r * ldb + (size_t)(i % cpr) * 16) = v;
    }
}

struct BpmPrepArgs {
    const uint64_t* QW;   // nq x W packed query words
    uint32_t W, D, qpad, nslab;   // nslab: 128-B slabs of 256 columns
    uint32_t nq;          // queries of the batch: blocks nq .. qpad - 1 write the zero image of a pad column
    uint8_t* img;
    float *sq, *bq, *marg2, *thr;
    uint32_t *count, *overflow;
};

__global__ void __launch_bounds__(256) k_bpm_prep_queries(BpmPrepArgs a) {
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t total = a.nslab * 128;   // bytes of this query's image
    if (a.nq && q >= a.nq) {   // (uniform) pad column of the query tile
        for (uint32_t i = tid; i < total; i += 256) a.img[((size_t)(i / 128) * a.qpad + q) * 128 + (i % 128)] = 0;
        return;
    }
    for (uint32_t i = tid; i < total; i += 256) {
        uint32_t byte = 0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t col = 2 * i + b;
            const uint32_t nib = col < a.D ? (((a.QW[(size_t)q * a.W + (col >> 6)] >> (col & 63)) & 1ull) ? 0x2u : 0xAu) : 0u;
            byte |= nib << (4 * b);
        }
        const uint32_t s = i / 128, k = i % 128, l = k >> 4, e = k & 15, p = l ^ ((q >> 1) & 7);
        a.img[(((size_t)s * a.qpad + q) * 8 + p) * 16 + e] = (uint8_t)byte;
    }
    if (tid == 0) {
        a.sq[q] = 1.0f;          // score = B_q + s_q * dot = the dot product itself (exact)
        a.bq[q] = 0.0f;
        a.marg2[q] = 0.0f;
        a.thr[q] = -LY_INF;
        a.count[q] = 0u;
        a.overflow[q] = 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan_binary: packed one-bit rows (ceil(D/64) u64 words per row).  Eight lanes own one row
// (16 B per lane per chunk -> a row's words are read as contiguous 128-B pieces), row words stay
// in registers across the query loop, packed queries + thresholds sit in LDS.  Distances follow
// packed_hamming / packed_jaccard / packed_dice (flat_mmap.rs:1298-1334) — integer popcounts,
// converted to f32 exactly as the reference does.  HBM-bound: algorithmic bytes = rows * W * 8.
// ------------------------------------------------------------------------------------------------
struct BinArgs {
    const uint64_t* P;  // rows x W
    uint32_t W;
    uint32_t row0, row1;
    const uint64_t* QW;  // nq x W
    uint32_t nq;
    const float* thr;
    uint64_t* cand;
    uint32_t* count;
    uint32_t cap;
    int emit_all;
    int strict_unused;
    const uint32_t* mask;  // filtered search: bit r = row r is in the subset (k_scan_binary_rows only)
};

constexpr int BIN_MAX_CHUNKS = 4;  // 4 chunks x 8 lanes x 2 words = 64 words = 4096 bits

template <int KIND>  // 0 hamming, 1 jaccard/tanimoto, 2 dice
__global__ void __launch_bounds__(256) k_scan_binary(BinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* qw = reinterpret_cast<uint64_t*>(smem);              // nq * W
    float* thr_l = reinterpret_cast<float*>(qw + (size_t)a.nq * a.W);  // nq
    const int tid = threadIdx.x, g = tid & 7;
    for (uint32_t i = tid; i < a.nq * a.W; i += 256) qw[i] = a.QW[i];
    for (uint32_t i = tid; i < a.nq; i += 256) thr_l[i] = a.thr[i];
    __syncthreads();
    const uint32_t nchunks = (a.W + 15) / 16;
    for (uint32_t rb = a.row0 + blockIdx.x * 32; rb < a.row1; rb += gridDim.x * 32) {
        const uint32_t row = rb + (tid >> 3);
        const bool valid = row < a.row1;
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
        for (uint32_t q = 0; q < a.nq; ++q) {
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
            c0 += __shfl_xor(c0, 1, 8);
            c0 += __shfl_xor(c0, 2, 8);
            c0 += __shfl_xor(c0, 4, 8);
            if (KIND != 0) {
                c1 += __shfl_xor(c1, 1, 8);
                c1 += __shfl_xor(c1, 2, 8);
                c1 += __shfl_xor(c1, 4, 8);
            }
            if (g == 0 && valid) {
                float dist;
                if (KIND == 0) dist = (float)c0;
                else if (KIND == 1) dist = c1 == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)c0, (float)c1));
                else dist = c1 == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)(2u * c0), (float)c1));
                if (a.emit_all || dist <= thr_l[q]) {
                    const uint32_t slot = a.emit_all ? (row - a.row0) : atomicAdd(&a.count[q], 1u);
                    if (slot < a.cap) a.cand[(size_t)q * a.cap + slot] = make_key(dist, row, true);
                }
            }
        }
    }
}

// k_scan_binary_wide: packed rows WIDER than 4096 bits (the register-resident kernels hold at most 64 words per row; the
// reference has no width limit: BinaryData rows are ceil(D/64) words, flat_mmap.rs:145-160).  Eight lanes own one row and
// walk its words with stride 8 for every query — row words come back through L1 / L2 once per query, query words and
// thresholds are read from global memory.  A correctness path for rare shapes, not a tuned one.
template <int KIND>  // 0 hamming, 1 jaccard/tanimoto, 2 dice
__global__ void __launch_bounds__(256) k_scan_binary_wide(BinArgs a) {
    const int tid = threadIdx.x, g = tid & 7;
    for (uint32_t rb = a.row0 + blockIdx.x * 32; rb < a.row1; rb += gridDim.x * 32) {
        const uint32_t row = rb + (tid >> 3);
        const bool valid = row < a.row1 && (!a.mask || ((a.mask[row >> 5] >> (row & 31)) & 1u));
        const uint64_t* rp = a.P + (size_t)(row < a.row1 ? row : a.row0) * a.W;
        uint32_t popr = 0;
        if (KIND == 2)
            for (uint32_t w = g; w < a.W; w += 8) popr += __popcll(rp[w]);
        for (uint32_t q = 0; q < a.nq; ++q) {
            const uint64_t* qp = a.QW + (size_t)q * a.W;
            uint32_t c0 = 0, c1 = 0;
            for (uint32_t w = g; w < a.W; w += 8) {
                const uint64_t x = qp[w], r = rp[w];
                if (KIND == 0) {
                    c0 += __popcll(x ^ r);
                } else if (KIND == 1) {
                    c0 += __popcll(x & r);
                    c1 += __popcll(x | r);
                } else {
                    c0 += __popcll(x & r);
                    c1 += __popcll(x);
                }
            }
            if (KIND == 2) c1 += popr;
            c0 += __shfl_xor(c0, 1, 8);
            c0 += __shfl_xor(c0, 2, 8);
            c0 += __shfl_xor(c0, 4, 8);
            if (KIND != 0) {
                c1 += __shfl_xor(c1, 1, 8);
                c1 += __shfl_xor(c1, 2, 8);
                c1 += __shfl_xor(c1, 4, 8);
            }
            if (g == 0 && row < a.row1) {
                float dist;
                if (KIND == 0) dist = (float)c0;
                else if (KIND == 1) dist = c1 == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)c0, (float)c1));
                else dist = c1 == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)(2u * c0), (float)c1));
                if (a.emit_all) {
                    const uint32_t slot = row - a.row0;
                    if (slot < a.cap) a.cand[(size_t)q * a.cap + slot] = valid ? make_key(dist, row, true) : KEY_SENTINEL;
                } else if (valid && dist <= a.thr[q]) {
                    const uint32_t slot = atomicAdd(&a.count[q], 1u);
                    if (slot < a.cap) a.cand[(size_t)q * a.cap + slot] = make_key(dist, row, true);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan_binary_rows: the BATCHED packed-binary scan (nq >= a few).  With many queries the 8-lanes-per-
// row kernel above is bound by its cross-lane reductions and LDS query reads, not by HBM; here ONE LANE
// OWNS ONE ROW (its words stay in VGPRs, <= 2*WCAP dwords) and the query words arrive as SCALAR loads
// (uniform address, constant address space -> s_load through the scalar cache), so a (row, query) pair
// costs exactly one v_xor/v_and + one v_bcnt per 32 bits and nothing else: the VALU floor of the
// problem (64 lane-ops per 1024-bit pair).  Jaccard / Dice use |x|q| = |x| + |q| - |x&q| (same
// integers as flat_mmap.rs:1298-1334) and only run the correctly rounded division when a cheap
// reciprocal estimate says the pair can pass the threshold.
// ------------------------------------------------------------------------------------------------
typedef const uint32_t __attribute__((address_space(4))) const_u32;
typedef const float __attribute__((address_space(4))) const_f32;

// subset row ids -> row bitmask (FlatMmap::search_filtered builds the same bitset, flat_mmap.rs:672-679);
// ids >= n are skipped like the reference skips them.  The mask must be zeroed first.
__global__ void __launch_bounds__(256) k_mask_build(const uint64_t* __restrict__ subset, uint64_t m, uint64_t n,
                                                    uint32_t* __restrict__ mask) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = subset[i];
        if (r < n) atomicOr(&mask[r >> 5], 1u << (r & 31));
    }
}

// IVF subset filter: the slab-position mask of a row mask (slab position p holds original row orig[p])
__global__ void __launch_bounds__(256) k_mask_permute(const uint32_t* __restrict__ rowmask, const uint32_t* __restrict__ orig,
                                                      uint64_t n, uint32_t* __restrict__ slabmask) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = orig[p];
        if ((rowmask[r >> 5] >> (r & 31)) & 1u) atomicOr(&slabmask[p >> 5], 1u << (p & 31));
    }
}

// BitSet words -> ascending row ids on the device (BitSet::to_vec): block b owns 256 words; pass 1 counts the
// set bits of each block, pass 2 (one block) turns the counts into offsets, pass 3 writes the ids.
__global__ void __launch_bounds__(256) k_bits_count(const uint64_t* __restrict__ words, uint64_t n_words, uint64_t n,
                                                    uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t red[4];
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t bits = w < n_words ? words[w] : 0ull;
    if (w * 64 + 64 > n) bits = w * 64 >= n ? 0ull : (bits & ((1ull << (n - w * 64)) - 1ull));
    uint32_t c = (uint32_t)__popcll(bits);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(1024) k_bits_offsets(uint32_t* __restrict__ block_counts, uint32_t nblocks) {
    // in-place exclusive scan by one block (nblocks <= a few thousand: 10M rows = 611 blocks)
    __shared__ uint32_t part[1024];
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per && b0 + i < nblocks; ++i) sum += block_counts[b0 + i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = 0; i < per && b0 + i < nblocks; ++i) {
        const uint32_t c = block_counts[b0 + i];
        block_counts[b0 + i] = run;
        run += c;
    }
}

__global__ void __launch_bounds__(256) k_bits_expand(const uint64_t* __restrict__ words, uint64_t n_words, uint64_t n,
                                                     const uint32_t* __restrict__ block_offsets, uint64_t* __restrict__ ids) {
    __shared__ uint32_t wave_base[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t bits = w < n_words ? words[w] : 0ull;
    if (w * 64 + 64 > n) bits = w * 64 >= n ? 0ull : (bits & ((1ull << (n - w * 64)) - 1ull));
    const uint32_t c = (uint32_t)__popcll(bits);
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_base[wave] = incl;
    __syncthreads();
    uint32_t base = block_offsets[blockIdx.x];
    for (int i = 0; i < wave; ++i) base += wave_base[i];
    uint32_t pos = base + incl - c;
    while (bits) {
        ids[pos++] = w * 64 + (uint64_t)__builtin_ctzll(bits);
        bits &= bits - 1;
    }
}

__global__ void __launch_bounds__(256) k_pad_words(const uint64_t* __restrict__ src, uint32_t W, uint64_t* __restrict__ dst,
                                                   uint32_t wcap, uint32_t nq) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * wcap) return;
    const uint32_t q = i / wcap, w = i % wcap;
    dst[i] = w < W ? src[(size_t)q * W + w] : 0ull;
}

constexpr int bin_rows_stride(int wcap) { return wcap * 8 + 16; }  // LDS row stride (bytes): +16 B keeps b128 row reads conflict-free

template <int KIND, int WCAP, bool ODD>
__global__ void __launch_bounds__(256) k_scan_binary_rows(BinArgs a) {
    // a.QW holds the queries PADDED to WCAP words each (zero words beyond a.W).
    // Each wave streams its 64 rows (one contiguous 64*W*8-byte span) with fully coalesced 16-B (8-B for odd W)
    // pieces, parks them in a wave-private padded LDS tile and every lane then reads ITS row back: the transpose
    // that turns coalesced HBM traffic into lane-per-row registers.  The pieces of the NEXT row block are
    // fetched into registers before the query loop of the current one (software pipeline).
    constexpr int RS = bin_rows_stride(WCAP);
    constexpr int NPC = WCAP >= 2 ? WCAP / 2 : 1;  // 16-B pieces per lane per block (W even)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* tile = smem + (size_t)wave * 64 * RS;
    const uint32_t W = a.W;
    const_u32* Q32 = (const_u32*)(uintptr_t)a.QW;
    const_f32* THR = (const_f32*)(uintptr_t)a.thr;
    constexpr bool odd = ODD;             // the host dispatches on W & 1
    constexpr bool PF = WCAP <= 32;       // software prefetch costs 2*WCAP staging VGPRs: not for the widest rows
    const bool fast = !ODD && W == (uint32_t)WCAP && WCAP >= 2;
    const uint32_t H = odd ? W : W / 2;       // pieces per row
    const uint32_t step_row = 64 / H, step_in = 64 % H;

    u32x4 nxt[ODD ? 1 : NPC];        // even W: 16-B pieces
    uint64_t nxt8[ODD ? WCAP : 1];   // odd W: 8-B pieces
    auto fetch = [&](uint32_t rb) {
        const uint32_t wrow0 = rb + wave * 64;
        const uint32_t nrows_w = wrow0 < a.row1 ? (a.row1 - wrow0 < 64u ? a.row1 - wrow0 : 64u) : 0u;
        const uint32_t npieces = nrows_w * H;
        const char* gbase = reinterpret_cast<const char*>(a.P + (size_t)wrow0 * W);
        if constexpr (!odd) {
#pragma unroll
            for (int it = 0; it < NPC; ++it) {
                const uint32_t p = lane + 64u * it;
                nxt[it] = u32x4{0u, 0u, 0u, 0u};
                if (p < npieces) nxt[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gbase) + p);
            }
        } else {
#pragma unroll
            for (int it = 0; it < WCAP; ++it) {
                const uint32_t p = lane + 64u * it;
                nxt8[it] = 0ull;
                if (p < npieces) nxt8[it] = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(gbase) + p);
            }
        }
    };
    auto park = [&]() {  // registers -> LDS tile, row-major with the padded stride
        if (fast) {
            if constexpr (!odd) {
                constexpr uint32_t HH = NPC;
#pragma unroll
                for (int it = 0; it < NPC; ++it) {
                    const uint32_t p = lane + 64u * it;
                    *reinterpret_cast<u32x4*>(tile + (p / HH) * RS + (p % HH) * 16) = nxt[it];
                }
            }
        } else {
            uint32_t prow = lane / H, pin = lane % H;
            if constexpr (!odd) {
#pragma unroll
                for (int it = 0; it < NPC; ++it) {
                    if (prow < 64u) *reinterpret_cast<u32x4*>(tile + prow * RS + pin * 16) = nxt[it];
                    prow += step_row; pin += step_in;
                    if (pin >= H) { pin -= H; prow += 1; }
                }
            } else {
#pragma unroll
                for (int it = 0; it < WCAP; ++it) {
                    if (prow < 64u) *reinterpret_cast<uint64_t*>(tile + prow * RS + pin * 8) = nxt8[it];
                    prow += step_row; pin += step_in;
                    if (pin >= H) { pin -= H; prow += 1; }
                }
            }
        }
    };

    uint32_t rb = a.row0 + blockIdx.x * 256;
    if (PF && rb < a.row1) fetch(rb);
    for (; rb < a.row1; rb += gridDim.x * 256) {
        const uint32_t row = rb + wave * 64 + lane;
        const bool valid = row < a.row1;
        if (!PF) fetch(rb);
        park();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t x[2 * WCAP];
        if (WCAP >= 2) {
#pragma unroll
            for (int c = 0; c < WCAP / 2; ++c) {
                u32x4 v = *reinterpret_cast<const u32x4*>(tile + lane * RS + c * 16);
                if (!fast) {  // words beyond W were never written
                    if ((uint32_t)(2 * c) >= W) { v[0] = 0; v[1] = 0; }
                    if ((uint32_t)(2 * c + 1) >= W) { v[2] = 0; v[3] = 0; }
                }
                x[4 * c] = v[0]; x[4 * c + 1] = v[1]; x[4 * c + 2] = v[2]; x[4 * c + 3] = v[3];
            }
        } else {
            const uint64_t v = *reinterpret_cast<const uint64_t*>(tile + lane * RS);
            x[0] = (uint32_t)v; x[1] = (uint32_t)(v >> 32);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next row block
        const uint32_t rb_next = rb + gridDim.x * 256;
        if (PF && rb_next < a.row1) fetch(rb_next);
        const bool in_subset = !a.mask || (valid && ((a.mask[row >> 5] >> (row & 31)) & 1u));
        uint32_t px = 0;
        if (KIND != 0) {
#pragma unroll
            for (int i = 0; i < 2 * WCAP; ++i) px += __popc(x[i]);
        }
        for (uint32_t q = 0; q < a.nq; ++q) {
            const_u32* qp = Q32 + (size_t)q * (2 * WCAP);
            uint32_t c0 = 0, pq = 0;
#pragma unroll
            for (int i = 0; i < 2 * WCAP; ++i) {
                const uint32_t qv = qp[i];
                c0 += __popc(KIND == 0 ? (x[i] ^ qv) : (x[i] & qv));
                if (KIND != 0) pq += __popc(qv);
            }
            const float thr = THR[q];
            float dist = (float)c0;
            bool pass;
            if (KIND == 0) {
                pass = dist <= thr;
            } else {
                const uint32_t den = KIND == 1 ? px + pq - c0 : px + pq;
                const uint32_t num = KIND == 1 ? c0 : 2u * c0;
                const float est = 1.0f - (float)num * __builtin_amdgcn_rcpf((float)den);  // |error| < 1e-6
                pass = den == 0 || est <= thr + 2e-6f;
            }
            if (valid && (a.emit_all || (pass && in_subset))) {
                if (KIND != 0) {
                    const uint32_t den = KIND == 1 ? px + pq - c0 : px + pq;
                    const uint32_t num = KIND == 1 ? c0 : 2u * c0;
                    dist = den == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)num, (float)den));
                }
                if (a.emit_all || dist <= thr) {
                    const uint32_t slot = a.emit_all ? (row - a.row0) : atomicAdd(&a.count[q], 1u);
                    if (slot < a.cap) a.cand[(size_t)q * a.cap + slot] = in_subset ? make_key(dist, row, true) : KEY_SENTINEL;
                }
            }
        }
    }
}

// ---- cross-lane helpers (k_select's small-k path, k_small_search) ----
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Cross-lane helpers of the fused search that stay off the LDS crossbar (ds_bpermute round trips were ~1 us per output
// rank): a uniform source lane is a v_readlane, the neighbour lane is a DPP wave shift, a wave-wide minimum is four DPP
// row rotations plus one v_readlane per row of 16 lanes.
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src_uniform) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src_uniform);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src_uniform);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v) {  // lane i <- lane i-1 (lane 0 keeps its own value)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {  // the minimum over the 64 lanes, in every lane
    uint64_t w;
    w = dpp_u64<0x128>(v); v = w < v ? w : v;  // row_ror:8
    w = dpp_u64<0x124>(v); v = w < v ? w : v;  // row_ror:4
    w = dpp_u64<0x122>(v); v = w < v ? w : v;  // row_ror:2
    w = dpp_u64<0x121>(v); v = w < v ? w : v;  // row_ror:1 -> every lane holds its row's minimum
    const uint64_t r0 = readlane_u64(v, 0), r1 = readlane_u64(v, 16), r2 = readlane_u64(v, 32), r3 = readlane_u64(v, 48);
    const uint64_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}

__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v) {  // the OR over the 64 lanes, in every lane (DPP row rotations + one readlane per row)
    v |= dpp_u64<0x128>(v);
    v |= dpp_u64<0x124>(v);
    v |= dpp_u64<0x122>(v);
    v |= dpp_u64<0x121>(v);
    return (readlane_u64(v, 0) | readlane_u64(v, 16)) | (readlane_u64(v, 32) | readlane_u64(v, 48));
}

// ------------------------------------------------------------------------------------------------
// Block-wide bitonic sort of npow2 u64 keys in LDS (ascending).
// ------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* keys, uint32_t npow2, int tid) {
    for (uint32_t size = 2; size <= npow2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < (npow2 >> 1); i += NT) {
                const uint32_t lo = 2 * i - (i & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t x = keys[lo], y = keys[hi];
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t n) {
    uint32_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

// Rescore keys[0..n) exactly (8 lanes per candidate); overwrite their score words.
// scr / scr_bytes: free LDS of the caller (16-B aligned).  With enough of it the rows go THROUGH LDS: the workgroup loads a chunk
// of candidate rows with coalesced 16-B loads (one global round trip per chunk, the next chunk's loads in flight while this one is
// scored) and the reference-order FMA chains of exact_score run out of LDS with the query staged next to them — straight from
// global memory a 768-d row is three dependent round trips of 2 x 32 strided 4-B loads per lane (k_select's threshold rescoring
// of 2k = 20 rows: 9 us -> 5.6 us, twice per shard step).  Used where the rows fit ONE batch; larger pools keep the direct loads.  Row stride D + 8 floats: the eight
// lane groups of a wave read eight different bank octets.
template <int NT>
__device__ __forceinline__ void rescore_keys(uint64_t* keys, uint32_t n, int metric, int ip_form,
                                             const float* qv, const float* V, uint32_t ld, uint32_t D,
                                             bool asc, int tid, bool neg = false, char* scr = nullptr, uint32_t scr_bytes = 0) {
    {
        constexpr uint32_t MAXV = 8;                       // 16-B loads per thread and batch: 21 rows of 768 floats (42 of 768 halves) in ONE global round trip
        // F16 shards (LYNSE_IPFORM_F16SEQ: V addresses f16 bits, ld counts floats = two halves): the rows are staged as halves and
        // exact_score_f16seq reads them from LDS — straight from global memory its 64-element steps are one dependent round trip each
        // (12 per 768-d row: the selects of an F16 shard took 72 / 130 us against 23 / 35 on an f32 shard)
        const bool f16 = ip_form == LYNSE_IPFORM_F16SEQ;
        const uint32_t row_b = D * (f16 ? 2u : 4u);        // bytes of a row
        const uint32_t vpr = row_b / 16u, stride_b = row_b + 32u;   // 16-B pieces per row; LDS row stride (the eight lane groups of a wave on different banks)
        const bool shape = scr && n && (D & (f16 ? 7u : 3u)) == 0 && (ld & 3u) == 0 && (reinterpret_cast<uintptr_t>(V) & 15u) == 0 &&
                           vpr <= MAXV * (uint32_t)NT / 8u && scr_bytes >= D * 4u + 8u * stride_b;   // (at least 8 rows per batch and per chunk, or the direct loads win)
        if (shape) {   // (uniform over the workgroup)
            float* q_l = reinterpret_cast<float*>(scr);
            char* rows_l = scr + (size_t)D * 4u;
            uint32_t R = (scr_bytes - D * 4u) / stride_b;                    // rows per LDS chunk
            R = R < (uint32_t)(NT / 8) ? R : (uint32_t)(NT / 8);
            uint32_t F = MAXV * (uint32_t)NT / vpr;                          // rows per batch of loads (held in registers until their chunk's turn)
            F = F < (uint32_t)(NT / 8) ? F : (uint32_t)(NT / 8);
            const int g = tid & 7;
            const uint32_t grp = tid >> 3;
            const char* Vb = reinterpret_cast<const char*>(V);
            for (uint32_t i = tid; i < D; i += NT) q_l[i] = qv[i];
            for (uint32_t f0 = 0; f0 < n; f0 += F) {
                const uint32_t fn = n - f0 < F ? n - f0 : F;
                f32x4 pre[MAXV];
                uint32_t prow4[MAXV / 4];   // row of the batch an element belongs to, one byte each (0xff: none; rows of a batch < NT / 8 <= 255)
#pragma unroll
                for (uint32_t u = 0; u < MAXV / 4; ++u) prow4[u] = 0xffffffffu;
#pragma unroll
                for (uint32_t u = 0; u < MAXV; ++u) {
                    const uint32_t t = (uint32_t)tid + u * NT;
                    if (t < fn * vpr) {
                        const uint32_t r = t / vpr, c = t - r * vpr;
                        prow4[u / 4] = (prow4[u / 4] & ~(0xffu << (8 * (u % 4)))) | (r << (8 * (u % 4)));
                        pre[u] = *reinterpret_cast<const f32x4*>(Vb + (size_t)key_row(keys[f0 + r]) * ld * 4u + c * 16u);
                    }
                }
                for (uint32_t c0 = 0; c0 < fn; c0 += R) {
                    const uint32_t rn = fn - c0 < R ? fn - c0 : R;
                    __syncthreads();   // the previous chunk has been scored
#pragma unroll
                    for (uint32_t u = 0; u < MAXV; ++u) {
                        const uint32_t r = (prow4[u / 4] >> (8 * (u % 4))) & 0xffu;
                        if (r >= c0 && r < c0 + rn) {   // (0xff never passes: c0 + rn <= fn <= NT / 8)
                            const uint32_t t = (uint32_t)tid + u * NT;
                            *reinterpret_cast<f32x4*>(rows_l + (size_t)(r - c0) * stride_b + (t - r * vpr) * 16u) = pre[u];
                        }
                    }
                    __syncthreads();
                    if (grp < rn) {
                        const uint32_t row = key_row(keys[f0 + c0 + grp]);
                        float sc = exact_score<16>(metric, ip_form, q_l, reinterpret_cast<const float*>(rows_l + (size_t)grp * stride_b), D, g);   // (LDS reads: 16 steps in flight are plenty)
                        if (neg) sc = -sc;
                        if (g == 0) keys[f0 + c0 + grp] = make_key(sc, row, asc);
                    }
                }
            }
            __syncthreads();
            return;
        }
    }
    const int g = tid & 7;
    const uint32_t grp = tid >> 3;
    const uint32_t rounds = (n + NT / 8 - 1) / (NT / 8);
    // two rows per lane group and trip: the loads of the second row are in flight while the first is reduced (a trip waits out
    // one global round trip either way; 163 survivors at k = 100 were three dependent trips)
    for (uint32_t r = 0; r < rounds; r += 2) {
        const uint32_t i0 = r * (NT / 8) + grp, i1 = i0 + NT / 8;
        const bool ok0 = i0 < n, ok1 = i1 < n;
        // rows past the end re-score row 0 of the list (uniform control flow inside exact_score) and are dropped
        const uint32_t row0 = key_row(keys[ok0 ? i0 : 0u]), row1 = key_row(keys[ok1 ? i1 : 0u]);
        float s0 = 0.0f, s1 = 0.0f;
        if (n) {
            s0 = exact_score<32>(metric, ip_form, qv, V + (size_t)row0 * ld, D, g);
            // (uniform: the second row of the trip exists for SOME lane group — a pool of <= NT / 8 rows, the usual case, used to score
            // a dummy row here: three more dependent round trips)
            if ((r + 1) * (uint32_t)(NT / 8) < n) s1 = exact_score<32>(metric, ip_form, qv, V + (size_t)row1 * ld, D, g);
        }
        if (neg) { s0 = -s0; s1 = -s1; }
        if (ok0 && g == 0) keys[i0] = make_key(s0, row0, asc);
        if (ok1 && g == 0) keys[i1] = make_key(s1, row1, asc);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Gather kernels of the "few matches" strategy of FlatMmap::search_filtered (direct_access_topk,
// flat_mmap.rs:5223-5274 — O(matches), no full scan): the listed rows of the f16 shadow (and their
// norms) are copied into a compact temporary store that the ordinary pipeline scans; the scan emits
// ORIGINAL row ids (ScanArgs::row_ids), so select / exact rescoring / final order work on the f32
// source rows unchanged.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_rows16(const _Float16* __restrict__ src, uint32_t ld16,
                                                       const uint64_t* __restrict__ ids, uint64_t m,
                                                       _Float16* __restrict__ dst) {
    const uint32_t cpr = ld16 / 8;
    const uint64_t total = m * cpr;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j = i / cpr;
        const uint32_t c = (uint32_t)(i % cpr) * 8;
        *reinterpret_cast<u32x4*>(dst + j * ld16 + c) = *reinterpret_cast<const u32x4*>(src + ids[j] * ld16 + c);
    }
}

__global__ void __launch_bounds__(256) k_gather_norms(const float* __restrict__ vn2, const float* __restrict__ vrinv,
                                                      const uint64_t* __restrict__ ids, uint64_t m,
                                                      float* __restrict__ o_vn2, float* __restrict__ o_vrinv,
                                                      uint32_t* __restrict__ o_ids32) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = ids[i];
        o_vn2[i] = vn2[r];
        o_vrinv[i] = vrinv[r];
        o_ids32[i] = (uint32_t)r;
    }
}

// ------------------------------------------------------------------------------------------------
// k_select: one block per query after each scan stage.  Sorts the candidate keys, sets the new
// per-query threshold and prunes.
//   float metrics: thr = tau -/+ 2E (tau = k-th best coarse score so far).  Any row whose exact
//     score could still reach the final top-k has coarse score within 2E of tau, so it survives.
//     If more than keep_max survive (huge margins / massive ties) the survivors are rescored
//     exactly on the spot and cut to the exact top-k (always valid: a row outside the exact top-k of
//     the rows seen so far can never enter the final top-k).
//   exact (binary) metrics: keep the top-k, later rows must be STRICTLY better (they have larger
//     row ids, so a tie loses) — the reference's strict admission rule (flat_mmap.rs:2170-2176).
// ------------------------------------------------------------------------------------------------
struct SelectArgs {
    uint64_t* cand;
    uint32_t* count;
    uint32_t* overflow;
    float* thr;
    const float* marg2;
    uint32_t k, cap, keep_max;
    int metric, ip_form, exact;
    int emit_all_n;  // >=0: stage 0 wrote exactly this many keys per query
    int keep_ties;   // IVF: rows are not scanned in id order -> the cut must let ties of the k-th score through
    int drop_sentinels;  // filtered search: the emit-all stage wrote KEY_SENTINEL for rows outside the subset
    int threshold_only;  // lane-max sample stage: derive the threshold, keep NO candidate (the rows are scanned again)
    int tighten;         // float coarse passes: threshold from the exact rescoring of the >= k best coarse rows (tau_x -/+ E)
    int hand_exact;      // k_select_final only: when every survivor was among the rows rescored for the threshold, they go on with those exact scores
    // segmented emission of the scan stage that ran before this select (ScanArgs::candB / segcnt): nseg segments of
    // `seg` slots per query, segcnt[q][s] keys in segment s.  nseg == 0: none.
    const uint64_t* candB;
    const uint8_t* segcnt;
    uint32_t seg, nseg;
    const float* Qf;
    const float* V;
    uint32_t ld, D;
    const uint32_t* abort_word;  // fused sample stage (ScanArgs::gsync + 2): != 0 -> the scan launch gave up, nothing it wrote is valid
    unsigned long long* stamps;  // debugging: [query][8] s_memtime stamps of the phases (nullptr = off)
    // != 0: `metric` only gives the KEY ORDER (M_IP: best-first); exact scores are those of metric neg_metric1 - 1, NEGATED —
    // the certified int8 pass of an ascending metric (squared L2 as an augmented inner product, k_i8c_prep_queries aug = 1)
    int neg_metric1;
    uint32_t lds_bytes;   // dynamic LDS of the launch (keys: cap x 8 B, the rest is scratch of the exact rescoring); 0 = cap x 8
};

// k_select finds the k-th best key with an 8-pass MSB radix select over the keys in LDS (256-bin
// histograms, wave-aggregated LDS atomics, one-wave prefix scan) instead of sorting them: the survivors
// only have to be FOUND, k_final orders them once at the end.  (A full bitonic sort of 8192 keys cost
// 97 us per stage; the radix select is ~10x cheaper.)  The rare compaction path (more than keep_max
// survivors: rescore exactly, cut to the exact top-k) still sorts.
// select_body: the whole of k_select for query q with the keys staged in `keys` (LDS, cap slots).  Returns the number of
// survivors it left in cand[q][0 .. ) (count[q] holds the same); *rescored = those keys already carry EXACT scores (the
// compaction path).  k_select is this and nothing else; k_select_final goes on to the exact rescoring + final order in the
// same launch.
template <int NT>
__device__ __forceinline__ uint32_t select_body(const SelectArgs& a, uint64_t* keys, const uint32_t q, bool* rescored) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_keep, s_rank;
    __shared__ uint64_t s_prefix;
    *rescored = false;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool asc = metric_ascending(a.metric);
    auto stamp = [&](int i) { if (a.stamps && tid == 0) a.stamps[(size_t)q * 8 + i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    if (a.abort_word && *a.abort_word) {  // the scan gave up: no candidates, the overflow flag sends the batch down the plan ladder
        if (tid == 0) { a.overflow[q] = 1u; a.count[q] = 0u; }
        return 0u;
    }
    uint32_t n = a.emit_all_n >= 0 ? (uint32_t)a.emit_all_n : a.count[q];
    if (n > a.cap) {
        if (tid == 0) a.overflow[q] = 1u;
        n = a.cap;
    }
    uint64_t* gkeys = a.cand + (size_t)q * a.cap;
    bool compacted = false;
    if (a.drop_sentinels) {  // load + drop the slots of rows outside the subset
        if (tid == 0) s_keep = 0;
        __syncthreads();
        // (eight loads in flight per thread: with one load per trip of the compaction the 4096 sample keys were eight
        // dependent global round trips, a third of this kernel's time)
        for (uint32_t i0 = 0; i0 < n; i0 += NT * 8) {
            uint64_t kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + u * NT + tid;
                kv[u] = i < n ? gkeys[i] : KEY_SENTINEL;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint64_t key = kv[u];
                const bool real = key != KEY_SENTINEL;
                const uint64_t m = __ballot(real);
                if (m) {
                    uint32_t base = 0;
                    const int leader = __builtin_ctzll(m);
                    if (lane == leader) base = atomicAdd(&s_keep, (uint32_t)__popcll(m));
                    base = __shfl(base, leader, 64);
                    if (real) keys[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
                }
            }
        }
        __syncthreads();
        n = s_keep;
        compacted = true;
        __syncthreads();
    }
    if (a.nseg) {  // gather the private segments of the scan stage behind the keys of the shared region
        __shared__ uint32_t s_wsum[NT / 64];
        if (!compacted)
            for (uint32_t i = tid; i < n; i += NT) keys[i] = gkeys[i];
        const uint32_t per = (a.nseg + NT - 1) / NT;
        const uint32_t s0 = tid * per < a.nseg ? tid * per : a.nseg;
        const uint32_t s1 = s0 + per < a.nseg ? s0 + per : a.nseg;
        const uint8_t* sc = a.segcnt + (size_t)q * a.nseg;
        uint32_t mine = 0;
        // up to 8 segments per thread with their counts and FIRST keys loaded together (most segments hold 0 or 1 key): two
        // global round trips for the whole gather instead of one per segment and key
        uint32_t cnt8[8];
        uint64_t first8[8];
        const bool fast = per <= 8;
        if (fast) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cnt8[u] = (s0 + u < s1) ? (uint32_t)sc[s0 + u] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) first8[u] = cnt8[u] ? a.candB[((size_t)q * a.nseg + s0 + u) * a.seg] : 0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u) mine += cnt8[u];
        } else {
            for (uint32_t sgm = s0; sgm < s1; ++sgm) mine += sc[sgm];
        }
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) {
            const uint32_t ws = s_wsum[w];
            if (w < (tid >> 6)) wbase += ws;
            total += ws;
        }
        uint32_t off = n + wbase + incl - mine;
        if (fast) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t c = cnt8[u];
                if (c) {
                    if (off < a.cap) keys[off] = first8[u];
                    ++off;
                    const uint64_t* src = a.candB + ((size_t)q * a.nseg + s0 + u) * a.seg;
                    for (uint32_t i = 1; i < c; ++i, ++off)
                        if (off < a.cap) keys[off] = src[i];
                }
            }
        } else {
            for (uint32_t sgm = s0; sgm < s1; ++sgm) {
                const uint32_t c = sc[sgm];
                const uint64_t* src = a.candB + ((size_t)q * a.nseg + sgm) * a.seg;
                for (uint32_t i = 0; i < c; ++i, ++off)
                    if (off < a.cap) keys[off] = src[i];
            }
        }
        if (n + total > a.cap) {
            if (tid == 0) a.overflow[q] = 1u;
            n = a.cap;
        } else {
            n += total;
        }
        compacted = true;  // the keys live in LDS only: every exit below writes them back
        __syncthreads();
    }
    stamp(1);
    if (n < a.k || a.k == 0) {  // fewer than k candidates so far: keep all, the threshold stays open
        if (compacted && !a.threshold_only)
            for (uint32_t i = tid; i < n; i += NT) gkeys[i] = keys[i];
        // the threshold is left as it is: open (set by the prep kernel) until a select has seen k keys — or already valid
        // for the whole shard when the threshold-only sample stage set it
        if (tid == 0) a.count[q] = a.threshold_only ? 0u : n;
        return a.threshold_only ? 0u : n;
    }
    if (!compacted)
        for (uint32_t i = tid; i < n; i += NT) keys[i] = gkeys[i];
    if (tid == 0) { s_keep = 0; s_prefix = 0; s_rank = a.k - 1; }
    __syncthreads();

    // (A small-k variant — per-wave insertion lists in registers, one key per lane, merged by wave 0 — was measured SLOWER than
    // the radix select on MI355X: ~150 cycles per serial insertion (readlane / DPP / scalar ping-pong), 4096 keys, k = 10:
    // 36k ticks against 28k; removed.)
    uint64_t kth = 0;
    float tau = 0.0f;
    float cut_exact = asc ? LY_INF : -LY_INF;   // tau_x -/+ E; unused when not tightened
    bool tightened = false;
    __shared__ uint64_t xs[256];   // the keys <= kth, rescored exactly (threshold tightening)
    __shared__ uint32_t s_x;
    uint32_t tight_m = 0;          // their number when ALL of them fitted xs
    {
        // ---- k-th smallest key (keys are unique: the low word is the row).  Float metrics only need the k-th SCORE (every
        // tie of it survives the margin cut anyway): four passes over the score word instead of eight over the whole key.
        const int last_shift = a.exact ? 0 : 32;
        if (a.k == 1u) {
            // the nearest row only (every k-means assignment pass, top-1 searches): the smallest key is a block-wide minimum, not
            // four / eight histogram passes (s_memtime stamps, 4096 keys: 23k of this kernel's 56k cycles were the radix select)
            __shared__ uint64_t s_wmin[NT / 64];
            uint64_t mn = ~0ull;
            for (uint32_t i = tid; i < n; i += NT) {
                const uint64_t kv = keys[i];
                mn = kv < mn ? kv : mn;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t other = __shfl_xor(mn, o, 64);
                mn = other < mn ? other : mn;
            }
            if (lane == 0) s_wmin[tid >> 6] = mn;
            __syncthreads();
            if (tid == 0) {
                uint64_t m = s_wmin[0];
#pragma unroll
                for (int w = 1; w < NT / 64; ++w) m = s_wmin[w] < m ? s_wmin[w] : m;
                s_prefix = a.exact ? m : (m & ~0xffffffffull);   // (what the passes down to last_shift leave)
            }
            __syncthreads();
        } else {
        // Round 5: digits every key shares are not worth a pass.  The candidates of a select are the rows near one threshold: their
        // score images agree in sign, exponent and the leading mantissa bits (s_memtime stamps: ~2.5k cycles per pass, four per select
        // and three selects per batch).  One OR-reduction of key ^ keys[0] finds the highest differing bit; the passes above it would
        // put every key into the same bin and leave the rank unchanged — their digits go into the prefix directly.
        int first_shift = 56;
        {
            __shared__ uint64_t s_wor[NT / 64];
            const uint64_t k0 = keys[0];
            uint64_t df = 0;
            for (uint32_t i = tid; i < n; i += NT) df |= keys[i] ^ k0;
            df = wave_or_u64(df);   // (DPP: six ds_bpermute round trips per 32-bit half cost as much as the pass they save)
            if (lane == 0) s_wor[tid >> 6] = df;
            __syncthreads();
            df = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) df |= s_wor[w];
            while (first_shift >= last_shift && (df >> first_shift) == 0ull) first_shift -= 8;   // (uniform)
            if (first_shift < 56 && tid == 0) s_prefix = k0 & (~0ull << (first_shift + 8));
            __syncthreads();
        }
        for (int shift = first_shift; shift >= last_shift; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const uint64_t prefix = s_prefix;
            const uint64_t himask = shift == 56 ? 0ull : (~0ull << (shift + 8));
            for (uint32_t i0 = 0; i0 < n; i0 += NT) {
                const uint32_t i = i0 + tid;
                const bool on = i < n && (keys[i] & himask) == prefix;
                const uint32_t bin = on ? (uint32_t)(keys[i] >> shift) & 255u : 0u;
                // the high bytes of the scores are (nearly) the same for every key: one atomic per wave for the
                // first lane's bin, plain atomics for the lanes that differ
                const uint64_t act = __ballot(on);
                if (act) {
                    const int leader = __builtin_ctzll(act);
                    const uint32_t first = __shfl(bin, leader, 64);
                    const uint64_t same = __ballot(on && bin == first);
                    if (lane == leader) atomicAdd(&hist[first], (uint32_t)__popcll(same));
                    if (on && bin != first) atomicAdd(&hist[bin], 1u);
                }
            }
            __syncthreads();
            if (tid < 64) {  // one wave: 4 bins per lane, inclusive scan over lanes, the bucket holding the rank
                const uint32_t rank = s_rank;
                const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const uint32_t sum = h0 + h1 + h2 + h3;
                uint32_t incl = sum;
    #pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += up;
                }
                const uint32_t excl = incl - sum;
                if (excl <= rank && rank < incl) {
                    uint32_t r = rank - excl, b = 4 * lane;
                    if (r >= h0) { r -= h0; ++b; if (r >= h1) { r -= h1; ++b; if (r >= h2) { r -= h2; ++b; } } }
                    s_rank = r;
                    s_prefix = prefix | ((uint64_t)b << shift);
                }
            }
            __syncthreads();
        }
        }
        stamp(2);
        kth = a.exact ? s_prefix : (s_prefix | 0xffffffffull);  // float: every row of the k-th score counts as <= kth
        tau = key_score(kth, asc);

        // ---- exact-rescored threshold (float coarse passes).  The cut tau -/+ 2E brackets the exact k-th best through the
        // COARSE k-th score tau: one E for the rows that realise tau, one for the row being tested.  Rescoring the >= k best
        // coarse rows exactly (a few dozen f32 rows per query) gives tau_x = the k-th best EXACT score among them — a valid lower
        // bound of the shard's exact k-th best whatever rows were picked — and every row that can still matter has a coarse
        // score within ONE E of it.  tau_x -/+ E is never looser than tau -/+ 2E (the k best coarse rows have exact scores
        // >= tau - E) and typically one E tighter: with the wide int8 margins that is ~5x fewer keys emitted, kept and
        // rescored in every later stage.
        if (!a.exact && a.tighten && a.k <= 128u) {
            const uint32_t mx = a.k * 2u < 256u ? a.k * 2u : 256u;
            if (tid == 0) s_x = 0;
            __syncthreads();
            for (uint32_t i0 = 0; i0 < n; i0 += NT) {   // (one LDS atomic per wave and trip, not one per key: they serialise)
                const uint32_t i = i0 + tid;
                const uint64_t key = i < n ? keys[i] : KEY_SENTINEL;
                const bool in = i < n && key <= kth;
                const uint64_t m = __ballot(in);
                if (m) {
                    uint32_t base = 0;
                    const int leader = __builtin_ctzll(m);
                    if (lane == leader) base = atomicAdd(&s_x, (uint32_t)__popcll(m));
                    base = __shfl(base, leader, 64);
                    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (in && slot < mx) xs[slot] = key;
                }
            }
            __syncthreads();
            const uint32_t m = s_x < mx ? s_x : mx;   // >= k: at least k keys are <= the k-th smallest
            if (m >= a.k) {
                // (scratch: the LDS behind the n keys of this query)
                const uint32_t used = (n * 8u + 15u) & ~15u, lds_all = a.lds_bytes ? a.lds_bytes : a.cap * 8u;
                rescore_keys<NT>(xs, m, a.neg_metric1 ? a.neg_metric1 - 1 : a.metric, a.ip_form, a.Qf + (size_t)q * a.D, a.V, a.ld, a.D, asc, tid, a.neg_metric1 != 0,
                                 reinterpret_cast<char*>(keys) + used, lds_all > used ? lds_all - used : 0u);
                // the k-th best of the m <= 256 rescored keys by RANK (keys are unique: the row is part of the key): one thread per key
                // counts the smaller ones — m broadcast LDS reads instead of the 36 barrier-separated steps of a 256-key bitonic sort
                __shared__ float s_taux;
                __syncthreads();
                if ((uint32_t)tid < m) {
                    const uint64_t mine = xs[tid];
                    uint32_t rank = 0;
                    for (uint32_t i = 0; i < m; ++i) rank += xs[i] < mine ? 1u : 0u;
                    if (rank == a.k - 1) s_taux = key_score(mine, asc);
                }
                __syncthreads();
                const float tau_x = s_taux;
                const float e1 = 0.5f * a.marg2[q];
                cut_exact = asc ? tau_x + e1 : tau_x - e1;
                tightened = tau_x == tau_x;  // (NaN scores: keep the coarse rule)
                tight_m = s_x <= mx ? m : 0u;
            }
            __syncthreads();
        }
    }
    auto cut_of = [&](float t, float m2) -> float {   // the tighter of the coarse rule and the exact-rescored rule
        const float c = asc ? t + m2 : t - m2;
        if (!tightened) return c;
        return asc ? (cut_exact < c ? cut_exact : c) : (cut_exact > c ? cut_exact : c);
    };

    stamp(3);
    float thr_new;
    uint32_t keep;
    bool sorted_path = false;
    if (a.threshold_only) {  // k rows at least as good as tau exist: tau -/+ 2E is a valid cut for every row of the shard
        if (tid == 0) {
            a.count[q] = 0u;
            const float m2 = a.exact ? 0.0f : a.marg2[q];
            a.thr[q] = cut_of(tau, m2);
        }
        return 0u;
    }
    if (a.exact) {
        // FLAT scans rows in ascending id order, so a later tie of the k-th score can never win: strict
        // cut.  Packed-binary IVF scans slabs (keys carry ORIGINAL ids): a later tie with a smaller id
        // must still get in -> non-strict cut.
        thr_new = a.keep_ties ? tau : (asc ? nextafterf(tau, -LY_INF) : nextafterf(tau, LY_INF));
        keep = a.k;
    } else {
        const float m2 = a.marg2[q];
        thr_new = cut_of(tau, m2);
        uint32_t local = 0;
        for (uint32_t i = tid; i < n; i += NT) {
            const float sc = key_score(keys[i], asc);
            local += (keys[i] <= kth || (asc ? (sc <= thr_new) : (sc >= thr_new))) ? 1u : 0u;
        }
        for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
        if (lane == 0 && local) atomicAdd(&s_keep, local);
        __syncthreads();
        keep = s_keep;
        sorted_path = keep > a.keep_max;
        // Nothing but the keys <= kth survived and all of them were rescored for the threshold a moment ago (a decisive best row:
        // every k-means assignment, clustered collections): they go on with those exact scores, the final rescoring reads no row again
        if (a.hand_exact && tightened && tight_m && keep == tight_m) {
            __syncthreads();
            for (uint32_t i = tid; i < tight_m; i += NT) gkeys[i] = xs[i];
            if (tid == 0) {
                a.count[q] = keep;
                a.thr[q] = thr_new;
            }
            *rescored = true;
            return keep;
        }
    }

    if (sorted_path) {
        // ---- compaction: too many survivors (huge margins / massive ties): rescore them exactly and cut to the
        // exact top-k (+ ties of the k-th score when rows are not scanned in id order)
        const float m2 = a.marg2[q];
        const uint32_t np2 = next_pow2(n < 2 ? 2 : n);
        __syncthreads();
        for (uint32_t i = n + tid; i < np2; i += NT) keys[i] = KEY_SENTINEL;
        bitonic_sort_lds<NT>(keys, np2, tid);  // survivors are a prefix of the sorted keys
        rescore_keys<NT>(keys, keep, a.neg_metric1 ? a.neg_metric1 - 1 : a.metric, a.ip_form, a.Qf + (size_t)q * a.D, a.V, a.ld, a.D, asc, tid, a.neg_metric1 != 0);
        for (uint32_t i = keep + tid; i < np2; i += NT) keys[i] = KEY_SENTINEL;
        bitonic_sort_lds<NT>(keys, np2, tid);
        const float xk = key_score(keys[a.k - 1], asc);
        uint32_t kept = a.k;
        if (a.keep_ties) {
            if (tid == 0) s_keep = 0;
            __syncthreads();
            uint32_t ties = 0;
            for (uint32_t i = a.k + tid; i < keep; i += NT) ties += key_score(keys[i], asc) == xk ? 1u : 0u;
            if (ties) atomicAdd(&s_keep, ties);
            __syncthreads();
            kept += s_keep;
        }
        for (uint32_t i = tid; i < kept; i += NT) gkeys[i] = keys[i];
        if (tid == 0) {
            a.count[q] = kept;
            a.thr[q] = asc ? xk + m2 : xk - m2;
        }
        *rescored = true;
        return kept;
    }

    stamp(4);
    // ---- write the survivors back, compacted (order is irrelevant: one atomic per wave reserves the slots)
    __syncthreads();
    if (tid == 0) s_keep = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += NT) {
        const uint32_t i = i0 + tid;
        bool pass = false;
        uint64_t key = 0;
        if (i < n) {
            key = keys[i];
            if (a.exact) {
                pass = key <= kth;
            } else {
                const float sc = key_score(key, asc);
                pass = key <= kth || (asc ? (sc <= thr_new) : (sc >= thr_new));
            }
        }
        const uint64_t m = __ballot(pass);
        if (m) {
            uint32_t base = 0;
            const int leader = __builtin_ctzll(m);
            if (lane == leader) base = atomicAdd(&s_keep, (uint32_t)__popcll(m));
            base = __shfl(base, leader, 64);
            if (pass) gkeys[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
        }
    }
    if (tid == 0) {
        a.count[q] = keep;
        a.thr[q] = thr_new;
    }
    stamp(5);
    return keep;
}

template <int NT>
__global__ void __launch_bounds__(NT) k_select(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bool rescored;
    (void)select_body<NT>(a, reinterpret_cast<uint64_t*>(smem), blockIdx.x, &rescored);
}

// ------------------------------------------------------------------------------------------------
// k_final: one block per query.  Rescores the surviving pool exactly in the reference's
// accumulation order (float metrics), sorts by (score,row) and writes the top-k:
// rows (u64, after the shard row map), distances (f32) and the count.  Output order is the
// canonical (distance, row ascending) order of vector_store.rs:953-970.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// k_small_search — the whole search in ONE launch, for a few queries over a shard small enough that launch latency,
// not bandwidth, is the cost (config 1: FLAT-IP 100k x 128, one query, k = 10 — the staged pipeline spends ~20 us
// scanning and ~50 us in eight dependent launches).  Every row is scored EXACTLY from the f32 rows with the reference's
// accumulation order (exact_score, 8 lanes per row) — no coarse pass, no margin, no rescoring; each wave keeps its k best
// keys in registers (one key per lane, sorted; insertion = one shuffle), a workgroup merges its waves by rank, and the
// LAST workgroup to finish (atomic ticket) merges the per-workgroup lists with k rounds of a block-wide minimum and writes
// the outputs in the canonical (distance, row) order.  Limits: k <= 64, nq <= SMALL_MAX_Q, gridDim.x <= SMALL_NT.
// ------------------------------------------------------------------------------------------------
constexpr int SMALL_NT = 512;     // 8 waves per workgroup
constexpr int SMALL_MAX_Q = 4;
constexpr int SMALL_MAX_K = 64;

struct SmallArgs {
    const float* V;          // rows
    uint32_t ld, D;
    uint32_t n;
    const float* Qf;         // nq x D
    uint32_t nq, k, out_k;
    int metric, ip_form;
    uint64_t row_stride, row_offset;
    uint64_t* part;          // [nq][gridDim.x][k] sorted keys of every workgroup
    uint32_t* ticket;        // arrival counter (left at zero)
    uint64_t* out_rows;
    float* out_dists;
    uint32_t* out_counts;
    uint32_t* overflow;      // cleared: this path cannot overflow
    // IVF mode (probes != nullptr): query q scans the rows of its nprobe probed lists instead of rows [0, n) — slab
    // positions list_off[c] .. list_off[c+1] of every probed list c < nlist, keys carry orig[position] (the canonical
    // order and the output use the original row, like k_final).  flag_empty: overflow[q] = 1 when the probed lists hold
    // no row at all (IVFIndex then falls back to every row, ivf.rs:258-265: the host reruns the general path).
    const uint64_t* probes;  // [nq][nprobe] list ids (the fused centroid ranking's out_rows)
    uint32_t nprobe, nlist;
    const uint64_t* list_off;
    const uint32_t* orig;
    int flag_empty;
    unsigned long long* dbg;   // development (LYNSE_HIP_SMALL_DBG=1): s_memtime stamps [workgroup][4] + the last workgroup's [2]; NULL otherwise
};

__global__ void __launch_bounds__(SMALL_NT, 4) k_small_search(SmallArgs a) {   // (4 waves per SIMD = two workgroups per CU: the grid of run_small is sized for that)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 4 + 0] = wall_clock64();
    float* qs = reinterpret_cast<float*>(smem);                                   // D floats: the query being scanned
    uint64_t* wl = reinterpret_cast<uint64_t*>(smem + (size_t)((a.D + 3) / 4 * 4) * 4);  // [8 waves][64] keys; later the merge lists
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane & 7, grp = lane >> 3;
    const bool asc = a.metric != M_IP;
    const uint32_t k = a.k;
    constexpr int NWAVE = SMALL_NT / 64;
    const bool ivf = a.probes != nullptr;
    const uint32_t parts = gridDim.x;
    const uint32_t rows_per_pass = parts * NWAVE * 16;   // a group of 8 lanes scores TWO rows per pass (two load streams in flight)
    // IVF mode: the probed lists of the query laid end to end — s_pre[i] = rows before list i, s_start[i] = its slab position
    __shared__ uint32_t s_pre[SMALL_MAX_K + 1], s_start[SMALL_MAX_K];
    auto enter_lists = [&](uint32_t q) -> uint32_t {  // (called by every thread; ends with the arrays visible to all)
        __syncthreads();
        if (wave == 0) {
            uint32_t len = 0, start = 0;
            if ((uint32_t)lane < a.nprobe) {
                const uint64_t c = a.probes[(size_t)q * a.nprobe + lane];
                if (c < a.nlist) { start = (uint32_t)a.list_off[c]; len = (uint32_t)(a.list_off[c + 1] - a.list_off[c]); }
            }
            uint32_t inc = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64); if (lane >= o) inc += t; }
            if (lane < SMALL_MAX_K) { s_pre[lane + 1] = inc; s_start[lane] = start; }
            if (lane == 0) s_pre[0] = 0;
        }
        __syncthreads();
        return s_pre[a.nprobe < (uint32_t)SMALL_MAX_K ? a.nprobe : (uint32_t)SMALL_MAX_K];
    };
    auto slab_pos = [&](uint32_t v) -> uint32_t {  // v-th row of the concatenated lists -> slab position (v < total)
        uint32_t lo = 0, hi = a.nprobe;  // last i with s_pre[i] <= v
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_pre[mid] <= v) lo = mid; else hi = mid; }
        return s_start[lo] + (v - s_pre[lo]);
    };
    // (round 6, measured and removed: touching the rows of the first pass — one dword per 128-B line — before the query has reached LDS,
    // so that the scan's first loads find their lines on the way: C1 31.9 / 32.7 against 31.2 / 31.9 us — nothing; the scan is bandwidth-bound)
    for (uint32_t q = 0; q < a.nq; ++q) {
        __syncthreads();
        for (uint32_t i = tid; i < a.D; i += SMALL_NT) qs[i] = a.Qf[(size_t)q * a.D + i];
        __syncthreads();
        const uint32_t n_rows = ivf ? enter_lists(q) : a.n;
        uint64_t L = KEY_SENTINEL;  // lane j holds the wave's j-th best key (ascending = best first)
        uint64_t thr = KEY_SENTINEL;
        auto offer = [&](uint64_t key) {
            unsigned long long m = __ballot(g == 0 && key < thr);
            while (m) {  // rare once the list has warmed up
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint64_t x = readlane_u64(key, src);
                if (x < thr) {
                    const uint64_t prev = wave_shr1_u64(L);
                    if (L > x) L = (lane == 0 || prev <= x) ? x : prev;
                    thr = readlane_u64(L, (int)k - 1);
                }
            }
        };
        for (uint32_t r0 = (blockIdx.x * NWAVE + wave) * 16; r0 < n_rows; r0 += rows_per_pass) {
            const uint32_t ra = r0 + grp, rb = r0 + 8 + grp;
            // out-of-range rows re-read the last row (uniform control flow inside exact_score) and are dropped afterwards
            uint32_t ca = ra < n_rows ? ra : n_rows - 1, cb = rb < n_rows ? rb : n_rows - 1;
            if (ivf) { ca = slab_pos(ca); cb = slab_pos(cb); }
            const float sa = exact_score<16>(a.metric, a.ip_form, qs, a.V + (size_t)ca * a.ld, a.D, g);
            const float sb = exact_score<16>(a.metric, a.ip_form, qs, a.V + (size_t)cb * a.ld, a.D, g);
            if (ivf && a.orig) { ca = a.orig[ca]; cb = a.orig[cb]; }
            else if (!ivf) { ca = ra; cb = rb; }
            offer(ra < n_rows ? make_key(sa, ca, asc) : KEY_SENTINEL);
            offer(rb < n_rows ? make_key(sb, cb, asc) : KEY_SENTINEL);
        }
        wl[wave * 64 + lane] = lane < (int)k ? L : KEY_SENTINEL;
        __syncthreads();
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 4 + 1] = wall_clock64();
        {   // merge the waves by rank: one key per thread, keys are unique apart from the sentinel.  Every wave list is sorted
            // (slots >= k hold the sentinel), so the rank of a key is the sum of eight branch-free binary searches — seven
            // dependent LDS reads with the eight lists interleaved, instead of 8 k sequential reads per thread.
            auto rank_of = [&](uint64_t x) -> uint32_t {
                uint32_t pos[NWAVE];
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) pos[w] = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) pos[w] += (wl[w * 64 + pos[w] + step - 1] < x) ? (uint32_t)step : 0u;
                }
                uint32_t r = 0;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) r += pos[w] + ((wl[w * 64 + pos[w]] < x) ? 1u : 0u);  // pos <= 63
                return r;
            };
            const uint64_t mine = wl[tid];
            uint64_t* dst = a.part + ((size_t)q * parts + blockIdx.x) * k;  // [nq][workgroups][k]: a query's lists are contiguous
            // 8-byte agent-scope atomics on both sides of the hand-off (write-through stores, L2-served loads): no cache
            // write-back / invalidate fences around the ticket (cdna_hip_programming.md G16, 'valid forms')
            if (mine != KEY_SENTINEL) {
                const uint32_t rank = rank_of(mine);
                if (rank < k) __hip_atomic_store(&dst[rank], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // slots past the number of real keys hold the sentinel
            if (tid < (int)k) {
                const uint32_t real = rank_of(KEY_SENTINEL);
                if ((uint32_t)tid >= real) __hip_atomic_store(&dst[tid], KEY_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- the last workgroup to arrive merges the per-workgroup lists: agent-scope atomic stores of the keys -> every wave
    // drains its stores -> barrier -> ticket; the last arriver reads the lists with agent-scope atomic loads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (a.dbg) a.dbg[(size_t)blockIdx.x * 4 + 2] = wall_clock64();
        const uint32_t t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == parts - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {
    // Merge of the gridDim.x sorted lists WITHOUT a tournament (two k-round tournaments — a DPP wave minimum and a dependent LDS
    // read per rank and level — were 7.8 us of the 24.6 us kernel on 100k x 128, whatever the number of lists):
    //   1. one head (best key) per list and thread; its rank among the 64 heads of the wave by v_readlane counting; the wave's
    //      kout smallest heads land rank-ordered in wh[wave][.];
    //   2. T = the kout-th smallest of all heads (rank over the eight sorted wave rows: branch-free binary searches).  kout lists
    //      start at or below T, so kout keys <= T exist and no key above T is among the kout best;  fewer than kout non-empty
    //      lists: T = the sentinel (every real key is a candidate);
    //   3. the (at most kout) lists whose head is <= T read their keys <= T (agent-scope loads, 8 in flight) into the candidate
    //      array in LDS — a few dozen keys on ordinary data, kout * k at the very worst;
    //   4. every candidate's rank by counting (keys are unique): rank < kout writes output slot `rank`.
    const uint32_t nlist = parts;                     // <= SMALL_NT: one list per thread
    uint64_t* wh = wl;                                // [NWAVE][64]
    uint64_t* cnd = wl + (size_t)NWAVE * 64;          // candidates (the launch sizes the dynamic LDS: min(k, nlist) * k slots, rounded up to a power of two)
    __shared__ uint64_t s_T;
    __shared__ uint32_t s_m;
    auto rank_rows = [&](uint64_t x) -> uint32_t {    // keys of wh smaller than x (every row sorted, sentinel-padded)
        uint32_t pos[NWAVE];
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) pos[w] = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) pos[w] += (wh[w * 64 + pos[w] + step - 1] < x) ? (uint32_t)step : 0u;
        }
        uint32_t r = 0;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) r += pos[w] + ((wh[w * 64 + pos[w]] < x) ? 1u : 0u);
        return r;
    };
    for (uint32_t q = 0; q < a.nq; ++q) {
        const uint32_t n_rows = ivf ? enter_lists(q) : a.n;
        const uint32_t kout = k < n_rows ? k : n_rows;
        const uint64_t* src = a.part + (size_t)q * nlist * k;
        __syncthreads();
        const bool owner = (uint32_t)tid < nlist;
        const uint64_t head = owner ? __hip_atomic_load(&src[(size_t)tid * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : KEY_SENTINEL;
        // (round 6) few lists with many keys — the centroid ranking of an IVF search: 8..64 lists x nprobe = 32 keys: EVERY key of every list is
        // fetched here, in the round trip of the heads (<= 4 per thread), instead of one dependent round trip per 8 keys of a list below the
        // cut (that merge took 10 us of the 26-us ranking launch with 8 lists, k = 32)
        const bool all_keys = nlist * k <= 4u * (uint32_t)SMALL_NT;
        uint64_t allk[4] = {KEY_SENTINEL, KEY_SENTINEL, KEY_SENTINEL, KEY_SENTINEL};
        if (all_keys) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = (uint32_t)tid + (uint32_t)u * (uint32_t)SMALL_NT;
                if (i < nlist * k) allk[u] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        wh[tid] = KEY_SENTINEL;
        if (tid == 0) { s_T = KEY_SENTINEL; s_m = 0; }
        uint32_t hr = 0;   // heads of this wave below mine
#pragma unroll
        for (int j = 0; j < 64; ++j) hr += readlane_u64(head, j) < head ? 1u : 0u;
        __syncthreads();
        if (head != KEY_SENTINEL && hr < kout) wh[wave * 64 + hr] = head;
        __syncthreads();
        {
            const uint64_t mine = wh[tid];
            if (mine != KEY_SENTINEL && kout && rank_rows(mine) == kout - 1) s_T = mine;
        }
        __syncthreads();
        const uint64_t T = s_T;
        if (all_keys) {   // (uniform) at most kout lists start at or below T: <= min(k, lists) * k candidates, the slots the launch sized
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (allk[u] != KEY_SENTINEL && allk[u] <= T) cnd[atomicAdd(&s_m, 1u)] = allk[u];
        } else if (head != KEY_SENTINEL && head <= T) {
            cnd[atomicAdd(&s_m, 1u)] = head;
            // (loading the first 16 keys of every list up front instead — one round trip, no second one here — was slower: 8192
            // strided 8-byte loads against 512 + a few dozen; the merge 6.4 us against 5.1 us)
            for (uint32_t j0 = 1; j0 < k; j0 += 8) {
                uint64_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = j0 + u < k ? __hip_atomic_load(&src[(size_t)tid * k + j0 + u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : KEY_SENTINEL;
                bool more = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    more = more && v[u] != KEY_SENTINEL && v[u] <= T;   // (the list is sorted: the first key above T ends it)
                    if (more) cnd[atomicAdd(&s_m, 1u)] = v[u];
                }
                if (!more) break;
            }
        }
        __syncthreads();
        const uint32_t m = s_m;
        if (m <= 512u) {
            for (uint32_t i = tid; i < m; i += SMALL_NT) {
                const uint64_t mine = cnd[i];
                uint32_t r = 0;
                for (uint32_t j = 0; j < m; ++j) r += cnd[j] < mine ? 1u : 0u;
                if (r < kout) {
                    a.out_rows[(size_t)q * a.out_k + r] = (uint64_t)key_row(mine) * a.row_stride + a.row_offset;
                    a.out_dists[(size_t)q * a.out_k + r] = key_score(mine, asc);
                }
            }
        } else {
            // rows stored in score order put whole lists below T (k = 64: up to 63 * 64 + 1 candidates, m^2 / 512 compares per
            // thread): sort instead (the candidate slots are sized to a power of two)
            const uint32_t np2 = next_pow2(m);
            for (uint32_t i = m + tid; i < np2; i += SMALL_NT) cnd[i] = KEY_SENTINEL;
            bitonic_sort_lds<SMALL_NT>(cnd, np2, tid);
            for (uint32_t r = tid; r < kout; r += SMALL_NT) {
                a.out_rows[(size_t)q * a.out_k + r] = (uint64_t)key_row(cnd[r]) * a.row_stride + a.row_offset;
                a.out_dists[(size_t)q * a.out_k + r] = key_score(cnd[r], asc);
            }
        }
        // short results are padded like k_final's (row ~0, the worst distance of the metric)
        for (uint32_t i = kout + tid; i < a.out_k; i += SMALL_NT) {
            a.out_rows[(size_t)q * a.out_k + i] = ~0ull;
            a.out_dists[(size_t)q * a.out_k + i] = asc ? LY_INF : -LY_INF;
        }
        if (tid == 0) { a.out_counts[q] = kout; a.overflow[q] = (ivf && a.flag_empty && n_rows == 0) ? 1u : 0u; }
    }
    if (tid == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 4 + 3] = wall_clock64();
    }   // s_last
}

// (round 6, measured and removed: k_small_search_ivf2 — the centroid ranking and the probed lists of a few-query IVF search in ONE launch, the
// ranking's probe list handed over grid-wide through a flag in device memory (release fence in the merging workgroup, relaxed polls with
// s_sleep in the others, a bounded wait with a two-launch fallback).  1.6M x 768, nlist 1024, nprobe 32, one query: 73.3-73.8 us against
// 69.8-71.5 us for the two launches, same box, alternating — a grid-wide hand-over costs more than the kernel boundary it replaces, as the
// fused sample stage of round 2 already found (first versions: an agent-scope release / acquire fence in EVERY wave = an L2 write-back /
// invalidate each: 110 us; acquire-ordered polls: 180 us).  scripts/gpu_r6_k1.sh)

struct FinalArgs {
    const uint64_t* cand;
    const uint32_t* count;
    uint32_t k, cap;
    uint32_t out_k;  // row stride of the output arrays (caller's k >= effective k)
    int metric, ip_form, exact;
    const float* Qf;
    const float* V;
    uint32_t ld, D;
    uint64_t row_stride, row_offset;
    const uint32_t* orig_ids;  // IVF: slab position -> original row (tie-break and output use the original row)
    uint64_t* out_rows;
    float* out_dists;
    uint32_t* out_counts;
    uint32_t* out_counts2;   // optional second destination of the counts (the caller's device array; the first one sits next to
                             // the overflow flags and comes back in the pinned header)
    unsigned long long* pool_total;
    // searches in flight (lynse_hip_flat_search_submit_*): the per-query overflow flags of k_select are OR-ed into ONE status
    // word that travels with the result block (the host looks at it when the ticket is waited for); NULL otherwise
    const uint32_t* overflow;
    uint32_t* any_overflow;
    // pinned host header of the search context: counts in [0, hdr_q), overflow flags in [hdr_q, 2 hdr_q) — written here instead
    // of being copied back by a copy kernel behind the search; NULL = not wanted
    uint32_t* h_hdr;
    uint32_t hdr_q;
    // batched Hamming on the matrix pipe: the keys carry the +-1 dot product (best-first as an IP score); the distance
    // written out is (ham_dim - dot) / 2 — exact (integers below 2^24).  0 = off
    uint32_t ham_dim;
    int neg_metric1;   // as SelectArgs::neg_metric1: rescoring with metric neg_metric1 - 1, negated; the distances written out are negated back
    uint32_t lds_bytes;   // as SelectArgs::lds_bytes
    // The NaN rule of the boundary (include/lynse_hip.h): a NaN score is reported as the WORST value of the metric and ranks behind every
    // better score, ties by ascending row (make_key).  A query with a NaN element — cosine: or an infinite one — scores NaN against EVERY row,
    // and nothing ever passes a threshold of the staged pipeline for it: its answer is written here directly, rows 0 .. min(k, nan_rows) - 1 at
    // the worst value (what the rule gives, and the rows the reference's first fill keeps, flat_mmap.rs:2141-2149).  nan_rows = rows of the
    // store for an unfiltered FLAT search over f32 / f16 rows, 0 = rule off (subset filters, IVF slabs, binary metrics).
    uint32_t nan_rows;
};

// k_rescore_pool: the exact rescoring of k_final spread over gridDim.y blocks per query (IVF on tightly clustered
// data keeps ~1000 candidates per query inside the certified margin: one block per query took 150-280 us).
// Rescored keys are written back in place; k_final then runs with exact = 1.
template <int NT>
__global__ void __launch_bounds__(NT) k_rescore_pool(FinalArgs a) {
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x, g = tid & 7;
    const bool asc = metric_ascending(a.metric);
    uint32_t n = a.count[q];
    if (n > a.cap) n = a.cap;
    uint64_t* keys = const_cast<uint64_t*>(a.cand) + (size_t)q * a.cap;
    const float* qv = a.Qf + (size_t)q * a.D;
    for (uint32_t i = blockIdx.y * (NT / 8) + (tid >> 3); i < n; i += gridDim.y * (NT / 8)) {
        const uint32_t row = key_row(keys[i]);
        float s = exact_score<32>(a.neg_metric1 ? a.neg_metric1 - 1 : a.metric, a.ip_form, qv, a.V + (size_t)row * a.ld, a.D, g);
        if (a.neg_metric1) s = -s;
        if (g == 0) keys[i] = make_key(s, row, asc);
    }
}

// final_body: k_final for query q — n survivors in cand[q][0 .. n), `keys` = cap slots of LDS.
template <int NT>
__device__ __forceinline__ void final_body(const FinalArgs& a, uint64_t* keys, const uint32_t q, uint32_t n, const bool exact, const bool same_launch = false) {
    const int tid = threadIdx.x;
    const bool asc = metric_ascending(a.metric);
    if (a.nan_rows) {   // (uniform) FinalArgs::nan_rows: the query whose every score is NaN
        const int real_metric = a.neg_metric1 ? a.neg_metric1 - 1 : a.metric;
        const float* qv = a.Qf + (size_t)q * a.D;
        int bad = 0;
        for (uint32_t i = tid; i < a.D; i += NT) { const float x = qv[i]; bad |= (x != x || (real_metric == M_COS && fabsf(x) == LY_INF)) ? 1 : 0; }
        if (__syncthreads_or(bad)) {
            const uint32_t cnt = a.k < a.nan_rows ? a.k : a.nan_rows;
            for (uint32_t i = tid; i < a.out_k; i += NT) {
                a.out_rows[(size_t)q * a.out_k + i] = i < cnt ? (uint64_t)i * a.row_stride + a.row_offset : ~0ull;
                a.out_dists[(size_t)q * a.out_k + i] = metric_ascending(real_metric) ? LY_INF : -LY_INF;
            }
            if (tid == 0) {
                a.out_counts[q] = cnt;
                if (a.out_counts2) a.out_counts2[q] = cnt;
                if (a.h_hdr) { a.h_hdr[q] = cnt; a.h_hdr[a.hdr_q + q] = 0u; }   // (nothing overflowed: there is nothing to answer again)
            }
            return;
        }
    }
    if (n > a.cap) n = a.cap;
    const uint32_t np2 = next_pow2(n < 2 ? 2 : n);
    const uint64_t* src = a.cand + (size_t)q * a.cap;
    // same_launch: the keys were stored by THIS workgroup a barrier ago — read them past the vector L1 (sc1: L2-served)
    for (uint32_t i = tid; i < np2; i += NT)
        keys[i] = i < n ? (same_launch ? __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[i]) : KEY_SENTINEL;
    __syncthreads();
    // (the survivors of a batch — a few dozen rows — are scored straight from global memory on f32 shards: staged through LDS in chunks
    // they cost one global round trip per chunk, measured 14 us against 12 us for 53 rows of 768 floats.  F16 shards stage them: half the
    // bytes per row, and the f16 kernels' sequential sums are a dependent round trip per 64 elements otherwise)
    if (!exact) {
        const bool f16 = a.ip_form == LYNSE_IPFORM_F16SEQ;
        const uint32_t used = (np2 > 512u ? np2 : 512u) * 8u, lds_all = a.lds_bytes ? a.lds_bytes : a.cap * 8u;   // (the rank sort below writes keys[256 .. 512))
        rescore_keys<NT>(keys, n, a.neg_metric1 ? a.neg_metric1 - 1 : a.metric, a.ip_form, a.Qf + (size_t)q * a.D, a.V, a.ld, a.D, asc, tid, a.neg_metric1 != 0,
                         f16 ? reinterpret_cast<char*>(keys) + used : nullptr, (f16 && lds_all > used) ? lds_all - used : 0u);
    }
    if (a.orig_ids) {  // canonical order is (distance, ORIGINAL row): swap the row word before the final sort
        for (uint32_t i = tid; i < n; i += NT) keys[i] = (keys[i] & 0xffffffff00000000ull) | a.orig_ids[key_row(keys[i])];
        __syncthreads();
    }
    // a few dozen survivors (the usual pool): rank sort — every thread counts the keys below its own (n broadcast LDS reads,
    // keys are unique) — instead of the 28 barrier-separated steps of a 128-key bitonic network
    if (n <= 256u && a.cap >= 512u) {
        uint64_t* sorted = keys + 256;
        uint64_t mine = KEY_SENTINEL;
        uint32_t rank = 0;
        if ((uint32_t)tid < n) {
            mine = keys[tid];
            for (uint32_t i = 0; i < n; ++i) rank += (keys[i] < mine || (keys[i] == mine && i < (uint32_t)tid)) ? 1u : 0u;   // (a permutation whatever the keys)
        }
        __syncthreads();
        if ((uint32_t)tid < n) sorted[rank] = mine;
        __syncthreads();
        keys = sorted;
    } else {
        bitonic_sort_lds<NT>(keys, np2, tid);
    }
    const uint32_t cnt = n < a.k ? n : a.k;
    for (uint32_t i = tid; i < a.out_k; i += NT) {
        if (i < cnt) {
            a.out_rows[(size_t)q * a.out_k + i] = (uint64_t)key_row(keys[i]) * a.row_stride + a.row_offset;
            const float sc = key_score(keys[i], asc);
            a.out_dists[(size_t)q * a.out_k + i] = a.ham_dim ? ((float)a.ham_dim - sc) * 0.5f : (a.neg_metric1 ? 0.0f - sc : sc);   // (0 - (+0) = +0: a zero distance keeps its sign)
        } else {
            a.out_rows[(size_t)q * a.out_k + i] = ~0ull;
            a.out_dists[(size_t)q * a.out_k + i] = (a.ham_dim || a.neg_metric1 || asc) ? LY_INF : -LY_INF;
        }
    }
    if (tid == 0) {
        a.out_counts[q] = cnt;
        if (a.out_counts2) a.out_counts2[q] = cnt;
        if (a.pool_total) atomicAdd(a.pool_total, (unsigned long long)n);
        if (a.any_overflow && a.overflow[q]) atomicOr(a.any_overflow, 1u);
        if (a.h_hdr) { a.h_hdr[q] = cnt; a.h_hdr[a.hdr_q + q] = a.overflow[q]; }
    }
}

template <int NT>
__global__ void __launch_bounds__(NT) k_final(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    final_body<NT>(a, reinterpret_cast<uint64_t*>(smem), blockIdx.x, a.count[blockIdx.x], a.exact != 0);
}

// k_select_final: the select behind the LAST scan stage, the exact rescoring of its survivors and the final order in ONE
// launch per batch (they were three: k_select, k_rescore_pool / the rescoring half of k_final, k_final — 40 us of a 345 us
// step on a 1.25M-row shard, two kernel boundaries and two trips of the survivors through HBM).  The survivors go to
// cand[q] as before (same workgroup: visible behind the barrier) and come back into LDS for the rescoring and the sort.
struct TailArgs {
    SelectArgs s;
    FinalArgs f;
};

template <int NT>
__global__ void __launch_bounds__(NT) k_select_final(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    const uint32_t q = blockIdx.x;
    bool rescored;
    const uint32_t n = select_body<NT>(a.s, keys, q, &rescored);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's survivor stores have reached L2 (they are read back sc1)
    __syncthreads();
    final_body<NT>(a.f, keys, q, n, a.f.exact != 0 || rescored, true);
    if (a.s.stamps && threadIdx.x == 0) a.s.stamps[(size_t)q * 8 + 6] = __builtin_amdgcn_s_memtime();
}

// ------------------------------------------------------------------------------------------------
// k_assign_pick — the arg-best epilogue of the k-means ASSIGNMENT pass (kmeans::assign_metric, kmeans.rs:237-264).
// The assignment scores every data row ("query") against every centroid ("row" of the centroid store).  Its scan runs in the
// lane-max mode of k_scan_h16 (emit_all = 2): per centroid tile every lane keeps the best two centroids of those it scored for a data
// row, so the nkeys = tiles x 16 (IP tiling) / x 8 (L2, cosine tiling) keys of a data row hold its two best COARSE centroids (the
// runner-up overall is either the best of its lane or the second of the winner's lane).  One wave per data row finds them:
//   gap(best, runner-up) > 2E (the certified margin of k_prep_queries)  =>  exact(best) >= coarse(best) - E > coarse(c) + E >= exact(c)
//   for every other centroid c: the row is assigned WITHOUT exact rescoring, to the centroid the reference's strict first-smaller rule
//   picks (no tie is possible);
//   otherwise the row goes onto the redo list and is answered by the exact top-1 search (a handful on any real data).
// It replaces, in this shape, the emit-all scan (4096 keys per data row: 268 MB per 8192-row launch) + k_select_final<512> (446 us per
// launch, 60 % of the training time: profiles/r04_c4_train_kernel_stats.csv).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_assign_pick(const uint64_t* __restrict__ cand, uint32_t cap, uint32_t nkeys, const float* __restrict__ marg2,
                                                     int asc, uint32_t nq, uint32_t q_base, uint32_t* __restrict__ out_ids,
                                                     uint32_t* __restrict__ redo_list, uint32_t* __restrict__ redo_count,
                                                     const float* __restrict__ Qf, const float* __restrict__ V, uint32_t ld, uint32_t D,
                                                     int metric, int ip_form) {
    // Near-ties (round 5, second step): with more lists than the data has clusters the sibling centroids of a cluster score within the
    // f16 margin of each other for MOST rows (6.25M x 768, 1024 clusters under 4096 lists: the redo path answered so many rows that
    // training took 19 s).  The lane-max keys come in pairs — slot 2j = the best, slot 2j + 1 = the second best centroid of one scan
    // lane — so the set S of centroids within 2E of the best coarse score is COMPLETE whenever no second-best key lies inside the
    // margin (then every scan lane holds at most one member of S, its best).  The members of S (<= ASSIGN_CMAX) are scored exactly with
    // the reference's single-row kernels, 8 lanes per candidate, and the canonical minimum (score, centroid id) — the reference's
    // first-strictly-smaller rule — is the assignment.  Only rows with a hidden candidate or an over-full S go to the redo list.
    constexpr uint32_t ASSIGN_CMAX = 16;
    __shared__ uint64_t s_c[4][ASSIGN_CMAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * 4 + (uint32_t)w;
    if (q >= nq) return;   // (uniform per wave)
    const bool up = asc != 0;
    const uint64_t* keys = cand + (size_t)q * cap;
    uint64_t b1 = KEY_SENTINEL;
    for (uint32_t i0 = 0; i0 < nkeys; i0 += 256) {   // four loads in flight per lane
        uint64_t kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + u * 64 + lane; kv[u] = i < nkeys ? keys[i] : KEY_SENTINEL; }
#pragma unroll
        for (int u = 0; u < 4; ++u) b1 = kv[u] < b1 ? kv[u] : b1;
    }
    const uint64_t best = wave_min_u64(b1);
    const float sb = key_score(best, up), m2 = marg2[q];
    bool hidden = false;
    uint32_t ncand = 0;
    if (best != KEY_SENTINEL) {
        for (uint32_t i0 = 0; i0 < nkeys; i0 += 64) {   // (the keys again: L2-resident, 2 KB per row)
            const uint32_t i = i0 + lane;
            const uint64_t key = i < nkeys ? keys[i] : KEY_SENTINEL;
            const float sc = key_score(key, up);
            const bool in = key != KEY_SENTINEL && (up ? (sc - sb <= m2) : (sb - sc <= m2));   // (NaN compares false)
            const bool second = (i & 1u) != 0u;
            hidden = hidden || __ballot(in && second) != 0ull;
            const uint64_t m = __ballot(in && !second);
            if (in && !second) {
                const uint32_t pos = ncand + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < ASSIGN_CMAX) s_c[w][pos] = key;
            }
            ncand += (uint32_t)__popcll(m);
        }
    }
    // (sb - sb <= m2 holds for the best key itself unless the score or the margin is NaN / the difference of infinities: then ncand = 0)
    const bool decided = best != KEY_SENTINEL && !hidden && ncand >= 1 && ncand <= ASSIGN_CMAX;
    if (!decided) {   // (uniform)
        if (lane == 0) {
            out_ids[q_base + q] = 0xffffffffu;
            redo_list[atomicAdd(redo_count, 1u)] = q_base + q;
        }
        return;
    }
    if (ncand == 1) {   // the best coarse centroid is alone inside the margin: exact(best) > exact(c) for every other c, no rescoring
        if (lane == 0) out_ids[q_base + q] = key_row(best);
        return;
    }
    __builtin_amdgcn_wave_barrier();
    const float* qv = Qf + (size_t)(q_base + q) * D;
    const int g = lane & 7;
    uint64_t bestx = KEY_SENTINEL;
    for (uint32_t r0 = 0; r0 < ncand; r0 += 8) {   // 8 candidates per trip, 8 lanes each (uniform control flow inside exact_score)
        const uint32_t j = r0 + (uint32_t)(lane >> 3);
        const uint32_t crow = key_row(s_c[w][j < ncand ? j : 0u]);
        const float sc = exact_score<32>(metric, ip_form, qv, V + (size_t)crow * ld, D, g);
        const uint64_t key = j < ncand ? make_key(sc, crow, up) : KEY_SENTINEL;
        const uint64_t mn = wave_min_u64(key);
        bestx = mn < bestx ? mn : bestx;
    }
    if (lane == 0) out_ids[q_base + q] = key_row(bestx);
}

// ------------------------------------------------------------------------------------------------
// k_merge: k-way merge of per-shard result blocks after the RCCL all-gather — one block per query.
// (score image, global row) pairs are sorted with an LDS bitonic network under the canonical
// (distance in metric order, row ascending) order: VectorStore::merge_results (vector_store.rs:953-970),
// cluster::merge_search_blocks (cluster.rs:327-393).
// ------------------------------------------------------------------------------------------------
struct MergeArgs {
    const char* blocks;
    uint64_t block_bytes, rows_off, dists_off, counts_off;
    uint32_t n_lists, k;
    int metric;
    uint64_t* out_rows;
    float* out_dists;
    uint32_t* out_counts;
    // searches in flight: every block carries a status word at status_off (bit 0: a query of that shard overflowed its
    // candidate buffers); their OR goes to out_status (pinned host memory, read when the ticket is waited for)
    uint64_t status_off;
    uint32_t* out_status;
};

template <int NT>
__global__ void __launch_bounds__(NT) k_merge(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t total = a.n_lists * a.k;
    const uint32_t np2 = next_pow2(total < 2 ? 2 : total);
    uint64_t* ids = reinterpret_cast<uint64_t*>(smem);
    uint32_t* ords = reinterpret_cast<uint32_t*>(ids + np2);
    __shared__ uint32_t s_n;
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const bool asc = metric_ascending(a.metric);
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (uint32_t i = tid; i < np2; i += NT) {
        uint64_t id = ~0ull;
        uint32_t o = 0xffffffffu;
        if (i < total) {
            const uint32_t l = i / a.k, j = i % a.k;
            const char* blk = a.blocks + (size_t)l * a.block_bytes;
            const uint32_t cnt = reinterpret_cast<const uint32_t*>(blk + a.counts_off)[q];
            if (j < cnt) {
                id = reinterpret_cast<const uint64_t*>(blk + a.rows_off)[(size_t)q * a.k + j];
                const float d = reinterpret_cast<const float*>(blk + a.dists_off)[(size_t)q * a.k + j];
                o = (uint32_t)(make_key(d, 0u, asc) >> 32);
                atomicAdd(&s_n, 1u);
            }
        }
        ids[i] = id;
        ords[i] = o;
    }
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < (np2 >> 1); i += NT) {
                const uint32_t lo = 2 * i - (i & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t ox = ords[lo], oy = ords[hi];
                const uint64_t ix = ids[lo], iy = ids[hi];
                const bool gt = ox > oy || (ox == oy && ix > iy);
                if (gt == up) { ords[lo] = oy; ords[hi] = ox; ids[lo] = iy; ids[hi] = ix; }
            }
        }
    }
    __syncthreads();
    const uint32_t n = s_n;
    const uint32_t cnt = n < a.k ? n : a.k;
    for (uint32_t i = tid; i < a.k; i += NT) {
        if (i < cnt) {
            a.out_rows[(size_t)q * a.k + i] = ids[i];
            a.out_dists[(size_t)q * a.k + i] = key_score((uint64_t)ords[i] << 32, asc);
        } else {
            a.out_rows[(size_t)q * a.k + i] = ~0ull;
            a.out_dists[(size_t)q * a.k + i] = asc ? LY_INF : -LY_INF;
        }
    }
    if (tid == 0) a.out_counts[q] = cnt;
    if (q == 0 && tid == 0 && a.out_status) {
        uint32_t st = 0;
        for (uint32_t l = 0; l < a.n_lists; ++l) st |= *reinterpret_cast<const uint32_t*>(a.blocks + (size_t)l * a.block_bytes + a.status_off);
        *a.out_status = st;
    }
}

}  // namespace lynse
