// common.h — shared host/device definitions for liblynse_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lynse_hip.h"

namespace lynse {

// Metric classes used by the kernels.
enum : int { M_IP = 0, M_L2 = 1, M_COS = 2, M_HAMMING = 3, M_JACCARD = 4, M_DICE = 5, M_TANIMOTO = 6 };

__host__ __device__ inline bool metric_ascending(int m) { return m != M_IP; }  // distance/mod.rs:111-116
__host__ __device__ inline bool metric_binary(int m) { return m >= M_HAMMING; }  // distance/mod.rs:161-166

// ---- candidate keys -------------------------------------------------------------------------
// A candidate is one u64: high word = order-preserving image of the f32 score arranged so that an
// ASCENDING u64 sort is best-first for the metric, low word = local row id.  Sorting keys therefore
// yields the canonical (distance in metric order, row ascending) order of
// VectorStore::merge_results (vector_store.rs:953-970).
__host__ __device__ inline uint32_t f32_to_ord(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ inline float ord_to_f32(uint32_t o) {
    union { float f; uint32_t u; } c;
    c.u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return c.f;
}
__host__ __device__ inline uint64_t make_key(float s, uint32_t row, bool asc) {
    if (s != s) s = asc ? __builtin_huge_valf() : -__builtin_huge_valf();  // NaN sorts last
    s = s + 0.0f;                                                         // -0.0 -> +0.0
    uint32_t o = f32_to_ord(s);
    if (!asc) o = ~o;
    return ((uint64_t)o << 32) | (uint64_t)row;
}
__host__ __device__ inline float key_score(uint64_t key, bool asc) {
    uint32_t o = (uint32_t)(key >> 32);
    if (!asc) o = ~o;
    return ord_to_f32(o);
}
__host__ __device__ inline uint32_t key_row(uint64_t key) { return (uint32_t)key; }

constexpr uint64_t KEY_SENTINEL = ~0ull;

// ---- geometry of the f16 MFMA scan ------------------------------------------------------------
constexpr int SCAN_BK = 64;         // K elements per slab
constexpr int SCAN_LDK = 72;        // LDS / Q-image row stride in halves (144 B: conflict-free ds_read_b128)
constexpr int SCAN_BR = 128;        // rows per tile
constexpr int SCAN_BQ_LARGE = 256;  // queries per tile, large-batch config
constexpr int SCAN_BQ_SMALL = 32;   // queries per tile, small-batch config

}  // namespace lynse
