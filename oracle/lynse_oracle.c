/*
 * lynse_oracle.c — CPU restatement of the LynseDB FLAT / IVF-Flat search hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see lynse_oracle.h).  Not part of the shipped
 * product; never linked into liblynse_hip.so.
 *
 * Pinning: the reference (Rust) cannot be compiled in the build image, so the
 * restatement is pinned against the known-answer tests of the reference's own
 * test-suite (transcribed in tests/test_oracle_kat.py, listed in SURVEY.md
 * §8c) and against golden vectors produced by importing the reference's
 * pure-Python modules (tests/golden/).
 *
 * Build: gcc -O2 -mavx2 -mfma -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * A portable lane-emulation build (-DLO_NO_INTRINSICS) gives bit-identical
 * results; tests compare the two.
 */
#define _GNU_SOURCE
#include "lynse_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

#if defined(__AVX2__) && defined(__FMA__) && !defined(LO_NO_INTRINSICS)
#include <immintrin.h>
#define LO_AVX2 1
#else
#define LO_AVX2 0
#endif

int lo_has_avx2_fma(void) { return LO_AVX2; }

/* ------------------------------------------------------------------ metrics */

int lo_metric_is_ascending(int metric) { return metric != LO_IP; } /* distance/mod.rs:111-116 */

int lo_metric_is_binary(int metric) { /* distance/mod.rs:161-166 */
    return metric == LO_HAMMING || metric == LO_JACCARD || metric == LO_DICE ||
           metric == LO_TANIMOTO;
}

static void lower_copy(const char *s, char *out, size_t cap) {
    size_t i = 0;
    for (; s[i] && i + 1 < cap; ++i) out[i] = (char)tolower((unsigned char)s[i]);
    out[i] = 0;
}

int lo_metric_from_str(const char *s) { /* distance/mod.rs:39-63 */
    char b[64];
    lower_copy(s, b, sizeof b);
    static const struct { const char *name; int m; } tab[] = {
        {"ip", LO_IP}, {"inner_product", LO_IP}, {"inner", LO_IP}, {"dot", LO_IP},
        {"l2", LO_L2}, {"l2sq", LO_L2}, {"l2_squared", LO_L2}, {"euclidean", LO_L2},
        {"cosine", LO_COS}, {"cos", LO_COS}, {"cosine_distance", LO_COS},
        {"hamming", LO_HAMMING}, {"jaccard", LO_JACCARD},
        {"dice", LO_DICE}, {"sorensen", LO_DICE}, {"sorensen_dice", LO_DICE},
        {"sorensen-dice", LO_DICE}, {"tanimoto", LO_TANIMOTO},
    };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; ++i)
        if (strcmp(b, tab[i].name) == 0) return tab[i].m;
    return -1;
}

int lo_metric_from_index_mode(const char *s) { /* distance/mod.rs:67-107 (in-scope tokens) */
    char b[128];
    size_t i = 0;
    for (; s[i] && i + 1 < sizeof b; ++i) b[i] = (char)toupper((unsigned char)s[i]);
    b[i] = 0;
    int has_tanimoto = 0, has_jaccard = 0, has_hamming = 0, has_dice = 0, has_l2 = 0, has_cos = 0,
        has_ip = 0, has_other = 0;
    char *save = NULL;
    for (char *tok = strtok_r(b, "-", &save); tok; tok = strtok_r(NULL, "-", &save)) {
        if (!strcmp(tok, "TANIMOTO")) has_tanimoto = 1;
        else if (!strcmp(tok, "JACCARD")) has_jaccard = 1;
        else if (!strcmp(tok, "HAMMING")) has_hamming = 1;
        else if (!strcmp(tok, "DICE") || !strcmp(tok, "SORENSEN")) has_dice = 1;
        else if (!strcmp(tok, "L2") || !strcmp(tok, "L2SQ")) has_l2 = 1;
        else if (!strcmp(tok, "COS") || !strcmp(tok, "COSINE")) has_cos = 1;
        else if (!strcmp(tok, "IP")) has_ip = 1;
        else if (!strcmp(tok, "JENSENSHANNON") || !strcmp(tok, "JS") || !strcmp(tok, "CHEBYSHEV") ||
                 !strcmp(tok, "CHEBYCHEV") || !strcmp(tok, "LINF") || !strcmp(tok, "CANBERRA") ||
                 !strcmp(tok, "BRAYCURTIS") || !strcmp(tok, "HAVERSINE") || !strcmp(tok, "GEO") ||
                 !strcmp(tok, "CORRELATION") || !strcmp(tok, "PEARSON") ||
                 !strcmp(tok, "HELLINGER") || !strcmp(tok, "WASSERSTEIN") ||
                 !strcmp(tok, "WASSERSTEIN1D") || !strcmp(tok, "EMD") || !strcmp(tok, "L1") ||
                 !strcmp(tok, "MANHATTAN") || !strcmp(tok, "CITYBLOCK"))
            has_other = 1; /* out-of-scope metric families take precedence in the reference */
    }
    if (has_other) return -1;
    if (has_tanimoto) return LO_TANIMOTO;
    if (has_jaccard) return LO_JACCARD;
    if (has_hamming) return LO_HAMMING;
    if (has_dice) return LO_DICE;
    if (has_l2) return LO_L2;
    if (has_cos) return LO_COS;
    if (has_ip) return LO_IP;
    return -1;
}

/* --------------------------------------------------------- distance kernels */

#if LO_AVX2
/* Horizontal sum exactly as the reference: lo128+hi128, + movehdup, + movehl. */
static inline float hsum256(__m256 acc) {
    __m128 hi = _mm256_extractf128_ps(acc, 1);
    __m128 lo = _mm256_castps256_ps128(acc);
    __m128 sum128 = _mm_add_ps(lo, hi);
    __m128 shuf = _mm_movehdup_ps(sum128);
    __m128 sums = _mm_add_ps(sum128, shuf);
    __m128 shuf2 = _mm_movehl_ps(sums, sums);
    __m128 result = _mm_add_ss(sums, shuf2);
    return _mm_cvtss_f32(result);
}
#else
/* Portable emulation of one 8-lane accumulator. */
typedef struct { float l[8]; } lanes8;
static inline lanes8 lanes_zero(void) { lanes8 z; memset(&z, 0, sizeof z); return z; }
static inline float hsum_lanes(lanes8 a) {
    float t0 = a.l[0] + a.l[4], t1 = a.l[1] + a.l[5], t2 = a.l[2] + a.l[6], t3 = a.l[3] + a.l[7];
    return (t0 + t1) + (t2 + t3);
}
#endif

/* simd.rs:1343-1396 */
float lo_ip_single(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8, rem = n % 8;
    size_t dbl = chunks / 2, single = chunks % 2;
    float sum;
#if LO_AVX2
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    for (size_t i = 0; i < dbl; ++i) {
        size_t base = i * 16;
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + base), _mm256_loadu_ps(b + base), acc0);
        acc1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + base + 8), _mm256_loadu_ps(b + base + 8), acc1);
    }
    if (single) {
        size_t base = dbl * 16;
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + base), _mm256_loadu_ps(b + base), acc0);
    }
    acc0 = _mm256_add_ps(acc0, acc1);
    sum = hsum256(acc0);
#else
    lanes8 acc0 = lanes_zero(), acc1 = lanes_zero();
    for (size_t i = 0; i < dbl; ++i) {
        size_t base = i * 16;
        for (int j = 0; j < 8; ++j) acc0.l[j] = fmaf(a[base + j], b[base + j], acc0.l[j]);
        for (int j = 0; j < 8; ++j) acc1.l[j] = fmaf(a[base + 8 + j], b[base + 8 + j], acc1.l[j]);
    }
    if (single) {
        size_t base = dbl * 16;
        for (int j = 0; j < 8; ++j) acc0.l[j] = fmaf(a[base + j], b[base + j], acc0.l[j]);
    }
    for (int j = 0; j < 8; ++j) acc0.l[j] = acc0.l[j] + acc1.l[j];
    sum = hsum_lanes(acc0);
#endif
    size_t base = chunks * 8;
    for (size_t i = 0; i < rem; ++i) {
        float p = a[base + i] * b[base + i]; /* separate mul + add: Rust never contracts */
        sum = sum + p;
    }
    return sum;
}

/* One row of simd.rs:1452-1525: a single 8-lane accumulator, 8 elements per step. */
float lo_ip_batch8_row(const float *q, const float *v, size_t n) {
    size_t chunks = n / 8, rem = n % 8;
    float sum;
#if LO_AVX2
    __m256 acc = _mm256_setzero_ps();
    for (size_t i = 0; i < chunks; ++i)
        acc = _mm256_fmadd_ps(_mm256_loadu_ps(q + i * 8), _mm256_loadu_ps(v + i * 8), acc);
    sum = hsum256(acc);
#else
    lanes8 acc = lanes_zero();
    for (size_t i = 0; i < chunks; ++i)
        for (int j = 0; j < 8; ++j) acc.l[j] = fmaf(q[i * 8 + j], v[i * 8 + j], acc.l[j]);
    sum = hsum_lanes(acc);
#endif
    size_t base = chunks * 8;
    for (size_t i = 0; i < rem; ++i) {
        float p = q[base + i] * v[base + i];
        sum = sum + p;
    }
    return sum;
}

/* simd.rs:1452-1525 — eight rows sharing each query load. */
void lo_ip_batch8(const float *q, const float *v0, const float *v1, const float *v2,
                  const float *v3, const float *v4, const float *v5, const float *v6,
                  const float *v7, size_t n, float out[8]) {
    const float *v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#if LO_AVX2
    size_t chunks = n / 8, rem = n % 8;
    __m256 acc[8];
    for (int r = 0; r < 8; ++r) acc[r] = _mm256_setzero_ps();
    for (size_t i = 0; i < chunks; ++i) {
        __m256 qv = _mm256_loadu_ps(q + i * 8);
        for (int r = 0; r < 8; ++r)
            acc[r] = _mm256_fmadd_ps(qv, _mm256_loadu_ps(v[r] + i * 8), acc[r]);
    }
    for (int r = 0; r < 8; ++r) out[r] = hsum256(acc[r]);
    size_t base = chunks * 8;
    for (size_t i = 0; i < rem; ++i) {
        float qq = q[base + i];
        for (int r = 0; r < 8; ++r) {
            float p = qq * v[r][base + i];
            out[r] = out[r] + p;
        }
    }
#else
    for (int r = 0; r < 8; ++r) out[r] = lo_ip_batch8_row(q, v[r], n);
#endif
}

/* simd.rs:1529-1581 */
float lo_l2_single(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8, rem = n % 8;
    size_t dbl = chunks / 2, single = chunks % 2;
    float sum;
#if LO_AVX2
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    for (size_t i = 0; i < dbl; ++i) {
        size_t base = i * 16;
        __m256 d0 = _mm256_sub_ps(_mm256_loadu_ps(a + base), _mm256_loadu_ps(b + base));
        acc0 = _mm256_fmadd_ps(d0, d0, acc0);
        __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(a + base + 8), _mm256_loadu_ps(b + base + 8));
        acc1 = _mm256_fmadd_ps(d1, d1, acc1);
    }
    if (single) {
        size_t base = dbl * 16;
        __m256 d = _mm256_sub_ps(_mm256_loadu_ps(a + base), _mm256_loadu_ps(b + base));
        acc0 = _mm256_fmadd_ps(d, d, acc0);
    }
    acc0 = _mm256_add_ps(acc0, acc1);
    sum = hsum256(acc0);
#else
    lanes8 acc0 = lanes_zero(), acc1 = lanes_zero();
    for (size_t i = 0; i < dbl; ++i) {
        size_t base = i * 16;
        for (int j = 0; j < 8; ++j) {
            float d = a[base + j] - b[base + j];
            acc0.l[j] = fmaf(d, d, acc0.l[j]);
        }
        for (int j = 0; j < 8; ++j) {
            float d = a[base + 8 + j] - b[base + 8 + j];
            acc1.l[j] = fmaf(d, d, acc1.l[j]);
        }
    }
    if (single) {
        size_t base = dbl * 16;
        for (int j = 0; j < 8; ++j) {
            float d = a[base + j] - b[base + j];
            acc0.l[j] = fmaf(d, d, acc0.l[j]);
        }
    }
    for (int j = 0; j < 8; ++j) acc0.l[j] = acc0.l[j] + acc1.l[j];
    sum = hsum_lanes(acc0);
#endif
    size_t base = chunks * 8;
    for (size_t i = 0; i < rem; ++i) {
        float d = a[base + i] - b[base + i];
        float p = d * d;
        sum = sum + p;
    }
    return sum;
}

/* simd.rs:1585-1636 */
float lo_cos_single(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8, rem = n % 8;
    float dot, na, nb;
#if LO_AVX2
    __m256 d = _mm256_setzero_ps(), x = _mm256_setzero_ps(), y = _mm256_setzero_ps();
    for (size_t i = 0; i < chunks; ++i) {
        __m256 va = _mm256_loadu_ps(a + i * 8), vb = _mm256_loadu_ps(b + i * 8);
        d = _mm256_fmadd_ps(va, vb, d);
        x = _mm256_fmadd_ps(va, va, x);
        y = _mm256_fmadd_ps(vb, vb, y);
    }
    dot = hsum256(d);
    na = hsum256(x);
    nb = hsum256(y);
#else
    lanes8 d = lanes_zero(), x = lanes_zero(), y = lanes_zero();
    for (size_t i = 0; i < chunks; ++i)
        for (int j = 0; j < 8; ++j) {
            float va = a[i * 8 + j], vb = b[i * 8 + j];
            d.l[j] = fmaf(va, vb, d.l[j]);
            x.l[j] = fmaf(va, va, x.l[j]);
            y.l[j] = fmaf(vb, vb, y.l[j]);
        }
    dot = hsum_lanes(d);
    na = hsum_lanes(x);
    nb = hsum_lanes(y);
#endif
    size_t base = chunks * 8;
    for (size_t i = 0; i < rem; ++i) {
        float p0 = a[base + i] * b[base + i];
        dot = dot + p0;
        float p1 = a[base + i] * a[base + i];
        na = na + p1;
        float p2 = b[base + i] * b[base + i];
        nb = nb + p2;
    }
    float prod = na * nb;
    float denom = sqrtf(prod);
    if (denom < 1e-30f) return 1.0f;
    float ratio = dot / denom;
    return 1.0f - ratio;
}

/* simd.rs:1223-1250 — scalar fallback, f64 accumulation in 8-chunks. */
float lo_ip_scalar(const float *a, const float *b, size_t n) {
    double sum = 0.0;
    size_t chunks = n / 8, rem = n % 8;
    for (size_t i = 0; i < chunks; ++i) {
        double local = 0.0;
        for (int j = 0; j < 8; ++j) local += (double)a[i * 8 + j] * (double)b[i * 8 + j];
        sum += local;
    }
    for (size_t i = 0; i < rem; ++i) sum += (double)a[chunks * 8 + i] * (double)b[chunks * 8 + i];
    return (float)sum;
}

/* simd.rs:1296-1317 */
float lo_l2_scalar(const float *a, const float *b, size_t n) {
    double sum = 0.0;
    size_t chunks = n / 8, rem = n % 8;
    for (size_t i = 0; i < chunks; ++i) {
        double local = 0.0;
        for (int j = 0; j < 8; ++j) {
            double d = (double)(a[i * 8 + j] - b[i * 8 + j]);
            local += d * d;
        }
        sum += local;
    }
    for (size_t i = 0; i < rem; ++i) {
        double d = (double)(a[chunks * 8 + i] - b[chunks * 8 + i]);
        sum += d * d;
    }
    return (float)sum;
}

/* simd.rs:1319-1337 */
float lo_cos_scalar(const float *a, const float *b, size_t n) {
    double dot = 0.0, na = 0.0, nb = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double ai = a[i], bi = b[i];
        dot += ai * bi;
        na += ai * ai;
        nb += bi * bi;
    }
    double denom = sqrt(na * nb);
    if (denom < 1e-30) return 1.0f;
    return 1.0f - (float)(dot / denom);
}

/* simd.rs:175-187 */
float lo_hamming_f32(const float *a, const float *b, size_t n) {
    uint32_t count = 0;
    for (size_t i = 0; i < n; ++i) count += ((a[i] > 0.5f) != (b[i] > 0.5f));
    return (float)count;
}

/* simd.rs:190-209 */
float lo_jaccard_f32(const float *a, const float *b, size_t n) {
    uint32_t inter = 0, uni = 0;
    for (size_t i = 0; i < n; ++i) {
        int ab = a[i] > 0.5f, bb = b[i] > 0.5f;
        if (ab || bb) {
            uni += 1;
            if (ab && bb) inter += 1;
        }
    }
    if (uni == 0) return 0.0f;
    return 1.0f - ((float)inter / (float)uni);
}

/* simd.rs:717-736 */
float lo_dice_f32(const float *a, const float *b, size_t n) {
    uint32_t inter = 0, ca = 0, cb = 0;
    for (size_t i = 0; i < n; ++i) {
        int ab = a[i] > 0.5f, bb = b[i] > 0.5f;
        ca += ab;
        cb += bb;
        inter += (ab && bb);
    }
    uint32_t total = ca + cb;
    if (total == 0) return 0.0f;
    return 1.0f - (float)(2 * inter) / (float)total;
}

/* simd.rs:750-763 (threshold 0.5), identical to flat_mmap.rs:1284-1290 */
void lo_pack_binary_f32(const float *src, size_t dim, uint64_t *words) {
    size_t w = (dim + 63) / 64;
    memset(words, 0, w * sizeof(uint64_t));
    for (size_t i = 0; i < dim; ++i)
        if (src[i] > 0.5f) words[i / 64] |= (uint64_t)1 << (i % 64);
}

/* simd.rs:766-771, flat_mmap.rs:1298-1304 */
float lo_packed_hamming(const uint64_t *a, const uint64_t *b, size_t words) {
    uint32_t s = 0;
    for (size_t i = 0; i < words; ++i) s += (uint32_t)__builtin_popcountll(a[i] ^ b[i]);
    return (float)s;
}

/* simd.rs:774-786, flat_mmap.rs:1306-1319 */
float lo_packed_jaccard(const uint64_t *a, const uint64_t *b, size_t words) {
    uint32_t inter = 0, uni = 0;
    for (size_t i = 0; i < words; ++i) {
        inter += (uint32_t)__builtin_popcountll(a[i] & b[i]);
        uni += (uint32_t)__builtin_popcountll(a[i] | b[i]);
    }
    if (uni == 0) return 0.0f;
    return 1.0f - (float)inter / (float)uni;
}

/* simd.rs:789-801, flat_mmap.rs:1321-1334 */
float lo_packed_dice(const uint64_t *a, const uint64_t *b, size_t words) {
    uint32_t inter = 0, count = 0;
    for (size_t i = 0; i < words; ++i) {
        inter += (uint32_t)__builtin_popcountll(a[i] & b[i]);
        count += (uint32_t)__builtin_popcountll(a[i]) + (uint32_t)__builtin_popcountll(b[i]);
    }
    if (count == 0) return 0.0f;
    return 1.0f - (float)(2 * inter) / (float)count;
}

/* distance/mod.rs:193-213 */
float lo_compute_distance(const float *a, const float *b, size_t n, int metric) {
    switch (metric) {
    case LO_IP: return lo_ip_single(a, b, n);
    case LO_L2: return lo_l2_single(a, b, n);
    case LO_COS: return lo_cos_single(a, b, n);
    case LO_HAMMING: return lo_hamming_f32(a, b, n);
    case LO_JACCARD: case LO_TANIMOTO: return lo_jaccard_f32(a, b, n);
    case LO_DICE: return lo_dice_f32(a, b, n);
    default: return NAN;
    }
}

typedef float (*packed_fn)(const uint64_t *, const uint64_t *, size_t);
static packed_fn packed_distance_fn(int metric) { /* flat_mmap.rs:1336-1343 */
    switch (metric) {
    case LO_HAMMING: return lo_packed_hamming;
    case LO_JACCARD: case LO_TANIMOTO: return lo_packed_jaccard;
    case LO_DICE: return lo_packed_dice;
    default: return lo_packed_hamming;
    }
}

/* -------------------------------------------------------------------- top-k */

typedef struct { float dist; uint32_t idx; } entry_t;

/* partial_cmp(..).unwrap_or(Equal): NaN compares Equal. Returns <0,0,>0. */
static inline int fcmp(float a, float b) { return (a < b) ? -1 : (a > b) ? 1 : 0; }

/* Stable insertion sort by distance only — stands in for sort_unstable_by at
 * flat_mmap.rs:2141-2149 (Rust's sort_unstable is an insertion sort for short
 * slices; tie order beyond that is not pinned by the reference). */
static void sort_entries(entry_t *e, size_t n, int asc) {
    for (size_t i = 1; i < n; ++i) {
        entry_t x = e[i];
        size_t j = i;
        while (j > 0) {
            int c = asc ? fcmp(x.dist, e[j - 1].dist) : fcmp(e[j - 1].dist, x.dist);
            if (c < 0) { e[j] = e[j - 1]; --j; } else break;
        }
        e[j] = x;
    }
}

typedef struct {
    entry_t *top;
    size_t len, k;
    float threshold;
    int filled, asc;
} topk_t;

static void topk_init(topk_t *t, entry_t *buf, size_t k, int asc) {
    t->top = buf; t->len = 0; t->k = k; t->asc = asc; t->filled = 0;
    t->threshold = asc ? INFINITY : -INFINITY;
}

/* flat_mmap.rs:2132-2166 */
static void topk_insert(topk_t *t, entry_t e) {
    if (!t->filled) {
        t->top[t->len++] = e;
        if (t->len == t->k) {
            sort_entries(t->top, t->len, t->asc);
            t->threshold = t->top[t->k - 1].dist;
            t->filled = 1;
        }
    } else {
        size_t k = t->k;
        t->top[k - 1] = e;
        size_t j = k - 1;
        if (t->asc) {
            while (j > 0 && t->top[j].dist < t->top[j - 1].dist) {
                entry_t tmp = t->top[j]; t->top[j] = t->top[j - 1]; t->top[j - 1] = tmp; --j;
            }
        } else {
            while (j > 0 && t->top[j].dist > t->top[j - 1].dist) {
                entry_t tmp = t->top[j]; t->top[j] = t->top[j - 1]; t->top[j - 1] = tmp; --j;
            }
        }
        t->threshold = t->top[k - 1].dist;
    }
}

/* flat_mmap.rs:2170-2176 */
static inline int passes(const topk_t *t, float d) {
    return t->asc ? (d < t->threshold) : (d > t->threshold);
}

static inline void topk_offer(topk_t *t, float d, uint32_t idx) {
    if (!t->filled || passes(t, d)) {
        entry_t e = {d, idx};
        topk_insert(t, e);
    }
}

static void topk_finish(topk_t *t) { /* flat_mmap.rs:5033-5039 */
    if (!t->filled && t->len > 0) sort_entries(t->top, t->len, t->asc);
}

typedef float (*dist_fn)(const float *, const float *, size_t);

static dist_fn float_dist_fn(int metric) {
    switch (metric) {
    case LO_IP: return lo_ip_single;
    case LO_L2: return lo_l2_single;
    case LO_COS: return lo_cos_single;
    case LO_HAMMING: return lo_hamming_f32;
    case LO_JACCARD: case LO_TANIMOTO: return lo_jaccard_f32;
    case LO_DICE: return lo_dice_f32;
    default: return lo_ip_single;
    }
}

/* flat_mmap.rs:4985-5044 */
static size_t fused_topk_seq(const float *q, const float *c, size_t dim, size_t n, size_t k,
                             int asc, dist_fn f, size_t base_idx, entry_t *out) {
    topk_t t;
    topk_init(&t, out, k, asc);
    for (size_t i = 0; i < n; ++i) topk_offer(&t, f(q, c + i * dim, dim), (uint32_t)(base_idx + i));
    topk_finish(&t);
    return t.len;
}

/* flat_mmap.rs:2179-2256 */
static size_t ip_scan_chunk_topk(const float *q, const float *chunk, size_t dim, size_t n_in_chunk,
                                 size_t k, size_t base_idx, entry_t *out) {
    topk_t t;
    topk_init(&t, out, k, 0);
    size_t blocks8 = n_in_chunk / 8;
    for (size_t blk = 0; blk < blocks8; ++blk) {
        const float *b = chunk + blk * 8 * dim;
        float d[8];
        lo_ip_batch8(q, b, b + dim, b + 2 * dim, b + 3 * dim, b + 4 * dim, b + 5 * dim,
                     b + 6 * dim, b + 7 * dim, dim, d);
        for (int j = 0; j < 8; ++j) topk_offer(&t, d[j], (uint32_t)(base_idx + blk * 8 + j));
    }
    for (size_t i = blocks8 * 8; i < n_in_chunk; ++i)
        topk_offer(&t, lo_ip_single(q, chunk + i * dim, dim), (uint32_t)(base_idx + i));
    topk_finish(&t);
    return t.len;
}

/* One chunk of fused_topk_parallel (flat_mmap.rs:4890-4975): rows in pairs, odd remainder. */
static size_t generic_scan_chunk_topk(const float *q, const float *chunk, size_t dim,
                                      size_t n_in_chunk, size_t k, int asc, dist_fn f,
                                      size_t base_idx, entry_t *out) {
    topk_t t;
    topk_init(&t, out, k, asc);
    size_t pairs = n_in_chunk / 2;
    for (size_t i = 0; i < pairs; ++i) {
        float d0 = f(q, chunk + (2 * i) * dim, dim);
        float d1 = f(q, chunk + (2 * i + 1) * dim, dim);
        topk_offer(&t, d0, (uint32_t)(base_idx + 2 * i));
        topk_offer(&t, d1, (uint32_t)(base_idx + 2 * i + 1));
    }
    if (n_in_chunk % 2 == 1)
        topk_offer(&t, f(q, chunk + (n_in_chunk - 1) * dim, dim),
                   (uint32_t)(base_idx + n_in_chunk - 1));
    topk_finish(&t);
    return t.len;
}

/* flat_mmap.rs:5183-5214 */
static size_t merge_topk_results(entry_t *const *chunks, const size_t *lens, size_t n_chunks,
                                 size_t k, int asc, entry_t *out) {
    topk_t t;
    topk_init(&t, out, k, asc);
    for (size_t c = 0; c < n_chunks; ++c)
        for (size_t i = 0; i < lens[c]; ++i) topk_offer(&t, chunks[c][i].dist, chunks[c][i].idx);
    topk_finish(&t);
    return t.len;
}

typedef struct {
    /* shared scan description */
    const float *q; const float *c; size_t dim, n, k; int metric, asc;
    const uint64_t *pq; const uint64_t *prow; size_t words; /* packed variant */
    size_t chunk_rows, n_chunks;
    entry_t **chunk_out; size_t *chunk_len;
    volatile size_t next; /* chunk dispenser for the mt variant */
} scan_t;

static void scan_one_chunk(scan_t *s, size_t ci) {
    size_t start = ci * s->chunk_rows;
    size_t rows = s->n - start < s->chunk_rows ? s->n - start : s->chunk_rows;
    if (s->prow) { /* flat_mmap.rs:1381-1406 */
        packed_fn f = packed_distance_fn(s->metric);
        topk_t t;
        topk_init(&t, s->chunk_out[ci], s->k, 1);
        for (size_t i = 0; i < rows; ++i)
            topk_offer(&t, f(s->pq, s->prow + (start + i) * s->words, s->words),
                       (uint32_t)(start + i));
        /* the packed path does not sort under-full chunks (flat_mmap.rs:1404) */
        s->chunk_len[ci] = t.len;
    } else if (s->metric == LO_IP) {
        s->chunk_len[ci] = ip_scan_chunk_topk(s->q, s->c + start * s->dim, s->dim, rows, s->k,
                                              start, s->chunk_out[ci]);
    } else {
        s->chunk_len[ci] = generic_scan_chunk_topk(s->q, s->c + start * s->dim, s->dim, rows, s->k,
                                                   s->asc, float_dist_fn(s->metric), start,
                                                   s->chunk_out[ci]);
    }
}

static void *scan_worker(void *arg) {
    scan_t *s = (scan_t *)arg;
    for (;;) {
        size_t ci = __atomic_fetch_add(&s->next, 1, __ATOMIC_RELAXED);
        if (ci >= s->n_chunks) break;
        scan_one_chunk(s, ci);
    }
    return NULL;
}

/* ---- persistent worker pool (the timed CPU baseline only) ----
 * The reference scans on rayon's GLOBAL pool (RAYON_NUM_THREADS; scripts/perf_gate_local.py:218 defaults it to 4): threads
 * are created once, not per query.  lo_pool_start(n) creates n workers pinned to cpus 0..n-1 and parked on a condition
 * variable; while the pool runs, the *_mt entry points hand chunk ci to worker ci % n (static, so the worker that
 * first-touched a chunk of the sample in lo_fill_uniform_mt is the one that scans it).  Results do not depend on the
 * pool: chunking and merge order are those of run_chunked either way. */
static struct {
    pthread_t *th; int n; int started;
    pthread_mutex_t mu; pthread_cond_t cv_go, cv_done;
    unsigned long gen; int pending, stop;
    void (*fn)(void *arg, int worker, int n_workers); void *arg;
} g_pool = {0};

typedef struct { int id; } pool_worker_t;

static void *pool_main(void *p) {
    const int id = ((pool_worker_t *)p)->id;
    free(p);
    unsigned long seen = 0;
    for (;;) {
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.gen == seen && !g_pool.stop) pthread_cond_wait(&g_pool.cv_go, &g_pool.mu);
        if (g_pool.stop) { pthread_mutex_unlock(&g_pool.mu); return NULL; }
        seen = g_pool.gen;
        void (*fn)(void *, int, int) = g_pool.fn;
        void *arg = g_pool.arg;
        const int n = g_pool.n;
        pthread_mutex_unlock(&g_pool.mu);
        fn(arg, id, n);
        pthread_mutex_lock(&g_pool.mu);
        if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.cv_done);
        pthread_mutex_unlock(&g_pool.mu);
    }
}

void lo_pool_stop(void) {
    if (!g_pool.started) return;
    pthread_mutex_lock(&g_pool.mu);
    g_pool.stop = 1;
    pthread_cond_broadcast(&g_pool.cv_go);
    pthread_mutex_unlock(&g_pool.mu);
    for (int i = 0; i < g_pool.n; ++i) pthread_join(g_pool.th[i], NULL);
    free(g_pool.th);
    pthread_mutex_destroy(&g_pool.mu); pthread_cond_destroy(&g_pool.cv_go); pthread_cond_destroy(&g_pool.cv_done);
    memset(&g_pool, 0, sizeof g_pool);
}

int lo_pool_start(int n_threads) {
    lo_pool_stop();
    if (n_threads < 1) return 0;
    pthread_mutex_init(&g_pool.mu, NULL); pthread_cond_init(&g_pool.cv_go, NULL); pthread_cond_init(&g_pool.cv_done, NULL);
    g_pool.th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    g_pool.n = n_threads;
    /* the cpus this process may run on (a container's cpuset can be narrower than the machine) */
    cpu_set_t allowed;
    int cpus[CPU_SETSIZE], ncpu = 0;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &allowed)) cpus[ncpu++] = c;
    for (int i = 0; i < n_threads; ++i) {
        pool_worker_t *w = (pool_worker_t *)malloc(sizeof *w);
        w->id = i;
        pthread_create(&g_pool.th[i], NULL, pool_main, w);
        if (ncpu > 0) {  /* best effort: keeps a worker (and the pages it first-touched) on one NUMA node */
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(cpus[i % ncpu], &set);
            (void)pthread_setaffinity_np(g_pool.th[i], sizeof set, &set);
        }
    }
    g_pool.started = 1;
    return n_threads;
}

static void pool_run(void (*fn)(void *, int, int), void *arg) {
    pthread_mutex_lock(&g_pool.mu);
    g_pool.fn = fn; g_pool.arg = arg; g_pool.pending = g_pool.n; g_pool.gen += 1;
    pthread_cond_broadcast(&g_pool.cv_go);
    while (g_pool.pending) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
}

static void pool_scan(void *arg, int worker, int n_workers) {
    scan_t *s = (scan_t *)arg;
    for (size_t ci = (size_t)worker; ci < s->n_chunks; ci += (size_t)n_workers) scan_one_chunk(s, ci);
}

typedef struct { float *dst; size_t n, dim, chunk_rows; uint64_t seed; } fill_t;

static void pool_fill(void *arg, int worker, int n_workers) {
    fill_t *f = (fill_t *)arg;
    const size_t n_chunks = (f->n + f->chunk_rows - 1) / f->chunk_rows;
    for (size_t ci = (size_t)worker; ci < n_chunks; ci += (size_t)n_workers) {
        const size_t r0 = ci * f->chunk_rows, r1 = r0 + f->chunk_rows < f->n ? r0 + f->chunk_rows : f->n;
        uint64_t x = f->seed * 0x9E3779B97F4A7C15ull + (uint64_t)ci * 0xD1B54A32D192ED03ull + 1;
        float *p = f->dst + r0 * f->dim;
        for (size_t i = 0, m = (r1 - r0) * f->dim; i < m; ++i) {  /* xorshift64*: uniform [0,1) with 24 random bits */
            x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
            p[i] = (float)((x * 0x2545F4914F6CDD1Dull) >> 40) * (1.0f / 16777216.0f);
        }
    }
}

/* Fill an UNTOUCHED buffer with uniform [0,1) rows using the pool, chunked exactly like the scan (n / threads rows, at
 * least 512): every page is first touched by the worker that will scan it.  Needs a running pool. */
int lo_fill_uniform_mt(float *dst, size_t n, size_t dim, uint64_t seed) {
    if (!g_pool.started || !dst) return -1;
    fill_t f;
    f.dst = dst; f.n = n; f.dim = dim; f.seed = seed;
    f.chunk_rows = n / (size_t)g_pool.n;
    if (f.chunk_rows < 512) f.chunk_rows = 512;
    pool_run(pool_fill, &f);
    return 0;
}

static size_t run_chunked(scan_t *s, int n_threads, int real_threads, uint32_t *out_idx,
                          float *out_dist) {
    size_t min_chunk = s->prow ? 1024 : 512; /* flat_mmap.rs:1379 / :4857 */
    size_t t = n_threads > 0 ? (size_t)n_threads : 1;
    s->chunk_rows = s->n / t;
    if (s->chunk_rows < min_chunk) s->chunk_rows = min_chunk;
    s->n_chunks = (s->n + s->chunk_rows - 1) / s->chunk_rows;
    s->chunk_out = (entry_t **)malloc(s->n_chunks * sizeof(entry_t *));
    s->chunk_len = (size_t *)calloc(s->n_chunks, sizeof(size_t));
    entry_t *pool = (entry_t *)malloc(s->n_chunks * s->k * sizeof(entry_t));
    for (size_t i = 0; i < s->n_chunks; ++i) s->chunk_out[i] = pool + i * s->k;
    s->next = 0;
    if (real_threads && t > 1 && g_pool.started && (size_t)g_pool.n == t) {
        pool_run(pool_scan, s);  /* persistent pool (timed baseline) */
    } else if (real_threads && t > 1) {
        pthread_t *th = (pthread_t *)malloc(t * sizeof(pthread_t));
        for (size_t i = 0; i < t; ++i) pthread_create(&th[i], NULL, scan_worker, s);
        for (size_t i = 0; i < t; ++i) pthread_join(th[i], NULL);
        free(th);
    } else {
        for (size_t i = 0; i < s->n_chunks; ++i) scan_one_chunk(s, i);
    }
    entry_t *merged = (entry_t *)malloc(s->k * sizeof(entry_t));
    size_t len = merge_topk_results(s->chunk_out, s->chunk_len, s->n_chunks, s->k, s->asc, merged);
    for (size_t i = 0; i < len; ++i) { out_idx[i] = merged[i].idx; out_dist[i] = merged[i].dist; }
    free(merged); free(pool); free(s->chunk_len); free(s->chunk_out);
    return len;
}

static size_t packed_search_impl(const uint64_t *query, const uint64_t *rows, size_t words, size_t n,
                                 size_t k, int metric, int n_threads, int real_threads,
                                 uint32_t *out_idx, float *out_dist) {
    if (n == 0 || k == 0) return 0;
    if (k > n) k = n;
    if (n < 4096) { /* flat_mmap.rs:1354-1375 */
        packed_fn f = packed_distance_fn(metric);
        entry_t *buf = (entry_t *)malloc(k * sizeof(entry_t));
        topk_t t;
        topk_init(&t, buf, k, 1);
        for (size_t i = 0; i < n; ++i) topk_offer(&t, f(query, rows + i * words, words), (uint32_t)i);
        for (size_t i = 0; i < t.len; ++i) { out_idx[i] = buf[i].idx; out_dist[i] = buf[i].dist; }
        size_t len = t.len;
        free(buf);
        return len;
    }
    scan_t s;
    memset(&s, 0, sizeof s);
    s.pq = query; s.prow = rows; s.words = words; s.n = n; s.k = k; s.metric = metric; s.asc = 1;
    return run_chunked(&s, n_threads, real_threads, out_idx, out_dist);
}

size_t lo_packed_binary_search(const uint64_t *query, const uint64_t *rows, size_t words, size_t n,
                               size_t k, int metric, int n_threads, uint32_t *out_idx,
                               float *out_dist) {
    return packed_search_impl(query, rows, words, n, k, metric, n_threads, 0, out_idx, out_dist);
}

size_t lo_packed_binary_search_mt(const uint64_t *query, const uint64_t *rows, size_t words,
                                  size_t n, size_t k, int metric, int n_threads,
                                  uint32_t *out_idx, float *out_dist) {
    return packed_search_impl(query, rows, words, n, k, metric, n_threads, 1, out_idx, out_dist);
}

static size_t flat_search_impl(const float *query, const float *cands, size_t dim, size_t n,
                               size_t k, int metric, int n_threads, int real_threads,
                               uint32_t *out_idx, float *out_dist) {
    if (n == 0 || k == 0) return 0; /* flat_mmap.rs:832-835 */
    if (k > n) k = n;               /* :836 */
    if (lo_metric_is_binary(metric)) { /* :839-845 — ensure_binary + pack_binary_query */
        size_t words = (dim + 63) / 64;
        uint64_t *packed = (uint64_t *)malloc(n * words * sizeof(uint64_t));
        uint64_t *pq = (uint64_t *)malloc(words * sizeof(uint64_t));
        for (size_t i = 0; i < n; ++i) lo_pack_binary_f32(cands + i * dim, dim, packed + i * words);
        lo_pack_binary_f32(query, dim, pq);
        size_t len = packed_search_impl(pq, packed, words, n, k, metric, n_threads, real_threads,
                                        out_idx, out_dist);
        free(pq); free(packed);
        return len;
    }
    int asc = lo_metric_is_ascending(metric);
    if (n < 4096) { /* :4852-4854 / :4884-4886 */
        entry_t *buf = (entry_t *)malloc(k * sizeof(entry_t));
        size_t len = fused_topk_seq(query, cands, dim, n, k, asc, float_dist_fn(metric), 0, buf);
        for (size_t i = 0; i < len; ++i) { out_idx[i] = buf[i].idx; out_dist[i] = buf[i].dist; }
        free(buf);
        return len;
    }
    scan_t s;
    memset(&s, 0, sizeof s);
    s.q = query; s.c = cands; s.dim = dim; s.n = n; s.k = k; s.metric = metric; s.asc = asc;
    return run_chunked(&s, n_threads, real_threads, out_idx, out_dist);
}

size_t lo_flat_search(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                      int metric, int n_threads, uint32_t *out_idx, float *out_dist) {
    return flat_search_impl(query, cands, dim, n, k, metric, n_threads, 0, out_idx, out_dist);
}

size_t lo_flat_search_mt(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                         int metric, int n_threads, uint32_t *out_idx, float *out_dist) {
    return flat_search_impl(query, cands, dim, n, k, metric, n_threads, 1, out_idx, out_dist);
}

/* ---- distance::top_k_search (distance/mod.rs:304-422) ---- */

typedef struct { float d; uint32_t i; } pair_t;

static inline int cmp_pair(pair_t a, pair_t b, int asc) { /* :356-362 */
    return asc ? fcmp(a.d, b.d) : fcmp(b.d, a.d);
}

static void quickselect_k(pair_t *arr, size_t n, size_t k, int asc) { /* :304-352 */
    if (n <= k || k == 0) return;
    size_t target = k - 1, lo = 0, hi = n - 1;
    while (lo < hi) {
        if (hi - lo >= 2) {
            size_t mid = lo + (hi - lo) / 2;
            pair_t t;
            if (cmp_pair(arr[lo], arr[mid], asc) > 0) { t = arr[lo]; arr[lo] = arr[mid]; arr[mid] = t; }
            if (cmp_pair(arr[lo], arr[hi], asc) > 0) { t = arr[lo]; arr[lo] = arr[hi]; arr[hi] = t; }
            if (cmp_pair(arr[mid], arr[hi], asc) > 0) { t = arr[mid]; arr[mid] = arr[hi]; arr[hi] = t; }
            t = arr[mid]; arr[mid] = arr[hi]; arr[hi] = t;
        }
        pair_t pivot = arr[hi];
        size_t store = lo;
        for (size_t j = lo; j < hi; ++j) {
            if (cmp_pair(arr[j], pivot, asc) <= 0) {
                pair_t t = arr[store]; arr[store] = arr[j]; arr[j] = t;
                ++store;
            }
        }
        pair_t t = arr[store]; arr[store] = arr[hi]; arr[hi] = t;
        if (store == target) return;
        else if (store < target) lo = store + 1;
        else hi = store - 1;
    }
}

static void sort_pairs(pair_t *p, size_t n, int asc) { /* stands in for sort_unstable_by :408-412 */
    for (size_t i = 1; i < n; ++i) {
        pair_t x = p[i];
        size_t j = i;
        while (j > 0 && cmp_pair(x, p[j - 1], asc) < 0) { p[j] = p[j - 1]; --j; }
        p[j] = x;
    }
}

size_t lo_top_k_search(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                       int metric, uint32_t *out_idx, float *out_dist) {
    if (k > n) k = n;
    if (n == 0 || k == 0) return 0;
    int asc = lo_metric_is_ascending(metric);
    pair_t *pairs = (pair_t *)malloc(n * sizeof(pair_t));
    for (size_t i = 0; i < n; ++i) {
        pairs[i].d = lo_compute_distance(query, cands + i * dim, dim, metric);
        pairs[i].i = (uint32_t)i;
    }
    quickselect_k(pairs, n, k, asc);
    sort_pairs(pairs, k, asc);
    for (size_t i = 0; i < k; ++i) { out_idx[i] = pairs[i].i; out_dist[i] = pairs[i].d; }
    free(pairs);
    return k;
}

/* ---- canonical exact top-k: (distance in metric order, id ascending) ---- */

typedef struct { float d; uint64_t id; } cpair_t;
static int g_cmp_asc; /* qsort context (single-threaded use) */
static int cmp_canonical(const void *pa, const void *pb) {
    const cpair_t *a = (const cpair_t *)pa, *b = (const cpair_t *)pb;
    int c = g_cmp_asc ? fcmp(a->d, b->d) : fcmp(b->d, a->d);
    if (c) return c;
    return (a->id < b->id) ? -1 : (a->id > b->id) ? 1 : 0;
}

void lo_all_distances(const float *query, const float *cands, size_t dim, size_t n, int metric,
                      int ip_form, float *out) {
    int form = ip_form;
    if (form == LO_IPFORM_AUTO) form = n < 4096 ? LO_IPFORM_SINGLE : LO_IPFORM_BATCH8;
    if (lo_metric_is_binary(metric)) {
        size_t words = (dim + 63) / 64;
        uint64_t *pq = (uint64_t *)malloc(words * 8), *pr = (uint64_t *)malloc(words * 8);
        packed_fn f = packed_distance_fn(metric);
        lo_pack_binary_f32(query, dim, pq);
        for (size_t i = 0; i < n; ++i) {
            lo_pack_binary_f32(cands + i * dim, dim, pr);
            out[i] = f(pq, pr, words);
        }
        free(pq); free(pr);
        return;
    }
    for (size_t i = 0; i < n; ++i) {
        const float *v = cands + i * dim;
        if (metric == LO_IP)
            out[i] = form == LO_IPFORM_BATCH8 ? lo_ip_batch8_row(query, v, dim)
                                              : lo_ip_single(query, v, dim);
        else
            out[i] = lo_compute_distance(query, v, dim, metric);
    }
}

static size_t canonical_from_dists(const float *d, size_t n, size_t k, int metric,
                                   uint32_t *out_idx, float *out_dist) {
    if (k > n) k = n;
    if (n == 0 || k == 0) return 0;
    cpair_t *p = (cpair_t *)malloc(n * sizeof(cpair_t));
    for (size_t i = 0; i < n; ++i) { p[i].d = d[i]; p[i].id = i; }
    g_cmp_asc = lo_metric_is_ascending(metric);
    qsort(p, n, sizeof(cpair_t), cmp_canonical);
    for (size_t i = 0; i < k; ++i) { out_idx[i] = (uint32_t)p[i].id; out_dist[i] = p[i].d; }
    free(p);
    return k;
}

size_t lo_canonical_topk(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                         int metric, int ip_form, uint32_t *out_idx, float *out_dist) {
    if (n == 0 || k == 0) return 0;
    float *d = (float *)malloc(n * sizeof(float));
    lo_all_distances(query, cands, dim, n, metric, ip_form, d);
    size_t r = canonical_from_dists(d, n, k, metric, out_idx, out_dist);
    free(d);
    return r;
}

size_t lo_canonical_topk_packed(const uint64_t *query, const uint64_t *rows, size_t words,
                                size_t n, size_t k, int metric, uint32_t *out_idx,
                                float *out_dist) {
    if (n == 0 || k == 0) return 0;
    float *d = (float *)malloc(n * sizeof(float));
    packed_fn f = packed_distance_fn(metric);
    for (size_t i = 0; i < n; ++i) d[i] = f(query, rows + i * words, words);
    size_t r = canonical_from_dists(d, n, k, metric, out_idx, out_dist);
    free(d);
    return r;
}

/* ------------------------------------------------------------- f16 storage (VectorDtype::F16) */

/* simd::inner_product_f16 / l2_squared_f16 / cosine_distance_f16 (src/distance/simd.rs:805-846): plain sequential
 * f32 sums over the DECODED f16 candidate (`cand` holds f16::to_f32 values), separate multiply and add. */
float lo_distance_f16(const float *query, const float *cand, size_t dim, int metric) {
    if (metric == LO_IP) {
        float sum = 0.0f;
        for (size_t i = 0; i < dim; ++i) { float p = query[i] * cand[i]; sum = sum + p; }
        return sum;
    }
    if (metric == LO_L2) {
        float sum = 0.0f;
        for (size_t i = 0; i < dim; ++i) { float d = query[i] - cand[i]; float p = d * d; sum = sum + p; }
        return sum;
    }
    float dot = 0.0f, nq = 0.0f, nc = 0.0f;
    for (size_t i = 0; i < dim; ++i) {
        float q = query[i], c = cand[i];
        float a = q * c; dot = dot + a;
        float b = q * q; nq = nq + b;
        float e = c * c; nc = nc + e;
    }
    if (nq == 0.0f || nc == 0.0f) return 1.0f;
    float den = sqrtf(nq) * sqrtf(nc);
    return 1.0f - dot / den;
}

/* f32 -> f16 -> f32, round to nearest even: encode_f32_slice_as_le_bytes(.., F16) + f16::to_f32 (src/storage/dtype.rs) */
static uint16_t f32_to_f16_bits(float f) { /* IEEE binary16, round to nearest even (half::f16::from_f32) */
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, exp = (x >> 23) & 0xffu, man = x & 0x7fffffu;
    if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0u)); /* inf / nan */
    int e = (int)exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);                                             /* overflow -> inf */
    if (e <= 0) {                                                                                /* subnormal / zero */
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t half = man >> shift, rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) half += 1;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (man >> 13), rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half += 1; /* may carry into the exponent (-> inf): correct */
    return (uint16_t)(sign | half);
}
static float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else { int e = -1; do { ++e; man <<= 1; } while (!(man & 0x400u)); x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13); }
    } else if (exp == 31) x = sign | 0x7f800000u | (man << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}
void lo_round_f16(const float *in, size_t n, float *out) {
    for (size_t i = 0; i < n; ++i) out[i] = f16_bits_to_f32(f32_to_f16_bits(in[i]));
}
void lo_f16_bits(const float *in, size_t n, uint16_t *out) {
    for (size_t i = 0; i < n; ++i) out[i] = f32_to_f16_bits(in[i]);
}

size_t lo_canonical_topk_f16(const float *query, const float *cands_decoded, size_t dim, size_t n, size_t k,
                             int metric, uint32_t *out_idx, float *out_dist) {
    if (n == 0 || k == 0) return 0;
    float *d = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) d[i] = lo_distance_f16(query, cands_decoded + i * dim, dim, metric);
    size_t r = canonical_from_dists(d, n, k, metric, out_idx, out_dist);
    free(d);
    return r;
}

/* ------------------------------------------------------------- SQ8 two-pass FLAT (FLAT-*-SQ8) */

/* SQ8Data::from_f32_parallel (flat_mmap.rs:5685-5737): per-dimension min / scale = 255/(max-min) (0 when the range is
 * <= 1e-30), codes = clamp(round((v - min) * scale), 0, 255) with f32::round (half away from zero). */
void lo_sq8_fit(const float *data, size_t n, size_t dim, float *mins, float *scales) {
    float *maxs = (float *)malloc(dim * sizeof(float));
    for (size_t d = 0; d < dim; ++d) { mins[d] = INFINITY; maxs[d] = -INFINITY; }
    for (size_t v = 0; v < n; ++v)
        for (size_t d = 0; d < dim; ++d) {
            float val = data[v * dim + d];
            if (val < mins[d]) mins[d] = val;
            if (val > maxs[d]) maxs[d] = val;
        }
    for (size_t d = 0; d < dim; ++d) {
        float range = maxs[d] - mins[d];
        scales[d] = range > 1e-30f ? 255.0f / range : 0.0f;
    }
    free(maxs);
}

static inline uint8_t sq8_code(float val, float mn, float sc) {
    float t = val - mn;
    float q = roundf(t * sc);
    if (!(q == q)) return 0;   /* NaN: clamp keeps NaN, `as u8` saturates it to 0 */
    if (q < 0.0f) q = 0.0f;
    if (q > 255.0f) q = 255.0f;
    return (uint8_t)q;
}

void lo_sq8_quantize(const float *data, size_t n, size_t dim, const float *mins, const float *scales,
                     uint8_t *out) { /* also SQ8Data::quantize_query (:5741-5750) with n = 1 */
    for (size_t v = 0; v < n; ++v)
        for (size_t d = 0; d < dim; ++d) out[v * dim + d] = sq8_code(data[v * dim + d], mins[d], scales[d]);
}

typedef struct { float s; uint32_t row; } sq8c_t;
static int g_sq8_asc;
static int cmp_sq8(const void *a, const void *b) {
    const sq8c_t *x = (const sq8c_t *)a, *y = (const sq8c_t *)b;
    if (x->s != y->s) return g_sq8_asc ? (x->s < y->s ? -1 : 1) : (x->s > y->s ? -1 : 1);
    return x->row < y->row ? -1 : x->row > y->row;
}

/* sq8_two_pass_search (flat_mmap.rs:5868-5926): pass 1 = integer scores over the u8 codes (u32 dot for IP, u32 squared
 * L2 for L2 AND cosine), converted to f32 like `dist_fn(..) as f32`, top n_cand = max(k*20, 200) (cosine: max(k*100, 500)),
 * capped at n; pass 2 = exact single-row-kernel distances of the candidates, sorted, truncated to k.
 * The reference cuts ties at the n_cand boundary (and orders equal exact distances) by heap / unstable-sort accident;
 * this is the CANONICAL variant: (score, row ascending) at the cut, (distance, row ascending) at the end. */
size_t lo_sq8_search_canonical(const float *query, const float *cands, const uint8_t *codes, const float *mins,
                               const float *scales, size_t dim, size_t n, size_t k, int metric,
                               uint32_t *out_idx, float *out_dist) {
    if (n == 0 || k == 0) return 0;
    if (k > n) k = n; /* FlatMmap::search clamps k first (flat_mmap.rs:836) */
    uint8_t *qu = (uint8_t *)malloc(dim);
    lo_sq8_quantize(query, 1, dim, mins, scales, qu);
    int ip = metric == LO_IP;
    size_t n_cand = ip || metric == LO_L2 ? k * 20 : k * 100;
    size_t floor_c = (ip || metric == LO_L2) ? 200 : 500;
    if (n_cand < floor_c) n_cand = floor_c;
    if (n_cand > n) n_cand = n;
    sq8c_t *sc = (sq8c_t *)malloc(n * sizeof(sq8c_t));
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *r = codes + i * dim;
        uint32_t acc = 0;
        if (ip) for (size_t d = 0; d < dim; ++d) acc += (uint32_t)qu[d] * (uint32_t)r[d];
        else for (size_t d = 0; d < dim; ++d) { int32_t df = (int32_t)qu[d] - (int32_t)r[d]; acc += (uint32_t)(df * df); }
        sc[i].s = (float)acc;
        sc[i].row = (uint32_t)i;
    }
    g_sq8_asc = !ip;
    qsort(sc, n, sizeof(sq8c_t), cmp_sq8);
    cpair_t *p = (cpair_t *)malloc(n_cand * sizeof(cpair_t));
    for (size_t i = 0; i < n_cand; ++i) {
        p[i].id = sc[i].row;
        p[i].d = lo_compute_distance(query, cands + (size_t)sc[i].row * dim, dim, metric);
    }
    g_cmp_asc = lo_metric_is_ascending(metric);
    qsort(p, n_cand, sizeof(cpair_t), cmp_canonical);
    if (k > n_cand) k = n_cand;
    for (size_t i = 0; i < k; ++i) { out_idx[i] = (uint32_t)p[i].id; out_dist[i] = p[i].d; }
    free(p); free(sc); free(qu);
    return k;
}

/* ------------------------------------------------------------- filtered search */

/* FlatMmap::search_filtered (flat_mmap.rs:491-815), f32 rows / packed-binary rows.
 * k = min(k, m) (:504); rows >= n are skipped; subsets of <= 50,000 ids take the direct random-access
 * loop in SUBSET order (direct_access_topk :5223-5274), larger ones the bitset scan in ROW order,
 * chunked like the unfiltered scan (fused_topk_parallel_filtered :5439-5556; n < 4096 sequential).
 * Every distance uses the SINGLE-row kernels (simd::inner_product_f32 ...).  Packed metrics always
 * take the subset-order loop with strict `<` admission (packed_binary_search_filtered :1411-1444). */
size_t lo_flat_search_filtered(const float *query, const float *cands, size_t dim, size_t n, size_t k,
                               int metric, const uint64_t *subset, size_t m, int n_threads,
                               uint32_t *out_idx, float *out_dist) {
    if (n == 0 || k == 0 || m == 0) return 0;
    if (k > m) k = m;
    int asc = lo_metric_is_ascending(metric);
    entry_t *buf = (entry_t *)malloc(k * sizeof(entry_t));
    topk_t t;
    size_t len = 0;
    if (m <= 50000) {
        topk_init(&t, buf, k, asc);
        for (size_t j = 0; j < m; ++j) {
            size_t idx = (size_t)subset[j];
            if (idx >= n) continue;
            topk_offer(&t, lo_compute_distance(query, cands + idx * dim, dim, metric), (uint32_t)idx);
        }
        topk_finish(&t);
        len = t.len;
    } else {
        uint64_t max_id = 0;
        for (size_t j = 0; j < m; ++j) if (subset[j] > max_id) max_id = subset[j];
        size_t words = (size_t)(max_id / 64) + 1;
        uint64_t *bits = (uint64_t *)calloc(words, 8);
        for (size_t j = 0; j < m; ++j)
            if (subset[j] < n) bits[subset[j] / 64] |= 1ull << (subset[j] % 64);
        if (n < 4096) {
            topk_init(&t, buf, k, asc);
            for (size_t i = 0; i < n; ++i) {
                if (i > max_id || !((bits[i / 64] >> (i % 64)) & 1)) continue;
                topk_offer(&t, lo_compute_distance(query, cands + i * dim, dim, metric), (uint32_t)i);
            }
            topk_finish(&t);
            len = t.len;
        } else {
            size_t T = n_threads > 0 ? (size_t)n_threads : 1;
            size_t chunk = n / T;
            if (chunk < 512) chunk = 512;
            size_t n_chunks = (n + chunk - 1) / chunk;
            entry_t *pool = (entry_t *)malloc(n_chunks * k * sizeof(entry_t));
            entry_t **outs = (entry_t **)malloc(n_chunks * sizeof(entry_t *));
            size_t *lens = (size_t *)calloc(n_chunks, sizeof(size_t));
            for (size_t c = 0; c < n_chunks; ++c) {
                outs[c] = pool + c * k;
                topk_t tc;
                topk_init(&tc, outs[c], k, asc);
                size_t end = (c + 1) * chunk < n ? (c + 1) * chunk : n;
                for (size_t i = c * chunk; i < end; ++i) {
                    if (i > max_id || !((bits[i / 64] >> (i % 64)) & 1)) continue;
                    topk_offer(&tc, lo_compute_distance(query, cands + i * dim, dim, metric), (uint32_t)i);
                }
                topk_finish(&tc);
                lens[c] = tc.len;
            }
            len = merge_topk_results(outs, lens, n_chunks, k, asc, buf);
            free(lens); free(outs); free(pool);
        }
        free(bits);
    }
    for (size_t i = 0; i < len; ++i) { out_idx[i] = buf[i].idx; out_dist[i] = buf[i].dist; }
    free(buf);
    return len;
}

size_t lo_packed_search_filtered(const uint64_t *query, const uint64_t *rows, size_t words, size_t n,
                                 size_t k, int metric, const uint64_t *subset, size_t m,
                                 uint32_t *out_idx, float *out_dist) {
    if (n == 0 || k == 0 || m == 0) return 0;
    if (k > m) k = m;
    packed_fn f = packed_distance_fn(metric);
    entry_t *buf = (entry_t *)malloc(k * sizeof(entry_t));
    topk_t t;
    topk_init(&t, buf, k, 1);
    for (size_t j = 0; j < m; ++j) {
        size_t idx = (size_t)subset[j];
        if (idx >= n) continue;
        topk_offer(&t, f(query, rows + idx * words, words), (uint32_t)idx);
    }
    size_t len = t.len; /* no final sort of an under-full result (:1441-1443) */
    for (size_t i = 0; i < len; ++i) { out_idx[i] = buf[i].idx; out_dist[i] = buf[i].dist; }
    free(buf);
    return len;
}

/* Canonical filtered answer: the subset as a SET of valid rows, exact single-row-kernel distances,
 * (distance, row) order, k = min(k, m) like the reference. */
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
size_t lo_canonical_topk_filtered(const float *query, const float *cands, const uint64_t *packed_query,
                                  const uint64_t *packed_rows, size_t words, size_t dim, size_t n,
                                  size_t k, int metric, const uint64_t *subset, size_t m,
                                  uint32_t *out_idx, float *out_dist) {
    if (n == 0 || k == 0 || m == 0) return 0;
    if (k > m) k = m;
    uint64_t *ids = (uint64_t *)malloc(m * 8);
    memcpy(ids, subset, m * 8);
    qsort(ids, m, 8, cmp_u64);
    size_t u = 0;
    for (size_t j = 0; j < m; ++j)
        if (ids[j] < n && (u == 0 || ids[u - 1] != ids[j])) ids[u++] = ids[j];
    cpair_t *p = (cpair_t *)malloc((u ? u : 1) * sizeof(cpair_t));
    packed_fn f = packed_rows ? packed_distance_fn(metric) : NULL;
    for (size_t j = 0; j < u; ++j) {
        p[j].id = ids[j];
        p[j].d = packed_rows ? f(packed_query, packed_rows + ids[j] * words, words)
                             : lo_compute_distance(query, cands + ids[j] * dim, dim, metric);
    }
    g_cmp_asc = lo_metric_is_ascending(metric);
    qsort(p, u, sizeof(cpair_t), cmp_canonical);
    if (k > u) k = u;
    for (size_t i = 0; i < k; ++i) { out_idx[i] = (uint32_t)p[i].id; out_dist[i] = p[i].d; }
    free(p); free(ids);
    return k;
}

/* vector_store.rs:953-970 */
size_t lo_merge_results(const uint64_t *ids, const float *dists, size_t n, size_t k, int metric,
                        uint64_t *out_ids, float *out_dists) {
    if (n == 0 || k == 0) return 0;
    cpair_t *p = (cpair_t *)malloc(n * sizeof(cpair_t));
    for (size_t i = 0; i < n; ++i) { p[i].d = dists[i]; p[i].id = ids[i]; }
    g_cmp_asc = lo_metric_is_ascending(metric);
    qsort(p, n, sizeof(cpair_t), cmp_canonical);
    if (k > n) k = n;
    for (size_t i = 0; i < k; ++i) { out_ids[i] = p[i].id; out_dists[i] = p[i].d; }
    free(p);
    return k;
}

/* ------------------------------------------------------------------ k-means */

typedef struct { uint64_t s; } fastrng_t;
static inline double rng_next(fastrng_t *r) { /* kmeans.rs:29-35 */
    r->s = r->s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(r->s >> 33) / (double)((uint64_t)1 << 31);
}

void lo_fastrng_stream(uint64_t seed, size_t count, double *out) {
    fastrng_t r = {seed};
    for (size_t i = 0; i < count; ++i) out[i] = rng_next(&r);
}

static size_t adaptive_init_sample_size(size_t n, size_t k) { /* kmeans.rs:50-53 */
    size_t s = k * 32;
    if (s < 2048) s = 2048;
    if (s > 10000) s = 10000;
    return n < s ? n : s;
}

void lo_kmeans_assign(const float *data, size_t n, size_t dim, const float *centroids,
                      size_t n_centroids, int metric, uint32_t *assignments) { /* :237-264 */
    int asc = lo_metric_is_ascending(metric);
    for (size_t i = 0; i < n; ++i) {
        const float *v = data + i * dim;
        size_t best = 0;
        float best_rank = 3.402823466e+38f; /* f32::MAX */
        for (size_t c = 0; c < n_centroids; ++c) {
            float raw = lo_compute_distance(v, centroids + c * dim, dim, metric);
            float rank = asc ? raw : -raw;
            if (rank < best_rank) { best_rank = rank; best = c; }
        }
        assignments[i] = (uint32_t)best;
    }
}

static void kmeans_init(const float *data, size_t n, size_t dim, size_t k, int metric,
                        float *centroids) { /* kmeans.rs:141-196 */
    fastrng_t rng = {42};
    size_t sample_n = adaptive_init_sample_size(n, k);
    size_t *idx = (size_t *)malloc(n * sizeof(size_t));
    for (size_t i = 0; i < n; ++i) idx[i] = i;
    if (sample_n < n) { /* :38-47 partial Fisher-Yates */
        for (size_t i = 0; i < sample_n; ++i) {
            size_t span = n - i;
            size_t off = (size_t)(rng_next(&rng) * (double)span);
            if (off > span - 1) off = span - 1;
            size_t j = i + off, t = idx[i];
            idx[i] = idx[j]; idx[j] = t;
        }
    }
    float *sample = (float *)malloc(sample_n * dim * sizeof(float));
    for (size_t s = 0; s < sample_n; ++s)
        memcpy(sample + s * dim, data + idx[s] * dim, dim * sizeof(float));
    int asc = lo_metric_is_ascending(metric);
    size_t first = (size_t)(rng_next(&rng) * (double)sample_n) % sample_n; /* :165 */
    memcpy(centroids, sample + first * dim, dim * sizeof(float));
    float *min_ranks = (float *)malloc(sample_n * sizeof(float));
    for (size_t i = 0; i < sample_n; ++i) min_ranks[i] = 3.402823466e+38f;
    for (size_t c = 1; c < k; ++c) {
        const float *cen = centroids + (c - 1) * dim;
        for (size_t i = 0; i < sample_n; ++i) {
            float raw = lo_compute_distance(sample + i * dim, cen, dim, metric);
            float rank = asc ? raw : -raw;
            if (rank < min_ranks[i]) min_ranks[i] = rank;
        }
        /* Iterator::max_by keeps the LAST maximal element (:185-190) */
        size_t best = 0;
        for (size_t i = 1; i < sample_n; ++i)
            if (fcmp(min_ranks[i], min_ranks[best]) >= 0) best = i;
        memcpy(centroids + c * dim, sample + best * dim, dim * sizeof(float));
    }
    free(min_ranks); free(sample); free(idx);
}

size_t lo_kmeans_train(const float *data, size_t n, size_t dim, size_t requested, size_t max_iter,
                       int metric, float *centroids, uint32_t *assignments) { /* kmeans.rs:74-139 */
    size_t k = requested < n ? requested : n;
    if (n == 0 || k == 0 || dim == 0) return 0;
    kmeans_init(data, n, dim, k, metric, centroids);
    uint32_t *cur = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *nxt = (uint32_t *)malloc(n * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) cur[i] = UINT32_MAX; /* usize::MAX sentinel */
    float *sums = (float *)malloc(k * dim * sizeof(float));
    uint32_t *counts = (uint32_t *)malloc(k * sizeof(uint32_t));
    for (size_t it = 0; it < max_iter; ++it) {
        lo_kmeans_assign(data, n, dim, centroids, k, metric, nxt);
        int changed = 0;
        for (size_t i = 0; i < n; ++i) if (nxt[i] != cur[i]) { changed = 1; break; }
        memcpy(cur, nxt, n * sizeof(uint32_t));
        memset(sums, 0, k * dim * sizeof(float));
        memset(counts, 0, k * sizeof(uint32_t));
        for (size_t i = 0; i < n; ++i) { /* :273-286 sequential branch */
            size_t c = cur[i];
            counts[c] += 1;
            for (size_t d = 0; d < dim; ++d) sums[c * dim + d] = sums[c * dim + d] + data[i * dim + d];
        }
        size_t max_c = 0; uint32_t max_count = 0; /* max_by_key keeps the LAST maximum (:105-110) */
        for (size_t c = 0; c < k; ++c) if (counts[c] >= max_count) { max_count = counts[c]; max_c = c; }
        for (size_t c = 0; c < k; ++c) {
            if (counts[c] > 0) {
                float inv = 1.0f / (float)counts[c];
                for (size_t d = 0; d < dim; ++d) centroids[c * dim + d] = sums[c * dim + d] * inv;
            } else if (max_count > 1) {
                for (size_t d = 0; d < dim; ++d) {
                    float f = 1e-4f * (float)d;
                    float g = 1.0f + f;
                    centroids[c * dim + d] = centroids[max_c * dim + d] * g;
                }
            }
        }
        if (!changed) break;
    }
    lo_kmeans_assign(data, n, dim, centroids, k, metric, assignments); /* :133 */
    free(counts); free(sums); free(nxt); free(cur);
    return k;
}

/* Row-sharded training (no reference counterpart: the reference's cluster mode trains one index per shard; SURVEY 8(e) names the
 * all-reduce of the centroid sums and counts).  kmeans_train over the UNION of `world` row shards (global row g on rank g % world)
 * with ONE difference: the centroid sums of an iteration are formed per rank (sequential over that rank's members in ascending
 * row order, kmeans.rs:273-286 on the shard) and the per-rank sums are then added in RANK order, ((p0 + p1) + p2) + ... with f32
 * adds.  This order is the SPECIFICATION of the sharded training at every world size: the product gathers the per-rank sums and
 * adds them in this order (an all-reduce would associate differently per chunk from three ranks on).  Init, assignment, counts,
 * the empty-cluster rule and the stop test see the whole collection. */
size_t lo_kmeans_train_sharded(const float *data, size_t n, size_t dim, size_t requested, size_t max_iter,
                               int metric, size_t world, float *centroids, uint32_t *assignments) {
    size_t k = requested < n ? requested : n;
    if (n == 0 || k == 0 || dim == 0 || world == 0) return 0;
    kmeans_init(data, n, dim, k, metric, centroids);
    uint32_t *cur = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *nxt = (uint32_t *)malloc(n * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) cur[i] = UINT32_MAX;
    float *sums = (float *)malloc(k * dim * sizeof(float));
    float *part = (float *)malloc(k * dim * sizeof(float));
    uint32_t *counts = (uint32_t *)malloc(k * sizeof(uint32_t));
    for (size_t it = 0; it < max_iter; ++it) {
        lo_kmeans_assign(data, n, dim, centroids, k, metric, nxt);
        int changed = 0;
        for (size_t i = 0; i < n; ++i) if (nxt[i] != cur[i]) { changed = 1; break; }
        memcpy(cur, nxt, n * sizeof(uint32_t));
        memset(counts, 0, k * sizeof(uint32_t));
        for (size_t i = 0; i < n; ++i) counts[cur[i]] += 1;
        for (size_t r = 0; r < world; ++r) {
            memset(part, 0, k * dim * sizeof(float));
            for (size_t i = r; i < n; i += world) {
                size_t c = cur[i];
                for (size_t d = 0; d < dim; ++d) part[c * dim + d] = part[c * dim + d] + data[i * dim + d];
            }
            if (r == 0) memcpy(sums, part, k * dim * sizeof(float));
            else for (size_t j = 0; j < k * dim; ++j) sums[j] = sums[j] + part[j];
        }
        size_t max_c = 0; uint32_t max_count = 0;
        for (size_t c = 0; c < k; ++c) if (counts[c] >= max_count) { max_count = counts[c]; max_c = c; }
        for (size_t c = 0; c < k; ++c) {
            if (counts[c] > 0) {
                float inv = 1.0f / (float)counts[c];
                for (size_t d = 0; d < dim; ++d) centroids[c * dim + d] = sums[c * dim + d] * inv;
            } else if (max_count > 1) {
                for (size_t d = 0; d < dim; ++d) {
                    float f = 1e-4f * (float)d;
                    float g = 1.0f + f;
                    centroids[c * dim + d] = centroids[max_c * dim + d] * g;
                }
            }
        }
        if (!changed) break;
    }
    lo_kmeans_assign(data, n, dim, centroids, k, metric, assignments);
    free(counts); free(part); free(sums); free(nxt); free(cur);
    return k;
}

/* ---------------------------------------------------------------------- IVF */

typedef struct { float d; uint32_t i; } rank_t;
static int g_rank_asc;
static int cmp_rank_stable(const void *pa, const void *pb) { /* stable sort_by emulation: (d, i) */
    const rank_t *a = (const rank_t *)pa, *b = (const rank_t *)pb;
    int c = g_rank_asc ? fcmp(a->d, b->d) : fcmp(b->d, a->d);
    if (c) return c;
    return (a->i < b->i) ? -1 : (a->i > b->i) ? 1 : 0;
}

static size_t ivf_search_impl(const float *query, const float *data, const uint64_t *packed, size_t words,
                     size_t dim, size_t n, const float *centroids, size_t nlist,
                     const uint64_t *list_offsets, const uint32_t *list_rows, size_t nprobe,
                     size_t k, int metric, const uint8_t *in_subset, uint64_t *out_ids, float *out_dist,
                     uint32_t *out_probed);

size_t lo_ivf_search(const float *query, const float *data, const uint64_t *packed, size_t words,
                     size_t dim, size_t n, const float *centroids, size_t nlist,
                     const uint64_t *list_offsets, const uint32_t *list_rows, size_t nprobe,
                     size_t k, int metric, uint64_t *out_ids, float *out_dist,
                     uint32_t *out_probed) {
    return ivf_search_impl(query, data, packed, words, dim, n, centroids, nlist, list_offsets, list_rows, nprobe, k,
                           metric, NULL, out_ids, out_dist, out_probed);
}

/* IVFIndex::search with SearchParams.subset (ivf.rs:251-265): probed candidates are intersected with the
 * subset; when nothing is left the whole corpus restricted to the subset is scored instead. */
size_t lo_ivf_search_filtered(const float *query, const float *data, const uint64_t *packed, size_t words,
                              size_t dim, size_t n, const float *centroids, size_t nlist,
                              const uint64_t *list_offsets, const uint32_t *list_rows, size_t nprobe,
                              size_t k, int metric, const uint64_t *subset, size_t m, uint64_t *out_ids,
                              float *out_dist) {
    uint8_t *in = (uint8_t *)calloc(n ? n : 1, 1);
    for (size_t j = 0; j < m; ++j) if (subset[j] < n) in[subset[j]] = 1;
    size_t r = ivf_search_impl(query, data, packed, words, dim, n, centroids, nlist, list_offsets, list_rows, nprobe, k,
                               metric, in, out_ids, out_dist, NULL);
    free(in);
    return r;
}

static size_t ivf_search_impl(const float *query, const float *data, const uint64_t *packed, size_t words,
                     size_t dim, size_t n, const float *centroids, size_t nlist,
                     const uint64_t *list_offsets, const uint32_t *list_rows, size_t nprobe,
                     size_t k, int metric, const uint8_t *in_subset, uint64_t *out_ids, float *out_dist,
                     uint32_t *out_probed) { /* ivf.rs:181-348 */
    if (n == 0) return 0;
    if (nprobe < 1) nprobe = 1; /* :192-196 (caller resolves the stored default) */
    int asc = lo_metric_is_ascending(metric);
    int binary = lo_metric_is_binary(metric);
    int routing = binary ? LO_L2 : metric; /* :81-87 */
    /* NOTE: for binary metrics the reference routes with the {0,1}-dequantised
     * query (ivf.rs:211-216, BinaryQuantizer); the caller passes that as `query`
     * and the packed query is derived from it (threshold 0.5 on {0,1} values). */
    rank_t *cd = (rank_t *)malloc(nlist * sizeof(rank_t));
    for (size_t c = 0; c < nlist; ++c) {
        cd[c].d = lo_compute_distance(query, centroids + c * dim, dim, routing);
        cd[c].i = (uint32_t)c;
    }
    g_rank_asc = lo_metric_is_ascending(routing);
    qsort(cd, nlist, sizeof(rank_t), cmp_rank_stable); /* stable sort_by :237-241 */
    size_t np = nprobe < nlist ? nprobe : nlist;
    size_t total = 0;
    for (size_t p = 0; p < np; ++p) {
        if (out_probed) out_probed[p] = cd[p].i;
        total += (size_t)(list_offsets[cd[p].i + 1] - list_offsets[cd[p].i]);
    }
    uint32_t *cand = (uint32_t *)malloc((total > n ? total : n) * sizeof(uint32_t) + 4);
    {
        size_t w = 0;
        for (size_t p = 0; p < np; ++p)
            for (uint64_t j = list_offsets[cd[p].i]; j < list_offsets[cd[p].i + 1]; ++j)
                if (!in_subset || in_subset[list_rows[j]]) cand[w++] = list_rows[j]; /* :251-256 retain */
        total = w;
    }
    if (total == 0) { /* :258-265 fall back to the (filtered) full corpus */
        for (size_t i = 0; i < n; ++i)
            if (!in_subset || in_subset[i]) cand[total++] = (uint32_t)i;
    }
    free(cd);
    if (total == 0) { free(cand); return 0; } /* :267-269 */
    size_t pool = k < total ? k : total; /* :271-275, no exact-rerank for None/Binary */
    cpair_t *sc = (cpair_t *)malloc(total * sizeof(cpair_t));
    uint64_t *pq = NULL;
    if (binary && packed) {
        pq = (uint64_t *)malloc(words * 8);
        lo_pack_binary_f32(query, dim, pq);
    }
    packed_fn pf = packed_distance_fn(metric);
    for (size_t j = 0; j < total; ++j) {
        size_t c = cand[j];
        sc[j].d = pq ? pf(pq, packed + c * words, words)
                     : lo_compute_distance(query, data + c * dim, dim, metric);
        sc[j].id = c; /* canonical tie-break: original row id (the reference's sort_unstable leaves ties unpinned) */
    }
    g_cmp_asc = asc;
    qsort(sc, total, sizeof(cpair_t), cmp_canonical);
    for (size_t j = 0; j < pool; ++j) { out_ids[j] = sc[j].id; out_dist[j] = sc[j].d; }
    free(pq); free(sc); free(cand);
    return pool;
}

static int cmp_float_asc(const void *a, const void *b) {
    return fcmp(*(const float *)a, *(const float *)b);
}

int lo_binary_fit(const float *data, size_t n, size_t dim, float *thresholds) { /* quantizer/mod.rs:321-357 */
    for (size_t d = 0; d < dim; ++d) thresholds[d] = 0.5f;
    if (n == 0 || dim == 0) return 1;
    int already_binary = 1;
    for (size_t i = 0; i < n * dim; ++i)
        if (!(data[i] == 0.0f || data[i] == 1.0f)) { already_binary = 0; break; }
    if (already_binary) return 1;
    float *col = (float *)malloc(n * sizeof(float));
    for (size_t d = 0; d < dim; ++d) {
        for (size_t i = 0; i < n; ++i) col[i] = data[i * dim + d];
        qsort(col, n, sizeof(float), cmp_float_asc);
        float mn = col[0], mx = col[n - 1], med = col[n / 2];
        if (med <= mn || med >= mx) {
            float s = mn + mx;
            thresholds[d] = 0.5f * s;
        } else {
            thresholds[d] = med;
        }
    }
    free(col);
    return 0;
}

void lo_binary_quantize(const float *data, size_t n, size_t dim, const float *thresholds, float *out) {
    for (size_t i = 0; i < n; ++i)
        for (size_t d = 0; d < dim; ++d) out[i * dim + d] = data[i * dim + d] > thresholds[d] ? 1.0f : 0.0f;
}

void lo_ivf_flat_layout(const uint32_t *assignments, size_t n, size_t nlist, uint64_t *offsets,
                        uint32_t *original_ids) { /* ivf_flat_mmap.rs:105-130 */
    uint64_t *sizes = (uint64_t *)calloc(nlist, sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) sizes[assignments[i]] += 1;
    offsets[0] = 0;
    for (size_t p = 0; p < nlist; ++p) offsets[p + 1] = offsets[p] + sizes[p];
    uint64_t *wp = (uint64_t *)malloc(nlist * sizeof(uint64_t));
    memcpy(wp, offsets, nlist * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) original_ids[wp[assignments[i]]++] = (uint32_t)i;
    free(wp); free(sizes);
}

size_t lo_ivf_routing_dims(const float *centroids, size_t dim, size_t nlist, uint32_t *out_dims) {
    /* ivf_flat_mmap.rs:309-348 */
    if (nlist == 0 || dim == 0 || !(dim >= 64 && nlist >= 64)) return 0;
    float *sums = (float *)calloc(dim, sizeof(float)), *sq = (float *)calloc(dim, sizeof(float));
    for (size_t c = 0; c < nlist; ++c)
        for (size_t d = 0; d < dim; ++d) {
            float v = centroids[c * dim + d];
            sums[d] = sums[d] + v;
            float p = v * v;
            sq[d] = sq[d] + p;
        }
    float inv_k = 1.0f / (float)nlist;
    size_t keep = dim < 16 ? dim : 16;
    rank_t *dv = (rank_t *)malloc(dim * sizeof(rank_t));
    for (size_t d = 0; d < dim; ++d) {
        float mean = sums[d] * inv_k;
        float a = sq[d] * inv_k, b = mean * mean;
        dv[d].d = a - b;
        dv[d].i = (uint32_t)d;
    }
    g_rank_asc = 0; /* descending variance; ties -> lower dim (select_nth_unstable is unpinned) */
    qsort(dv, dim, sizeof(rank_t), cmp_rank_stable);
    /* selected.sort_unstable() -> ascending dim ids */
    for (size_t i = 0; i < keep; ++i) out_dims[i] = dv[i].i;
    for (size_t i = 1; i < keep; ++i) {
        uint32_t x = out_dims[i];
        size_t j = i;
        while (j > 0 && out_dims[j - 1] > x) { out_dims[j] = out_dims[j - 1]; --j; }
        out_dims[j] = x;
    }
    free(dv); free(sq); free(sums);
    return keep;
}

size_t lo_ivf_flat_probe(const float *query, const float *centroids, size_t dim, size_t nlist,
                         size_t nprobe, int metric, const uint32_t *routing_dims,
                         size_t n_routing, uint32_t *out_parts) { /* ivf_flat_mmap.rs:381-444 */
    if (nprobe >= nlist) {
        for (size_t c = 0; c < nlist; ++c) out_parts[c] = (uint32_t)c;
        return nlist;
    }
    if (metric == LO_IP && dim >= 64 && nlist >= 64 && n_routing > 0) {
        size_t shortlist = nprobe * 3;
        if (shortlist < 24) shortlist = 24;
        if (shortlist > 96) shortlist = 96;
        if (shortlist > nlist) shortlist = nlist;
        rank_t *best = (rank_t *)malloc(shortlist * sizeof(rank_t));
        size_t len = 0;
        for (size_t c = 0; c < nlist; ++c) {
            const float *cen = centroids + c * dim;
            float score = 0.0f; /* coarse_ip_score :350-357 */
            for (size_t r = 0; r < n_routing; ++r) {
                float p = query[routing_dims[r]] * cen[routing_dims[r]];
                score = score + p;
            }
            if (len < shortlist) { best[len].d = score; best[len].i = (uint32_t)c; ++len; continue; }
            size_t worst = 0; float ws = best[0].d; /* shortlist_insert :359-378 */
            for (size_t i = 1; i < shortlist; ++i) if (best[i].d < ws) { ws = best[i].d; worst = i; }
            if (score > ws) { best[worst].d = score; best[worst].i = (uint32_t)c; }
        }
        for (size_t i = 0; i < len; ++i)
            best[i].d = lo_compute_distance(query, centroids + (size_t)best[i].i * dim, dim, metric);
        g_rank_asc = 0;
        qsort(best, len, sizeof(rank_t), cmp_rank_stable);
        size_t np = nprobe < len ? nprobe : len;
        for (size_t i = 0; i < np; ++i) out_parts[i] = best[i].i;
        free(best);
        return np;
    }
    rank_t *d = (rank_t *)malloc(nlist * sizeof(rank_t));
    for (size_t c = 0; c < nlist; ++c) {
        d[c].d = lo_compute_distance(query, centroids + c * dim, dim, metric);
        d[c].i = (uint32_t)c;
    }
    g_rank_asc = lo_metric_is_ascending(metric);
    qsort(d, nlist, sizeof(rank_t), cmp_rank_stable);
    for (size_t i = 0; i < nprobe; ++i) out_parts[i] = d[i].i;
    free(d);
    return nprobe;
}

size_t lo_ivf_flat_search(const float *query, const float *slab_data, size_t dim, size_t n,
                          const float *centroids, size_t nlist, const uint64_t *offsets,
                          const uint32_t *original_ids, const uint32_t *routing_dims,
                          size_t n_routing, size_t nprobe, size_t k, int metric,
                          uint32_t *out_ids, float *out_dist) { /* ivf_flat_mmap.rs:225-304 */
    if (n == 0 || k == 0) return 0;
    if (k > n) k = n;
    if (nprobe < 1) nprobe = 1;
    if (nprobe > nlist) nprobe = nlist;
    uint32_t *parts = (uint32_t *)malloc(nlist * sizeof(uint32_t));
    size_t np = lo_ivf_flat_probe(query, centroids, dim, nlist, nprobe, metric, routing_dims,
                                  n_routing, parts);
    size_t total = 0;
    for (size_t p = 0; p < np; ++p) total += (size_t)(offsets[parts[p] + 1] - offsets[parts[p]]);
    if (total == 0) { free(parts); return 0; }
    cpair_t *best = (cpair_t *)malloc(total * sizeof(cpair_t));
    size_t w = 0;
    for (size_t p = 0; p < np; ++p)
        for (uint64_t j = offsets[parts[p]]; j < offsets[parts[p] + 1]; ++j) {
            best[w].d = lo_compute_distance(query, slab_data + j * dim, dim, metric);
            best[w].id = original_ids[j];
            ++w;
        }
    free(parts);
    if (k > total) k = total;
    g_cmp_asc = lo_metric_is_ascending(metric);
    qsort(best, total, sizeof(cpair_t), cmp_canonical);
    for (size_t i = 0; i < k; ++i) { out_ids[i] = (uint32_t)best[i].id; out_dist[i] = best[i].d; }
    free(best);
    return k;
}
