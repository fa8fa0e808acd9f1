// lynse_hip.hip — host side of liblynse_hip.so: handles, HBM layout, stage planning, launches and
// the C ABI declared in include/lynse_hip.h.  gfx950 only; there is no CPU fallback anywhere in
// this file — every compute entry needs a HIP device.
//
// Reference call path replaced:
//   Collection::search / batch_search (engine.rs:4697-4833, :5352-5498)
//     -> VectorStore::search (vector_store.rs:972-1004) -> FlatMmap::search (flat_mmap.rs:824-923)
//     -> exact_flat_search / packed_binary_search (:1173-1230, :1345-1409) -> simd kernels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include <sched.h>

#include "kernels.h"
#include "scan_qs.h"
#include "scan_qh.h"
#include "ivf.h"

using namespace lynse;

// ------------------------------------------------------------------------------------ errors ----
static thread_local std::string g_last_error;

static int set_error(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define LY_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            int _code = (_e == hipErrorOutOfMemory) ? LYNSE_ERR_OUT_OF_MEMORY : LYNSE_ERR_DEVICE; \
            return set_error(_code, std::string(#expr) + ": " + hipGetErrorString(_e));       \
        }                                                                                     \
    } while (0)

#define LY_TRY(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != LYNSE_OK) return _rc; \
    } while (0)

// hipMemset on device memory may return before the fill has run (it is ordered on the NULL stream), and the search contexts'
// streams are created hipStreamNonBlocking: they do NOT wait for the null stream.  A kernel on a context stream could therefore
// start before the one-time clears of a fresh workspace had landed — seen as a first search on a new index that came back with
// zero results, once in ~10^5 cases (the arrival counter of k_small_search held garbage).  Every synchronous fill waits here.
static int memset_done(void* p, int v, size_t n) {
    LY_HIP(hipMemset(p, v, n));
    LY_HIP(hipStreamSynchronize(nullptr));
    return LYNSE_OK;
}

// ... and the same for blocking host-to-device copies of set-up data (a pageable source is staged and the DMA may still be in
// flight when hipMemcpy returns; the kernels that read the data run on non-blocking streams)
static int h2d_done(void* dst, const void* src, size_t n) {
    LY_HIP(hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
    LY_HIP(hipStreamSynchronize(nullptr));
    return LYNSE_OK;
}

// The end of a blocking search: hipStreamSynchronize spins on the completion signal for ~100 us and then parks the thread on an
// interrupt (ROCclr) — a batch over 10M rows runs 1.8 ms, so every blocking step paid the wake-up (~20-25 us between k_select_final
// and the next batch's first kernel in the kernel trace).  Poll the stream instead: a tight poll for the first 200 us (the latency-shaped
// searches end inside it), then a poll with sched_yield() between the queries — a thread that has something to enqueue, or another
// blocked reader of the same handle, gets the core (eight blocked readers were eight spinning cores) — for up to
// LYNSE_HIP_SPIN_US in all (default 5 ms: any batch this library answers at the BASELINE sizes), then block as before.
static int stream_wait(hipStream_t st) {
    static const int64_t spin_us = []() { const char* e = getenv("LYNSE_HIP_SPIN_US"); return e ? (int64_t)atoll(e) : (int64_t)5000; }();
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        bool polite = false;
        for (uint32_t it = 0;; ++it) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return LYNSE_OK;
            if (q != hipErrorNotReady) return set_error(LYNSE_ERR_DEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(q));
            if (polite) sched_yield();
            if (polite || (it & 15u) == 15u) {
                const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                if (us > spin_us) break;
                if (us > 200) polite = true;
            }
        }
    }
    LY_HIP(hipStreamSynchronize(st));
    return LYNSE_OK;
}

extern "C" int lynse_hip_abi_version(void) { return LYNSE_HIP_ABI_VERSION; }

extern "C" size_t lynse_hip_last_error(char* buf, size_t cap) {
    if (buf && cap) {
        size_t n = std::min(cap - 1, g_last_error.size());
        memcpy(buf, g_last_error.data(), n);
        buf[n] = 0;
    }
    return g_last_error.size();
}

extern "C" int lynse_hip_device_count(int* out) {
    if (!out) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out = 0;
        return set_error(LYNSE_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *out = n;
    return LYNSE_OK;
}

// ----------------------------------------------------------------------------------- metrics ----
static std::string lower(const char* s) {
    std::string r(s ? s : "");
    for (auto& c : r) c = (char)tolower((unsigned char)c);
    return r;
}

extern "C" int lynse_hip_metric_from_str(const char* name, int* out) {  // distance/mod.rs:39-63
    if (!name || !out) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    const std::string s = lower(name);
    static const struct { const char* n; int m; } tab[] = {
        {"ip", M_IP}, {"inner_product", M_IP}, {"inner", M_IP}, {"dot", M_IP},
        {"l2", M_L2}, {"l2sq", M_L2}, {"l2_squared", M_L2}, {"euclidean", M_L2},
        {"cosine", M_COS}, {"cos", M_COS}, {"cosine_distance", M_COS},
        {"hamming", M_HAMMING}, {"jaccard", M_JACCARD},
        {"dice", M_DICE}, {"sorensen", M_DICE}, {"sorensen_dice", M_DICE}, {"sorensen-dice", M_DICE},
        {"tanimoto", M_TANIMOTO},
    };
    for (auto& t : tab)
        if (s == t.n) { *out = t.m; return LYNSE_OK; }
    // valid names of the reference's other nine metrics (distance/mod.rs:46-60): recognised, but not part of this path
    for (const char* o : {"l1", "manhattan", "cityblock", "haversine", "haversine_m", "haversine-m", "geo", "correlation", "pearson",
                          "hellinger", "wasserstein", "wasserstein1d", "wasserstein_1d", "wasserstein-1d", "emd", "jensen_shannon",
                          "jensen-shannon", "jensenshannon", "js", "chebyshev", "chebychev", "linf", "l_inf", "l-infinity", "canberra",
                          "bray_curtis", "bray-curtis", "braycurtis"})
        if (s == o)
            return set_error(LYNSE_ERR_UNSUPPORTED, std::string("metric '") + name + "' is a LynseDB metric outside the GPU FLAT / IVF path "
                                                    "(supported: ip, l2, cosine, hamming, jaccard, dice, tanimoto)");
    return set_error(LYNSE_ERR_UNKNOWN_METRIC, std::string("Unknown metric: ") + name);
}

extern "C" int lynse_hip_metric_from_index_mode(const char* mode, int* out) {  // distance/mod.rs:67-107
    if (!mode || !out) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    std::string up(mode);
    for (auto& c : up) c = (char)toupper((unsigned char)c);
    std::vector<std::string> tok;
    size_t p = 0;
    while (p <= up.size()) {
        size_t e = up.find('-', p);
        if (e == std::string::npos) e = up.size();
        tok.push_back(up.substr(p, e - p));
        p = e + 1;
    }
    auto has = [&](const char* v) { return std::find(tok.begin(), tok.end(), v) != tok.end(); };
    // metric families outside this path take precedence in the reference's chain
    if ((has("JENSEN") && has("SHANNON")) || (has("BRAY") && has("CURTIS")))
        return set_error(LYNSE_ERR_UNSUPPORTED, std::string("index mode names a LynseDB metric outside the GPU FLAT / IVF path: ") + mode);
    for (const char* o : {"JENSENSHANNON", "JS", "CHEBYSHEV", "CHEBYCHEV", "LINF", "CANBERRA", "BRAYCURTIS"})
        if (has(o)) return set_error(LYNSE_ERR_UNSUPPORTED, std::string("index mode names a LynseDB metric outside the GPU FLAT / IVF path: ") + mode);
    int m = -1;
    if (has("TANIMOTO")) m = M_TANIMOTO;
    else if (has("JACCARD")) m = M_JACCARD;
    else if (has("HAMMING")) m = M_HAMMING;
    else if (has("DICE") || has("SORENSEN")) m = M_DICE;
    else {
        for (const char* o : {"HAVERSINE", "GEO", "CORRELATION", "PEARSON", "HELLINGER", "WASSERSTEIN",
                              "WASSERSTEIN1D", "EMD", "L1", "MANHATTAN", "CITYBLOCK"})
            if (has(o)) return set_error(LYNSE_ERR_UNSUPPORTED, std::string("index mode names a LynseDB metric outside the GPU FLAT / IVF path: ") + mode);
        if (has("L2") || has("L2SQ")) m = M_L2;
        else if (has("COS") || has("COSINE")) m = M_COS;
        else if (has("IP")) m = M_IP;
    }
    if (m < 0) return set_error(LYNSE_ERR_UNKNOWN_METRIC, std::string("Unknown index mode: ") + mode);
    *out = m;
    return LYNSE_OK;
}

extern "C" int lynse_hip_metric_is_ascending(int metric) { return metric_ascending(metric) ? 1 : 0; }
extern "C" int lynse_hip_metric_is_binary(int metric) { return (metric >= M_HAMMING && metric <= M_TANIMOTO) ? 1 : 0; }

static bool metric_valid(int m) { return m >= M_IP && m <= M_TANIMOTO; }

// ------------------------------------------------------------------------------------ handle ----
constexpr size_t H_OUT_BYTES = 96 * 1024;  // results up to this size come back through pinned memory (no pageable-copy stalls)

struct Workspace {
    uint32_t qcap = 0, cap = 0, D = 0, W = 0, kcap = 0;
    uint64_t* cand = nullptr;
    uint64_t* candB = nullptr;   // segmented emission: [QCHUNK][SEG_KEYS] keys (ScanArgs::candB)
    uint8_t* segcnt = nullptr;   // [QCHUNK][SEG_MAX] per-segment key counts
    uint32_t *count = nullptr, *overflow = nullptr;
    float *thr = nullptr, *qinv = nullptr, *qn2 = nullptr, *qrinv = nullptr, *marg2 = nullptr;
    _Float16* Q16 = nullptr;
    size_t q16_halves = 0;
    float* Qf = nullptr;
    uint64_t* QW = nullptr;
    uint64_t* QWp = nullptr;  // queries padded to a power-of-two word count (k_scan_binary_rows)
    uint64_t* out_rows = nullptr;
    float* out_dists = nullptr;
    uint32_t* out_counts = nullptr;  // out_counts[QCHUNK] and overflow[QCHUNK] share one allocation: one readback for both
    uint32_t* h_hdr = nullptr;       // pinned host mirror of that pair
    uint8_t* h_out = nullptr;        // pinned staging for small host-API results (rows then dists), H_OUT_BYTES
    unsigned long long* pool_total = nullptr;
    uint32_t* gsync = nullptr;        // hand-over words of the fused sample stage (ScanArgs::gsync)
    int* dyn = nullptr;               // self-tightening scan (ScanArgs::dyn_*): [qcap] tau, [qcap] margin, [qcap][32] partition maxima
    uint64_t* small_part = nullptr;   // k_small_search: [workgroups][SMALL_MAX_Q][SMALL_MAX_K] keys
    uint32_t* small_ticket = nullptr;
    void release() {
        if (h_hdr) (void)hipHostFree(h_hdr);
        if (h_out) (void)hipHostFree(h_out);
        for (void* p : {(void*)cand, (void*)candB, (void*)segcnt, (void*)count, (void*)thr, (void*)qinv, (void*)qn2,
                        (void*)qrinv, (void*)marg2, (void*)Q16, (void*)Qf, (void*)QW, (void*)QWp, (void*)out_rows,
                        (void*)out_dists, (void*)out_counts, (void*)pool_total, (void*)small_part, (void*)small_ticket, (void*)gsync, (void*)dyn})
            if (p) (void)hipFree(p);
        *this = Workspace();
    }
};

struct lynse_hip_flat {
    uint32_t dim = 0, ld = 0, words = 0;
    int device = 0;
    int num_cu = 256;
    // Reader / writer discipline (VectorIndex: Send + Sync, Arc<RwLock<Collection>> with inner.read() on the search path,
    // src/python/mod.rs:950, :1187): unfiltered searches hold `rw` SHARED and run concurrently, each on its own search
    // context (workspace + stream + event pool) taken from `ctx`; append / finalize / lazy builds / the subset-filtered
    // paths (they borrow per-handle scratch) hold it EXCLUSIVE and use context 0.  The context of the calling thread is
    // selected by a thread-local slot (cur()), so the helpers below keep reading h->ws / cur(h).stream through it.
    std::shared_mutex rw;
    struct Ctx {
        hipStream_t stream = nullptr;
        Workspace ws;
        std::vector<hipEvent_t> ev_pool;
        // filtered search: row bitmask of the subset + staging for the subset ids — per context, so that masked filtered
        // searches run under the shared lock like unfiltered ones (the gathered-rows strategy swaps handle fields: exclusive)
        uint32_t* d_mask = nullptr;
        uint64_t mask_words = 0;
        uint64_t* d_subset = nullptr;
        uint64_t subset_cap = 0;
    };
    static constexpr int MAX_CTX = 8;
    Ctx ctx[MAX_CTX];
    int n_ctx = 4;                       // LYNSE_HIP_CONTEXTS (1..8): searches in flight per handle
    std::mutex ctx_mu;
    std::condition_variable ctx_cv;
    uint32_t ctx_busy = 0;               // bit s: context s is taken
    std::mutex prof_mu;                  // profile counters are shared by the contexts
    std::atomic<int> inflight{0};        // tickets of lynse_hip_flat_search_submit_* not yet waited for (async_host.inc)

    uint64_t n = 0, capacity = 0;
    float* rows = nullptr;       // capacity x ld f32, row-major (pad columns zero); nullptr for an F16 shard
    // VectorDtype::F16 shards (src/storage/dtype.rs, flat_mmap.rs:187-221) keep their rows as f16 bits ONLY: capacity x ld16
    // halves, pad columns zero — the bytes of an F16 segment file, the source of every exact score (sequential f32 sums over
    // the exactly decoded halves, simd.rs:805-846).  With sv == 1 the coarse-pass shadow rows16 is this very buffer.
    _Float16* rows_h = nullptr;
    bool shadow_alias = false;   // rows16 == rows_h (not separately owned)
    // f16 shadow of the rows for the coarse scan (k_scan_h16): (half)(v * sv16), pitch ld16 halves, built at finalize
    _Float16* rows16 = nullptr;
    uint32_t ld16 = 0;
    uint64_t n16 = 0, cap16 = 0;
    float sv16 = 0.0f;
    uint64_t n_packed = 0, packed_capacity = 0;
    uint64_t* packed = nullptr;  // n x words u64
    bool packed_only = false;

    uint64_t n_stats = 0;        // rows covered by vn2/vrinv/stats
    uint64_t stats_capacity = 0;
    float *vn2 = nullptr, *vrinv = nullptr;
    uint32_t* d_stats = nullptr;  // 5 words, see k_row_stats
    int rows_integer = 0, rows_nonneg = 0;   // every element of the shard is an integer / non-negative (k_row_stats stats[4]; k_prep_queries' exactness rule)
    float amax = 0.f, vmax = 0.f, vmin = 0.f, sv = 1.f;
    int cos_degenerate = 0;

    uint64_t row_stride = 1, row_offset = 0;
    int ip_form = LYNSE_IPFORM_AUTO;
    int no_fused = 0;            // lynse_hip_flat_set_fused_search(h, 0): always the staged pipeline (tests, A/B)
    uint32_t qchunk = 256;       // queries per pipeline pass (QCHUNK); the k-means assignment step widens it on its centroid store:
                                 // thousands of rows per launch against <= cap centroids (k_scan_h16's blockIdx.y)
    int dtype = LYNSE_DTYPE_F32;  // F16: rows hold f16-representable values, distances use the f16 kernels' sequential sums
    uint32_t stage0_rows = 4096, growth = 8, cap = 16384;

    // SQ8 two-pass mode (FLAT-*-SQ8): signed codes (code - 128), per-dimension min / scale, per-row sums; built lazily
    int8_t* sq8 = nullptr;
    uint32_t ld8 = 0;
    uint64_t n_sq8 = 0, sq8_cap = 0;
    float *sq8_mins = nullptr, *sq8_scales = nullptr;
    int *sq8_sum = nullptr, *sq8_sum2 = nullptr;
    uint32_t* sq8_mm = nullptr;  // 2 x dim ordered-int min / max
    uint32_t* sq8_stats = nullptr;  // [0] max row L1 of the signed codes, [1] non-finite elements (k_sq8_quantize)
    uint32_t* sq8_mm_prev = nullptr;  // scratch of the incremental append: the min / max table before the new rows were merged + a flag word
    uint32_t sq8_a1 = 0, sq8_a2sq = 0;   // max row L1 / sum of squares of the signed codes
    float sq8_eps2 = 0.0f;                 // bound of the max row sum of squared quantisation residuals (k_sq8_quantize stats[3])
    bool sq8_finite = false;
    // the L2 form of the certified int8 pass: SQ8 codes of the AUGMENTED rows [v, |v|^2] (k_i8c_prep_queries, aug = 1), pitch
    // ld8a = round_up(dim + 1, 128) (whole 128-column slabs: the non-ragged scan kernels), built lazily on the first L2 batch
    // of 33..256 queries
    int8_t* sq8a = nullptr;
    uint32_t ld8a = 0, aug_cols = 1;     // aug_cols: columns carrying |v|^2 (a power of two <= 32 that fits the slab padding)
    uint64_t n_sq8a = 0, sq8a_cap = 0;
    float *sq8a_mins = nullptr, *sq8a_scales = nullptr;
    uint32_t *sq8a_mm = nullptr, *sq8a_stats = nullptr;
    uint32_t sq8a_a1 = 0, sq8a_a2sq = 0;   // max row L1 / sum of squares of the signed codes
    float sq8a_eps2 = 0.0f;                 // bound of the max row sum of squared quantisation residuals (k_sq8_quantize stats[3])
    bool sq8a_finite = false;
    std::atomic<int> i8c_strikes_l2{0};
    // the cosine form: SQ8 codes of the UNIT rows fl(v_d * rinv[row]) (pitch ld8), built lazily on the first cosine batch of
    // 33..256 queries
    int8_t* sq8c = nullptr;
    uint64_t n_sq8c = 0, sq8c_cap = 0;
    float *sq8c_mins = nullptr, *sq8c_scales = nullptr;
    uint32_t *sq8c_mm = nullptr, *sq8c_stats = nullptr;
    uint32_t sq8c_a1 = 0, sq8c_a2sq = 0;   // max row L1 / sum of squares of the signed codes
    float sq8c_eps2 = 0.0f;                 // bound of the max row sum of squared quantisation residuals (k_sq8_quantize stats[3])
    bool sq8c_finite = false;
    std::atomic<int> i8c_strikes_cos{0};
    // certified int8 coarse pass (FLAT-IP batches of 33..256 queries): -1 = off (env / strikes), else overflow strikes so far
    std::atomic<int> i8c_strikes{0};     // (searches under the SHARED lock bump it)
    // batched Hamming on the matrix pipe: one signed byte per bit (+1 / -1) of the packed rows, pitch ld8; built lazily on the
    // first Hamming batch of >= bin_mfma_minq() queries (4x the packed words: the price of feeding the matrix pipe from HBM)
    uint8_t* bpm = nullptr;              // FP4 nibbles (+1.0 / -1.0 per bit), pitch ld_bpm = round_up(dim, 256) / 2 bytes
    uint32_t ld_bpm = 0;
    uint64_t n_bpm = 0, bpm_cap = 0;
    bool bpm_failed = false;             // the copy did not fit: the popcount kernels answer
    // gathered ("few matches") filtered path: compact copy of the listed shadow rows, their norms and 32-bit ids
    _Float16* g_rows16 = nullptr;
    float *g_vn2 = nullptr, *g_vrinv = nullptr;
    uint32_t* g_ids32 = nullptr;
    uint64_t g_cap = 0;

    std::atomic<bool> profiling{false};   // (read by searches without the lock: atomics)
    std::atomic<uint32_t> prof_rate{1};              // every prof_rate-th search records its events (lynse_hip_flat_profile_enable(h, n))
    std::atomic<uint32_t> prof_seq{0};
    lynse_hip_profile prof{};
};

static inline bool is_f16(const lynse_hip_flat* h) { return h->dtype == LYNSE_DTYPE_F16; }
// the row matrix the exact-scoring kernels read: f32 rows, or the f16 bits of an F16 shard handed in as float* with a pitch
// of ld16 / 2 floats (exact_score_f16seq turns the pointer back; LYNSE_IPFORM_F16SEQ goes with it)
static inline const float* score_rows(const lynse_hip_flat* h) { return is_f16(h) ? reinterpret_cast<const float*>(h->rows_h) : h->rows; }
static inline uint32_t score_ld(const lynse_hip_flat* h) { return is_f16(h) ? h->ld16 / 2 : h->ld; }

static thread_local bool tl_prof = false;   // the search running on this thread records profile events (profile_begin_search)
static thread_local int tl_ctx_slot = 0;   // the search context of the calling thread (0 outside a concurrent search)
static inline lynse_hip_flat::Ctx& cur(lynse_hip_flat* h) { return h->ctx[tl_ctx_slot]; }
static inline const lynse_hip_flat::Ctx& cur(const lynse_hip_flat* h) { return h->ctx[tl_ctx_slot]; }

// Takes a free search context for the calling thread (blocks while all n_ctx are busy); context 0's stream is created with
// the handle, the others on first use.
struct CtxLease {
    lynse_hip_flat* h = nullptr;
    int slot = 0, prev = 0;
    int acquire(lynse_hip_flat* hh) {
        h = hh;
        {
            std::unique_lock<std::mutex> lk(h->ctx_mu);
            for (;;) {
                int s = 0;
                while (s < h->n_ctx && ((h->ctx_busy >> s) & 1u)) ++s;
                if (s < h->n_ctx) { slot = s; h->ctx_busy |= 1u << s; break; }
                // every context leased by a TICKET (only a wait() of the caller frees those): an error, not a deadlock
                if (h->inflight.load(std::memory_order_acquire) >= h->n_ctx) {
                    h = nullptr;
                    return set_error(LYNSE_ERR_INVALID_ARGUMENT, "every search context of the handle is held by a ticket: wait for one first (LYNSE_HIP_CONTEXTS)");
                }
                h->ctx_cv.wait(lk);
            }
        }
        prev = tl_ctx_slot;
        tl_ctx_slot = slot;
        if (!h->ctx[slot].stream && hipStreamCreateWithFlags(&h->ctx[slot].stream, hipStreamNonBlocking) != hipSuccess) {
            release();
            return set_error(LYNSE_ERR_DEVICE, "hipStreamCreate failed");
        }
        return LYNSE_OK;
    }
    void release() {
        if (!h) return;
        tl_ctx_slot = prev;
        {
            std::lock_guard<std::mutex> lk(h->ctx_mu);
            h->ctx_busy &= ~(1u << slot);
        }
        h->ctx_cv.notify_one();
        h = nullptr;
    }
    ~CtxLease() { release(); }
};


// Writers — append / finalize / lazy builds / setters / the subset-filtered and large-k searches — hold `rw` EXCLUSIVE and
// refuse to start while tickets of lynse_hip_flat_search_submit_* are outstanding: a ticket does not keep the reader lock
// (it may be waited for by another thread, and one thread may hold several), it is counted in `inflight` under the
// shared lock, so a writer that owns the lock sees every ticket submitted before it.
static int writer_lock(lynse_hip_flat* h, std::unique_lock<std::shared_mutex>& lk) {
    lk = std::unique_lock<std::shared_mutex>(h->rw);
    if (h->inflight.load(std::memory_order_acquire) != 0) {
        lk.unlock();
        return set_error(LYNSE_ERR_INVALID_ARGUMENT, "searches are in flight on this handle: wait for the outstanding tickets first");
    }
    return LYNSE_OK;
}
#define LY_WRITER(h, lk) std::unique_lock<std::shared_mutex> lk; LY_TRY(writer_lock((h), lk))

// decides whether the search starting on this thread is a timed one (profiling on, and its turn at the sampling rate)
static bool profile_begin_search(lynse_hip_flat* h) {
    const uint32_t rate = h->prof_rate.load();
    tl_prof = h->profiling.load() && (rate <= 1 || h->prof_seq.fetch_add(1) % rate == 0);
    return tl_prof;
}

static int use_device(const lynse_hip_flat* h) {
    LY_HIP(hipSetDevice(h->device));
    return LYNSE_OK;
}

static uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

extern "C" int lynse_hip_flat_create(uint32_t dim, int device, lynse_hip_flat** out) {
    if (!out) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (dim == 0) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "dimension must be greater than zero");
    int ndev = 0;
    LY_TRY(lynse_hip_device_count(&ndev));
    if (ndev <= 0) return set_error(LYNSE_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    LY_HIP(hipSetDevice(device));
    auto* h = new lynse_hip_flat();
    h->dim = dim;
    h->ld = round_up(dim, 4);
    h->ld16 = round_up(dim, 8);
    // 48..63 and 96..127 columns: the f16 rows are padded (zeros) to a whole 64-element slab — at most a third more shadow bytes, and the
    // float batches of such a shard (deep-96, GloVe-100 ...) run the whole-slab kernels: k_scan_qh (scan_qh.h) and the non-ragged
    // k_scan_h16 variants.  LYNSE_HIP_SHADOW_PAD=0: the 8-element pitch (A/B, tests; read when the handle is created)
    if (((dim >= 48 && dim < 64) || (dim >= 96 && dim < 128)) && !(getenv("LYNSE_HIP_SHADOW_PAD") && atoi(getenv("LYNSE_HIP_SHADOW_PAD")) == 0))
        h->ld16 = round_up(dim, 64);
    h->ld8 = round_up(dim, 16);
    h->ld_bpm = round_up(dim, 256) / 2;
    h->ld8a = round_up(dim + 1, 128);
    while (h->aug_cols * 2 <= 32 && dim + h->aug_cols * 2 <= h->ld8a) h->aug_cols *= 2;
    h->words = (dim + 63) / 64;
    h->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        h->num_cu = prop.multiProcessorCount;
    if (const char* ce = getenv("LYNSE_HIP_CONTEXTS")) h->n_ctx = std::max(1, std::min((int)lynse_hip_flat::MAX_CTX, atoi(ce)));
    hipError_t e = hipStreamCreateWithFlags(&h->ctx[0].stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return set_error(LYNSE_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    e = hipMalloc(&h->d_stats, 5 * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipStreamDestroy(cur(h).stream);
        delete h;
        return set_error(LYNSE_ERR_OUT_OF_MEMORY, "hipMalloc(stats)");
    }
    const uint32_t init[5] = {0u, 0u, 0x7f800000u, 0u, 0u};
    (void)h2d_done(h->d_stats, init, sizeof init);
    *out = h;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_destroy(lynse_hip_flat* h) {
    if (!h) return LYNSE_OK;
    (void)hipSetDevice(h->device);
    for (auto& c : h->ctx) {
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        c.ws.release();
        if (c.d_mask) (void)hipFree(c.d_mask);
        if (c.d_subset) (void)hipFree(c.d_subset);
        for (auto e : c.ev_pool) (void)hipEventDestroy(e);
    }
    if (h->shadow_alias) h->rows16 = nullptr;
    for (void* p : {(void*)h->rows, (void*)h->rows_h, (void*)h->rows16, (void*)h->packed, (void*)h->vn2, (void*)h->vrinv, (void*)h->d_stats,
                    (void*)h->g_rows16, (void*)h->g_vn2, (void*)h->g_vrinv, (void*)h->g_ids32,
                    (void*)h->bpm, (void*)h->sq8c, (void*)h->sq8c_mins, (void*)h->sq8c_scales, (void*)h->sq8c_mm, (void*)h->sq8c_stats, (void*)h->sq8a, (void*)h->sq8a_mins, (void*)h->sq8a_scales, (void*)h->sq8a_mm, (void*)h->sq8a_stats,
                    (void*)h->sq8, (void*)h->sq8_mins, (void*)h->sq8_scales, (void*)h->sq8_sum, (void*)h->sq8_sum2, (void*)h->sq8_mm, (void*)h->sq8_stats, (void*)h->sq8_mm_prev})
        if (p) (void)hipFree(p);
    for (auto& c : h->ctx)
        if (c.stream) (void)hipStreamDestroy(c.stream);
    delete h;
    return LYNSE_OK;
}

static int realloc_rows_h(lynse_hip_flat* h, uint64_t cap) {   // F16 shard: capacity of the f16 row matrix
    _Float16* nr = nullptr;
    LY_HIP(hipMalloc(&nr, (size_t)cap * h->ld16 * sizeof(_Float16) + 256));
    if (h->rows_h && h->n)
        LY_HIP(hipMemcpyAsync(nr, h->rows_h, (size_t)h->n * h->ld16 * sizeof(_Float16), hipMemcpyDeviceToDevice, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    if (h->rows_h) (void)hipFree(h->rows_h);
    h->rows_h = nr;
    h->capacity = cap;
    if (h->shadow_alias) { h->rows16 = h->rows_h; h->cap16 = cap; }
    return LYNSE_OK;
}

static int grow_rows(lynse_hip_flat* h, uint64_t need) {
    if (need <= h->capacity) return LYNSE_OK;
    uint64_t cap = std::max<uint64_t>(need, h->capacity + h->capacity / 2);
    cap = std::max<uint64_t>(cap, 1024);
    if (is_f16(h)) return realloc_rows_h(h, cap);
    float* nr = nullptr;
    LY_HIP(hipMalloc(&nr, (size_t)cap * h->ld * sizeof(float)));
    if (h->rows && h->n)
        LY_HIP(hipMemcpyAsync(nr, h->rows, (size_t)h->n * h->ld * sizeof(float), hipMemcpyDeviceToDevice, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    if (h->rows) (void)hipFree(h->rows);
    h->rows = nr;
    h->capacity = cap;
    return LYNSE_OK;
}

static int grow_packed(lynse_hip_flat* h, uint64_t need) {
    if (need <= h->packed_capacity) return LYNSE_OK;
    uint64_t cap = std::max<uint64_t>(need, h->packed_capacity + h->packed_capacity / 2);
    cap = std::max<uint64_t>(cap, 1024);
    uint64_t* np = nullptr;
    LY_HIP(hipMalloc(&np, (size_t)cap * h->words * sizeof(uint64_t)));
    if (h->packed && h->n_packed)
        LY_HIP(hipMemcpyAsync(np, h->packed, (size_t)h->n_packed * h->words * 8, hipMemcpyDeviceToDevice, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    if (h->packed) (void)hipFree(h->packed);
    h->packed = np;
    h->packed_capacity = cap;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_reserve(lynse_hip_flat* h, uint64_t rows) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    if (rows > 0xfffffff0ull) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "row count exceeds the u32 id capacity of one shard");
    if (h->packed_only) return grow_packed(h, rows);
    if (rows > h->capacity && is_f16(h)) return realloc_rows_h(h, rows);
    if (rows > h->capacity) {
        // exact reservation (no 1.5x slack): callers reserve to size a shard to its HBM budget
        float* nr = nullptr;
        LY_HIP(hipMalloc(&nr, (size_t)rows * h->ld * sizeof(float)));
        if (h->rows && h->n)
            LY_HIP(hipMemcpyAsync(nr, h->rows, (size_t)h->n * h->ld * sizeof(float), hipMemcpyDeviceToDevice, cur(h).stream));
        LY_HIP(hipStreamSynchronize(cur(h).stream));
        if (h->rows) (void)hipFree(h->rows);
        h->rows = nr;
        h->capacity = rows;
    }
    return LYNSE_OK;
}

static int copy_rows_kernel(lynse_hip_flat* h, float* dst, uint32_t dp, const float* src, uint32_t sp, uint32_t width,
                            uint64_t n, int zero_pad) {
    const uint64_t total = n * (zero_pad ? dp : width);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, (uint64_t)h->num_cu * 32);
    hipLaunchKernelGGL(k_copy_rows, dim3(std::max<uint32_t>(blocks, 1)), dim3(256), 0, cur(h).stream, dst, dp, src, sp, width, n, zero_pad);
    LY_HIP(hipGetLastError());
    return LYNSE_OK;
}

// F16 shard: rows [first, first + n) decoded to f32 (exact) into dst (pitch dp floats)
static int decode_rows_kernel(lynse_hip_flat* h, float* dst, uint32_t dp, uint64_t first, uint64_t n) {
    const uint64_t total = n * h->dim;
    hipLaunchKernelGGL(k_f16_to_f32_rows, dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((total + 255) / 256, (uint64_t)h->num_cu * 32))), dim3(256), 0,
                       cur(h).stream, dst, dp, h->rows_h + (size_t)first * h->ld16, h->ld16, h->dim, n);
    LY_HIP(hipGetLastError());
    return LYNSE_OK;
}

constexpr uint64_t STAGE_BYTES = 64ull << 20;  // dense staging buffer for padded-layout host transfers

static int append_f32_impl(lynse_hip_flat* h, const float* src, uint64_t n, hipMemcpyKind kind) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    if (n == 0) return LYNSE_OK;
    if (!src) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "rows is NULL");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    if (h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "store holds packed rows; cannot append f32 rows");
    if (h->n + n > 0xfffffff0ull) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "row count exceeds the u32 id capacity of one shard");
    LY_TRY(grow_rows(h, h->n + n));
    if (is_f16(h)) {
        // f32 in (host or device) -> f16 bits (RNE): device rows are converted in place, host rows through a staging buffer
        _Float16* dsth = h->rows_h + (size_t)h->n * h->ld16;
        auto convert = [&](const float* d_src, uint64_t nr, uint64_t r) -> int {
            const uint64_t total = nr * h->ld16;
            hipLaunchKernelGGL(k_f32_to_f16_rows, dim3((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((total + 255) / 256, (uint64_t)h->num_cu * 32))), dim3(256), 0,
                               cur(h).stream, dsth + r * h->ld16, h->ld16, d_src, h->dim, h->dim, nr);
            LY_HIP(hipGetLastError());
            return LYNSE_OK;
        };
        if (kind == hipMemcpyDeviceToDevice) {
            LY_TRY(convert(src, n, 0));
        } else {
            const uint64_t chunk = std::max<uint64_t>(1, STAGE_BYTES / ((uint64_t)h->dim * 4));
            float* stage = nullptr;
            LY_HIP(hipMalloc(&stage, (size_t)std::min<uint64_t>(chunk, n) * h->dim * 4));
            for (uint64_t r = 0; r < n; r += chunk) {
                const uint64_t nr = std::min<uint64_t>(chunk, n - r);
                int rc = hipMemcpyAsync(stage, src + r * h->dim, (size_t)nr * h->dim * 4, kind, cur(h).stream) == hipSuccess ? convert(stage, nr, r)
                                                                                                                       : set_error(LYNSE_ERR_DEVICE, "host-to-device copy failed");
                if (rc == LYNSE_OK && hipStreamSynchronize(cur(h).stream) != hipSuccess) rc = set_error(LYNSE_ERR_DEVICE, "stream sync failed");
                if (rc != LYNSE_OK) { (void)hipFree(stage); return rc; }
            }
            (void)hipFree(stage);
        }
        LY_HIP(hipStreamSynchronize(cur(h).stream));
        h->n += n;
        return LYNSE_OK;
    }
    float* dst = h->rows + (size_t)h->n * h->ld;
    if (h->ld == h->dim) {
        LY_HIP(hipMemcpyAsync(dst, src, (size_t)n * h->dim * sizeof(float), kind, cur(h).stream));
    } else if (kind == hipMemcpyDeviceToDevice) {
        LY_TRY(copy_rows_kernel(h, dst, h->ld, src, h->dim, h->dim, n, 1));
    } else {
        const uint64_t chunk = std::max<uint64_t>(1, STAGE_BYTES / ((uint64_t)h->dim * 4));
        float* stage = nullptr;
        LY_HIP(hipMalloc(&stage, (size_t)std::min<uint64_t>(chunk, n) * h->dim * 4));
        for (uint64_t r = 0; r < n; r += chunk) {
            const uint64_t nr = std::min<uint64_t>(chunk, n - r);
            hipError_t e = hipMemcpyAsync(stage, src + r * h->dim, (size_t)nr * h->dim * 4, kind, cur(h).stream);
            int rc = e == hipSuccess ? copy_rows_kernel(h, dst + r * h->ld, h->ld, stage, h->dim, h->dim, nr, 1)
                                     : set_error(LYNSE_ERR_DEVICE, hipGetErrorString(e));
            if (rc == LYNSE_OK && hipStreamSynchronize(cur(h).stream) != hipSuccess) rc = set_error(LYNSE_ERR_DEVICE, "stream sync failed");
            if (rc != LYNSE_OK) { (void)hipFree(stage); return rc; }
        }
        (void)hipFree(stage);
    }
    LY_HIP(hipStreamSynchronize(cur(h).stream));   // (F16 shards returned above: they keep their rows as f16 bits only)
    h->n += n;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_set_dtype(lynse_hip_flat* h, int dtype) {
    if (!h || (dtype != LYNSE_DTYPE_F32 && dtype != LYNSE_DTYPE_F16)) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "bad dtype");
    LY_WRITER(h, lk);
    if (h->n || h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "the dtype is fixed once rows are stored");
    h->dtype = dtype;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_append_f16_bits(lynse_hip_flat* h, const uint16_t* rows, uint64_t n) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    if (h->dtype != LYNSE_DTYPE_F16) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "append_f16_bits needs an F16 shard (lynse_hip_flat_set_dtype)");
    if (n == 0) return LYNSE_OK;
    if (!rows) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "rows is NULL");
    // decode on the host in bounded chunks (f16 -> f32 is exact), then the ordinary append
    const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / ((uint64_t)h->dim * 4));
    std::vector<float> buf((size_t)std::min<uint64_t>(chunk, n) * h->dim);
    for (uint64_t r = 0; r < n; r += chunk) {
        const uint64_t nr = std::min<uint64_t>(chunk, n - r);
        const _Float16* src = reinterpret_cast<const _Float16*>(rows + r * h->dim);
        for (uint64_t i = 0; i < nr * h->dim; ++i) buf[i] = (float)src[i];
        LY_TRY(append_f32_impl(h, buf.data(), nr, hipMemcpyHostToDevice));
    }
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_append_f32(lynse_hip_flat* h, const float* rows, uint64_t n) {
    return append_f32_impl(h, rows, n, hipMemcpyHostToDevice);
}
extern "C" int lynse_hip_flat_append_f32_device(lynse_hip_flat* h, const float* d_rows, uint64_t n) {
    return append_f32_impl(h, d_rows, n, hipMemcpyDeviceToDevice);
}

static int append_packed_impl(lynse_hip_flat* h, const uint64_t* src, uint64_t n, hipMemcpyKind kind) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    if (n == 0) return LYNSE_OK;
    if (!src) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "words is NULL");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    if (h->n > 0 && !h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "store holds f32 rows; cannot append packed rows");
    if (h->n + n > 0xfffffff0ull) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "row count exceeds the u32 id capacity of one shard");
    h->packed_only = true;
    LY_TRY(grow_packed(h, h->n + n));
    LY_HIP(hipMemcpyAsync(h->packed + (size_t)h->n * h->words, src, (size_t)n * h->words * 8, kind, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    h->n += n;
    h->n_packed = h->n;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_append_packed_u64(lynse_hip_flat* h, const uint64_t* words, uint64_t n) {
    return append_packed_impl(h, words, n, hipMemcpyHostToDevice);
}
extern "C" int lynse_hip_flat_append_packed_u64_device(lynse_hip_flat* h, const uint64_t* d_words, uint64_t n) {
    return append_packed_impl(h, d_words, n, hipMemcpyDeviceToDevice);
}

// Row statistics for rows [n_stats, n): norms + collection stats (locked by caller).
static int ensure_shadow_locked(lynse_hip_flat* h);
static bool shadow_ready(const lynse_hip_flat* h) { return h->packed_only || h->n == 0 || (h->rows16 && h->n16 == h->n && h->sv16 == h->sv); }
constexpr int LY_RESTART_EXCLUSIVE = 10001;   // search_impl_once: the f16 shadow is missing and only a shared lock is held — run again under the writer lock
static int scan_variant();

static int finalize_locked(lynse_hip_flat* h) {
    if (h->packed_only || h->n_stats == h->n) return LYNSE_OK;
    if (h->n > h->stats_capacity) {
        uint64_t cap = std::max<uint64_t>(h->n, h->capacity);
        float *a = nullptr, *b = nullptr;
        // +256: the scan kernel's per-tile norm DMA reads a whole 256-row window
        LY_HIP(hipMalloc(&a, ((size_t)cap + 256) * sizeof(float)));
        LY_HIP(hipMalloc(&b, ((size_t)cap + 256) * sizeof(float)));
        LY_HIP(hipMemsetAsync(a, 0, ((size_t)cap + 256) * sizeof(float), cur(h).stream));
        LY_HIP(hipMemsetAsync(b, 0, ((size_t)cap + 256) * sizeof(float), cur(h).stream));
        if (h->n_stats) {
            LY_HIP(hipMemcpyAsync(a, h->vn2, (size_t)h->n_stats * 4, hipMemcpyDeviceToDevice, cur(h).stream));
            LY_HIP(hipMemcpyAsync(b, h->vrinv, (size_t)h->n_stats * 4, hipMemcpyDeviceToDevice, cur(h).stream));
            LY_HIP(hipStreamSynchronize(cur(h).stream));
        }
        if (h->vn2) (void)hipFree(h->vn2);
        if (h->vrinv) (void)hipFree(h->vrinv);
        h->vn2 = a;
        h->vrinv = b;
        h->stats_capacity = cap;
    }
    const uint64_t nnew = h->n - h->n_stats;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((nnew + 3) / 4, (uint64_t)h->num_cu * 16);
    if (is_f16(h))
        hipLaunchKernelGGL(k_row_stats<_Float16>, dim3(blocks), dim3(256), 0, cur(h).stream, h->rows_h, h->ld16, h->dim,
                           (uint32_t)h->n_stats, (uint32_t)h->n, h->vn2, h->vrinv, h->d_stats);
    else
        hipLaunchKernelGGL(k_row_stats<float>, dim3(blocks), dim3(256), 0, cur(h).stream, h->rows, h->ld, h->dim,
                           (uint32_t)h->n_stats, (uint32_t)h->n, h->vn2, h->vrinv, h->d_stats);
    LY_HIP(hipGetLastError());
    uint32_t st[5];
    LY_HIP(hipMemcpyAsync(st, h->d_stats, sizeof st, hipMemcpyDeviceToHost, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    float f[3];
    memcpy(f, st, sizeof f);
    h->amax = f[0];
    h->vmax = std::sqrt(f[1]);
    h->vmin = (st[2] == 0x7f800000u) ? 0.0f : std::sqrt(f[2]);
    h->cos_degenerate = st[3] ? 1 : 0;
    h->rows_integer = ((st[4] & 1u) == 0u && getenv("LYNSE_HIP_NO_EXACT_INT") == nullptr) ? 1 : 0;
    h->rows_nonneg = (st[4] & 2u) == 0u ? 1 : 0;
    int e = (h->amax > 0.0f && std::isfinite(h->amax)) ? std::ilogb(h->amax) : 13;
    e = std::max(-100, std::min(100, e));
    // no scaling when max|v| in [2^-6, 2^15): the f16 subnormal floor (2^-25 absolute) is then far
    // below u*|v| for every element that matters, and the hot loop saves the multiply
    h->sv = (e >= -6 && e <= 14) ? 1.0f : std::ldexp(1.0f, 13 - e);
    h->n_stats = h->n;
    // (the f16 shadow is a LAZY copy since round 4 — ensure_shadow_locked, built by the first search that runs the f16 coarse pass: a
    // shard whose batches all take the certified int8 pass never pays its 2 B per element: 10M x 768 = 15.4 GB of the 53.9 GB)
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_finalize(lynse_hip_flat* h) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    return finalize_locked(h);
}

// ensure_binary (flat_mmap.rs:388-401): lazily pack f32 rows [n_packed, n).
static int ensure_packed_locked(lynse_hip_flat* h) {
    if (h->packed_only || h->n_packed == h->n) return LYNSE_OK;
    LY_TRY(grow_packed(h, std::max<uint64_t>(h->n, h->capacity)));
    const uint64_t nnew = h->n - h->n_packed;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((nnew + 3) / 4, (uint64_t)h->num_cu * 16);
    if (is_f16(h))
        hipLaunchKernelGGL(k_pack_bits<_Float16>, dim3(blocks), dim3(256), 0, cur(h).stream,
                           h->rows_h + (size_t)h->n_packed * h->ld16, h->ld16, h->dim, (uint32_t)nnew,
                           h->packed + (size_t)h->n_packed * h->words, h->words);
    else
        hipLaunchKernelGGL(k_pack_bits<float>, dim3(blocks), dim3(256), 0, cur(h).stream,
                           h->rows + (size_t)h->n_packed * h->ld, h->ld, h->dim, (uint32_t)nnew,
                           h->packed + (size_t)h->n_packed * h->words, h->words);
    LY_HIP(hipGetLastError());
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    h->n_packed = h->n;
    return LYNSE_OK;
}

// Batched Hamming on the int8 MFMA (kernels.h, k_bits_to_pm1): batches of at least LYNSE_HIP_BIN_MFMA_MINQ (default 72)
// queries, unfiltered, Hamming only (Jaccard / Dice need the row popcounts in the score: the popcount kernels keep them).
static uint32_t bin_mfma_minq() {
    static const uint32_t v = []() { const char* e = getenv("LYNSE_HIP_BIN_MFMA_MINQ"); return e ? (uint32_t)atoi(e) : 72u; }();
    return v;   // (the MFMA pass costs the same for 33 or 256 queries — 1.9 ms at 12.5M x 1024 bits —, the popcount kernel ~27 us per query: they cross near 70)
}
static bool bin_mfma_eligible(const lynse_hip_flat* h, int metric, bool filtered, uint64_t nq) {
    return metric == M_HAMMING && !filtered && bin_mfma_minq() > 0 && nq >= bin_mfma_minq() && nq > SCAN_BQ_SMALL && !h->bpm_failed &&
           h->n >= 65536 && scan_variant() == 3;
}
static int ensure_bpm_locked(lynse_hip_flat* h) {
    if (h->bpm_failed || (h->bpm && h->n_bpm == h->n)) return LYNSE_OK;
    LY_TRY(ensure_packed_locked(h));
    if (h->bpm_cap < h->n) {
        const uint64_t cap = std::max<uint64_t>(h->n, std::max<uint64_t>(h->capacity, h->packed_capacity));
        uint8_t* nb = nullptr;
        if (hipMalloc(&nb, (size_t)cap * h->ld_bpm + 256) != hipSuccess) {  // no room for the 4x copy: not an error
            (void)hipGetLastError();
            h->bpm_failed = true;
            return LYNSE_OK;
        }
        if (h->bpm && h->n_bpm)
            LY_HIP(hipMemcpyAsync(nb, h->bpm, (size_t)h->n_bpm * h->ld_bpm, hipMemcpyDeviceToDevice, cur(h).stream));
        LY_HIP(hipStreamSynchronize(cur(h).stream));
        if (h->bpm) (void)hipFree(h->bpm);
        h->bpm = nb;
        h->bpm_cap = cap;
    }
    const uint64_t pieces = (h->n - h->n_bpm) * (h->ld_bpm / 16);
    hipLaunchKernelGGL(k_bits_to_fp4, dim3((uint32_t)std::min<uint64_t>((pieces + 255) / 256, (uint64_t)h->num_cu * 32)), dim3(256), 0, cur(h).stream,
                       h->packed, h->words, h->dim, h->n_bpm, h->n, h->bpm, h->ld_bpm);
    LY_HIP(hipGetLastError());
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    h->n_bpm = h->n;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_set_row_map(lynse_hip_flat* h, uint64_t stride, uint64_t offset) {
    if (!h || stride == 0) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "bad row map");
    LY_WRITER(h, lk);
    h->row_stride = stride;
    h->row_offset = offset;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_set_ip_form(lynse_hip_flat* h, int f) {
    if (!h || f < LYNSE_IPFORM_AUTO || f > LYNSE_IPFORM_BATCH8) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "bad ip form");
    LY_WRITER(h, lk);
    h->ip_form = f;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_set_fused_search(lynse_hip_flat* h, int on) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    LY_WRITER(h, lk);
    h->no_fused = on ? 0 : 1;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_set_plan(lynse_hip_flat* h, uint32_t stage0_rows, uint32_t growth, uint32_t cap) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    if (cap < 256 || cap > 16384 || (cap & (cap - 1))) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "cap must be a power of two in [256,16384]");
    if (stage0_rows == 0 || stage0_rows > cap || growth < 2) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "bad stage plan");
    LY_WRITER(h, lk);
    (void)hipSetDevice(h->device);
    if (cap != h->cap)
        for (auto& c : h->ctx) c.ws.release();
    h->stage0_rows = stage0_rows;
    h->growth = growth;
    h->cap = cap;
    return LYNSE_OK;
}

extern "C" uint64_t lynse_hip_flat_len(const lynse_hip_flat* h) { return h ? h->n : 0; }
extern "C" uint32_t lynse_hip_flat_dim(const lynse_hip_flat* h) { return h ? h->dim : 0; }
extern "C" int lynse_hip_flat_device(const lynse_hip_flat* h) { return h ? h->device : -1; }

extern "C" int lynse_hip_flat_read_rows(const lynse_hip_flat* hc, uint64_t first, uint64_t n, float* out) {
    auto* h = const_cast<lynse_hip_flat*>(hc);
    if (!h || (!out && n)) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    std::unique_lock<std::shared_mutex> lk(h->rw);
    LY_TRY(use_device(h));
    if (h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "store holds packed rows only");
    if (first + n > h->n) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "row range out of bounds");
    if (n == 0) return LYNSE_OK;
    if (!is_f16(h) && h->ld == h->dim) {
        LY_HIP(hipMemcpyAsync(out, h->rows + (size_t)first * h->ld, (size_t)n * h->dim * 4, hipMemcpyDeviceToHost, cur(h).stream));
        LY_HIP(hipStreamSynchronize(cur(h).stream));
        return LYNSE_OK;
    }
    const uint64_t chunk = std::max<uint64_t>(1, STAGE_BYTES / ((uint64_t)h->dim * 4));
    float* stage = nullptr;
    LY_HIP(hipMalloc(&stage, (size_t)std::min<uint64_t>(chunk, n) * h->dim * 4));
    for (uint64_t r = 0; r < n; r += chunk) {
        const uint64_t nr = std::min<uint64_t>(chunk, n - r);
        int rc = is_f16(h) ? decode_rows_kernel(h, stage, h->dim, first + r, nr)
                           : copy_rows_kernel(h, stage, h->dim, h->rows + (size_t)(first + r) * h->ld, h->ld, h->dim, nr, 0);
        if (rc == LYNSE_OK && hipMemcpyAsync(out + r * h->dim, stage, (size_t)nr * h->dim * 4, hipMemcpyDeviceToHost, cur(h).stream) != hipSuccess)
            rc = set_error(LYNSE_ERR_DEVICE, "device-to-host copy failed");
        if (rc == LYNSE_OK && hipStreamSynchronize(cur(h).stream) != hipSuccess) rc = set_error(LYNSE_ERR_DEVICE, "stream sync failed");
        if (rc != LYNSE_OK) { (void)hipFree(stage); return rc; }
    }
    (void)hipFree(stage);
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_copy_rows_device(const lynse_hip_flat* hc, uint64_t first, uint64_t n, float* d_out) {
    auto* h = const_cast<lynse_hip_flat*>(hc);
    if (!h || (!d_out && n)) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    std::unique_lock<std::shared_mutex> lk(h->rw);
    LY_TRY(use_device(h));
    if (h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "store holds packed rows only");
    if (first + n > h->n) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "row range out of bounds");
    if (n == 0) return LYNSE_OK;
    if (is_f16(h))
        LY_TRY(decode_rows_kernel(h, d_out, h->dim, first, n));
    else if (h->ld == h->dim)
        LY_HIP(hipMemcpyAsync(d_out, h->rows + (size_t)first * h->ld, (size_t)n * h->dim * 4, hipMemcpyDeviceToDevice, cur(h).stream));
    else
        LY_TRY(copy_rows_kernel(h, d_out, h->dim, h->rows + (size_t)first * h->ld, h->ld, h->dim, n, 0));
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_read_packed(lynse_hip_flat* h, uint64_t first, uint64_t n, uint64_t* out) {
    if (!h || (!out && n)) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    if (first + n > h->n) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "row range out of bounds");
    LY_TRY(ensure_packed_locked(h));
    if (n == 0) return LYNSE_OK;
    LY_HIP(hipMemcpyAsync(out, h->packed + (size_t)first * h->words, (size_t)n * h->words * 8, hipMemcpyDeviceToHost, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));
    return LYNSE_OK;
}

// ---------------------------------------------------------------------------------- profiling ----
extern "C" int lynse_hip_flat_profile_enable(lynse_hip_flat* h, int on) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    std::unique_lock<std::shared_mutex> lk(h->rw);
    h->profiling = on != 0;
    h->prof_rate = on > 1 ? (uint32_t)on : 1u;   // on = n > 1: every n-th search is timed (HIP events between the kernels cost a few us each)
    h->prof_seq.store(0);
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_profile_get(lynse_hip_flat* h, lynse_hip_profile* out, int reset) {
    if (!h || !out) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    std::unique_lock<std::shared_mutex> lk(h->rw);
    LY_TRY(use_device(h));
    unsigned long long pool_sum = 0;
    for (auto& c : h->ctx) {  // every search context counts its own rescored candidates
        if (!c.ws.pool_total) continue;
        unsigned long long pool = 0;
        LY_HIP(hipStreamSynchronize(c.stream));
        LY_HIP(hipMemcpy(&pool, c.ws.pool_total, 8, hipMemcpyDeviceToHost));
        pool_sum += pool;
        if (reset) LY_TRY(memset_done(c.ws.pool_total, 0, 8));
    }
    h->prof.pool_entries = pool_sum;
    *out = h->prof;
    if (reset) h->prof = lynse_hip_profile{};
    return LYNSE_OK;
}

// ------------------------------------------------------------------------------------- search ----
constexpr uint32_t QCHUNK = 256;
constexpr int SEL_NT = 512;
constexpr uint32_t SEG_KEYS = 32768;  // keys per query in the segmented-emission buffer
constexpr uint32_t SEG_MAX = 2048;    // most segments per query (grid x row-waves: 512 x 4 for the <= 32-query kernel)

// segments of one scan launch: grid x WR of min(255, SEG_KEYS / nseg) slots
static void seg_geometry(uint32_t grid, uint32_t wr, uint32_t* nseg, uint32_t* seg) {
    *nseg = grid * wr;
    *seg = 0;
    static const int off = []() { const char* e = getenv("LYNSE_HIP_NO_SEGMENTS"); return e ? atoi(e) : 0; }();
    if (off || *nseg == 0 || *nseg > SEG_MAX) { *nseg = 0; return; }
    *seg = std::min<uint32_t>(255u, SEG_KEYS / *nseg);
}

static size_t scan_lds_bytes(int bq) { return (size_t)2 * (SCAN_BR + bq) * SCAN_LDK * sizeof(_Float16); }

template <typename K>
static int set_max_lds(K kernel, size_t bytes) {
    LY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return LYNSE_OK;
}

static int ensure_workspace(lynse_hip_flat* h, uint32_t k /* caller's k = output stride */) {
    Workspace& w = cur(h).ws;
    const uint32_t nslab = (h->dim + SCAN_BK - 1) / SCAN_BK;
    const uint32_t QC = std::max<uint32_t>(h->qchunk, QCHUNK);
    if (w.cand && w.cap == h->cap && w.D == h->dim && w.kcap >= k && w.qcap == QC) return LYNSE_OK;
    w.release();
    w.qcap = QC;
    w.cap = h->cap;
    w.D = h->dim;
    w.W = h->words;
    w.kcap = std::max<uint32_t>(k, 128);
    LY_HIP(hipMalloc(&w.cand, (size_t)QC * w.cap * 8));
    LY_HIP(hipMalloc(&w.candB, (size_t)QCHUNK * SEG_KEYS * 8));
    LY_HIP(hipMalloc(&w.segcnt, (size_t)QCHUNK * SEG_MAX));
    LY_HIP(hipMalloc(&w.count, (size_t)QC * 4));
    LY_HIP(hipMalloc(&w.thr, (size_t)QC * 4));
    LY_HIP(hipMalloc(&w.qinv, (size_t)QC * 4));
    LY_HIP(hipMalloc(&w.qn2, (size_t)QC * 4));
    LY_HIP(hipMalloc(&w.qrinv, (size_t)QC * 4));
    LY_HIP(hipMalloc(&w.marg2, (size_t)QC * 4));
    w.q16_halves = (size_t)nslab * QC * SCAN_LDK;
    LY_HIP(hipMalloc(&w.Q16, w.q16_halves * sizeof(_Float16)));
    LY_TRY(memset_done(w.Q16, 0, w.q16_halves * sizeof(_Float16)));
    LY_HIP(hipMalloc(&w.Qf, (size_t)QC * h->dim * 4));
    LY_HIP(hipMalloc(&w.QW, (size_t)QCHUNK * h->words * 8));
    {
        uint32_t wcap = 1;
        while (wcap < h->words) wcap <<= 1;
        LY_HIP(hipMalloc(&w.QWp, (size_t)QCHUNK * wcap * 8));
    }
    LY_HIP(hipMalloc(&w.out_rows, (size_t)QC * w.kcap * 8));
    LY_HIP(hipMalloc(&w.out_dists, (size_t)QC * w.kcap * 4));
    LY_HIP(hipMalloc(&w.out_counts, (size_t)2 * QC * 4));
    w.overflow = w.out_counts + QC;
    LY_HIP(hipHostMalloc(&w.h_hdr, (size_t)2 * QC * 4, hipHostMallocDefault));
    LY_HIP(hipHostMalloc(&w.h_out, H_OUT_BYTES, hipHostMallocDefault));
    LY_HIP(hipMalloc(&w.pool_total, 8));
    LY_TRY(memset_done(w.pool_total, 0, 8));
    LY_HIP(hipMalloc(&w.dyn, (size_t)QC * 34 * sizeof(int)));
    LY_HIP(hipMalloc(&w.gsync, 256 + 1024 * 64));   // hand-over words + (debugging) 8 time stamps per workgroup / per (stage, query)
    LY_TRY(memset_done(w.gsync, 0, 256 + 1024 * 64));
    LY_HIP(hipMalloc(&w.small_part, (size_t)SMALL_NT * SMALL_MAX_Q * SMALL_MAX_K * 8));
    LY_HIP(hipMalloc(&w.small_ticket, 16));
    LY_TRY(memset_done(w.small_ticket, 0, 16));
    return LYNSE_OK;
}

struct Stage {
    uint32_t r0, r1;
    uint32_t sample_tiles = 0, sample_stride = 0;  // sample stage: sample_tiles tiles of tile_rows rows at t * sample_stride
};

// level 0: sampled plan (k_scan_h16 only) — stage 0 scans `cap/2` rows taken as evenly spread tiles, so the first
//          threshold is representative of the whole shard whatever the insertion order (a collection sorted by
//          similarity to the query would overflow every stage of the contiguous plan); the following stages grow
//          by min(16, cap/(8k)) — measured best on 10M and 1.25M rows (larger growth: fewer launches but thousands
//          of emitted keys per query in the first full stage);
// level 1: contiguous plan [0,S0) [S0,8*S0) ... (all other kernels; retry after an overflow of level 0);
// level 2: exhaustive plan, stages of cap/2 rows — cannot overflow.
static std::vector<Stage> make_plan(const lynse_hip_flat* h, uint32_t k, int level, uint32_t tile_rows, bool threshold_only_sample = false, uint32_t growth_hint = 0) {
    std::vector<Stage> plan;
    const uint64_t n = h->n;
    if (n == 0) return plan;
    if (level >= 2) {
        const uint32_t step = h->cap / 2;
        for (uint64_t r = 0; r < n; r += step) plan.push_back({(uint32_t)r, (uint32_t)std::min<uint64_t>(n, r + step)});
        return plan;
    }
    if (level == 0 && tile_rows && n > 4ull * h->cap) {
        static const uint32_t s_env = []() { const char* e = getenv("LYNSE_HIP_SAMPLE_ROWS"); return e ? (uint32_t)atoi(e) : 0u; }();
        // emit-all sample: every sample row becomes a key (S <= cap / 2).  Threshold-only sample: 16 keys per tile, and a
        // tile per CU costs the same latency as 32 tiles — a larger sample gives a tighter first threshold for free
        // (its rows are scanned again: keep it <= n / 16).
        uint32_t S = s_env ? std::min<uint32_t>(s_env, h->cap / 2) : h->cap / 2;
        if (threshold_only_sample) {
            static const uint32_t big_env = []() { const char* e = getenv("LYNSE_HIP_SAMPLE_ROWS_TO"); return e ? (uint32_t)atoi(e) : 0u; }();
            const uint64_t want = big_env ? big_env : (uint64_t)h->num_cu * tile_rows;
            S = (uint32_t)std::max<uint64_t>(S, std::min<uint64_t>(std::min<uint64_t>(want, big_env ? n / 4 : n / 16), (uint64_t)h->cap / 16 * tile_rows));
            S = S / tile_rows * tile_rows;
        }
        const uint32_t nt = S / tile_rows;
        const uint64_t stride = (n - tile_rows) / (nt - 1) / tile_rows * tile_rows;
        if (stride > tile_rows) {
            Stage s0{0u, (uint32_t)n};
            s0.sample_tiles = nt;
            s0.sample_stride = (uint32_t)stride;
            plan.push_back(s0);
            static const uint64_t genv = []() { const char* e = getenv("LYNSE_HIP_SAMPLE_GROWTH"); return e ? (uint64_t)atoi(e) : 0ull; }();
            const uint64_t gmax = genv ? genv : growth_hint ? growth_hint : (threshold_only_sample ? 32ull : 16ull);  // measured best on 10M and 1.25M rows
            const uint64_t g = std::max<uint64_t>(2, std::min<uint64_t>(gmax, h->cap / (8ull * std::max<uint32_t>(k, 1))));
            uint64_t seen = S, b = 0;
            while (b < n) {
                uint64_t nb = std::min<uint64_t>(n, (seen * g + tile_rows - 1) / tile_rows * tile_rows);
                if (n - nb < nb / 8) nb = n;  // do not leave a sliver for an extra launch
                plan.push_back({(uint32_t)b, (uint32_t)nb});
                seen = nb;
                b = nb;
            }
            return plan;
        }
    }
    uint64_t s0 = std::max<uint64_t>(h->stage0_rows, std::min<uint64_t>(h->cap, 4ull * k));
    s0 = (s0 + 255) / 256 * 256;  // stage boundaries on 256-row tiles (mask words, tile grid of the sample stage)
    s0 = std::min<uint64_t>(std::min<uint64_t>(s0, h->cap), n);
    uint64_t g = std::max<uint64_t>(2, std::min<uint64_t>(h->growth, h->cap / (4ull * std::max<uint32_t>(k, 1))));
    uint64_t b = s0;
    plan.push_back({0u, (uint32_t)b});
    while (b < n) {
        uint64_t nb = std::min<uint64_t>(n, b * g);
        plan.push_back({(uint32_t)b, (uint32_t)nb});
        b = nb;
    }
    return plan;
}

#ifdef LYNSE_EXPERIMENTS  // launchers of the superseded scan kernels (A/B references)
template <int WQ, int WR, int TQ, int TR, int PD>
static int launch_scan(lynse_hip_flat* h, const ScanArgs& a, int metric, uint32_t grid, hipStream_t st) {
    constexpr int NT = WQ * WR * 64;
    constexpr int BQ = WQ * TQ * 32;
    const size_t lds = scan_lds_bytes(BQ);
    static std::atomic<bool> attr_done[3] = {false, false, false};
    auto go = [&](auto kern, int slot) -> int {
        if (!attr_done[slot]) {
            LY_TRY(set_max_lds(kern, lds));
            attr_done[slot] = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    (void)h;
    switch (metric) {
    case M_IP: return go(k_scan_f16<WQ, WR, TQ, TR, M_IP, PD>, 0);
    case M_L2: return go(k_scan_f16<WQ, WR, TQ, TR, M_L2, PD>, 1);
    default: return go(k_scan_f16<WQ, WR, TQ, TR, M_COS, PD>, 2);
    }
}

template <int WQ, int WR, int TQ, int TR, int NS>
static int launch_scan_glds(const ScanArgs& a, int metric, bool scale, uint32_t grid, hipStream_t st) {
    constexpr int NT = WQ * WR * 64;
    constexpr int BQ = WQ * TQ * 32;
    constexpr int BR = WR * TR * 32;
    const size_t lds = (size_t)NS * (BR * GL_BK * 4 + BQ * GL_BK * 2) + (metric == M_IP ? 0 : (size_t)NS * 1024);  // + per-tile norm ring
    static std::atomic<bool> attr_done[12] = {false};
    static const int nt_hint = []() { const char* e = getenv("LYNSE_HIP_SCAN_NT"); return e ? atoi(e) : 1; }();
    auto go = [&](auto kern, int slot) -> int {
        if (!attr_done[slot]) {
            LY_TRY(set_max_lds(kern, lds));
            attr_done[slot] = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
#define LY_GO(M, S, H, slot) return go(k_scan_glds<WQ, WR, TQ, TR, M, S, NS, H>, slot)
    if (nt_hint) {
        if (scale) {
            switch (metric) { case M_IP: LY_GO(M_IP, true, 2, 0); case M_L2: LY_GO(M_L2, true, 2, 1); default: LY_GO(M_COS, true, 2, 2); }
        }
        switch (metric) { case M_IP: LY_GO(M_IP, false, 2, 3); case M_L2: LY_GO(M_L2, false, 2, 4); default: LY_GO(M_COS, false, 2, 5); }
    }
    if (scale) {
        switch (metric) { case M_IP: LY_GO(M_IP, true, 0, 6); case M_L2: LY_GO(M_L2, true, 0, 7); default: LY_GO(M_COS, true, 0, 8); }
    }
    switch (metric) { case M_IP: LY_GO(M_IP, false, 0, 9); case M_L2: LY_GO(M_L2, false, 0, 10); default: LY_GO(M_COS, false, 0, 11); }
#undef LY_GO
}

#endif  // LYNSE_EXPERIMENTS

// debugging only (not in the header): per-query thresholds / shared-region counts / overflow flags and the hand-over words of
// context 0's workspace
extern "C" int lynse_hip_debug_workspace(lynse_hip_flat* h, float* thr, uint32_t* count, uint32_t* overflow, uint32_t* gsync, uint32_t nq) {
    if (!h || !h->ctx[0].ws.thr) return 1;
    Workspace& w = h->ctx[0].ws;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->ctx[0].stream);
    if (thr && hipMemcpy(thr, w.thr, nq * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    if (count && hipMemcpy(count, w.count, nq * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    if (overflow && hipMemcpy(overflow, w.overflow, nq * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    if (gsync && hipMemcpy(gsync, w.gsync, 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    return 0;
}
extern "C" int lynse_hip_debug_fs_stamps(lynse_hip_flat* h, unsigned long long* out /* [512][8] */) {
    if (!h || !h->ctx[0].ws.gsync) return 1;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->ctx[0].stream);
    return hipMemcpy(out, h->ctx[0].ws.gsync + 64, 512 * 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
extern "C" int lynse_hip_debug_sel_stamps(lynse_hip_flat* h, unsigned long long* out /* [4][256][8] */) {
    if (!h || !h->ctx[0].ws.gsync) return 1;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->ctx[0].stream);
    return hipMemcpy(out, h->ctx[0].ws.gsync + 64, 1024 * 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}

static unsigned long long* g_dbg_ptr = nullptr;
extern "C" int lynse_hip_debug_phase_cycles(unsigned long long* out, int n) {  // experiments only (not in the header)
    if (!g_dbg_ptr) return 1;
    return hipMemcpy(out, g_dbg_ptr, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}

// 3 = LDS-DMA ring kernel over the f16 shadow rows (default), 0 = LDS-DMA ring kernel over the f32 rows,
// 1 / 2 = register-staged kernel with prefetch depth 1 / 2
static int scan_variant() {
#ifdef LYNSE_EXPERIMENTS
    static const int v = []() { const char* e = getenv("LYNSE_HIP_SCAN_VARIANT"); return e ? atoi(e) : 3; }();
    return v;
#else
    return 3;  // the superseded variants are only compiled with make EXPERIMENTS=1
#endif
}

// k_scan_h16 launcher: LDS = NSV row stages + NSQ query stages + the norm ring (NSV+1 slots of 1 KiB) when it fits.
// The 256 x 256 tilings compile ONE epilogue per emission mode (EMIT template parameter: 0 threshold stages, 1 emit-all
// first stage, 2 lane-max sample stage — one body with all three spills, kernels.h) and only the tiling / metric pairs
// the product path uses: <2,4,4,2> for unfiltered IP, <4,2,2,4> for L2 / cosine and every subset-filtered scan.
// The <= 32-query kernel and the IVF work-list kernel (<1,4,1,1>) keep the runtime emission switch (no spills there).
template <int WQ, int WR, int TQ, int TR, int NSV, int NSQ, bool TILED>
static int launch_scan_h16(const ScanArgs& a, int metric, uint32_t grid, hipStream_t st, uint32_t grid_y = 1) {
    constexpr int BQ = WQ * TQ * 32, BR = WR * TR * 32;
    constexpr size_t rings = (size_t)(NSV * BR + NSQ * BQ) * (HK * 2);
    const size_t lds = (rings + (NSV + 1) * 1024 <= 160 * 1024) ? rings + (NSV + 1) * 1024 : rings;
    static std::atomic<bool> attr_done[64] = {false};
    auto go = [&](auto kern, int slot) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid, grid_y), dim3(WQ * WR * 64), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    const bool filt = a.mask != nullptr || a.row_ids != nullptr;
    const bool ragged = a.ld16 % HK != 0;
#ifdef LYNSE_EXPERIMENTS
    if constexpr (WQ == 2 && WR == 4 && !TILED) {  // timing experiments (LYNSE_HIP_DEBUG_FLAGS bits 8..11 = DBG << 8)
        if (metric == M_IP && !filt && !ragged && ((a.debug_flags >> 8) & 15)) {
            auto ex = [&](auto kern) -> int {
                LY_TRY(set_max_lds(kern, lds));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(WQ * WR * 64), lds, st, a);
                return LYNSE_OK;
            };
            switch ((a.debug_flags >> 8) & 15) {
            case 1: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 1>)); break;
            case 2: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 2>)); break;
            case 3: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 3>)); break;
            case 7: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 7>)); break;
            case 11: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 11>)); break;
            case 12: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 12>)); break;
            case 15: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 15>)); break;
            case 14: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 14>)); break;
            case 13: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 13>)); break;
            case 4: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 4>)); break;
            case 8: LY_TRY(ex(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false, 8>)); break;
            default: return set_error(LYNSE_ERR_INVALID_ARGUMENT, "unknown experiment");
            }
            LY_HIP(hipGetLastError());
            return LYNSE_OK;
        }
    }
#endif
#ifdef LYNSE_EXPERIMENTS
    if constexpr (WQ == 4 && WR == 2 && TR == 4 && !TILED) {  // decomposition of the low-D L2 scan (C3): LYNSE_HIP_DEBUG_FLAGS bits 8..12 = DBG << 8
        const int dbg = (a.debug_flags >> 8) & 31;
        if (metric == M_L2 && !filt && !ragged && dbg && a.emit_all == 0) {
            auto ex = [&](auto kern) -> int {
                LY_TRY(set_max_lds(kern, lds));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(WQ * WR * 64), lds, st, a);
                LY_HIP(hipGetLastError());
                return LYNSE_OK;
            };
            switch (dbg) {
            case 16: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 16, false, 0, 0, 0, true>);   // no epilogue
            case 17: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 17, false, 0, 0, 0, true>);   // + no MFMA (DMA + LDS reads)
            case 19: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 19, false, 0, 0, 0, true>);   // DMA only
            case 20: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 20, false, 0, 0, 0, true>);   // no epilogue, no query DMA
            case 24: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 24, false, 0, 0, 0, true>);   // no epilogue, no row DMA
            case 28: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 28, false, 0, 0, 0, true>);   // MFMA + LDS reads only
            case 30: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 30, false, 0, 0, 0, true>);   // MFMA only
            case 12: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 12, false, 0, 0, 0, true>);   // MFMA + LDS reads + epilogue (no DMA)
            case 4: return ex(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, false, false, 4, false, 0, 0, 0, true>);     // everything but the query DMA
            default: return set_error(LYNSE_ERR_INVALID_ARGUMENT, "unknown experiment");
            }
        }
    }
#endif
    if constexpr (WQ * WR == 8 && !TILED) {
        // one kernel per (metric, ragged / filter variant, emission mode)
        auto by_emit = [&](auto mtag, auto ragtag, auto filtag, int base) -> int {
            constexpr int M = decltype(mtag)::value;
            constexpr bool RG = decltype(ragtag)::value, FL = decltype(filtag)::value;
            if constexpr (WR < 4 && !FL && M != M_IP) {  // many survivors per block expected: one-pass epilogue (kernels.h, DENSE)
                if (a.emit_all == 0 && a.dense) return go(k_scan_h16<WQ, WR, TQ, TR, M, NSV, NSQ, 2, false, RG, 0, FL, 0, 0, 0, true>, base + 27 + (RG ? 1 : 0));
            }
            if (a.emit_all == 0) return go(k_scan_h16<WQ, WR, TQ, TR, M, NSV, NSQ, 2, false, RG, 0, FL, 0, 0>, base);
            if constexpr (!FL) {  // (the lane-max sample stage exists for the unfiltered tilings only)
                if (a.emit_all == 2) return go(k_scan_h16<WQ, WR, TQ, TR, M, NSV, NSQ, 2, false, RG, 0, FL, 0, 2>, base + 2);
            }
            return go(k_scan_h16<WQ, WR, TQ, TR, M, NSV, NSQ, 2, false, RG, 0, FL, 0, 1>, base + 1);
        };
        auto by_variant = [&](auto mtag, int base) -> int {
            if (filt) {  // subset filter compiled in (always the ragged-capable variant)
                if constexpr (WR >= 4) return set_error(LYNSE_ERR_INTERNAL, "subset-filtered scans run on the <4,2,2,4> tiling");
                else return by_emit(mtag, std::true_type{}, std::true_type{}, base + 6);
            }
            if (ragged) return by_emit(mtag, std::true_type{}, std::false_type{}, base + 3);
            return by_emit(mtag, std::false_type{}, std::false_type{}, base);  // no ragged last slab: branch-free DMA issue
        };
        if constexpr (WR >= 4) {  // <2,4,4,2>: unfiltered IP only
#ifndef LYNSE_EXPERIMENTS
            if (metric != M_IP) return set_error(LYNSE_ERR_INTERNAL, "the <2,4,4,2> tiling is compiled for IP only");
#else
            if (metric == M_L2) return by_variant(std::integral_constant<int, M_L2>{}, 9);
            if (metric == M_COS) return by_variant(std::integral_constant<int, M_COS>{}, 18);
#endif
            return by_variant(std::integral_constant<int, M_IP>{}, 0);
        } else {
            switch (metric) {
            case M_IP:
#ifndef LYNSE_EXPERIMENTS
                if (!filt) return set_error(LYNSE_ERR_INTERNAL, "unfiltered IP scans run on the <2,4,4,2> tiling");
                return by_emit(std::integral_constant<int, M_IP>{}, std::true_type{}, std::true_type{}, 6);
#else
                return by_variant(std::integral_constant<int, M_IP>{}, 0);
#endif
            case M_L2: return by_variant(std::integral_constant<int, M_L2>{}, 9);
            default: return by_variant(std::integral_constant<int, M_COS>{}, 18);
            }
        }
    } else {
        if (!ragged && !filt) {
            switch (metric) {
            case M_IP: return go(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, false>, 3);
            case M_L2: return go(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, TILED, false>, 4);
            default: return go(k_scan_h16<WQ, WR, TQ, TR, M_COS, NSV, NSQ, 2, TILED, false>, 5);
            }
        }
        if (filt) {  // subset filter compiled in (always the ragged-capable variant: one instantiation per metric)
            switch (metric) {
            case M_IP: return go(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, true, 0, true>, 6);
            case M_L2: return go(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, TILED, true, 0, true>, 7);
            default: return go(k_scan_h16<WQ, WR, TQ, TR, M_COS, NSV, NSQ, 2, TILED, true, 0, true>, 8);
            }
        }
        switch (metric) {
        case M_IP: return go(k_scan_h16<WQ, WR, TQ, TR, M_IP, NSV, NSQ, 2, TILED, true>, 0);
        case M_L2: return go(k_scan_h16<WQ, WR, TQ, TR, M_L2, NSV, NSQ, 2, TILED, true>, 1);
        default: return go(k_scan_h16<WQ, WR, TQ, TR, M_COS, NSV, NSQ, 2, TILED, true>, 2);
        }
    }
}

// certified int8 coarse pass for batches of <= 32 queries: the 128-row x 32-query tiling (two workgroups per CU, 3 + 3-stage
// rings) over the SQ8 codes — HBM-bound like its f16 twin, at half the bytes; emission mode at run time (EMIT = -1)
static int launch_scan_i8c_small(const ScanArgs& a, uint32_t grid, hipStream_t st, bool l2n = false, bool filt = false) {
    constexpr size_t lds = (size_t)(3 * 128 + 3 * 32) * 128;
    static std::atomic<bool> attr_done[4] = {false, false, false, false};
    if (filt) {   // masked scan of a small batch (row bitmask in the epilogue): whole slabs
        if (a.ld16 % 128 != 0 || a.row_ids || l2n) return set_error(LYNSE_ERR_INTERNAL, "the masked int8 scan needs whole 128-column slabs and a row bitmask");
        auto kern = k_scan_h16<1, 4, 1, 1, M_IP, 3, 3, 2, false, false, 0, true, 2>;
        if (!attr_done[3]) { LY_TRY(set_max_lds(kern, lds)); attr_done[3] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    }
    if (l2n) {   // plain-code L2 (I8Q = 4): + the norm ring
        constexpr size_t ldsn = lds + 4 * 1024;
        auto kern = k_scan_h16<1, 4, 1, 1, M_L2, 3, 3, 2, false, false, 0, false, 4>;
        if (!attr_done[2]) { LY_TRY(set_max_lds(kern, ldsn)); attr_done[2] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsn, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    }
    auto go = [&](auto kern, int slot) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    if (a.ld16 % 128 == 0) return go(k_scan_h16<1, 4, 1, 1, M_IP, 3, 3, 2, false, false, 0, false, 2>, 0);
    return go(k_scan_h16<1, 4, 1, 1, M_IP, 3, 3, 2, false, true, 0, false, 2>, 1);
}

// Squared L2 on the query-stationary tiling (scan_qs.h, MET = 1): threshold stages of unfiltered batches of 129..256 queries over code
// rows of 768 bytes (64-row tiles).  Rows per tile, or 0.
static uint32_t qs_l2_rows(const ScanArgs& a) {
    const char* e = getenv("LYNSE_HIP_QS");
    if (e && atoi(e) == 0) return 0;
    if (a.emit_all != 0 || a.qpad != 256 || a.nq > 256 || a.tile_stride != 0 || a.skip_stride != 0 || a.mask || a.row_ids || a.row1 <= a.row0) return 0;
    if (a.ld16 == 768 && a.nslab == 6) return 64;
    // 256 / 384 / 512 / 640 columns (LYNSE_HIP_QS_WIDTHS=0 or an A/B variant of LYNSE_HIP_QS: 768 only)
    const char* w = getenv("LYNSE_HIP_QS_WIDTHS");
    if ((w && atoi(w) == 0) || (e && atoi(e) != 1)) return 0;
    // (6M rows, 256 queries: 256 / 384 columns 0.772 / 0.916 ms on the <4,2,2,4> tiling against 0.773 / 0.929 here — the float epilogue weighs more the
    // fewer MFMAs a tile has; 512 / 640 columns 1.125 / 1.347 -> 1.106 / 1.307; 100 queries: 0.65 / 0.80 / 0.94 / 1.11 -> 0.61 / 0.72 / 0.83 / 0.94)
    if (a.nslab >= 2 && a.nslab <= 5 && a.ld16 == a.nslab * 128u && (a.nslab >= 4 || a.nq <= 128)) return 64;
    return 0;
}
static int launch_scan_qs_l2(const ScanArgs& a, uint32_t grid, hipStream_t st) {
    static std::atomic<bool> attr_done[8] = {false};
    auto go = [&](auto kern, int slot, size_t lds) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    switch (a.nslab) {
    case 2: return go(k_scan_qs<2, 2, 2, 3, false, 8, 0, 1, 0, 1>, 2, (size_t)3 * 2 * 64 * 128 + 4 * 256);
    case 3: return go(k_scan_qs<3, 2, 3, 3, false, 8, 0, 1, 0, 1>, 3, (size_t)3 * 3 * 64 * 128 + 4 * 256);
    case 4: return go(k_scan_qs<4, 2, 4, 3, false, 8, 0, 1, 0, 1>, 4, (size_t)3 * 4 * 64 * 128 + 4 * 256);
    case 5: return go(k_scan_qs<5, 2, 5, 3, false, 8, 0, 1, 0, 1>, 5, (size_t)3 * 5 * 64 * 128 + 4 * 256);
    default: return go(k_scan_qs<6, 2, 6, 3, false, 8, 0, 1, 0, 1>, 6, (size_t)3 * 6 * 64 * 128 + 4 * 256);
    }
}

// squared L2 on the plain SQ8 codes (kernels.h, I8Q = 4): the <4,2,2,4> L2 tiling with 2 + 2-stage rings and the norm ring, int8
// operands, float epilogue; whole 128-column slabs
static int launch_scan_i8l2(const ScanArgs& a, uint32_t grid, hipStream_t st, bool qs = false) {
    if (qs) return launch_scan_qs_l2(a, grid, st);
    constexpr size_t lds = (size_t)(2 * 256 + 2 * 256) * 128 + 3 * 1024;
    static std::atomic<bool> attr_done[4] = {false, false, false, false};
    auto go = [&](auto kern, int slot) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    if (a.ld16 % 128 != 0) return set_error(LYNSE_ERR_INTERNAL, "the plain-code L2 scan needs whole 128-column slabs");
    if (a.emit_all == 1) return go(k_scan_h16<4, 2, 2, 4, M_L2, 2, 2, 2, false, false, 0, false, 4, 1>, 0);
    if (a.emit_all == 2) return go(k_scan_h16<4, 2, 2, 4, M_L2, 2, 2, 2, false, false, 0, false, 4, 2>, 1);
    if (a.dense) return go(k_scan_h16<4, 2, 2, 4, M_L2, 2, 2, 2, false, false, 0, false, 4, 0, 0, true>, 2);
    return go(k_scan_h16<4, 2, 2, 4, M_L2, 2, 2, 2, false, false, 0, false, 4, 0>, 3);
}

// certified int8 coarse pass for batches of 33..64 queries (128 rows x 64 queries, 4 waves, two workgroups per CU, 3 + 3 stages)
// and of 65..128 queries (256 rows x 128 queries, 8 waves as 2 x 4, 3 + 2 stages): per row byte they do a quarter / half of the
// MFMA and fragment-read work of the 256-query tiling, which a batch of 40 or 100 queries would otherwise pay in full
static int launch_scan_i8c_mid(const ScanArgs& a, uint32_t grid, hipStream_t st, bool wide, bool l2n = false, bool filt = false) {
    static std::atomic<bool> attr_done[8] = {false, false, false, false, false, false, false, false};
    const bool rag = a.ld16 % 128 != 0;
    if (filt) {   // masked scan (row bitmask in the epilogue): whole slabs
        if (rag || a.row_ids || l2n) return set_error(LYNSE_ERR_INTERNAL, "the masked int8 scan needs whole 128-column slabs and a row bitmask");
        auto gof = [&](auto kern, int slot, size_t lds, uint32_t threads) -> int {
            if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, a);
            LY_HIP(hipGetLastError());
            return LYNSE_OK;
        };
        if (!wide) return gof(k_scan_h16<1, 4, 2, 1, M_IP, 3, 3, 2, false, false, 0, true, 2>, 6, (size_t)(3 * 128 + 3 * 64) * 128, 256);
        return gof(k_scan_h16<2, 4, 2, 2, M_IP, 3, 2, 2, false, false, 0, true, 2>, 7, (size_t)(3 * 256 + 2 * 128) * 128, 512);
    }
    if (l2n) {   // plain-code L2 (I8Q = 4): whole slabs, + the norm ring
        auto gol = [&](auto kern, int slot, size_t lds, uint32_t threads) -> int {
            if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, a);
            LY_HIP(hipGetLastError());
            return LYNSE_OK;
        };
        if (!wide) return gol(k_scan_h16<1, 4, 2, 1, M_L2, 3, 3, 2, false, false, 0, false, 4>, 4, (size_t)(3 * 128 + 3 * 64) * 128 + 4 * 1024, 256);
        return gol(k_scan_h16<2, 4, 2, 2, M_L2, 3, 2, 2, false, false, 0, false, 4>, 5, (size_t)(3 * 256 + 2 * 128) * 128 + 4 * 1024, 512);
    }
    auto go = [&](auto kern, int slot, size_t lds, uint32_t threads) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    if (!wide) {
        constexpr size_t lds = (size_t)(3 * 128 + 3 * 64) * 128;
        if (!rag) return go(k_scan_h16<1, 4, 2, 1, M_IP, 3, 3, 2, false, false, 0, false, 2>, 0, lds, 256);
        return go(k_scan_h16<1, 4, 2, 1, M_IP, 3, 3, 2, false, true, 0, false, 2>, 1, lds, 256);
    }
    constexpr size_t lds = (size_t)(3 * 256 + 2 * 128) * 128;
    if (!rag) return go(k_scan_h16<2, 4, 2, 2, M_IP, 3, 2, 2, false, false, 0, false, 2>, 2, lds, 512);
    return go(k_scan_h16<2, 4, 2, 2, M_IP, 3, 2, 2, false, true, 0, false, 2>, 3, lds, 512);
}

// certified int8 coarse pass: the 256 x 256 IP tiling with 3 + 2-stage rings over 128-element slabs, one kernel per
// (ragged last slab, emission mode)
// The query-stationary tiling of the certified int8 pass (scan_qs.h): threshold stages of an unfiltered FLAT batch over whole
// 768-byte code rows — 8 waves x 32 register-resident queries, 64-row tiles, rows-only LDS ring.  LYNSE_HIP_QS=0: the 256 x 256
// tile of k_scan_h16 (A/B, tests; read per call).
// LYNSE_HIP_QS selects the instantiation (A/B; the default is the fastest measured): 1 = 64-row tiles, whole-K stages, 3-stage ring,
// ping-pong roles; 2 = 32-row tiles, 6-stage ring, fragments prefetched across the barrier; 3 = as 1 without the ping-pong roles.
static int qs_variant() { const char* e = getenv("LYNSE_HIP_QS"); return e ? atoi(e) : 1; }
static uint32_t qs_rows(int v) { return v == 2 ? 32u : 64u; }   // rows per tile
// Self-tightening single-launch scan (scan_qs.h, STS; LYNSE_HIP_STS=1 — OFF by default): unfiltered FLAT batches of 129..256 queries on
// the certified int8 pass (IP, cosine) over whole 768-byte code rows, k <= 32 (one partition maximum per wanted neighbour), shards
// large enough that every workgroup helper covers a query (8 grid >= nq) — first plan level only, the overflow ladder keeps the
// staged plans.  Bit-identical results (tests), and MEASURED no faster than the staged plan it replaces (MI355X, 10M x 768 x 256,
// scripts/qs_microbench.hip at the margin the benchmark data has, 2E = 0.76 sd of the scores): one launch 1986 us + query image +
// final select against 37 + 31 + ~1880 + 31 us of sample stage, selects and threshold stages.  Without the exact-rescored
// threshold the selects of the staged plan compute (tau_x - E instead of tau - 2E) the band of rows that must be kept is ~5x wider:
// ~4800 keys per query are emitted (10,000 for the unluckiest query; the cap is 16,384 — 6 of 256 queries of the C2 test overflowed
// and went down the ladder), 15-20 % of all tile epilogues take the slow path.
// Row widths the query-stationary tiling is instantiated for: whole 128-column slabs, 2..6 of them (256 / 384 / 512 / 640 / 768 columns of
// codes: the B operand of a wave is NSLAB x 16 registers).  The masked (MSK) and plain-L2 (MET) forms and the A/B variants exist for 768
// columns only.  LYNSE_HIP_QS_WIDTHS=0: 768 columns only (A/B; read per call).
static bool qs_width_ok(uint32_t ld, uint32_t nslab, bool only768 = false) {
    if (ld == 768 && nslab == 6) return true;
    if (only768 || qs_variant() != 1) return false;
    const char* e = getenv("LYNSE_HIP_QS_WIDTHS");
    if (e && atoi(e) == 0) return false;
    return ((nslab >= 2 && nslab <= 5) || nslab == 8) && ld == nslab * 128u;   // (8 slabs = 1024 columns: 32-row tiles, B = 128 registers)
}
static bool sts_env_on() { const char* e = getenv("LYNSE_HIP_STS"); return e && atoi(e) != 0; }
static bool qs_scan_ok(const ScanArgs& a, bool fs, bool filt, bool f4) {
    const int v = qs_variant();
    if (v < 1 || v > 3) return false;
    // filt: the MASKED threshold stages of a subset-filtered search (row bitmask; k_scan_qs<.., MSK>) — LYNSE_HIP_QS_MASKED=0: the DENSE
    // masked epilogue of k_scan_h16 (A/B; read per call)
    if (filt) {
        const char* e = getenv("LYNSE_HIP_QS_MASKED");
        if ((e && atoi(e) == 0) || v == 2 || !a.mask || a.row_ids) return false;
    }
    // f4: batched Hamming on the FP4 MFMA (k_scan_qs<.., F4>): 1024-bit fingerprints = 512 B of nibbles = 4 slabs; its plans carry an emit-all
    // sample whose tiles the kernel skips (a.skip_stride).  LYNSE_HIP_QS_F4=0: the 256 x 256 tile of k_scan_h16<.., I8Q = 3> (A/B; read per call)
    if (f4) {
        const char* e = getenv("LYNSE_HIP_QS_F4");
        return !(e && atoi(e) == 0) && v == 1 && !fs && !filt && a.emit_all == 0 && (a.nslab == 2 || a.nslab == 4 || a.nslab == 8) && a.ld16 == a.nslab * 128u &&   // 512 / 1024 / 2048-bit rows
               a.qpad == 256 && a.nq <= 256 &&
               a.tile_stride == 0 && !a.mask && !a.row_ids && a.row1 > a.row0;
    }
    return !fs && a.emit_all == 0 && qs_width_ok(a.ld16, a.nslab, filt) && a.qpad == 256 && a.nq <= 256 && a.tile_stride == 0 &&
           (filt || (a.skip_stride == 0 && !a.mask)) && !a.row_ids && a.row1 > a.row0;
}
// LYNSE_HIP_SCAN_CUS: workgroups (= CUs: one 144-KB workgroup per CU) of the persistent threshold-stage scans.  The scan is bound by the
// power the chip may draw, not by its CU count: a few CUs left free cost it little and let the short latency-bound kernels of ANOTHER
// batch in flight (query image, sample stage, selects, final rescoring) run beside it instead of between its launches.
static uint32_t qs_grid(const ScanArgs& a, uint32_t num_cu) {
    const uint32_t rt = a.nslab == 8 ? 32u : qs_rows(qs_variant());   // (1024 columns: 32-row tiles)
    const char* e = getenv("LYNSE_HIP_SCAN_CUS");
    const uint32_t cus = e && atoi(e) > 0 ? std::min<uint32_t>((uint32_t)atoi(e), num_cu) : num_cu;
    return std::min<uint32_t>((a.row1 - a.row0 + rt - 1) / rt, cus);
}
static int launch_scan_qs(const ScanArgs& a, uint32_t grid, hipStream_t st) {
    static std::atomic<bool> attr_done[5] = {false, false, false, false, false};
    if (a.dyn_thr) {   // self-tightening thresholds: the whole shard in one launch (scan_qs.h, STS)
        auto kern = k_scan_qs<6, 2, 6, 3, false, 8, 0, 0, 1>;
        constexpr size_t lds = (size_t)3 * 6 * 64 * 128 + QS_STS_LDS;
        if (!attr_done[4]) { LY_TRY(set_max_lds(kern, lds)); attr_done[4] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    }
    auto go = [&](auto kern, int slot, size_t lds) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    if (a.mask) return go(k_scan_qs<6, 2, 6, 3, false, 8, 0, 1, 0, 0, 0, 1>, 0, (size_t)3 * 6 * 64 * 128);   // masked threshold stages (MSK)
    static std::atomic<bool> attr_w[8] = {false};
    auto gow = [&](auto kern, int slot, size_t lds) -> int {
        if (!attr_w[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_w[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    switch (a.nslab) {   // the narrower code rows (qs_width_ok)
    case 2: return gow(k_scan_qs<2, 2, 2, 3, false, 8, 0, 1>, 2, (size_t)3 * 2 * 64 * 128);
    case 3: return gow(k_scan_qs<3, 2, 3, 3, false, 8, 0, 1>, 3, (size_t)3 * 3 * 64 * 128);
    case 4: return gow(k_scan_qs<4, 2, 4, 3, false, 8, 0, 1>, 4, (size_t)3 * 4 * 64 * 128);
    case 5: return gow(k_scan_qs<5, 2, 5, 3, false, 8, 0, 1>, 5, (size_t)3 * 5 * 64 * 128);
    case 8: return gow(k_scan_qs<8, 1, 8, 4, false, 8, 0, 1>, 7, (size_t)4 * 8 * 32 * 128);
    default: break;
    }
    switch (qs_variant()) {
    case 2: return go(k_scan_qs<6, 1, 6, 6, true, 8, 0, 0>, 2, (size_t)6 * 6 * 32 * 128);
    case 3: return go(k_scan_qs<6, 2, 6, 3, false, 8, 0, 0>, 3, (size_t)3 * 6 * 64 * 128);
    default: return go(k_scan_qs<6, 2, 6, 3, false, 8, 0, 1>, 1, (size_t)3 * 6 * 64 * 128);
    }
}

// The threshold-only SAMPLE stage on the query-stationary tiling (scan_qs.h, SMP = 1): the sample rows as 64-row tiles, four per workgroup
// on a full chip (one 144-KB workgroup per CU), 4 keys per workgroup and query = 1024 keys per query.  LYNSE_HIP_QS_SAMPLE=0: the 256 x 256 sample tiles of k_scan_h16 (A/B; read per call).
static bool qs_sample_ok(const ScanArgs& a, bool fs, bool filt, bool f4, uint32_t plan_tile) {
    const char* e = getenv("LYNSE_HIP_QS_SAMPLE");
    const int v = qs_variant();
    if ((e && atoi(e) == 0) || (v != 1 && v != 3)) return false;
    return !fs && !filt && !f4 && a.emit_all == 2 && plan_tile == 256 && a.nslab != 8 && qs_width_ok(a.ld16, a.nslab) && a.qpad == 256 && a.nq <= 256 && a.tile_stride >= 256 &&
           a.skip_stride == 0 && !a.mask && !a.row_ids && a.row1 > a.row0;
}
static int launch_scan_qs_sample(const ScanArgs& a, uint32_t grid, hipStream_t st) {
    static std::atomic<bool> attr_done[8] = {false};
    auto go = [&](auto kern, int slot, size_t lds) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    switch (a.nslab) {
    case 2: return go(k_scan_qs<2, 2, 2, 3, false, 8, 0, 0, 0, 0, 1>, 2, (size_t)3 * 2 * 64 * 128);
    case 3: return go(k_scan_qs<3, 2, 3, 3, false, 8, 0, 0, 0, 0, 1>, 3, (size_t)3 * 3 * 64 * 128);
    case 4: return go(k_scan_qs<4, 2, 4, 3, false, 8, 0, 0, 0, 0, 1>, 4, (size_t)3 * 4 * 64 * 128);
    case 5: return go(k_scan_qs<5, 2, 5, 3, false, 8, 0, 0, 0, 0, 1>, 5, (size_t)3 * 5 * 64 * 128);
    default: return go(k_scan_qs<6, 2, 6, 3, false, 8, 0, 0, 0, 0, 1>, 6, (size_t)3 * 6 * 64 * 128);
    }
}

// The query-stationary tiling of the f16 shadow at low dimension (scan_qh.h): threshold stages of an unfiltered FLAT batch of 33..256
// queries on the float path over rows of 1 or 2 whole 64-element slabs (64 / 128 columns: BASELINE config 3).  ON by default since the
// deferred emission of round 5 (LYNSE_HIP_QH=0: k_scan_h16's 256 x 256 tile, A/B; read per call): bit-identical results (tests), C3
// 0.190-0.194 against 0.213-0.219 ms per batch on the same box; DESIGN 4.3 has the phase table and what the first versions lost to
// (keys stored straight to global memory stalled the ring; a grouped slow path as long as the MFMAs of its step).
constexpr int QH_DEFAULT = 1;
static int qh_variant() { const char* e = getenv("LYNSE_HIP_QH"); return e ? atoi(e) : QH_DEFAULT; }   // 0 = k_scan_h16, 1 = k_scan_qh
// the batch SHAPE k_scan_qh takes (shared with i8c_eligible, which sends FLAT-IP batches of this shape to the float pass only because
// this kernel runs them: ADVICE r5): 33..256 queries — one 256-query chunk —, rows of one or two whole 64-element slabs
static bool qh_shape(uint32_t ld16, uint64_t nq) { return qh_variant() == 1 && nq > SCAN_BQ_SMALL && nq <= QCHUNK && (ld16 == 64u || ld16 == 128u); }
static bool qh_scan_ok(const ScanArgs& a, bool filt, uint32_t qchunks, int level) {
    if (level != 0) return false;   // (the plan ladder's later levels run k_scan_h16: k_scan_qh may hand a batch with massive ties to it)
    return qh_shape(a.ld16, a.nq) && !filt && a.emit_all == 0 && qchunks == 1 && a.qpad == 256 && a.tile_stride == 0 && a.skip_stride == 0 && !a.mask && !a.row_ids &&
           !a.tiles && a.row1 > a.row0 && (a.nslab == 1 || a.nslab == 2) && a.ld16 == a.nslab * 64u;
}
static uint32_t qh_grid(const ScanArgs& a, uint32_t num_cu, uint32_t* segs_per_wg) {
    *segs_per_wg = 1u;   // (one segment per workgroup and query: the deferred emission of scan_qh.h)
    return std::min<uint32_t>((a.row1 - a.row0 + 63u) / 64u, 2u * num_cu);
}
static int launch_scan_qh(const ScanArgs& a, int metric, uint32_t grid, hipStream_t st) {
    static std::atomic<bool> attr_done[16] = {false};
    auto go = [&](auto kern, int slot, size_t lds) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    // 32 queries per wave, 64-row tiles, two workgroups per CU (<= 80 KB of LDS each)
    constexpr size_t stg1 = 8 * 144 * 24 + 1024, nrm1 = 4 * 256 + stg1;   // (three ring stages: the LDS of the fourth is the staging regions')
    if (a.nslab == 1) {
        switch (metric) {
        case M_IP: return go(k_scan_qh<1, M_IP, 3, 2, 1>, 6, (size_t)3 * 1 * 64 * 128 + stg1);
        case M_L2: return go(k_scan_qh<1, M_L2, 3, 2, 1>, 7, (size_t)3 * 1 * 64 * 128 + nrm1);
        default: return go(k_scan_qh<1, M_COS, 3, 2, 1>, 8, (size_t)3 * 1 * 64 * 128 + nrm1);
        }
    }
    switch (metric) {
    case M_IP: return go(k_scan_qh<2, M_IP, 3, 2, 1>, 9, (size_t)3 * 2 * 64 * 128 + stg1);
    case M_L2: return go(k_scan_qh<2, M_L2, 3, 2, 1>, 10, (size_t)3 * 2 * 64 * 128 + nrm1);
    default: return go(k_scan_qh<2, M_COS, 3, 2, 1>, 11, (size_t)3 * 2 * 64 * 128 + nrm1);
    }
}

// The threshold-only SAMPLE stage of the same plans on k_scan_qh<.., SMP> (scan_qh.h): the sample rows as 64-row tiles, two workgroups per CU,
// 4 keys per workgroup and query.  LYNSE_HIP_QH_SAMPLE=0: the 256 x 256 sample tiles of k_scan_h16<.., EMIT = 2> (A/B; read per call).
static bool qh_sample_ok(const ScanArgs& a, bool filt, uint32_t qchunks, int level, uint32_t plan_tile) {
    const char* e = getenv("LYNSE_HIP_QH_SAMPLE");
    if ((e && atoi(e) == 0) || qh_variant() != 1 || level != 0) return false;
    return !filt && a.emit_all == 2 && plan_tile == 256 && qchunks == 1 && a.qpad == 256 && a.nq <= 256 && a.tile_stride >= 256 && a.skip_stride == 0 && !a.mask && !a.row_ids &&
           !a.tiles && a.row1 > a.row0 && (a.nslab == 1 || a.nslab == 2) && a.ld16 == a.nslab * 64u;
}
// -> the grid of the sample launch (0: the sample stage stays on k_scan_h16): two workgroups per CU, enough keys for k_select's threshold-only
// rule (>= 8 k), inside the candidate buffer, <= 32 tiles per workgroup (the packed position of a sample key, scan_qh.h)
static uint32_t qh_sample_grid(const ScanArgs& a, uint32_t sample_tiles, bool filt, uint32_t qchunks, int level, uint32_t plan_tile, uint32_t k, uint32_t cap, uint32_t num_cu) {
    if (!sample_tiles || !qh_sample_ok(a, filt, qchunks, level, plan_tile)) return 0;
    const uint32_t nt64 = sample_tiles * 4u, sgrid = std::min<uint32_t>(nt64, 2u * num_cu);
    const bool ok = (uint64_t)sgrid * 4u >= 8ull * k && sgrid * 4u <= cap && (nt64 + sgrid - 1) / sgrid <= 32u;
    return ok ? sgrid : 0u;
}
static int launch_scan_qh_sample(const ScanArgs& a, int metric, uint32_t grid, hipStream_t st) {
    static std::atomic<bool> attr_done[8] = {false};
    auto go = [&](auto kern, int slot, size_t lds) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    constexpr size_t stg1 = 8 * 144 * 24 + 1024, nrm1 = 4 * 256 + stg1;
    if (a.nslab == 1) {
        switch (metric) {
        case M_IP: return go(k_scan_qh<1, M_IP, 3, 2, 1, 1>, 0, (size_t)3 * 1 * 64 * 128 + stg1);
        case M_L2: return go(k_scan_qh<1, M_L2, 3, 2, 1, 1>, 1, (size_t)3 * 1 * 64 * 128 + nrm1);
        default: return go(k_scan_qh<1, M_COS, 3, 2, 1, 1>, 2, (size_t)3 * 1 * 64 * 128 + nrm1);
        }
    }
    switch (metric) {
    case M_IP: return go(k_scan_qh<2, M_IP, 3, 2, 1, 1>, 3, (size_t)3 * 2 * 64 * 128 + stg1);
    case M_L2: return go(k_scan_qh<2, M_L2, 3, 2, 1, 1>, 4, (size_t)3 * 2 * 64 * 128 + nrm1);
    default: return go(k_scan_qh<2, M_COS, 3, 2, 1, 1>, 5, (size_t)3 * 2 * 64 * 128 + nrm1);
    }
}

// (round 5, measured and removed: 1024-bit Hamming on a 64-queries-per-wave form of k_scan_qh — k_scan_qh<4, M_IP, 4, 8, QB = 2, one row block
// per wave, FP4 MFMA>: every row fragment read feeds two MFMAs, half the LDS fragment returns per MFMA.  Bit-identical, and SLOWER: C5 share
// 1.900 / 1.916 / 1.928 against 1.705 / 1.708 / 1.725 ms per batch, same box, alternating (scripts/gpu_r5_c5.sh): with one 32-row block per
// wave a step is 32 MFMAs between two barriers, and a wave's epilogue / DMA issue is no longer covered by its SIMD partner's MFMAs.)
static int launch_scan_i8c(const ScanArgs& a, uint32_t grid, hipStream_t st, bool fs = false, bool filt = false, bool f4 = false, bool qs = false) {
    constexpr size_t lds = (size_t)(3 * 256 + 2 * 256) * 128;
    if (qs && f4) {   // batched Hamming on the query-stationary tiling (scan_qs.h, F4)
        static std::atomic<bool> attr_f4[3] = {false, false, false};
        auto gof = [&](auto kern, int slot, size_t qlds) -> int {
            if (!attr_f4[slot]) { LY_TRY(set_max_lds(kern, qlds)); attr_f4[slot] = true; }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), qlds, st, a);
            LY_HIP(hipGetLastError());
            return LYNSE_OK;
        };
        if (a.nslab == 2) return gof(k_scan_qs<2, 2, 2, 3, false, 8, 0, 1, 0, 0, 0, 0, 1>, 0, (size_t)3 * 2 * 64 * 128);   // 512-bit rows
        if (a.nslab == 8) return gof(k_scan_qs<8, 1, 8, 4, false, 8, 0, 1, 0, 0, 0, 0, 1>, 2, (size_t)4 * 8 * 32 * 128);   // 2048-bit rows: 32-row tiles
        return gof(k_scan_qs<4, 2, 4, 3, false, 8, 0, 1, 0, 0, 0, 0, 1>, 1, (size_t)3 * 4 * 64 * 128);
    }
    if (qs) return launch_scan_qs(a, grid, st);
    static std::atomic<bool> attr_done[16] = {false};
    auto go = [&](auto kern, int slot) -> int {
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
#ifdef LYNSE_EXPERIMENTS
    if (a.ld16 % 128 == 0 && a.emit_all == 0 && ((a.debug_flags >> 8) & 31)) {  // timing experiments: DBG variants of the int8 kernel
        auto ex = [&](auto kern) -> int {
            LY_TRY(set_max_lds(kern, lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
            LY_HIP(hipGetLastError());
            return LYNSE_OK;
        };
        switch ((a.debug_flags >> 8) & 31) {
        case 3: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 3, false, 2, 0>);    // DMA only
        case 4: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 4, false, 2, 0>);    // no query-image DMA
        case 8: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 8, false, 2, 0>);    // no row DMA
        case 12: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 12, false, 2, 0>);  // MFMA + LDS reads only
        case 7: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 7, false, 2, 0>);    // row DMA only
        case 11: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 11, false, 2, 0>);  // query DMA only
        case 1: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 1, false, 2, 0>);    // no MFMA (DMA + LDS reads)
        case 13: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 13, false, 2, 0>);  // LDS reads only
        case 14: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 14, false, 2, 0>);  // MFMA only
        case 2: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 2, false, 2, 0>);    // no LDS reads (DMA + MFMA)
        case 16: return ex(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 16, false, 2, 0>);  // no epilogue
        default: return set_error(LYNSE_ERR_INVALID_ARGUMENT, "unknown experiment");
        }
    }
#endif
    if (f4) {   // batched Hamming: FP4 +-1 operands (kernels.h, I8Q = 3); whole 128-B slabs by construction, unfiltered
        if (a.ld16 % 128 != 0 || filt || fs) return set_error(LYNSE_ERR_INTERNAL, "the FP4 Hamming scan runs unfiltered over whole slabs");
        if (a.emit_all == 1) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 3, 1>, 11);
        if (a.emit_all == 2) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 3, 2>, 12);
        if (a.dense) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 3, 0, 0, true>, 13);
        return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 3, 0>, 14);
    }
    if (filt) {  // masked scan (subset as a row bitmask): emit-all sample with sentinels, DENSE threshold stages (kernels.h, FILT && I8C)
        if (a.ld16 % 128 != 0 || a.row_ids) return set_error(LYNSE_ERR_INTERNAL, "the masked int8 scan needs whole 128-column slabs and a row bitmask");
        if (a.emit_all == 1) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, true, 2, 1>, 9);
        if (a.emit_all != 0 || !a.dense) return set_error(LYNSE_ERR_INTERNAL, "the masked int8 scan runs emit-all and DENSE threshold stages");
        return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, true, 2, 0, 0, true>, 10);
    }
    // One wave per SIMD: 4 waves x (128 queries x 128 rows), accumulators in fixed AGPR tuples (kernels.h, AG) — the threshold
    // stages over whole 128-column slabs (LYNSE_HIP_AG=1; the sample stages keep the 8-wave kernels)
    const int ag_env = []() { const char* e = getenv("LYNSE_HIP_AG"); return e ? atoi(e) : 0; }();   // (read per call: A/B, tests)
    if (ag_env && a.ld16 % 128 == 0 && a.emit_all == 0 && !fs) {
        static std::atomic<bool> ag_attr[2] = {false, false};
        auto go4 = [&](auto kern, int slot) -> int {
            if (!ag_attr[slot]) { LY_TRY(set_max_lds(kern, lds)); ag_attr[slot] = true; }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
            LY_HIP(hipGetLastError());
            return LYNSE_OK;
        };
        if (a.dense) return go4(k_scan_h16<2, 2, 4, 4, M_IP, 3, 2, 2, false, false, 0, false, 2, 0, 0, true>, 0);
        return go4(k_scan_h16<2, 2, 4, 4, M_IP, 3, 2, 2, false, false, 0, false, 2, 0>, 1);
    }
    if (a.ld16 % 128 == 0) {
        static const int prio = []() { const char* e = getenv("LYNSE_HIP_PRIO"); return e ? atoi(e) : 0; }();
        static std::atomic<bool> prattr = false;
        if (a.emit_all == 0 && prio) { auto k = k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 2, 0, 0, false, true>; if (!prattr) { LY_TRY(set_max_lds(k, lds)); prattr = true; } hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a); LY_HIP(hipGetLastError()); return LYNSE_OK; }
        static const int place = []() { const char* e = getenv("LYNSE_HIP_PLACE"); return e ? atoi(e) : 0; }();
        static std::atomic<bool> pattr[2] = {false, false};
        if (a.emit_all == 0 && place == 1) { auto k = k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 2, 0, 1>; if (!pattr[0]) { LY_TRY(set_max_lds(k, lds)); pattr[0] = true; } hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a); LY_HIP(hipGetLastError()); return LYNSE_OK; }
        if (a.emit_all == 0 && place == 2) { auto k = k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 2, 0, 2>; if (!pattr[1]) { LY_TRY(set_max_lds(k, lds)); pattr[1] = true; } hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a); LY_HIP(hipGetLastError()); return LYNSE_OK; }
        if (fs) {  // fused sample stage: sample tile + in-launch thresholds + the first threshold stage (kernels.h, FS)
            if (a.emit_all != 0) return set_error(LYNSE_ERR_INTERNAL, "the fused sample stage is a threshold stage");
            if (!a.dense) return set_error(LYNSE_ERR_INTERNAL, "the fused sample stage is compiled with the DENSE epilogue");  // (the two-level body spills 12 B with it)
            return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 2, 0, 0, true, false, true, 1>, 7);
        }
        if (a.emit_all == 0 && a.dense) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 2, 0, 0, true>, 6);
        if (a.emit_all == 0) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 2, 0>, 0);
        if (a.emit_all == 1) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 2, 1>, 1);
        return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, false, 0, false, 2, 2>, 2);
    }
    if (a.emit_all == 0) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, true, 0, false, 2, 0>, 3);
    if (a.emit_all == 1) return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, true, 0, false, 2, 1>, 4);
    return go(k_scan_h16<2, 4, 4, 2, M_IP, 3, 2, 2, false, true, 0, false, 2, 2>, 5);
}

// f16 shadow rows [n16, n) (all rows again when the scale changed)
static int ensure_shadow_locked(lynse_hip_flat* h) {
    if (h->packed_only || h->n == 0) return LYNSE_OK;
    if (is_f16(h) && h->sv == 1.0f) {   // the f16 bits ARE the shadow (no scaling): one copy of the rows in HBM
        if (!h->shadow_alias && h->rows16) (void)hipFree(h->rows16);
        h->shadow_alias = true;
        h->rows16 = h->rows_h;
        h->cap16 = h->capacity;
        h->n16 = h->n;
        h->sv16 = h->sv;
        return LYNSE_OK;
    }
    if (h->shadow_alias) {              // the scale moved away from 1: the shadow becomes a buffer of its own
        h->shadow_alias = false;
        h->rows16 = nullptr;
        h->cap16 = 0;
        h->n16 = 0;
    }
    if (h->cap16 < h->n) {
        const uint64_t cap = std::max<uint64_t>(h->capacity, h->n);
        _Float16* nr = nullptr;
        LY_HIP(hipMalloc(&nr, (size_t)cap * h->ld16 * sizeof(_Float16) + 256));
        if (h->rows16 && h->n16 && h->sv16 == h->sv)
            LY_HIP(hipMemcpyAsync(nr, h->rows16, (size_t)h->n16 * h->ld16 * sizeof(_Float16), hipMemcpyDeviceToDevice, cur(h).stream));
        LY_HIP(hipStreamSynchronize(cur(h).stream));
        if (h->rows16) (void)hipFree(h->rows16);
        h->rows16 = nr;
        h->cap16 = cap;
    }
    if (h->sv16 != h->sv) h->n16 = 0;
    if (h->n16 < h->n) {
        const uint64_t chunks = (h->n - h->n16) * (h->ld16 / 8);
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((chunks + 255) / 256, (uint64_t)h->num_cu * 32);
        if (is_f16(h))
            hipLaunchKernelGGL(k_rows_to_f16<_Float16>, dim3(std::max<uint32_t>(blocks, 1)), dim3(256), 0, cur(h).stream, h->rows_h, h->ld16, h->dim,
                               h->n16, h->n, h->sv, h->rows16, h->ld16);
        else
            hipLaunchKernelGGL(k_rows_to_f16<float>, dim3(std::max<uint32_t>(blocks, 1)), dim3(256), 0, cur(h).stream, h->rows, h->ld, h->dim,
                               h->n16, h->n, h->sv, h->rows16, h->ld16);
        LY_HIP(hipGetLastError());
        LY_HIP(hipStreamSynchronize(cur(h).stream));
    }
    h->n16 = h->n;
    h->sv16 = h->sv;
    return LYNSE_OK;
}

static int launch_scan_binary(const BinArgs& a, int metric, uint32_t grid, size_t lds, hipStream_t st) {
    static std::atomic<bool> attr_done[3] = {false, false, false};
    auto go = [&](auto kern, int slot) -> int {
        if (!attr_done[slot]) {
            LY_TRY(set_max_lds(kern, 160 * 1024 - 64));
            attr_done[slot] = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    switch (metric) {
    case M_HAMMING: return go(k_scan_binary<0>, 0);
    case M_DICE: return go(k_scan_binary<2>, 2);
    default: return go(k_scan_binary<1>, 1);
    }
}

// rows wider than 4096 bits (k_scan_binary_wide): no LDS, no register-resident words
static int launch_scan_binary_wide(const BinArgs& a, int metric, uint32_t grid, hipStream_t st) {
    switch (metric) {
    case M_HAMMING: hipLaunchKernelGGL(k_scan_binary_wide<0>, dim3(grid), dim3(256), 0, st, a); break;
    case M_DICE: hipLaunchKernelGGL(k_scan_binary_wide<2>, dim3(grid), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL(k_scan_binary_wide<1>, dim3(grid), dim3(256), 0, st, a); break;
    }
    LY_HIP(hipGetLastError());
    return LYNSE_OK;
}

// lane-per-row batched kernel: WCAP = the power of two >= words
template <int KIND>
static int launch_scan_binary_rows_k(const BinArgs& a, uint32_t grid, hipStream_t st) {
    static std::atomic<bool> attr_done[12] = {false};
    auto go = [&](auto kern, int wcap, int slot) -> int {
        const size_t lds = (size_t)4 * 64 * bin_rows_stride(wcap);
        if (!attr_done[slot]) { LY_TRY(set_max_lds(kern, lds)); attr_done[slot] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
        LY_HIP(hipGetLastError());
        return LYNSE_OK;
    };
    if (a.W & 1u) {
        if (a.W <= 1) return go(k_scan_binary_rows<KIND, 1, true>, 1, 0);
        if (a.W <= 4) return go(k_scan_binary_rows<KIND, 4, true>, 4, 1);
        if (a.W <= 8) return go(k_scan_binary_rows<KIND, 8, true>, 8, 2);
        if (a.W <= 16) return go(k_scan_binary_rows<KIND, 16, true>, 16, 3);
        if (a.W <= 32) return go(k_scan_binary_rows<KIND, 32, true>, 32, 4);
        return go(k_scan_binary_rows<KIND, 64, true>, 64, 5);
    }
    if (a.W <= 2) return go(k_scan_binary_rows<KIND, 2, false>, 2, 6);
    if (a.W <= 4) return go(k_scan_binary_rows<KIND, 4, false>, 4, 7);
    if (a.W <= 8) return go(k_scan_binary_rows<KIND, 8, false>, 8, 8);
    if (a.W <= 16) return go(k_scan_binary_rows<KIND, 16, false>, 16, 9);
    if (a.W <= 32) return go(k_scan_binary_rows<KIND, 32, false>, 32, 10);
    return go(k_scan_binary_rows<KIND, 64, false>, 64, 11);
}
static int launch_scan_binary_rows(const BinArgs& a, int metric, uint32_t grid, hipStream_t st) {
    switch (metric) {
    case M_HAMMING: return launch_scan_binary_rows_k<0>(a, grid, st);
    case M_DICE: return launch_scan_binary_rows_k<2>(a, grid, st);
    default: return launch_scan_binary_rows_k<1>(a, grid, st);
    }
}

// dynamic LDS of the select / final kernels: the cap key slots + scratch for the LDS-staged exact rescoring (rescore_keys): 48 KB hold
// the query and 15 rows of 768 floats per chunk; one workgroup per query either way
constexpr size_t SEL_LDS_MAX = 150 * 1024;
static inline uint32_t sel_lds_bytes(uint32_t cap, uint32_t nq = 0) {
    // (a widened pass — thousands of queries per launch, the k-means assignment — is bound by how many of its one-per-query workgroups
    // fit a CU: no scratch there, 4096 key slots stay 32 KB instead of 80 KB)
    if (nq > 512) return cap * 8u;
    static const size_t extra = []() { const char* e = getenv("LYNSE_HIP_SEL_SCRATCH_KB"); return (size_t)(e ? atoi(e) : 48) * 1024; }();   // (0: the direct loads, A/B)
    return (uint32_t)std::min<size_t>((size_t)cap * 8 + extra, SEL_LDS_MAX);
}

static int get_event(lynse_hip_flat* h, size_t idx, hipEvent_t* out) {
    while (cur(h).ev_pool.size() <= idx) {
        hipEvent_t e;
        LY_HIP(hipEventCreate(&e));
        cur(h).ev_pool.push_back(e);
    }
    *out = cur(h).ev_pool[idx];
    return LYNSE_OK;
}

static bool l2_plain(const lynse_hip_flat* h, uint64_t nqc, bool masked = false);   // (defined with the other int8-pass rules)
// lynse_hip_flat_coarse_scores: one emit-all stage over the whole (small) shard, then stop — the candidate buffer then holds the
// coarse score of every (row, query) exactly as the scan kernels compute it
static thread_local bool tl_coarse_dump = false;
// flat_assign_top1_device: the lane-max scan of the k-means assignment over EVERY tile of the (small) centroid store, then stop —
// k_assign_pick reads the keys (kernels.h)
static thread_local bool tl_assign = false;

// One chunk (<= QCHUNK queries) whose inputs are already in the workspace (Qf for float metrics,
// QW for binary).  Results land in ws.out_*.  `level` selects the stage plan (make_plan).
static int run_chunk(lynse_hip_flat* h, uint32_t nq, uint32_t k, uint32_t out_k, int metric, int level, hipStream_t st,
                     size_t* ev_used, std::vector<std::pair<size_t, uint64_t>>* scan_events, bool* sampled_plan,
                     const uint32_t* mask = nullptr, const uint32_t* row_ids = nullptr, bool i8c = false,
                     uint64_t* r_dst = nullptr, float* d_dst = nullptr, uint32_t* c2_dst = nullptr, uint32_t* any_ovf = nullptr,
                     const float* qsrc = nullptr, bool hdr_direct = false, bool allow_sts = true, bool* used_sts = nullptr) {
    // allow_sts / used_sts: the self-tightening single-launch scan may answer this chunk / did (an overflow of it is retried on the
    // staged plan of the same coarse pass, without a strike)
    // qsrc: the float queries of the chunk when they already live in device memory that stays valid for the whole search (no
    // staging copy into the workspace); nullptr = w.Qf.  hdr_direct: k_final writes counts + overflow flags straight into the
    // pinned header of the context (no copy kernel behind the search)
    // any_ovf: device status word k_final ORs the overflow flags into (searches in flight), or nullptr
    // r_dst / d_dst / c2_dst: where k_final writes rows, distances (stride out_k) and a second copy of the counts — the
    // caller's device arrays or the pinned staging buffer; nullptr = the workspace (copied out by the caller)
    // i8c: the coarse pass streams the SQ8 codes (1 B / element) against the symmetric int8 query image with a certified
    // error bound (k_i8c_prep_queries) instead of the f16 shadow — FLAT-IP batches of 33..256 queries
    Workspace& w = cur(h).ws;
    const float* Qf = qsrc ? qsrc : w.Qf;
    const bool binary = metric >= M_HAMMING;
    const bool asc = metric_ascending(metric);
    const bool h16 = scan_variant() == 3;
    const bool glds = scan_variant() == 0;
    // batched Hamming on the int8 MFMA: the +-1 copy of the rows is there and the batch is large enough (search_impl builds it)
    const bool bin_mfma = h16 && bin_mfma_eligible(h, metric, mask != nullptr || row_ids != nullptr, nq) && h->bpm && h->n_bpm == h->n;
    if (bin_mfma) i8c = true;   // the scan side IS the certified-int8 IP scan (exact here: margin 0)
    // squared L2 on the certified int8 pass: an inner product of augmented vectors in the negated score space (kernels.h,
    // I8cPrepArgs::aug): scan, thresholds and selects run as a best-first IP search, exact scores are -|q - v|^2
    // squared L2 on the plain codes + exact row norms (l2_plain): the <4,2,2,4> L2 tiling with int8 operands and the float epilogue
    const bool l2n = i8c && !bin_mfma && metric == M_L2 && !row_ids && l2_plain(h, nq, mask != nullptr);
    const bool aug = i8c && !bin_mfma && metric == M_L2 && !l2n;
    const bool cosq = i8c && !bin_mfma && metric == M_COS;   // cosine distance: unit vectors, negated score space (I8cPrepArgs::cosine)
    const int key_metric = (bin_mfma || aug || cosq) ? (int)M_IP : metric;   // the order of the candidate keys
    const uint32_t nslab = bin_mfma ? h->ld_bpm / 128 : i8c ? ((aug ? h->dim + h->aug_cols : h->dim) + 127) / 128 : glds ? (h->dim + GL_BK - 1) / GL_BK : (h->dim + SCAN_BK - 1) / SCAN_BK;  // h16: HK == SCAN_BK == 64
    const bool small = nq <= SCAN_BQ_SMALL;
    // mid-size batches on the int8 codes: 33..64 queries -> the 128 x 64 tiling, 65..128 -> the 256 x 128 tiling (LYNSE_HIP_MID_TILINGS=0: off)
    const int mid_env = []() { const char* e = getenv("LYNSE_HIP_MID_TILINGS"); return e ? atoi(e) : 1; }();   // (read per call: tests flip it)
    // ... unless the query-stationary tiling takes the batch (768-column codes of the IP / cosine / plain-L2 forms; a row bitmask only on
    // the IP / cosine form): waves without queries skip their MFMAs there, and it beats both mid tilings from 33 queries on (10M x 768, IP:
    // 40 / 64 / 100 / 128 queries 1.28 / 1.32 / 1.48 / 1.49 -> 1.22 / 1.25 / 1.29 / 1.31 ms; L2 1.39 / 1.43 / 1.81 / 1.89 -> 1.34 / 1.38 / 1.57 /
    // 1.62).  LYNSE_HIP_QS_MID=0: the mid tilings (A/B, tests; read per call)
    const bool qs_mid = []() { const char* e = getenv("LYNSE_HIP_QS_MID"); return !e || atoi(e) != 0; }() && qs_variant() >= 1 && qs_variant() <= 3 && i8c && !bin_mfma &&
                        !small && !row_ids && h16 && !aug && qs_width_ok(h->ld8, (h->dim + 127) / 128, mask != nullptr) && !(l2n && (h->dim + 127) / 128 == 8) && nq <= 256 &&   // (plain-L2: not on the 32-row tiles of 1024 columns)
                        (h->ld8 == 768 || nq > 64) &&   // (narrower rows: the 128 x 64 tiling keeps 33..64 queries — 6M x 256, 40 queries: 0.34 ms against 0.37)
                        (!mask || (!l2n && qs_variant() != 2 && []() { const char* e = getenv("LYNSE_HIP_QS_MASKED"); return !e || atoi(e) != 0; }()));
    const bool mid_ok = mid_env && !qs_mid && i8c && !bin_mfma && !small && !row_ids && h16 && (!mask || (aug ? h->ld8a : h->ld8) % 128 == 0);
    const bool mid64 = mid_ok && nq <= 64, mid128 = mid_ok && !mid64 && nq <= 128;
    const bool narrow = small || mid64;   // 128-row tiles
    const uint32_t qpad = small ? SCAN_BQ_SMALL : mid64 ? 64u : mid128 ? 128u : round_up(nq, SCAN_BQ_LARGE);  // > 256 queries: a widened handle, qpad / 256 chunks per launch
    const uint32_t qchunks = (small || mid64 || mid128) ? 1u : qpad / SCAN_BQ_LARGE;
    if (qchunks > 1 && (binary || i8c || mask || row_ids || !h16 || h->n > w.cap || nq > w.qcap))
        return set_error(LYNSE_ERR_INTERNAL, "more than 256 queries per pass need the widened float pipeline over <= cap rows");
    int ip_form = h->ip_form;
    if (ip_form == LYNSE_IPFORM_AUTO) ip_form = h->n < 4096 ? LYNSE_IPFORM_SINGLE : LYNSE_IPFORM_BATCH8;
    if (mask || row_ids) ip_form = LYNSE_IPFORM_SINGLE;
    if (h->dtype == LYNSE_DTYPE_F16) ip_form = LYNSE_IPFORM_F16SEQ;  // every f16 path of the reference uses the sequential kernels  // search_filtered scores every row with the single-row kernels (flat_mmap.rs:553-560)

    static std::atomic<bool> sel_attr = false;
    if (!sel_attr) {
        LY_TRY(set_max_lds(k_select<SEL_NT>, SEL_LDS_MAX));
        LY_TRY(set_max_lds(k_final<SEL_NT>, SEL_LDS_MAX));
        LY_TRY(set_max_lds(k_select_final<SEL_NT>, SEL_LDS_MAX));
        sel_attr = true;
    }

    const uint32_t sts_ld8 = cosq ? h->ld8 : h->ld8;
    const bool sts = allow_sts && sts_env_on() && level == 0 && i8c && !bin_mfma && !aug && !l2n && !small && !mid64 && !mid128 && !mask && !row_ids && h16 &&
                     k >= 1 && k <= 32 && qs_variant() >= 1 && qs_variant() <= 3 && sts_ld8 == 768 && nslab == 6 && qpad == 256 && h->n >= 65536 &&
                     h->n < 0xffffff00ull && (metric == M_IP || cosq);
    if (used_sts) *used_sts = sts;
    if (bin_mfma) {
        // (the prep kernels write every byte of the image: the lines of their queries, pad columns included, and — blocks nq .. qpad - 1 of the
        // grid — zero lines for the pad queries of the tile; up to round 5 a hipMemsetAsync of the image ran in front: a launch of its own)
        BpmPrepArgs p{};
        p.QW = w.QW; p.W = h->words; p.D = h->dim; p.qpad = qpad; p.nslab = nslab; p.nq = nq; p.img = reinterpret_cast<uint8_t*>(w.Q16);
        p.sq = w.qinv; p.bq = w.qn2; p.marg2 = w.marg2; p.thr = w.thr; p.count = w.count; p.overflow = w.overflow;
        hipLaunchKernelGGL(k_bpm_prep_queries, dim3(qpad), dim3(256), 0, st, p);
        LY_HIP(hipGetLastError());
    } else if (binary) {
        std::vector<float> thr0(nq, INFINITY);
        LY_HIP(hipMemcpyAsync(w.thr, thr0.data(), nq * 4, hipMemcpyHostToDevice, st));
        LY_HIP(hipMemsetAsync(w.count, 0, nq * 4, st));
        LY_HIP(hipMemsetAsync(w.overflow, 0, nq * 4, st));
        LY_HIP(hipStreamSynchronize(st));  // thr0 is a stack/heap temporary
    } else if (i8c) {
        I8cPrepArgs p{};
        p.Q = Qf; p.D = h->dim; p.qpad = qpad; p.nslab = nslab; p.nq = nq; p.mins = aug ? h->sq8a_mins : (cosq ? h->sq8c_mins : h->sq8_mins);
        p.scales = aug ? h->sq8a_scales : (cosq ? h->sq8c_scales : h->sq8_scales);
        p.a1 = aug ? h->sq8a_a1 : (cosq ? h->sq8c_a1 : h->sq8_a1); p.vmax = h->vmax;
        static const bool cs_env = []() { const char* e = getenv("LYNSE_HIP_I8C_CS"); return !e || atoi(e) != 0; }();   // (0: the Hoelder terms alone, A/B)
        if (cs_env) { p.a2sq = aug ? h->sq8a_a2sq : (cosq ? h->sq8c_a2sq : h->sq8_a2sq); p.eps2 = aug ? h->sq8a_eps2 : (cosq ? h->sq8c_eps2 : h->sq8_eps2); } p.cosine = cosq ? 1 : 0; p.img = reinterpret_cast<int8_t*>(w.Q16); p.aug = aug ? (int)h->aug_cols : 0; p.l2n = l2n ? 1 : 0;
        p.sq = w.qinv; p.bq = w.qn2; p.marg2 = w.marg2; p.thr = w.thr; p.count = w.count; p.overflow = w.overflow;
        p.gsync = w.gsync;
        if (sts) {   // seed the partition maxima of the self-tightening scan from a few sample rows (valid thresholds from the first tile on)
            p.codes = cosq ? h->sq8c : h->sq8; p.ld8 = sts_ld8; p.n_rows = (uint32_t)h->n; p.tile_rows = 64; p.dyn_ks = k;
            p.seed_rows = std::min<uint32_t>(1024u, 32u * k);
            p.dyn_thr = w.dyn; p.dyn_marg = w.dyn + w.qcap; p.dyn_slot = w.dyn + 2 * (size_t)w.qcap;
        }
        hipLaunchKernelGGL(k_i8c_prep_queries, dim3(qpad), dim3(256), sts ? (size_t)nslab * 128 : 0, st, p);
        LY_HIP(hipGetLastError());
    } else {
        // queries with index >= nq inside the padded tile must be finite: blocks nq .. qpad - 1 of the grid write zero lines
        PrepArgs p{};
        p.Q = Qf; p.D = h->dim; p.nq = nq; p.qpad = qpad; p.nslab = nslab; p.metric = metric; p.layout = glds ? 1 : (h16 ? 2 : 0);
        p.sv = h->sv; p.vmax = h->vmax; p.vmin = h->vmin; p.cos_degenerate = h->cos_degenerate; p.rows_integer = h->rows_integer; p.rows_nonneg = h->rows_nonneg; p.amax_v = h->amax;
        p.Q16 = w.Q16; p.qinv = w.qinv; p.qn2 = w.qn2; p.qrinv = w.qrinv; p.marg2 = w.marg2; p.thr = w.thr;
        p.count = w.count; p.overflow = w.overflow;
        hipLaunchKernelGGL(k_prep_queries, dim3(qpad), dim3(256), 0, st, p);
        LY_HIP(hipGetLastError());
    }

    // the sampled plan needs the strided-tile support of k_scan_h16; every other kernel starts at level 1
    const uint32_t plan_tile = (h16 && !binary) ? (narrow ? 128u : 256u) : 0u;
    static const int no_sample = []() { const char* e = getenv("LYNSE_HIP_NO_SAMPLE_PLAN"); return e ? atoi(e) : 0; }();
    // wave tiling of the 256 x 256 tile (measured on MI355X, 10M x 768, 256 queries): IP is fastest with <2,4,4,2> and the
    // 3+2-stage split rings, L2 / cosine (norm ring in LDS, more registers in the epilogue) with <4,2,2,4> and 2+2 stages
#ifdef LYNSE_EXPERIMENTS
    static const int w16env = []() { const char* e = getenv("LYNSE_HIP_SCAN_W16"); return e ? atoi(e) : -1; }();
#else
    constexpr int w16env = -1;  // (the product build compiles each tiling only for the metrics it is the default of)
#endif
    const bool filt = mask != nullptr || row_ids != nullptr;
    // (the subset-filter variants of <2,4,4,2> spill 96 B into the MFMA loop: masked 10M x 768 scan 7.5 ms vs 4.9 ms with <4,2,2,4>)
    const int waves16 = l2n ? 0 : i8c ? 3 : (w16env >= 0 ? w16env : ((metric == M_IP && !filt) ? 3 : 0));
    static const int no_lane_max0 = []() { const char* e = getenv("LYNSE_HIP_NO_LANE_MAX"); return e ? atoi(e) : 0; }();
    // the subset-filter kernel variants carry no lane-max code (registers)
    // (k <= 16: the best tile alone supplies k keys, so even a shard sorted by score gets a tight threshold from its best
    // sample tile; up to k = 128 the sample as a whole supplies >= 8 k keys — a clustered shard may then overflow the first
    // stage and fall back to the contiguous plan)
    const bool can_threshold_only = h16 && !binary && !filt && !no_lane_max0 && k <= 128;
    Stage assign_stage{0u, (uint32_t)h->n};   // (tl_assign) every 256-row tile of the store as a "sample" tile: lane-max keys, no threshold
    assign_stage.sample_tiles = (uint32_t)((h->n + 255) / 256);
    assign_stage.sample_stride = 256;
    const std::vector<Stage> plan = tl_assign ? std::vector<Stage>{assign_stage}
                                    : (sts || tl_coarse_dump) ? std::vector<Stage>{Stage{0u, (uint32_t)h->n}}
                                        : make_plan(h, k, (level == 0 && (!plan_tile || no_sample)) ? 1 : level, plan_tile, can_threshold_only,
                                                    // large k on the float tilings: the epilogue's cost is the emissions (one sample threshold over 1M x 128,
                                                    // k = 100, admits 1400 rows per query: 87 of the stage's 163 us) — a second threshold stage a quarter of the
                                                    // way in halves them (same box, alternating: 0.241 -> 0.233 ms per batch)
                                                    (h16 && !i8c && !binary && k >= 64) ? 4u : 0u);
    const Stage sample = (!plan.empty() && plan[0].sample_tiles) ? plan[0] : Stage{0, 0};
    *sampled_plan = sample.sample_tiles != 0;
    // sampled plan: the sample stage only has to produce a threshold -> one key per lane (its best row) instead of every
    // score, as long as that leaves comfortably more than k keys per query (2 WR keys per tile and query)
    const uint32_t sample_keys_per_tile = (small || mid64 || mid128 || waves16 != 0) ? 16u : 8u;  // 2 WR lanes per query and tile x their best 2 rows (WR = 4: 16, <4,2,2,4>: 8)
    static const int no_lane_max = []() { const char* e = getenv("LYNSE_HIP_NO_LANE_MAX"); return e ? atoi(e) : 0; }();
    // k <= keys per tile: the best sample tile alone supplies k keys (a shard sorted by score still gets a tight threshold)
    const bool sample_threshold_only = h16 && !binary && !filt && sample.sample_tiles && !no_lane_max && sample_keys_per_tile &&
                                       (k <= sample_keys_per_tile || (uint64_t)sample.sample_tiles * sample_keys_per_tile >= 8ull * k);
    // Fused sample stage (k_scan_h16<.., FS>, LYNSE_HIP_FUSED_SAMPLE=1; OFF by default): the sample tiles are scored by the
    // workgroups of the FIRST threshold stage (one each) and the grid agrees on the thresholds inside that launch — no
    // sample launch, no k_select between.  Needs one sample tile per workgroup, every workgroup resident (grid = CUs) and
    // the register-resident query constants of the certified int8 pass.  Bit-identical results (tests), but MEASURED SLOWER
    // than the two launches it replaces (MI355X, 1.25M x 768 x 256: 366 us against 33 + 26 + 261 us; s_memtime stamps,
    // scripts/dbg_fs_stamps.py: first tile 32 us, hand-over 1 27 us, select 15 us, hand-over 2 17 us, restart): a grid-wide
    // hand-over drains the LDS-DMA ring of every CU and idles the chip twice, which costs more than a kernel boundary.
    const int fs_env = []() { const char* e = getenv("LYNSE_HIP_FUSED_SAMPLE"); return e ? atoi(e) : 0; }();   // (read per call: tests flip it)
    static const int dbg_env = []() { const char* e = getenv("LYNSE_HIP_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    const bool fs = fs_env != 0 && !dbg_env && i8c && !aug && !cosq && !l2n && !bin_mfma && !mid64 && !mid128 && h->ld8 % 128 == 0 && sample_threshold_only && plan.size() >= 2 && k <= 32 &&
                    sample.sample_tiles == (uint32_t)h->num_cu && (plan[1].r1 - plan[1].r0 + 255) / 256 >= (uint32_t)h->num_cu &&
                    (uint64_t)k * 50000ull > (uint64_t)sample.sample_tiles * plan_tile &&   // (the stage behind the sample runs the DENSE epilogue)
                    []() { const char* e = getenv("LYNSE_HIP_DENSE"); return !e || atoi(e) != 0; }();
    bool plan_used_segments = false, plan_used_qs = false, plan_qs_sample = false, plan_used_qh = false;
    // the select behind the last stage + exact rescoring + final order in one launch (k_select_final); LYNSE_HIP_FUSED_TAIL=0:
    // the three separate kernels (A/B)
    const int fused_tail_env = []() { const char* e = getenv("LYNSE_HIP_FUSED_TAIL"); return e ? atoi(e) : 1; }();   // (read per call: tests flip it)
    const bool fused_tail = fused_tail_env != 0 && !plan.empty();
    SelectArgs sa_last{};
    for (size_t si = 0; si < plan.size(); ++si) {
        const Stage s = plan[si];
        const bool emit_all = si == 0 && !sts;   // (sts: the one stage is a threshold stage — its thresholds live in w.dyn and tighten while it runs)
        if (fs && si == 0) continue;   // scored inside the launch of stage 1
        const bool fs_stage = fs && si == 1;
        uint32_t qs_sample_keys = 0;   // != 0: the sample stage ran on the query-stationary tiling and left this many keys per query
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (tl_prof) {
            LY_TRY(get_event(h, (*ev_used)++, &e0));
            LY_TRY(get_event(h, (*ev_used)++, &e1));
            LY_HIP(hipEventRecord(e0, st));
        }
        uint32_t st_nseg = 0, st_seg = 0;  // segmented emission of this stage (k_select gathers)
        int a_emit_all_last = -1;
        if (binary && !bin_mfma) {
            BinArgs b{};
            b.P = h->packed; b.W = h->words; b.row0 = s.r0; b.row1 = s.r1; b.QW = w.QW; b.nq = nq;
            b.thr = w.thr; b.cand = w.cand; b.count = w.count; b.cap = w.cap; b.emit_all = emit_all ? 1 : 0;
            b.mask = mask;
            static const int rows_minq = []() { const char* e = getenv("LYNSE_HIP_BIN_ROWS_MINQ"); return e ? atoi(e) : 1; }();
            if (h->words > 16u * BIN_MAX_CHUNKS) {  // wider than 4096 bits: the generic strided kernel
                const uint32_t groups = (s.r1 - s.r0 + 31) / 32;
                const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(groups, (uint32_t)h->num_cu * 8));
                LY_TRY(launch_scan_binary_wide(b, metric, grid, st));
            } else if ((int)nq >= rows_minq || mask) {  // batched: lane-per-row, scalar query words
                uint32_t wcap = 1;
                while (wcap < h->words) wcap <<= 1;
                if (wcap != h->words) {
                    if (si == 0) {
                        hipLaunchKernelGGL(k_pad_words, dim3((nq * wcap + 255) / 256), dim3(256), 0, st, w.QW, h->words, w.QWp, wcap, nq);
                        LY_HIP(hipGetLastError());
                    }
                    b.QW = w.QWp;
                }
                const uint32_t groups = (s.r1 - s.r0 + 255) / 256;
                const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(groups, (uint32_t)h->num_cu * 8));
                LY_TRY(launch_scan_binary_rows(b, metric, grid, st));
            } else {
                const uint32_t groups = (s.r1 - s.r0 + 31) / 32;
                const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(groups, (uint32_t)h->num_cu * 8));
                const size_t lds = (size_t)nq * h->words * 8 + (size_t)nq * 4;
                LY_TRY(launch_scan_binary(b, metric, grid, lds, st));
            }
        } else {
            ScanArgs a{};
            a.V = score_rows(h); a.ld = score_ld(h); a.D = h->dim; a.row0 = s.r0; a.row1 = s.r1; a.Q16 = w.Q16;
            a.mask = mask;
            a.row_ids = row_ids;
            a.tile_stride = s.sample_stride;  // 0 = contiguous
            if (!s.sample_tiles && sample.sample_tiles && !sample_threshold_only) { a.skip_stride = sample.sample_stride; a.skip_tiles = sample.sample_tiles; }
            static const int big_rows = []() { const char* e = getenv("LYNSE_HIP_SCAN_BR"); return e ? atoi(e) : 256; }();
            uint32_t tile_rows = (glds && !small && big_rows == 256) ? 256u : (uint32_t)SCAN_BR;
            if (h16) tile_rows = narrow ? 128u : 256u;
            a.V16 = h->rows16; a.ld16 = h->ld16;
            if (i8c) { a.V16 = reinterpret_cast<const _Float16*>(bin_mfma ? (const int8_t*)h->bpm : (aug ? h->sq8a : (cosq ? h->sq8c : h->sq8))); a.ld16 = bin_mfma ? h->ld_bpm : (aug ? h->ld8a : h->ld8); }
            if (glds && !small && big_rows == 192 && metric == M_IP) tile_rows = 192u;
            a.qpad = qpad; a.nq = nq; a.nslab = nslab; a.ntiles = (s.r1 - s.r0 + tile_rows - 1) / tile_rows;
            if (s.sample_tiles) a.ntiles = s.sample_tiles;
            a.qinv = w.qinv; a.qn2 = w.qn2; a.qrinv = w.qrinv; a.thr = w.thr; a.vn2 = h->vn2; a.vrinv = h->vrinv;
            a.sv = h->sv; a.vmax2 = h->vmax * h->vmax; a.cand = w.cand; a.count = w.count; a.cap = w.cap; a.emit_all = emit_all ? 1 : 0;
            // sampled plan: the sample stage only has to produce a threshold -> one key per lane (its best row) instead of
            // every score, as long as that leaves at least k keys per query
            if (sample_threshold_only && s.sample_tiles) a.emit_all = 2;
            static const int dbg = []() { const char* e = getenv("LYNSE_HIP_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
            a.debug_flags = dbg;
            if (dbg & 2) a.emit_all = 0;
            if (dbg & 64) {  // phase timing of the LAST (largest) stage: 256 blocks x 8 waves x 4 counters, printed by the host
                // one region of 8192 words per stage, the LAST stage first (scripts/phase_timing.py reads it at offset 0), cleared per batch
                static unsigned long long* d_dbg = nullptr;
                if (!d_dbg) LY_HIP(hipMalloc(&d_dbg, 4096 * 32 * 8));
                if (si == 0) LY_HIP(hipMemsetAsync(d_dbg, 0, 4096 * 32 * 8, st));
                a.dbg = d_dbg + (size_t)std::min<size_t>(plan.size() - 1 - si, 15) * 8192;
                g_dbg_ptr = d_dbg;
            }
            const int variant = scan_variant();
            if (i8c && small) {
                a.candB = w.candB; a.segcnt = w.segcnt;
                const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu * 2);
                if (!a.emit_all) seg_geometry(grid, 4, &a.nseg, &a.seg);
                LY_TRY(launch_scan_i8c_small(a, grid, st, l2n, filt));
            } else if (mid64 || mid128) {
                a.candB = w.candB; a.segcnt = w.segcnt;
                const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu * (mid64 ? 2u : 1u));
                if (!a.emit_all) seg_geometry(grid, 4, &a.nseg, &a.seg);
                LY_TRY(launch_scan_i8c_mid(a, grid, st, mid128, l2n, filt));
            } else if (l2n) {
                a.candB = w.candB; a.segcnt = w.segcnt;
                uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu);
                const uint64_t seen_before = s.sample_tiles ? 0 : (sample.sample_tiles ? std::max<uint64_t>((uint64_t)sample.sample_tiles * plan_tile, s.r0) : s.r0);
                a.dense = (!a.emit_all && seen_before) ? 1 : 0;   // (the DENSE float epilogue, like every L2 threshold stage of the f16 shadow)
                const uint32_t qs_rt = qs_l2_rows(a);   // threshold stages on the query-stationary tiling (scan_qs.h, MET = 1)
                if (qs_rt) {
                    grid = std::min<uint32_t>((a.row1 - a.row0 + qs_rt - 1) / qs_rt, (uint32_t)h->num_cu);
                    seg_geometry(grid, 2, &a.nseg, &a.seg);
                    plan_used_qs = true;
                } else if (!a.emit_all) seg_geometry(grid, a.dense ? 4 : 2, &a.nseg, &a.seg);
                LY_TRY(launch_scan_i8l2(a, grid, st, qs_rt != 0));
            } else if (i8c) {
                a.candB = w.candB; a.segcnt = w.segcnt;
                const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu);
                // DENSE epilogue while the threshold is loose (kernels.h): the int8 margin keeps ~5x the rows an exact threshold
                // would, so a 32-query x 64-row block holds a survivor with probability ~ 2048 * 5 k / rows-seen-before; above
                // ~0.2 one integer compare per accumulator beats level 1 + (mostly) level 2.  Segments are per wave half then.
                static const int dense_env = []() { const char* e = getenv("LYNSE_HIP_DENSE"); return e ? atoi(e) : -1; }();
                const uint64_t seen_before = s.sample_tiles ? 0 : (sample.sample_tiles ? std::max<uint64_t>((uint64_t)sample.sample_tiles * plan_tile, s.r0) : s.r0);
                a.dense = (!a.emit_all && a.ld16 % 128 == 0 && seen_before &&
                           (filt || (dense_env >= 0 ? dense_env != 0 : (uint64_t)k * 50000ull > seen_before))) ? 1 : 0;   // (masked: DENSE is the one epilogue compiled with the mask)
                const bool ag = getenv("LYNSE_HIP_AG") && atoi(getenv("LYNSE_HIP_AG")) && a.ld16 % 128 == 0 && !a.emit_all && !fs_stage && !filt && !bin_mfma;
                if (sts) {
                    a.dyn_thr = w.dyn; a.dyn_marg = w.dyn + w.qcap; a.dyn_slot = w.dyn + 2 * (size_t)w.qcap; a.dyn_ks = k;
                    a.dyn_warm = h->n >= 4000000ull ? 4u : 2u;   // tiles per workgroup that only feed the maxima first (scanned again at the end)
                }
                // the threshold-only sample stage on the query-stationary tiling (4 keys per workgroup and query; the same sample rows)
                if (s.sample_tiles && !sts && qs_sample_ok(a, fs_stage, filt, bin_mfma, plan_tile)) {
                    const uint32_t nt64 = s.sample_tiles * 4u;
                    const uint32_t sgrid = std::min<uint32_t>(nt64, (uint32_t)h->num_cu);   // one 144-KB workgroup per CU: a second round of workgroups pays the launch ramp and the query image again (40 us against 30)
                    if ((uint64_t)sgrid * 4u >= 8ull * k && sgrid * 4u <= w.cap && (nt64 + sgrid - 1) / sgrid <= 32u) {   // (<= 32 tiles per workgroup: the packed position of a sample key, scan_qs.h)
                        a.ntiles = nt64;
                        LY_TRY(launch_scan_qs_sample(a, sgrid, st));
                        qs_sample_keys = sgrid * 4u;
                        plan_qs_sample = true;
                    }
                }
                if (!qs_sample_keys) {
                const bool qs = sts || (!ag && qs_scan_ok(a, fs_stage, filt, bin_mfma));   // the query-stationary tiling (scan_qs.h): two segments per workgroup and query
                uint32_t launch_grid = qs ? qs_grid(a, (uint32_t)h->num_cu) : grid;
                if (sts) {   // contiguous chunks of dyn_pitch tiles per workgroup; the pitch coprime to the partition count (scan_qs.h)
                    const uint32_t nt = (a.row1 - a.row0 + 63) / 64;
                    uint32_t pitch = (nt + launch_grid - 1) / launch_grid;
                    auto gcd = [](uint32_t x, uint32_t y) { while (y) { const uint32_t t = x % y; x = y; y = t; } return x; };
                    while (gcd(pitch, k) != 1) ++pitch;
                    a.dyn_pitch = pitch;
                    launch_grid = (nt + pitch - 1) / pitch;
                }
                if (qs) { seg_geometry(launch_grid, 2, &a.nseg, &a.seg); plan_used_qs = true; }
                else
                if (!a.emit_all) seg_geometry(grid, (a.dense ? 8 : 4) / (ag ? 2 : 1), &a.nseg, &a.seg);   // (segments per workgroup: WR, or 2 WR wave halves with DENSE)
                if (fs_stage) {
                    a.fs_stride = sample.sample_stride; a.fs_rows = (uint32_t)h->n; a.gsync = w.gsync; a.Qf = Qf; a.marg2 = w.marg2;
                    a.thr_out = w.thr; a.k = k; a.ip_form = ip_form; a.metric = metric;
                    if (getenv("LYNSE_HIP_FS_STAMPS")) a.debug_flags |= 128;
                }
                LY_TRY(launch_scan_i8c(a, launch_grid, st, fs_stage, filt, bin_mfma, qs));
                }
            } else if (h16) {
                a.candB = w.candB; a.segcnt = w.segcnt;
                if (small) {
                    const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu * 2);
                    if (!a.emit_all) seg_geometry(grid, 4, &a.nseg, &a.seg);
                    LY_TRY((launch_scan_h16<1, 4, 1, 1, 3, 3, false>(a, metric, grid, st)));
                } else if (const uint32_t sgrid = qh_sample_grid(a, s.sample_tiles, filt, qchunks, level, plan_tile, k, w.cap, (uint32_t)h->num_cu)) {
                    // the threshold-only sample stage on k_scan_qh<.., SMP>: 4 keys per workgroup and query
                    a.ntiles = s.sample_tiles * 4u;
                    LY_TRY(launch_scan_qh_sample(a, metric, sgrid, st));
                    qs_sample_keys = sgrid * 4u;
                    plan_used_qh = true;
                } else if (qh_scan_ok(a, filt, qchunks, level)) {   // the query-stationary tiling of the low-dimensional f16 shadow (scan_qh.h): one segment per workgroup and query
                    uint32_t segs = 0;
                    const uint32_t grid = qh_grid(a, (uint32_t)h->num_cu, &segs);
                    seg_geometry(grid, segs, &a.nseg, &a.seg);
                    plan_used_qh = true;
                    LY_TRY(launch_scan_qh(a, metric, grid, st));
                } else {
                    const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu);
                    // DENSE epilogue (<4,2,2,4>, L2 / cosine, unfiltered; kernels.h).  Its segments are per wave half.
                    static const int dense_env = []() { const char* e = getenv("LYNSE_HIP_DENSE"); return e ? atoi(e) : -1; }();
                    const uint64_t seen_before = s.sample_tiles ? 0 : (sample.sample_tiles ? std::max<uint64_t>((uint64_t)sample.sample_tiles * plan_tile, s.r0) : s.r0);
                    // (since the grouped branches + shared, prefetched norm loads of the DENSE epilogue it beats the two-level filter
                    // at every threshold tightness measured: 1M x 128 and 4M x 128 / 768, k = 10 and 100)
                    a.dense = (!a.emit_all && waves16 == 0 && metric != M_IP && !filt && seen_before && (dense_env >= 0 ? dense_env != 0 : true)) ? 1 : 0;
                    if (!a.emit_all) seg_geometry(grid, a.dense ? 4 : ((waves16 == 3 || waves16 == 2) ? 4 : 2), &a.nseg, &a.seg);
                    if (qchunks > 1 && !(plan.size() == 1 && (a.emit_all == 1 || (tl_assign && a.emit_all == 2))))
                        return set_error(LYNSE_ERR_INTERNAL, "the widened pipeline runs single-stage emit-all plans only");
#ifdef LYNSE_EXPERIMENTS
                    if (waves16 == 2) { LY_TRY((launch_scan_h16<2, 4, 4, 2, 2, 2, false>(a, metric, grid, st))); }
                    else
#endif
                    if (waves16 == 3) { LY_TRY((launch_scan_h16<2, 4, 4, 2, 3, 2, false>(a, metric, grid, st, qchunks))); }
                    else { LY_TRY((launch_scan_h16<4, 2, 2, 4, 2, 2, false>(a, metric, grid, st, qchunks))); }
                }
            }
#ifdef LYNSE_EXPERIMENTS
            else if (glds) {
                const bool scale = h->sv != 1.0f;
                if (small) {
                    const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu * 2);
                    LY_TRY((launch_scan_glds<1, 4, 1, 1, 4>(a, metric, scale, grid, st)));
                } else if (tile_rows == 192) {
                    const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu);
                    LY_TRY((launch_scan_glds<4, 2, 2, 3, 4>(a, metric, scale, grid, st)));
                } else if (tile_rows == 256) {
                    static const int gw16 = []() { const char* e = getenv("LYNSE_HIP_SCAN_W16"); return e ? atoi(e) : 2; }();
                    const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu);
                    if (gw16 == 3) LY_TRY((launch_scan_glds<1, 8, 8, 1, 3>(a, metric, scale, grid, st)));
                    else if (gw16 == 0) LY_TRY((launch_scan_glds<4, 2, 2, 4, 3>(a, metric, scale, grid, st)));
                    else LY_TRY((launch_scan_glds<2, 4, 4, 2, 3>(a, metric, scale, grid, st)));
                } else {
                    const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu);
                    LY_TRY((launch_scan_glds<4, 2, 2, 2, 4>(a, metric, scale, grid, st)));
                }
            } else {
                if (small) {
                    const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu * 3);
                    if (variant == 1) LY_TRY((launch_scan<1, 4, 1, 1, 1>(h, a, metric, grid, st)));
                    else LY_TRY((launch_scan<1, 4, 1, 1, 2>(h, a, metric, grid, st)));
                } else {
                    const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu);
                    if (variant == 1) LY_TRY((launch_scan<4, 2, 2, 2, 1>(h, a, metric, grid, st)));
                    else LY_TRY((launch_scan<4, 2, 2, 2, 2>(h, a, metric, grid, st)));
                }
            }
#endif
            (void)variant;
            a_emit_all_last = (int)a.emit_all;
            st_nseg = a.seg ? a.nseg : 0; st_seg = a.seg;
            plan_used_segments = plan_used_segments || a.seg != 0;
        }
        if (tl_coarse_dump) return LYNSE_OK;   // (diagnostics: the emit-all stage has written one key per row and query; nothing else runs)
        if (tl_assign) {
            if (a_emit_all_last != 2) return set_error(LYNSE_ERR_INTERNAL, "the assignment scan must run in lane-max mode");
            return LYNSE_OK;   // (k_assign_pick reads the lane-max keys)
        }
        if (tl_prof) {
            LY_HIP(hipEventRecord(e1, st));
            scan_events->push_back({*ev_used - 2, s.sample_tiles ? (uint64_t)s.sample_tiles * plan_tile
                                                                  : (uint64_t)(s.r1 - s.r0) + (fs_stage ? (uint64_t)sample.sample_tiles * plan_tile : 0ull)});
        }
        SelectArgs sa{};
        sa.cand = w.cand; sa.count = w.count; sa.overflow = w.overflow; sa.thr = w.thr; sa.marg2 = w.marg2;
        sa.k = k; sa.cap = w.cap; sa.keep_max = w.cap / 2; sa.metric = key_metric; sa.ip_form = ip_form;   // (bin_mfma: best-first +-1 dot products; aug: negated squared L2)
        sa.neg_metric1 = (aug || cosq) ? metric + 1 : 0;
        sa.exact = binary ? 1 : 0;
        static const int tighten_env = []() { const char* e = getenv("LYNSE_HIP_TIGHTEN"); return e ? atoi(e) : 1; }();
        // (worth its ~k exact rescorings per query and stage where the margin is wide — the int8 pass — or k is small)
        sa.tighten = (!binary && tighten_env && (i8c || k <= 32)) ? 1 : 0;
        sa.drop_sentinels = (mask && emit_all) ? 1 : 0;
        sa.emit_all_n = emit_all ? (s.sample_tiles ? (int)(s.sample_tiles * plan_tile) : (int)(s.r1 - s.r0)) : -1;
        if (!binary && emit_all && sample_threshold_only) {
            sa.threshold_only = 1;
            sa.drop_sentinels = 1;  // a lane whose rows are all masked / out of range wrote the sentinel
            sa.emit_all_n = qs_sample_keys ? (int)qs_sample_keys : (int)(s.sample_tiles * sample_keys_per_tile);
        }
        sa.Qf = Qf; sa.V = score_rows(h); sa.ld = score_ld(h); sa.D = h->dim;
        sa.candB = w.candB; sa.segcnt = w.segcnt; sa.seg = st_seg; sa.nseg = st_nseg;
        sa.abort_word = fs_stage ? w.gsync + 2 : nullptr;
        if (getenv("LYNSE_HIP_SEL_STAMPS"))   // debugging: phase stamps of the select behind stage si ([stage][query][8] behind the fs stamps)
            sa.stamps = reinterpret_cast<unsigned long long*>(w.gsync + 64) + (size_t)si * 256 * 8;
        if (fs_stage && getenv("LYNSE_HIP_DEBUG_FS")) return LYNSE_OK;   // debugging: stop behind the fused stage (lynse_hip_debug_workspace)
        if (fused_tail && si + 1 == plan.size()) { sa_last = sa; continue; }  // the last select runs inside k_select_final
        // (round 5, tried: the survivors of the last select handed to the final rescoring in LDS instead of through cand[q] — a store, its
        // acknowledgement and a load less in the tail's chain: 10M 1.8216 / 1.8213 / 1.8527 against 1.8206 / 1.8455 / 1.8381 ms, shard
        // 0.2835 / 0.2875 / 0.2824 against 0.2877 / 0.2884 / 0.2950: nothing, removed again, scripts/gpu_r5_tail1.sh)
        // (round 5, tried: the LAST select without the exact-rescored threshold, its coarse-rule survivors all rescored in the final pass —
        // 116 instead of 35 rows per query, two rounds of the final rescoring: 10M step +10 us, shard step +6 us; dropped)
        sa.lds_bytes = sel_lds_bytes(w.cap, nq);
        hipLaunchKernelGGL(k_select<SEL_NT>, dim3(nq), dim3(SEL_NT), sa.lds_bytes, st, sa);
        LY_HIP(hipGetLastError());
    }
    if (tl_prof && (!binary || bin_mfma)) {   // (batched Hamming on the matrix pipe runs the float pipeline's plans)
        const uint64_t tiling = plan_used_qs ? 0x81u : plan_used_qh ? 0x82u : (small || mid64) ? 0x14u : ((mid128 || waves16 == 3 || waves16 == 2) ? 0x24u : 0x42u);   // (0x81: query-stationary threshold stages; 0x82: those of the low-dimensional f16 shadow, scan_qh.h)
        std::lock_guard<std::mutex> plk(h->prof_mu);
        h->prof.last_plan = (sample.sample_tiles ? 1u : 0u) | ((sample.sample_tiles && sample_threshold_only) ? 2u : 0u) | (i8c ? 4u : 0u) |
                            (plan_used_segments ? 8u : 0u) | (small ? 16u : 0u) | (fs ? 128u : 0u) | ((uint64_t)(plan.size() & 0xff) << 8) | (tiling << 16) |
                            (sts ? (1ull << 24) : 0ull) | (plan_qs_sample ? (1ull << 25) : 0ull);   // bit 24: self-tightening single-launch scan; bit 25: sample stage on the query-stationary tiling
    }
    FinalArgs fa{};
    fa.cand = w.cand; fa.count = w.count; fa.k = k; fa.out_k = out_k; fa.cap = w.cap; fa.metric = key_metric; fa.ip_form = ip_form;
    fa.ham_dim = bin_mfma ? h->dim : 0u;
    fa.neg_metric1 = (aug || cosq) ? metric + 1 : 0;
    fa.exact = binary ? 1 : 0; fa.Qf = Qf; fa.V = score_rows(h); fa.ld = score_ld(h); fa.D = h->dim;
    fa.row_stride = h->row_stride; fa.row_offset = h->row_offset;
    fa.nan_rows = (!binary && !mask && !row_ids && h->n <= 0xffffffffull) ? (uint32_t)h->n : 0u;   // (FinalArgs::nan_rows: the all-NaN query of an unfiltered FLAT search)
    fa.out_rows = r_dst ? r_dst : w.out_rows; fa.out_dists = d_dst ? d_dst : w.out_dists; fa.out_counts = w.out_counts;
    fa.out_counts2 = c2_dst;
    fa.pool_total = tl_prof ? w.pool_total : nullptr;
    fa.overflow = w.overflow; fa.any_overflow = any_ovf;
    if (hdr_direct) { fa.h_hdr = w.h_hdr; fa.hdr_q = w.qcap; }
    if (fused_tail) {
        TailArgs ta{sa_last, fa};
        ta.s.hand_exact = getenv("LYNSE_HIP_HAND_EXACT") ? atoi(getenv("LYNSE_HIP_HAND_EXACT")) : 1;   // (0: the final rescoring always reads its rows, A/B)
        ta.s.lds_bytes = ta.f.lds_bytes = sel_lds_bytes(w.cap, nq);
        hipLaunchKernelGGL(k_select_final<SEL_NT>, dim3(nq), dim3(SEL_NT), ta.s.lds_bytes, st, ta);
        LY_HIP(hipGetLastError());
        (void)asc;
        return LYNSE_OK;
    }
    if (i8c && !bin_mfma) {  // a few hundred survivors per query inside the int8 margin: spread their exact rescoring over the chip
        hipLaunchKernelGGL(k_rescore_pool<256>, dim3(nq, 4), dim3(256), 0, st, fa);
        LY_HIP(hipGetLastError());
        fa.exact = 1;
    }
    fa.lds_bytes = sel_lds_bytes(w.cap, nq);
    hipLaunchKernelGGL(k_final<SEL_NT>, dim3(nq), dim3(SEL_NT), fa.lds_bytes, st, fa);
    LY_HIP(hipGetLastError());
    (void)asc;
    return LYNSE_OK;
}

// kmeans::assign_metric (kmeans.rs:237-264) for the device k-means: the nearest centroid of every data row, WITHOUT the general
// top-1 search.  `h` is the centroid store (a widened handle: h->qchunk data rows per pass, <= cap centroids), d_q the data rows in
// device memory.  Per pass of <= qchunk data rows: query image + certified margin (k_prep_queries), ONE lane-max scan over all
// centroid tiles (run_chunk under tl_assign), k_assign_pick.  d_ids[i] = the centroid of row i, or 0xffffffff for the rows whose two
// best coarse centroids lie within the certified margin: their indices come back in d_redo[0 .. *d_redo_count) and the caller answers
// them with the exact top-1 search (lynse_hip_flat_search_f32_device), whose first-strictly-smaller rule they need.
// Returns LYNSE_ERR_UNSUPPORTED when the shape is not the one this path is built for (the caller falls back to the search).
static int flat_assign_top1_device(lynse_hip_flat* h, const float* d_q, uint64_t nq, int metric, uint32_t* d_ids, uint32_t* d_redo,
                                   uint32_t* d_redo_count) {
    if (!h || !d_q || !d_ids || !d_redo || !d_redo_count) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    if (metric < M_IP || metric > M_COS) return set_error(LYNSE_ERR_UNSUPPORTED, "float metrics only");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    if (scan_variant() != 3 || h->n == 0 || h->n > h->cap || h->qchunk <= QCHUNK || h->packed_only || h->dtype != LYNSE_DTYPE_F32)
        return set_error(LYNSE_ERR_UNSUPPORTED, "not the widened float shape of the assignment pass");
    struct Slot0 { int prev; Slot0() : prev(tl_ctx_slot) { tl_ctx_slot = 0; } ~Slot0() { tl_ctx_slot = prev; } } scope;
    LY_TRY(finalize_locked(h));
    if (metric == M_COS && h->cos_degenerate) return set_error(LYNSE_ERR_UNSUPPORTED, "degenerate norms: the exact search answers");
    LY_TRY(ensure_shadow_locked(h));
    LY_TRY(ensure_workspace(h, 1));
    Workspace& w = cur(h).ws;
    hipStream_t st = cur(h).stream;
    const uint32_t ntiles = (uint32_t)((h->n + 255) / 256);
    const uint32_t nkeys = ntiles * ((metric == M_IP) ? 16u : 8u);   // 2 WR lanes per tile and data row x their best two (<2,4,4,2> / <4,2,2,4>)
    if (nkeys > w.cap) return set_error(LYNSE_ERR_UNSUPPORTED, "lane-max keys exceed the candidate slots");
    LY_HIP(hipMemsetAsync(d_redo_count, 0, 4, st));
    size_t ev_used = 0;
    std::vector<std::pair<size_t, uint64_t>> scan_events;
    struct Flag { Flag() { tl_assign = true; } ~Flag() { tl_assign = false; } } flag;
    const bool prof_prev = tl_prof;
    tl_prof = false;
    int rc = LYNSE_OK;
    for (uint64_t q0 = 0; q0 < nq && rc == LYNSE_OK; q0 += w.qcap) {
        const uint32_t nqc = (uint32_t)std::min<uint64_t>(w.qcap, nq - q0);
        bool sampled = false;
        rc = run_chunk(h, nqc, 1, 1, metric, 0, st, &ev_used, &scan_events, &sampled, nullptr, nullptr, false, nullptr, nullptr, nullptr, nullptr,
                       d_q + q0 * h->dim, false, false, nullptr);
        if (rc != LYNSE_OK) break;
        int ip_form = h->ip_form;
        if (ip_form == LYNSE_IPFORM_AUTO) ip_form = h->n < 4096 ? LYNSE_IPFORM_SINGLE : LYNSE_IPFORM_BATCH8;   // (run_chunk's rule)
        hipLaunchKernelGGL(k_assign_pick, dim3((nqc + 3) / 4), dim3(256), 0, st, w.cand, w.cap, nkeys, w.marg2, metric_ascending(metric) ? 1 : 0,
                           nqc, (uint32_t)q0, d_ids, d_redo, d_redo_count, d_q, score_rows(h), score_ld(h), h->dim, metric, ip_form);
        if (hipGetLastError() != hipSuccess) rc = set_error(LYNSE_ERR_DEVICE, "k_assign_pick launch failed");
    }
    tl_prof = prof_prev;
    if (rc != LYNSE_OK) { (void)hipStreamSynchronize(st); return rc; }
    LY_HIP(hipStreamSynchronize(st));
    return LYNSE_OK;
}

// The fused single-launch search (k_small_search): a few queries over a shard small enough that the eight dependent
// launches of the staged pipeline, not the bytes, are the cost.  Exact scores straight from the f32 rows.
static bool small_path_ok(const lynse_hip_flat* h, uint64_t nq, uint32_t kk, int metric, bool filtered) {
    static const int env = []() { const char* e = getenv("LYNSE_HIP_FUSED"); return e ? atoi(e) : 1; }();
    static const uint64_t max_bytes = []() { const char* e = getenv("LYNSE_HIP_FUSED_MAX_MB"); return (uint64_t)(e ? atoi(e) : 64) << 20; }();
    return env && !h->no_fused && metric <= M_COS && !filtered && h->dtype == LYNSE_DTYPE_F32 && !h->packed_only && nq <= (uint64_t)SMALL_MAX_Q &&
           kk <= (uint32_t)SMALL_MAX_K && (uint64_t)h->n * h->ld * 4 * nq <= max_bytes &&
           (size_t)((h->dim + 3) / 4 * 4) * 4 + (size_t)128 * 1024 + 64 <= 150u * 1024u;
}

// SmallRows: the row matrix a fused search scans when it is not the handle's own (IVF: the centroid matrix, the slab with
// its row map); SmallIvf: the probed-lists mode of k_small_search.
struct SmallRows { const float* V; uint32_t ld; uint64_t n; int ip_form; uint64_t row_stride, row_offset; };
struct SmallIvf { const uint64_t* probes; uint32_t nprobe, nlist; const uint64_t* list_off; const uint32_t* orig; int flag_empty; };

static int run_small(lynse_hip_flat* h, uint32_t nq, uint32_t k, uint32_t out_k, int metric, hipStream_t st, size_t* ev_used,
                     std::vector<std::pair<size_t, uint64_t>>* scan_events, uint64_t* r_dst, float* d_dst, uint32_t* c_dst, uint32_t* o_dst,
                     const float* d_q, const SmallRows* rows = nullptr, const SmallIvf* ivf = nullptr) {
    // r_dst / d_dst / c_dst / o_dst: where the last workgroup writes rows, distances, counts and overflow flags — the
    // caller's device buffers, or pinned host memory (device-visible): no copy kernels behind the search
    Workspace& w = cur(h).ws;
    SmallArgs a{};
    a.V = h->rows; a.ld = h->ld; a.D = h->dim; a.n = (uint32_t)h->n; a.Qf = d_q; a.nq = nq; a.k = k; a.out_k = out_k; a.metric = metric;
    int ip_form = h->ip_form;
    if (ip_form == LYNSE_IPFORM_AUTO) ip_form = h->n < 4096 ? LYNSE_IPFORM_SINGLE : LYNSE_IPFORM_BATCH8;  // flat_mmap.rs:4852-4854
    a.ip_form = ip_form;
    a.row_stride = h->row_stride; a.row_offset = h->row_offset;
    uint64_t n_scan = h->n;  // rows one query scans (IVF mode: an upper bound, for the grid size only)
    if (rows) { a.V = rows->V; a.ld = rows->ld; a.n = (uint32_t)rows->n; a.ip_form = rows->ip_form; a.row_stride = rows->row_stride; a.row_offset = rows->row_offset; n_scan = rows->n; }
    if (ivf) { a.probes = ivf->probes; a.nprobe = ivf->nprobe; a.nlist = ivf->nlist; a.list_off = ivf->list_off; a.orig = ivf->orig; a.flag_empty = ivf->flag_empty; }
    a.part = w.small_part; a.ticket = w.small_ticket;
    a.out_rows = r_dst; a.out_dists = d_dst; a.out_counts = c_dst; a.overflow = o_dst;
    // two workgroups per CU (16 waves: the scan is a chain of dependent load batches per wave), one merge list per workgroup:
    // at most SMALL_NT lists (one per thread of the last workgroup) and 128 KB of them in the hand-over buffer
    uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>((uint64_t)std::min(2 * h->num_cu, SMALL_NT), 16384u / std::max<uint32_t>(k, 1)), (n_scan + 127) / 128));
    {   // whole passes for every workgroup (a pass = 128 rows per workgroup): 100k rows over 512 workgroups are 1.53 passes — half of the
        // workgroups idle through the second one and all 512 queue at the ticket; 391 workgroups x 2 passes: kernel 21.4 -> 19.7 us
        const uint64_t passes = (n_scan + (uint64_t)grid * 128 - 1) / ((uint64_t)grid * 128);
        if (passes > 1) grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(grid, (n_scan + 128 * passes - 1) / (128 * passes)));
    }
    if (const char* ge = getenv("LYNSE_HIP_SMALL_GRID")) grid = std::max(1u, std::min<uint32_t>(grid, (uint32_t)atoi(ge)));   // development
    auto pow2_at_least = [](uint32_t v) { uint32_t p = 2; while (p < v) p <<= 1; return p; };
    const size_t lds = (size_t)((h->dim + 3) / 4 * 4) * 4 + (size_t)SMALL_NT * 8 + (size_t)pow2_at_least(std::min<uint32_t>(k, grid) * k) * 8 + 64;  // the query, the wave lists (later the heads of the merge), the merge's candidate keys
    static std::atomic<bool> small_attr = false;
    if (!small_attr) { LY_TRY(set_max_lds(k_small_search, 160 * 1024 - 1024)); small_attr = true; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = tl_prof && ev_used && scan_events;
    if (timed) {
        LY_TRY(get_event(h, (*ev_used)++, &e0));
        LY_TRY(get_event(h, (*ev_used)++, &e1));
        LY_HIP(hipEventRecord(e0, st));
    }
    static const bool small_dbg = []() { const char* e = getenv("LYNSE_HIP_SMALL_DBG"); return e && atoi(e) != 0; }();
    static unsigned long long* d_dbg = nullptr;
    if (small_dbg) {   // development: where the time of the fused search goes (stderr; synchronises)
        if (!d_dbg) LY_HIP(hipMalloc(&d_dbg, (size_t)SMALL_NT * 4 * 8));
        LY_HIP(hipMemsetAsync(d_dbg, 0, (size_t)SMALL_NT * 4 * 8, st));
        a.dbg = d_dbg;
    }
    hipLaunchKernelGGL(k_small_search, dim3(grid), dim3(SMALL_NT), lds, st, a);
    LY_HIP(hipGetLastError());
    if (small_dbg) {
        std::vector<unsigned long long> hd((size_t)grid * 4);
        LY_HIP(hipMemcpyAsync(hd.data(), d_dbg, hd.size() * 8, hipMemcpyDeviceToHost, st));
        LY_HIP(hipStreamSynchronize(st));
        unsigned long long t0 = ~0ull, s_last = 0, scan_end = 0, tick = 0, fin = 0, scan_end_min = ~0ull, scan_sum = 0, merge_sum = 0;
        for (uint32_t b = 0; b < grid; ++b) {
            t0 = std::min(t0, hd[b * 4]); s_last = std::max(s_last, hd[b * 4]);
            scan_end = std::max(scan_end, hd[b * 4 + 1]); scan_end_min = std::min(scan_end_min, hd[b * 4 + 1]);
            tick = std::max(tick, hd[b * 4 + 2]); fin = std::max(fin, hd[b * 4 + 3]);
            scan_sum += hd[b * 4 + 1] - hd[b * 4]; merge_sum += hd[b * 4 + 2] - hd[b * 4 + 1];
        }
        fprintf(stderr, "k_small_search grid %u lds %zu (10 ns ticks from the first start): last start %llu, first scan end %llu, last scan end %llu, last ticket %llu, end %llu; "
                        "mean scan %llu, mean wave merge %llu\n", grid, lds, s_last - t0, scan_end_min - t0, scan_end - t0, tick - t0, fin - t0, scan_sum / grid, merge_sum / grid);
    }
    if (timed) {
        LY_HIP(hipEventRecord(e1, st));
        scan_events->push_back({*ev_used - 2, (uint64_t)h->n * nq});
        std::lock_guard<std::mutex> plk(h->prof_mu);
        h->prof.last_plan = 32u | (1u << 8);  // bit 5: fused single-launch search
    }
    return LYNSE_OK;
}

// ------------------------------------------------------------------------------ SQ8 two-pass ----
// ensure_sq8 (flat_mmap.rs:375-386) + SQ8Data::from_f32_parallel (:5685-5737): (re)built over ALL rows whenever rows were
// appended since the last build (min / max are collection-wide).
// Rows appended to a shard that already has its codes (Collection::flush hands over <= 10,000 rows at a time, src/engine.rs:93-94):
// the new rows' per-dimension min / max are MERGED into the stored table; when no entry moves, the collection-wide fit — and with
// it every existing code — is what a full rebuild would produce (SQ8Data::from_f32_parallel, flat_mmap.rs:5685-5737, fits min / max
// over all rows), so only the new rows are quantised (and the max row L1 norm / the non-finite count keep accumulating).  A moved
// entry means new scales: everything is coded again, from the already merged table.  Returns the first row that still needs codes
// (0 = all of them) and whether `mm` already covers every row.
static int sq8_merge_new_rows(lynse_hip_flat* h, uint32_t* mm, uint32_t D, uint64_t n_done, const float* row_scale, uint64_t* first_row, bool* mm_covers_all) {
    *first_row = 0;
    *mm_covers_all = false;
    if (n_done == 0 || n_done >= h->n) return LYNSE_OK;
    if (!h->sq8_mm_prev) LY_HIP(hipMalloc(&h->sq8_mm_prev, ((size_t)(h->dim + 64) * 2 + 1) * 4));
    hipStream_t st = cur(h).stream;
    uint32_t* flag = h->sq8_mm_prev + 2 * (size_t)D;
    LY_HIP(hipMemcpyAsync(h->sq8_mm_prev, mm, (size_t)D * 8, hipMemcpyDeviceToDevice, st));
    LY_HIP(hipMemsetAsync(flag, 0, 4, st));
    const uint64_t nn = h->n - n_done;
    const uint32_t gx = (D + 255) / 256;
    const uint32_t gy = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(nn / 256, 1), (uint64_t)h->num_cu * 8);
    if (is_f16(h)) hipLaunchKernelGGL(k_sq8_minmax<_Float16>, dim3(gx, gy), dim3(256), 0, st, (const _Float16*)h->rows_h + n_done * h->ld16, h->ld16, D, nn, mm, mm + D,
                                      row_scale ? row_scale + n_done : nullptr);
    else hipLaunchKernelGGL(k_sq8_minmax<float>, dim3(gx, gy), dim3(256), 0, st, h->rows + n_done * h->ld, h->ld, D, nn, mm, mm + D, row_scale ? row_scale + n_done : nullptr);
    hipLaunchKernelGGL(k_mm_changed, dim3((2 * D + 255) / 256), dim3(256), 0, st, mm, h->sq8_mm_prev, 2 * D, flag);
    LY_HIP(hipGetLastError());
    uint32_t moved = 1;
    LY_HIP(hipMemcpyAsync(&moved, flag, 4, hipMemcpyDeviceToHost, st));
    LY_HIP(hipStreamSynchronize(st));
    *mm_covers_all = true;
    if (!moved) *first_row = n_done;
    return LYNSE_OK;
}

static int ensure_sq8_locked(lynse_hip_flat* h) {
    if (h->n_sq8 == h->n && h->sq8) return LYNSE_OK;
    bool kept = h->sq8 != nullptr && h->sq8_mins != nullptr && h->sq8_cap >= h->n;   // the existing codes survive (no reallocation)
    if (h->sq8_cap < h->n) {
        for (void* p : {(void*)h->sq8, (void*)h->sq8_sum, (void*)h->sq8_sum2})
            if (p) (void)hipFree(p);
        h->sq8 = nullptr; h->sq8_sum = nullptr; h->sq8_sum2 = nullptr;
        const uint64_t cap = std::max<uint64_t>(h->capacity, h->n);
        LY_HIP(hipMalloc(&h->sq8, (size_t)cap * h->ld8 + 256));
        LY_HIP(hipMalloc(&h->sq8_sum, ((size_t)cap + 256) * 4));
        LY_HIP(hipMalloc(&h->sq8_sum2, ((size_t)cap + 256) * 4));
        LY_HIP(hipMemsetAsync(h->sq8_sum, 0, ((size_t)cap + 256) * 4, cur(h).stream));
        LY_HIP(hipMemsetAsync(h->sq8_sum2, 0, ((size_t)cap + 256) * 4, cur(h).stream));
        h->sq8_cap = cap;
    }
    if (!h->sq8_mins) {
        LY_HIP(hipMalloc(&h->sq8_mins, (size_t)h->dim * 4));
        LY_HIP(hipMalloc(&h->sq8_scales, (size_t)h->dim * 4));
        LY_HIP(hipMalloc(&h->sq8_mm, (size_t)h->dim * 8));
        LY_HIP(hipMalloc(&h->sq8_stats, 16));   // (four words: k_sq8_quantize)
    }
    uint64_t r0 = 0;
    bool mm_done = false;
    if (kept) LY_TRY(sq8_merge_new_rows(h, h->sq8_mm, h->dim, h->n_sq8, nullptr, &r0, &mm_done));
    const uint32_t gx = (h->dim + 255) / 256;
    std::vector<uint32_t> init;
    if (!mm_done) {
        init.resize((size_t)h->dim * 2);
        for (uint32_t d = 0; d < h->dim; ++d) { init[d] = f32_to_ord(INFINITY); init[h->dim + d] = f32_to_ord(-INFINITY); }
        LY_HIP(hipMemcpyAsync(h->sq8_mm, init.data(), init.size() * 4, hipMemcpyHostToDevice, cur(h).stream));
        const uint32_t gy = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(h->n / 256, 1), (uint64_t)h->num_cu * 8);
        if (is_f16(h)) hipLaunchKernelGGL(k_sq8_minmax<_Float16>, dim3(gx, gy), dim3(256), 0, cur(h).stream, (const _Float16*)h->rows_h, h->ld16, h->dim, h->n, h->sq8_mm, h->sq8_mm + h->dim);
        else hipLaunchKernelGGL(k_sq8_minmax<float>, dim3(gx, gy), dim3(256), 0, cur(h).stream, h->rows, h->ld, h->dim, h->n, h->sq8_mm, h->sq8_mm + h->dim);
    }
    if (r0 == 0) {   // a new fit: every row is coded (again)
        LY_HIP(hipMemsetAsync(h->sq8_stats, 0, 16, cur(h).stream));
        hipLaunchKernelGGL(k_sq8_scales, dim3(gx), dim3(256), 0, cur(h).stream, h->sq8_mm, h->sq8_mm + h->dim, h->dim, h->sq8_mins, h->sq8_scales);
    }
    const uint64_t nn = h->n - r0;
    const uint32_t qgrid = (uint32_t)std::min<uint64_t>((nn + 3) / 4, (uint64_t)h->num_cu * 16);
    if (is_f16(h)) hipLaunchKernelGGL(k_sq8_quantize<_Float16>, dim3(qgrid), dim3(256), 0, cur(h).stream,
                       (const _Float16*)h->rows_h + r0 * h->ld16, h->ld16, h->dim, nn, h->sq8_mins, h->sq8_scales, h->sq8 + r0 * h->ld8, h->ld8, h->sq8_sum + r0, h->sq8_sum2 + r0, h->sq8_stats);
    else hipLaunchKernelGGL(k_sq8_quantize<float>, dim3(qgrid), dim3(256), 0, cur(h).stream,
                       h->rows + r0 * h->ld, h->ld, h->dim, nn, h->sq8_mins, h->sq8_scales, h->sq8 + r0 * h->ld8, h->ld8, h->sq8_sum + r0, h->sq8_sum2 + r0, h->sq8_stats);
    LY_HIP(hipGetLastError());
    uint32_t qst[4] = {0, 0, 0, 0};
    LY_HIP(hipMemcpyAsync(qst, h->sq8_stats, 16, hipMemcpyDeviceToHost, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));  // `init` is a temporary
    h->sq8_a1 = qst[0];
    h->sq8_a2sq = qst[2];
    memcpy(&h->sq8_eps2, &qst[3], 4);
    h->sq8_finite = qst[1] == 0;
    h->n_sq8 = h->n;
    return LYNSE_OK;
}

// The augmented code set of the L2 form (rows [v, |v|^2]): rebuilt over all rows when rows were appended, like the SQ8 set.
static int ensure_sq8a_locked(lynse_hip_flat* h) {
    if (h->n_sq8a == h->n && h->sq8a) return LYNSE_OK;
    const uint32_t DA = h->dim + h->aug_cols;
    if (h->sq8a_cap < h->n) {
        if (h->sq8a) (void)hipFree(h->sq8a);
        h->sq8a = nullptr;
        const uint64_t cap = std::max<uint64_t>(h->capacity, h->n);
        LY_HIP(hipMalloc(&h->sq8a, (size_t)cap * h->ld8a + 256));
        h->sq8a_cap = cap;
    }
    if (!h->sq8a_mins) {
        LY_HIP(hipMalloc(&h->sq8a_mins, (size_t)DA * 4));
        LY_HIP(hipMalloc(&h->sq8a_scales, (size_t)DA * 4));
        LY_HIP(hipMalloc(&h->sq8a_mm, (size_t)DA * 8));
        LY_HIP(hipMalloc(&h->sq8a_stats, 16));   // (four words: k_sq8_quantize)
    }
    LY_HIP(hipMemsetAsync(h->sq8a_stats, 0, 16, cur(h).stream));
    std::vector<uint32_t> init((size_t)DA * 2);
    for (uint32_t d = 0; d < DA; ++d) { init[d] = f32_to_ord(INFINITY); init[DA + d] = f32_to_ord(-INFINITY); }
    LY_HIP(hipMemcpyAsync(h->sq8a_mm, init.data(), init.size() * 4, hipMemcpyHostToDevice, cur(h).stream));
    const uint32_t gx = (h->dim + 255) / 256;
    const uint32_t gy = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(h->n / 256, 1), (uint64_t)h->num_cu * 8);
    if (is_f16(h)) hipLaunchKernelGGL(k_sq8_minmax<_Float16>, dim3(gx, gy), dim3(256), 0, cur(h).stream, (const _Float16*)h->rows_h, h->ld16, h->dim, h->n, h->sq8a_mm, h->sq8a_mm + DA);
    else hipLaunchKernelGGL(k_sq8_minmax<float>, dim3(gx, gy), dim3(256), 0, cur(h).stream, h->rows, h->ld, h->dim, h->n, h->sq8a_mm, h->sq8a_mm + DA);
    hipLaunchKernelGGL(k_vec_minmax, dim3((uint32_t)std::min<uint64_t>((h->n + 255) / 256, (uint64_t)h->num_cu * 8)), dim3(256), 0, cur(h).stream,
                       h->vn2, h->n, h->sq8a_mm + h->dim, h->sq8a_mm + DA + h->dim, h->aug_cols);
    hipLaunchKernelGGL(k_sq8_scales, dim3((DA + 255) / 256), dim3(256), 0, cur(h).stream, h->sq8a_mm, h->sq8a_mm + DA, DA, h->sq8a_mins, h->sq8a_scales);
    if (is_f16(h)) hipLaunchKernelGGL(k_sq8_quantize<_Float16>, dim3((uint32_t)std::min<uint64_t>((h->n + 3) / 4, (uint64_t)h->num_cu * 16)), dim3(256), 0, cur(h).stream,
                       (const _Float16*)h->rows_h, h->ld16, h->dim, h->n, h->sq8a_mins, h->sq8a_scales, h->sq8a, h->ld8a, (int*)nullptr, (int*)nullptr, h->sq8a_stats, h->vn2, h->aug_cols);
    else hipLaunchKernelGGL(k_sq8_quantize<float>, dim3((uint32_t)std::min<uint64_t>((h->n + 3) / 4, (uint64_t)h->num_cu * 16)), dim3(256), 0, cur(h).stream,
                       h->rows, h->ld, h->dim, h->n, h->sq8a_mins, h->sq8a_scales, h->sq8a, h->ld8a, (int*)nullptr, (int*)nullptr, h->sq8a_stats, h->vn2, h->aug_cols);
    LY_HIP(hipGetLastError());
    uint32_t qst[4] = {0, 0, 0, 0};
    LY_HIP(hipMemcpyAsync(qst, h->sq8a_stats, 16, hipMemcpyDeviceToHost, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));  // `init` is a temporary
    h->sq8a_a1 = qst[0];
    h->sq8a_a2sq = qst[2];
    memcpy(&h->sq8a_eps2, &qst[3], 4);
    h->sq8a_finite = qst[1] == 0;
    h->n_sq8a = h->n;
    return LYNSE_OK;
}

// The code set of the cosine form: SQ8 codes of the rows scaled to unit norm with the stored reciprocal norm.
static int ensure_sq8c_locked(lynse_hip_flat* h) {
    if (h->n_sq8c == h->n && h->sq8c) return LYNSE_OK;
    const uint32_t D = h->dim;
    const bool kept = h->sq8c != nullptr && h->sq8c_mins != nullptr && h->sq8c_cap >= h->n;
    if (h->sq8c_cap < h->n) {
        if (h->sq8c) (void)hipFree(h->sq8c);
        h->sq8c = nullptr;
        const uint64_t cap = std::max<uint64_t>(h->capacity, h->n);
        LY_HIP(hipMalloc(&h->sq8c, (size_t)cap * h->ld8 + 256));
        h->sq8c_cap = cap;
    }
    if (!h->sq8c_mins) {
        LY_HIP(hipMalloc(&h->sq8c_mins, (size_t)D * 4));
        LY_HIP(hipMalloc(&h->sq8c_scales, (size_t)D * 4));
        LY_HIP(hipMalloc(&h->sq8c_mm, (size_t)D * 8));
        LY_HIP(hipMalloc(&h->sq8c_stats, 16));   // (four words: k_sq8_quantize)
    }
    uint64_t r0 = 0;
    bool mm_done = false;
    if (kept) LY_TRY(sq8_merge_new_rows(h, h->sq8c_mm, D, h->n_sq8c, h->vrinv, &r0, &mm_done));   // (appended rows inside the fitted ranges: only they are coded)
    const uint32_t gx = (D + 255) / 256;
    std::vector<uint32_t> init;
    if (!mm_done) {
        init.resize((size_t)D * 2);
        for (uint32_t d = 0; d < D; ++d) { init[d] = f32_to_ord(INFINITY); init[D + d] = f32_to_ord(-INFINITY); }
        LY_HIP(hipMemcpyAsync(h->sq8c_mm, init.data(), init.size() * 4, hipMemcpyHostToDevice, cur(h).stream));
        const uint32_t gy = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(h->n / 256, 1), (uint64_t)h->num_cu * 8);
        if (is_f16(h)) hipLaunchKernelGGL(k_sq8_minmax<_Float16>, dim3(gx, gy), dim3(256), 0, cur(h).stream, (const _Float16*)h->rows_h, h->ld16, D, h->n, h->sq8c_mm, h->sq8c_mm + D, h->vrinv);
        else hipLaunchKernelGGL(k_sq8_minmax<float>, dim3(gx, gy), dim3(256), 0, cur(h).stream, h->rows, h->ld, D, h->n, h->sq8c_mm, h->sq8c_mm + D, h->vrinv);
    }
    if (r0 == 0) {
        LY_HIP(hipMemsetAsync(h->sq8c_stats, 0, 16, cur(h).stream));
        hipLaunchKernelGGL(k_sq8_scales, dim3(gx), dim3(256), 0, cur(h).stream, h->sq8c_mm, h->sq8c_mm + D, D, h->sq8c_mins, h->sq8c_scales);
    }
    const uint64_t nn = h->n - r0;
    const uint32_t qgrid = (uint32_t)std::min<uint64_t>((nn + 3) / 4, (uint64_t)h->num_cu * 16);
    if (is_f16(h)) hipLaunchKernelGGL(k_sq8_quantize<_Float16>, dim3(qgrid), dim3(256), 0, cur(h).stream,
                       (const _Float16*)h->rows_h + r0 * h->ld16, h->ld16, D, nn, h->sq8c_mins, h->sq8c_scales, h->sq8c + r0 * h->ld8, h->ld8, (int*)nullptr, (int*)nullptr, h->sq8c_stats,
                       (const float*)nullptr, 0u, h->vrinv + r0);
    else hipLaunchKernelGGL(k_sq8_quantize<float>, dim3(qgrid), dim3(256), 0, cur(h).stream,
                       h->rows + r0 * h->ld, h->ld, D, nn, h->sq8c_mins, h->sq8c_scales, h->sq8c + r0 * h->ld8, h->ld8, (int*)nullptr, (int*)nullptr, h->sq8c_stats,
                       (const float*)nullptr, 0u, h->vrinv + r0);
    LY_HIP(hipGetLastError());
    uint32_t qst[4] = {0, 0, 0, 0};
    LY_HIP(hipMemcpyAsync(qst, h->sq8c_stats, 16, hipMemcpyDeviceToHost, cur(h).stream));
    LY_HIP(hipStreamSynchronize(cur(h).stream));  // `init` is a temporary
    h->sq8c_a1 = qst[0];
    h->sq8c_a2sq = qst[2];
    memcpy(&h->sq8c_eps2, &qst[3], 4);
    h->sq8c_finite = qst[1] == 0;
    h->n_sq8c = h->n;
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_sq8_params(lynse_hip_flat* h, float* mins, float* scales) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    if (h->packed_only || h->n == 0) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "no f32 rows");
    LY_TRY(finalize_locked(h));
    LY_TRY(ensure_sq8_locked(h));
    if (mins) LY_HIP(hipMemcpy(mins, h->sq8_mins, (size_t)h->dim * 4, hipMemcpyDeviceToHost));
    if (scales) LY_HIP(hipMemcpy(scales, h->sq8_scales, (size_t)h->dim * 4, hipMemcpyDeviceToHost));
    return LYNSE_OK;
}

// sq8_two_pass_search (flat_mmap.rs:5868-5926) for one chunk of <= 256 queries (queries in ws.Qf): pass 1 = exact integer
// scores of the u8 codes on the i8 MFMA, exact top-n_cand in (score, row) order with the staged strict thresholds of the
// packed-binary path; pass 2 = exact f32 rescoring of the n_cand rows, (distance, row) order, top k.
static int run_chunk_sq8(lynse_hip_flat* h, uint32_t nq, uint32_t k, uint32_t out_k, uint32_t n_cand, int metric, int level, hipStream_t st,
                         bool pass1_only = false) {
    // pass1_only: stop after pass 1 — the outputs are the top-k rows by the integer score of the codes, in (score, row) order,
    // with that score as the distance (the large-n_cand path merges such lists over row ranges; tests look at the ranking)
    Workspace& w = cur(h).ws;
    const bool ip = metric == M_IP;
    const int m1 = ip ? M_IP : M_L2;  // cosine ranks the codes by squared L2 too (:5887-5890)
    const uint32_t nslab = (h->dim + 127) / 128, qpad = SCAN_BQ_LARGE;
    static std::atomic<bool> attr = false;
    if (!attr) {
        LY_TRY(set_max_lds(k_select<SEL_NT>, SEL_LDS_MAX));
        LY_TRY(set_max_lds(k_final<SEL_NT>, SEL_LDS_MAX));
        attr = true;
    }
    LY_HIP(hipMemsetAsync(w.Q16, 0, (size_t)nslab * qpad * 128, st));
    hipLaunchKernelGGL(k_sq8_prep_queries, dim3(nq), dim3(256), 0, st, w.Qf, h->dim, qpad, nslab, h->sq8_mins, h->sq8_scales,
                       reinterpret_cast<int8_t*>(w.Q16), w.qn2, w.thr, w.count, w.overflow, ip ? 1 : 0);
    LY_HIP(hipGetLastError());
    const std::vector<Stage> plan = make_plan(h, n_cand, level >= 2 ? 2 : 1, 0);  // contiguous stages: rows in ascending order
    for (size_t si = 0; si < plan.size(); ++si) {
        const Stage s = plan[si];
        ScanArgs a{};
        a.V16 = reinterpret_cast<const _Float16*>(h->sq8); a.ld16 = h->ld8; a.D = h->dim; a.row0 = s.r0; a.row1 = s.r1;
        a.Q16 = w.Q16; a.qpad = qpad; a.nq = nq; a.nslab = nslab; a.ntiles = (s.r1 - s.r0 + 255) / 256;
        a.qinv = w.qinv; a.qn2 = w.qn2; a.qrinv = w.qrinv; a.thr = w.thr;
        a.vn2 = reinterpret_cast<const float*>(ip ? h->sq8_sum : h->sq8_sum2); a.vrinv = a.vn2;
        a.sv = 1.0f; a.cand = w.cand; a.count = w.count; a.cap = w.cap; a.emit_all = si == 0 ? 1 : 0;
        const uint32_t grid = std::min<uint32_t>(a.ntiles, (uint32_t)h->num_cu);
        a.candB = w.candB; a.segcnt = w.segcnt;
        if (!a.emit_all) seg_geometry(grid, 2, &a.nseg, &a.seg);
        constexpr size_t lds = (size_t)(2 * 256 + 2 * 256) * 128 + 3 * 1024;
        static std::atomic<bool> a2[2] = {false, false};
        if (ip) {
            auto kern = k_scan_h16<4, 2, 2, 4, M_IP, 2, 2, 2, false, true, 0, false, 1>;
            if (!a2[0]) { LY_TRY(set_max_lds(kern, lds)); a2[0] = true; }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        } else {
            auto kern = k_scan_h16<4, 2, 2, 4, M_L2, 2, 2, 2, false, true, 0, false, 1>;
            if (!a2[1]) { LY_TRY(set_max_lds(kern, lds)); a2[1] = true; }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
        }
        LY_HIP(hipGetLastError());
        SelectArgs sa{};
        sa.cand = w.cand; sa.count = w.count; sa.overflow = w.overflow; sa.thr = w.thr; sa.marg2 = w.marg2;
        sa.k = n_cand; sa.cap = w.cap; sa.keep_max = w.cap / 2; sa.metric = m1; sa.ip_form = LYNSE_IPFORM_SINGLE;
        sa.exact = 1; sa.emit_all_n = si == 0 ? (int)(s.r1 - s.r0) : -1;
        sa.Qf = w.Qf; sa.V = h->rows; sa.ld = h->ld; sa.D = h->dim;
        sa.candB = w.candB; sa.segcnt = w.segcnt; sa.seg = a.seg; sa.nseg = a.seg ? a.nseg : 0;
        sa.lds_bytes = sel_lds_bytes(w.cap, nq);
        hipLaunchKernelGGL(k_select<SEL_NT>, dim3(nq), dim3(SEL_NT), sa.lds_bytes, st, sa);
        LY_HIP(hipGetLastError());
    }
    FinalArgs fa{};
    fa.cand = w.cand; fa.count = w.count; fa.k = k; fa.out_k = out_k; fa.cap = w.cap; fa.metric = metric;
    fa.ip_form = LYNSE_IPFORM_SINGLE;  // pass 2 uses simd::inner_product_f32 & co (:5899-5906)
    fa.exact = 0; fa.Qf = w.Qf; fa.V = h->rows; fa.ld = h->ld; fa.D = h->dim;
    fa.row_stride = h->row_stride; fa.row_offset = h->row_offset;
    fa.out_rows = w.out_rows; fa.out_dists = w.out_dists; fa.out_counts = w.out_counts; fa.pool_total = nullptr;
    const char* dbg = getenv("LYNSE_HIP_SQ8_PASS1");  // tests: return the pass-1 ranking (integer scores) instead of rescoring
    if (pass1_only || (dbg && atoi(dbg))) {
        fa.metric = m1;
    } else {
        hipLaunchKernelGGL(k_rescore_pool<256>, dim3(nq, nq >= 64 ? 4 : 32), dim3(256), 0, st, fa);
        LY_HIP(hipGetLastError());
    }
    fa.exact = 1;
    fa.lds_bytes = sel_lds_bytes(w.cap, nq);
    hipLaunchKernelGGL(k_final<SEL_NT>, dim3(nq), dim3(SEL_NT), fa.lds_bytes, st, fa);
    LY_HIP(hipGetLastError());
    return LYNSE_OK;
}

static int search_sq8_large(lynse_hip_flat* h, std::unique_lock<std::shared_mutex>& lk, const float* queries, uint64_t nq, uint32_t k, uint32_t kk,
                            uint32_t n_cand, int metric, uint64_t* out_rows, float* out_dists, uint32_t* out_counts);

extern "C" int lynse_hip_flat_search_sq8_f32(lynse_hip_flat* h, const float* queries, uint64_t nq, uint32_t k, int metric,
                                             uint64_t* out_rows, float* out_dists, uint32_t* out_counts) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    if (!metric_valid(metric)) return set_error(LYNSE_ERR_UNKNOWN_METRIC, "Unknown metric id");
    // use_sq8 only changes ip / l2 / cosine (flat_mmap.rs:891-896); every other metric takes the ordinary path
    if (metric > M_COS) return lynse_hip_flat_search_f32(h, queries, nq, k, metric, out_rows, out_dists, out_counts);
    if (nq == 0) return LYNSE_OK;
    if (!queries || !out_counts || (k && (!out_rows || !out_dists))) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    if (h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "store holds packed rows; float metrics unavailable");
    if (h->dtype != LYNSE_DTYPE_F32) return set_error(LYNSE_ERR_UNSUPPORTED, "SQ8 mode on an F16 shard is not supported");
    if (h->n == 0 || k == 0) { memset(out_counts, 0, nq * 4); return LYNSE_OK; }
    const uint32_t kk = (uint32_t)std::min<uint64_t>(k, h->n);
    uint64_t n_cand = metric == M_COS ? std::max<uint64_t>((uint64_t)kk * 100, 500) : std::max<uint64_t>((uint64_t)kk * 20, 200);  // :5883-5893
    n_cand = std::min<uint64_t>(n_cand, h->n);
    LY_TRY(finalize_locked(h));
    LY_TRY(ensure_workspace(h, k));
    LY_TRY(ensure_sq8_locked(h));
    if (h->n > h->cap && n_cand > h->cap / 4)  // more candidates than one pass holds: row ranges + host merge (search_sq8_large)
        return search_sq8_large(h, lk, queries, nq, k, kk, (uint32_t)n_cand, metric, out_rows, out_dists, out_counts);
    Workspace& w = cur(h).ws;
    hipStream_t st = cur(h).stream;
    for (uint64_t q0 = 0; q0 < nq; q0 += QCHUNK) {
        const uint32_t nqc = (uint32_t)std::min<uint64_t>(QCHUNK, nq - q0);
        LY_HIP(hipMemcpyAsync(w.Qf, queries + q0 * h->dim, (size_t)nqc * h->dim * 4, hipMemcpyHostToDevice, st));
        for (int level = 1; level < 3; ++level) {
            LY_TRY(run_chunk_sq8(h, nqc, kk, k, (uint32_t)n_cand, metric, level, st));
            LY_HIP(hipMemcpyAsync(out_rows + q0 * k, w.out_rows, (size_t)nqc * k * 8, hipMemcpyDeviceToHost, st));  // speculative, as in search_impl
            LY_HIP(hipMemcpyAsync(out_dists + q0 * k, w.out_dists, (size_t)nqc * k * 4, hipMemcpyDeviceToHost, st));
            LY_HIP(hipMemcpyAsync(w.h_hdr, w.out_counts, 2 * QCHUNK * 4, hipMemcpyDeviceToHost, st));  // counts + overflow flags
            LY_HIP(hipStreamSynchronize(st));
            uint32_t nov = 0;
            for (uint32_t i = 0; i < nqc; ++i) nov += w.h_hdr[QCHUNK + i] ? 1 : 0;
            if (nov == 0) break;
            if (level == 2) return set_error(LYNSE_ERR_INTERNAL, "candidate overflow on the exhaustive plan");
        }
        memcpy(out_counts + q0, w.h_hdr, nqc * 4);
    }
    return LYNSE_OK;
}

// certified int8 coarse pass: FLAT-IP batches of 33..256 queries over an f32 shard (auto: shards of >= 64K rows;
// LYNSE_HIP_COARSE=i8 / f16 forces / disables it); strikes: overflows so far (3 turn it off), -1 = data not finite
static int coarse_env() {
    static const int v = []() { const char* e = getenv("LYNSE_HIP_COARSE"); return !e ? 0 : (!strcmp(e, "i8") ? 2 : (!strcmp(e, "f16") ? 1 : 0)); }();
    return v;
}
// per-metric state of the certified int8 pass: IP reads the SQ8 codes of the rows, squared L2 the codes of the augmented rows
static std::atomic<int>& i8c_strike_counter(lynse_hip_flat* h, int metric) {
    return metric == M_L2 ? h->i8c_strikes_l2 : (metric == M_COS ? h->i8c_strikes_cos : h->i8c_strikes);
}
// Squared L2 has two int8 forms: unfiltered FLAT batches over whole 128-column slabs run on the PLAIN codes with the exact f32
// row norms in a float epilogue (k_scan_h16<.., I8Q = 4>, every tiling: 1 B per element, no second code set, IP-sized margins);
// everything else (ragged widths, masked scans, IVF) on the codes of the AUGMENTED rows through the IP kernels.  nqc = 0: "not a FLAT
// batch" (IVF): the augmented form.  LYNSE_HIP_L2_PLAIN=0: always the augmented form.
static bool l2_plain(const lynse_hip_flat* h, uint64_t nqc, bool masked) {
    static const int on = []() { const char* e = getenv("LYNSE_HIP_L2_PLAIN"); return e ? atoi(e) : 1; }();
    // (1M x 128, k = 100: 0.244 ms on the f16 shadow, 0.277 on the codes with the <4,2,2,4> tiling and 0.291 with the query-stationary
    // L2 tiling (k_scan_qs<1,4,1,6,.., MET = 1>, built and measured in round 4: at k = 100 the first threshold lets 0.8 % of all
    // (row, query) pairs through and the scan is bound by the emission of ~8000 keys per query, not by the tiling) — from 256
    // dimensions on)
    return on && !masked && nqc >= 1 && nqc <= QCHUNK && h->ld8 % 128 == 0 && h->dim >= 256;
}
static bool i8c_codes_ready(const lynse_hip_flat* h, int metric, uint64_t nqc = 0, bool masked = false) {
    if (metric == M_L2 && !l2_plain(h, nqc, masked)) return h->sq8a && h->n_sq8a == h->n;
    if (metric == M_COS) return h->sq8c && h->n_sq8c == h->n;
    return h->sq8 && h->n_sq8 == h->n;
}
static bool i8c_codes_finite(const lynse_hip_flat* h, int metric, uint64_t nqc = 0, bool masked = false) {
    return (metric == M_L2 && !l2_plain(h, nqc, masked)) ? h->sq8a_finite : (metric == M_COS ? h->sq8c_finite : h->sq8_finite);
}
static int ensure_i8c_codes_locked(lynse_hip_flat* h, int metric, uint64_t nqc = 0, bool masked = false) {
    return (metric == M_L2 && !l2_plain(h, nqc, masked)) ? ensure_sq8a_locked(h) : (metric == M_COS ? ensure_sq8c_locked(h) : ensure_sq8_locked(h));
}
static void i8c_add_strike(lynse_hip_flat* h, int metric) {
    std::atomic<int>& c = i8c_strike_counter(h, metric);
    if (c.load() >= 0) c.fetch_add(1);
}
static bool i8c_eligible(const lynse_hip_flat* h, int metric, bool filtered, uint64_t nqc, bool view = false, bool masked = false, bool ivf = false) {
    // filtered: a subset filter on the gathered-rows strategy (never int8); masked: a subset filter as a row bitmask — the
    // masked int8 scan (whole 128-column slabs only)
    // view: a row-range view of search_large_k (h->n is the range): the SQ8 codes belong to the whole shard
    const int strikes = (metric == M_L2 ? h->i8c_strikes_l2 : (metric == M_COS ? h->i8c_strikes_cos : h->i8c_strikes)).load();
    static const int l2_off = []() { const char* e = getenv("LYNSE_HIP_COARSE_L2"); return e && !strcmp(e, "f16") ? 1 : 0; }();
    // squared L2 streams round_up(dim + 1, 128) bytes of augmented codes per row against 2 round_up(dim, 8) of the f16 shadow and
    // rescoring pools ~10x wider: it pays from ~256 dimensions on (MI355X: 10M x 768 x 256 3.97 -> 2.81 ms; 1M x 128, k = 100
    // 0.24 -> 0.34 ms: stays on the f16 pass)
    const bool l2_ok = metric == M_L2 && !l2_off &&
                       (l2_plain(h, nqc, masked) || (h->dim >= 256 && (uint64_t)h->ld8a * 4 <= (uint64_t)h->ld16 * 2 * 3));
    // cosine streams the codes of the unit rows (1 B per element, like IP); tiny-norm rows make the f16 pass go exhaustive and
    // are left to it
    const bool cos_ok = metric == M_COS && !l2_off && h->dim >= 256 && !h->cos_degenerate;
    if (masked && metric != M_L2 && h->ld8 % 128 != 0) return false;
    // batches of <= 32 queries (the 128 x 32 tiling, HBM-bound: half the bytes = half the time) from 256K rows on — below, the scan
    // is a few tens of microseconds either way and the exact few-query kernel often answers alone (LYNSE_HIP_COARSE_SMALLQ=0: off)
    static const int smallq = []() { const char* e = getenv("LYNSE_HIP_COARSE_SMALLQ"); return e ? atoi(e) : 1; }();
    const bool nq_ok = nqc > SCAN_BQ_SMALL || (smallq && nqc >= 1 && (coarse_env() == 2 || h->n >= 262144));   // (masked small batches too)
    // (F16 shards too: the codes are built from the exactly decoded halves, the exact rescoring uses the f16 kernels' sequential
    // sums — any summation order is inside the bound's rounding term)
    // FLAT-IP over rows of one or two whole 64-element slabs (64 / 128 columns, and the padded 48..63 / 96..127): batches of 33..256
    // queries run the float pass on the f16 shadow — k_scan_qh (scan_qh.h) is faster there than the 256 x 256 int8 tile over 128-byte code
    // rows and its margin ~2.5x tighter (round 5, 1M rows, 256 queries: 64 columns k = 10 / 100 0.222 / 0.318 -> 0.137 / 0.166 ms, 100 columns
    // 0.234 / 0.344 -> 0.169 / 0.188, 128 columns 0.163 / 0.273 -> 0.175 / 0.193; 100 queries 0.157 -> 0.127).  LYNSE_HIP_IP_LOWD=i8: the int8 pass
    // (read per call: tests; ivf: the list scans of an IVF store run the work-list tilings of k_scan_h16 either way — they keep the codes)
    const char* lowd_env = getenv("LYNSE_HIP_IP_LOWD");
    const bool lowd_i8 = lowd_env && !strcmp(lowd_env, "i8");
    // (gated on the very shape predicate qh_scan_ok applies — a batch k_scan_qh would not take must not lose the int8 pass to the slower
    // 256 x 256 f16 tile; `view` = the gathered rows of a subset filter, a widened handle never comes here with <= 256 queries)
    if (metric == M_IP && !lowd_i8 && !ivf && !masked && !filtered && !view && coarse_env() != 2 && h->qchunk <= QCHUNK && qh_shape(h->ld16, nqc))
        return false;
    return (metric == M_IP || l2_ok || cos_ok) && !filtered && !view && nq_ok &&
           scan_variant() == 3 && coarse_env() != 1 && strikes >= 0 && strikes < 3 && (coarse_env() == 2 || h->n >= 65536);
}

// Shared driver: q_src is nq x dim f32 (float metrics / f32 binary queries) or nq x words u64
// (pre-packed).  Sources and destinations are host or device pointers according to `on_device`.
static int search_large_k(lynse_hip_flat* h, const void* q_src, bool packed_queries, uint64_t nq, uint32_t k, int metric,
                          uint64_t* out_rows, float* out_dists, uint32_t* out_counts, bool on_device, hipStream_t user_stream,
                          const uint64_t* subset = nullptr, uint64_t n_subset = 0, bool filtered = false,
                          const uint64_t* bitset_words = nullptr, uint64_t n_words = 0);

// Adds one finished search to the handle's profile: the pipeline time between its two events and the HIP-event duration of
// every scan launch (events of the search context `cx`).
static int prof_accumulate(lynse_hip_flat* h, lynse_hip_flat::Ctx& cx, hipEvent_t ev_begin, hipEvent_t ev_end,
                           const std::vector<std::pair<size_t, uint64_t>>& scan_events, bool binary, uint64_t fallback_queries) {
    LY_HIP(hipEventSynchronize(ev_end));
    float ms = 0.f;
    LY_HIP(hipEventElapsedTime(&ms, ev_begin, ev_end));
    std::lock_guard<std::mutex> plk(h->prof_mu);
    h->prof.total_us += (double)ms * 1000.0;
    const uint64_t row_bytes = binary ? (uint64_t)h->words * 8 : (uint64_t)h->dim * 4;
    for (auto& se : scan_events) {
        float sms = 0.f;
        LY_HIP(hipEventElapsedTime(&sms, cx.ev_pool[se.first], cx.ev_pool[se.first + 1]));
        h->prof.scan_us += (double)sms * 1000.0;
        h->prof.scan_launches += 1;
        h->prof.scan_rows += se.second;
        h->prof.scan_bytes += se.second * row_bytes;
    }
    h->prof.searches += 1;
    h->prof.fallback_queries += fallback_queries;
    return LYNSE_OK;
}

static int search_impl_once(lynse_hip_flat* h, const void* q_src, bool packed_queries, uint64_t nq, uint32_t k,
                            int metric, uint64_t* out_rows, float* out_dists, uint32_t* out_counts,
                            bool on_device, hipStream_t user_stream, const uint64_t* subset, uint64_t n_subset,
                            bool filtered, const uint64_t* bitset_words, uint64_t n_words, bool caller_holds_exclusive, bool force_shadow);
static int search_impl(lynse_hip_flat* h, const void* q_src, bool packed_queries, uint64_t nq, uint32_t k,
                       int metric, uint64_t* out_rows, float* out_dists, uint32_t* out_counts,
                       bool on_device, hipStream_t user_stream, const uint64_t* subset = nullptr, uint64_t n_subset = 0,
                       bool filtered = false, const uint64_t* bitset_words = nullptr, uint64_t n_words = 0,
                       bool caller_holds_exclusive = false) {
    int rc = search_impl_once(h, q_src, packed_queries, nq, k, metric, out_rows, out_dists, out_counts, on_device, user_stream, subset, n_subset,
                              filtered, bitset_words, n_words, caller_holds_exclusive, false);
    // (a batch that started on the int8 pass under the shared lock and now needs the f16 shadow — an overflow, or a last chunk of a
    // different shape: once more from the top, under the writer lock, with the shadow built first)
    if (rc == LY_RESTART_EXCLUSIVE)
        rc = search_impl_once(h, q_src, packed_queries, nq, k, metric, out_rows, out_dists, out_counts, on_device, user_stream, subset, n_subset,
                              filtered, bitset_words, n_words, caller_holds_exclusive, true);
    return rc;
}
static int search_impl_once(lynse_hip_flat* h, const void* q_src, bool packed_queries, uint64_t nq, uint32_t k,
                            int metric, uint64_t* out_rows, float* out_dists, uint32_t* out_counts,
                            bool on_device, hipStream_t user_stream, const uint64_t* subset, uint64_t n_subset,
                            bool filtered, const uint64_t* bitset_words, uint64_t n_words, bool caller_holds_exclusive, bool force_shadow) {
    // filtered: the subset is either a list of row ids (`subset`, n_subset) or BitSet words (`bitset_words`; n_subset =
    // number of set bits below len, counted by the caller)
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    if (!metric_valid(metric)) return set_error(LYNSE_ERR_UNKNOWN_METRIC, "Unknown metric id");
    if (nq == 0) return LYNSE_OK;
    if (!q_src || !out_counts || (k && (!out_rows || !out_dists))) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    LY_TRY(use_device(h));
    profile_begin_search(h);
    const bool binary = metric >= M_HAMMING;
    // Locking: an unfiltered search over a shard whose derived data (row statistics, f16 shadow, packed words, SQ8 codes of
    // the int8 coarse pass) is up to date runs under the SHARED lock on a search context of its own — subset-filtered searches on
    // the masked-scan strategy included; anything that has to build or borrows per-handle state (lazy builds, the gathered-rows
    // strategy of a subset filter) runs EXCLUSIVE on context 0.
    std::shared_lock<std::shared_mutex> rlk(h->rw, std::defer_lock);
    std::unique_lock<std::shared_mutex> xlk(h->rw, std::defer_lock);
    CtxLease lease;
    auto derived_ready = [&]() {
        if (binary) {
            if (!(h->packed_only || (h->packed != nullptr && h->n_packed == h->n))) return false;
            return !bin_mfma_eligible(h, metric, filtered, std::min<uint64_t>(nq, QCHUNK)) || (h->bpm && h->n_bpm == h->n);   // (the +-1 copy is a lazy build too)
        }
        if (h->packed_only) return true;  // (rejected below)
        if (h->n_stats != h->n) return false;
        if (force_shadow && !shadow_ready(h)) return false;
        // (filtered: what matters is the MASKED int8 scan — a gathered-rows search goes exclusive anyway)
        // a batch that takes the certified int8 pass needs its codes; every other float batch the f16 shadow (a lazy copy too).  The
        // chunks of a large batch can differ (the last one may be small): the chunk loop checks again and restarts under the writer lock
        if (i8c_eligible(h, metric, false, std::min<uint64_t>(nq, QCHUNK), caller_holds_exclusive, filtered)) return i8c_codes_ready(h, metric, std::min<uint64_t>(nq, QCHUNK), filtered);
        return shadow_ready(h) || small_path_ok(h, std::min<uint64_t>(nq, QCHUNK), (uint32_t)std::min<uint64_t>(k, h->n), metric, filtered);
    };
    // k beyond the candidate capacity of one pass (k > cap / 4 over more than cap rows; the reference accepts any k, and its
    // server caps at MAX_TOP_K = 10,000, src/server/mod.rs:46): row ranges of `cap` rows, each answered exactly, merged.
    // (n and cap are read under the lock the search then runs under: an append between the decision and the search
    // cannot turn an ordinary search into an unsupported one.)
    auto needs_large_k = [&]() {
        if (h->n <= h->cap) return false;
        const uint64_t kk0 = std::min<uint64_t>(k, h->n);
        return (filtered ? std::min<uint64_t>(kk0, n_subset) : kk0) > h->cap / 4;
    };
    auto go_large_k = [&]() {
        return filtered ? search_large_k(h, q_src, packed_queries, nq, k, metric, out_rows, out_dists, out_counts, on_device, user_stream,
                                         subset, n_subset, true, bitset_words, n_words)
                        : search_large_k(h, q_src, packed_queries, nq, k, metric, out_rows, out_dists, out_counts, on_device, user_stream);
    };
    // filtered search, strategy (flat_mmap.rs:549-556 switches at a fixed 50,000 ids): gather the listed shadow rows and scan
    // only those when that is cheaper than a masked scan of the whole shard — bytes over measured rates on MI355X
    bool strategy_direct = false, strategy_known = false;
    auto choose_direct = [&]() -> bool {
        if (!filtered || binary) return false;
        const double rate = nq <= SCAN_BQ_SMALL ? 6.0e12 : 3.4e12;
        const double chunks = (double)((nq + QCHUNK - 1) / QCHUNK);
        const double row_b = (double)h->dim * 2.0;
        const double c_direct = (double)n_subset * row_b * 2.0 / 2.5e12 + (double)n_subset * row_b / rate * chunks + 120e-6;
        // (the masked scan of 33..256 queries streams the SQ8 codes when the certified int8 pass is available: 1 B per element
        // at ~3.6 TB/s, measured 10M x 768 x 256 with a 50 % subset: 2.2 ms)
        const bool mask_i8 = i8c_eligible(h, metric, false, std::min<uint64_t>(nq, QCHUNK), caller_holds_exclusive, true);
        const double c_scan = (mask_i8 ? (double)h->n * (double)h->dim / (nq <= SCAN_BQ_SMALL ? 6.5e12 : 3.6e12) : (double)h->n * row_b / rate) * chunks + 80e-6;
        const char* fe = getenv("LYNSE_HIP_FILTER_STRATEGY");  // tests: 1 = gathered rows, 2 = masked scan (read per call)
        const int force = fe ? atoi(fe) : 0;
        return (double)n_subset * row_b <= 8e9 && (force == 1 || (force == 0 && c_direct < c_scan));
    };
    if (!caller_holds_exclusive) {
        rlk.lock();
        if (needs_large_k()) { rlk.unlock(); return go_large_k(); }
        // (a MASKED filtered search only needs scratch of its own context: shared lock; the gathered-rows strategy points the
        // handle's scan-side fields at the compact copy for the duration of the call: exclusive)
        // (the strategy is decided ONCE, here: its inputs — the strike counters concurrent searches bump, an environment variable —
        // can change between two evaluations, and "masked" under the shared lock followed by "gathered" at the use site would run
        // the gathered-rows path, which repoints handle-wide scan fields, beside concurrent readers)
        strategy_direct = filtered && n_subset != 0 && choose_direct();
        strategy_known = true;
        if (!user_stream && derived_ready() && !(filtered && (n_subset == 0 || strategy_direct))) {
            LY_TRY(lease.acquire(h));
        } else {
            rlk.unlock();
            LY_TRY(writer_lock(h, xlk));  // (builds, subset filters, a caller's stream: exclusive, and no tickets outstanding)
            if (needs_large_k()) { xlk.unlock(); return go_large_k(); }
        }
    }
    hipStream_t st = user_stream ? user_stream : cur(h).stream;
    const hipMemcpyKind in_kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind out_kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;

    if (packed_queries && !binary) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "packed queries need a binary metric");
    if (!binary && h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "store holds packed rows; float metrics unavailable");

    if (filtered && !binary && scan_variant() != 3)
        return set_error(LYNSE_ERR_UNSUPPORTED, "filtered search needs the default scan kernel (LYNSE_HIP_SCAN_VARIANT=3)");
    if (filtered && n_subset && !subset && !bitset_words) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "subset is NULL");
    // empty store, k == 0 or an empty subset -> empty results, not an error (flat_mmap.rs:832-835, :498-500)
    if (h->n == 0 || k == 0 || (filtered && n_subset == 0)) {
        if (on_device) LY_HIP(hipMemsetAsync(out_counts, 0, nq * 4, st));
        else memset(out_counts, 0, nq * 4);
        if (on_device) LY_HIP(hipStreamSynchronize(st));
        return LYNSE_OK;
    }
    uint32_t kk = (uint32_t)std::min<uint64_t>(k, h->n);  // k.min(n), flat_mmap.rs:836
    if (filtered) kk = (uint32_t)std::min<uint64_t>(kk, n_subset);  // k.min(subset.len()), flat_mmap.rs:501
    if (h->n > h->cap && kk > h->cap / 4)
        return set_error(LYNSE_ERR_UNSUPPORTED, "k > cap/4 over more than cap rows is not supported together with a subset filter");
    // (rows wider than 4096 bits run on k_scan_binary_wide: no LDS tile, no register-resident words)

    if (binary) {
        LY_TRY(ensure_packed_locked(h));
        if (xlk.owns_lock() && bin_mfma_eligible(h, metric, filtered, std::min<uint64_t>(nq, QCHUNK))) LY_TRY(ensure_bpm_locked(h));
    } else {
        LY_TRY(finalize_locked(h));
        if (force_shadow && (xlk.owns_lock() || caller_holds_exclusive)) LY_TRY(ensure_shadow_locked(h));
    }
    LY_TRY(ensure_workspace(h, k));
    Workspace& w = cur(h).ws;
    hipStream_t st0 = user_stream ? user_stream : cur(h).stream;
    // the f16 shadow, for whoever is about to scan it: built here under the writer lock, or the whole search starts over under it
    auto need_shadow = [&]() -> int {
        if (binary || shadow_ready(h)) return LYNSE_OK;
        if (xlk.owns_lock() || caller_holds_exclusive) return ensure_shadow_locked(h);
        return LY_RESTART_EXCLUSIVE;
    };
    bool direct = false;
    std::vector<uint64_t> sorted_subset;
    auto& fc = cur(h);   // (this search's context: 0 under the exclusive lock, the leased one under the shared lock)
    if (filtered) {
        direct = strategy_known ? strategy_direct : choose_direct();   // (caller_holds_exclusive: evaluated here, under that lock)
        if (direct && !xlk.owns_lock() && !caller_holds_exclusive) return set_error(LYNSE_ERR_INTERNAL, "gathered-rows filter strategy without the exclusive lock");
        if (n_subset > fc.subset_cap && (direct || !bitset_words)) {
            if (fc.d_subset) (void)hipFree(fc.d_subset);
            fc.d_subset = nullptr;
            LY_HIP(hipMalloc(&fc.d_subset, n_subset * 8));
            fc.subset_cap = n_subset;
        }
        bool ids_on_device = false;
        if (bitset_words && direct) {  // BitSet::to_vec on the device: upload the words (they double as scratch in d_mask), expand
            const uint64_t nw = std::min<uint64_t>(n_words, (h->n + 63) / 64);
            const uint64_t words32 = (std::max<uint64_t>(h->n, h->capacity) + 511) / 32 + 8;
            const uint32_t nblocks = (uint32_t)((nw + 255) / 256);
            if (words32 + nblocks + 16 > fc.mask_words) {
                if (fc.d_mask) (void)hipFree(fc.d_mask);
                fc.d_mask = nullptr;
                LY_HIP(hipMalloc(&fc.d_mask, (words32 + nblocks + 16) * 4));
                fc.mask_words = words32 + nblocks + 16;
            }
            if (n_subset > fc.subset_cap) {
                if (fc.d_subset) (void)hipFree(fc.d_subset);
                fc.d_subset = nullptr;
                LY_HIP(hipMalloc(&fc.d_subset, n_subset * 8));
                fc.subset_cap = n_subset;
            }
            uint32_t* d_counts = fc.d_mask + words32;
            LY_HIP(hipMemcpyAsync(fc.d_mask, bitset_words, nw * 8, hipMemcpyHostToDevice, st0));
            const uint64_t* d_words = reinterpret_cast<const uint64_t*>(fc.d_mask);
            hipLaunchKernelGGL(k_bits_count, dim3(nblocks), dim3(256), 0, st0, d_words, nw, h->n, d_counts);
            hipLaunchKernelGGL(k_bits_offsets, dim3(1), dim3(1024), 0, st0, d_counts, nblocks);
            hipLaunchKernelGGL(k_bits_expand, dim3(nblocks), dim3(256), 0, st0, d_words, nw, h->n, d_counts, fc.d_subset);
            LY_HIP(hipGetLastError());
            ids_on_device = true;
        }
        const uint64_t* src = subset;
        if (direct && n_subset && !ids_on_device) {  // the direct path needs a set: sort + unique unless already strictly ascending (a BitSet's to_vec is)
            bool ascending = subset[n_subset - 1] < h->n;  // and every id valid
            for (uint64_t i = 1; i < n_subset && ascending; ++i) ascending = subset[i - 1] < subset[i];
            if (!ascending) {
                sorted_subset.assign(subset, subset + n_subset);
                std::sort(sorted_subset.begin(), sorted_subset.end());
                sorted_subset.erase(std::unique(sorted_subset.begin(), sorted_subset.end()), sorted_subset.end());
                while (!sorted_subset.empty() && sorted_subset.back() >= h->n) sorted_subset.pop_back();  // rows >= n are skipped (:5243)
                src = sorted_subset.data();
                n_subset = sorted_subset.size();
            }
        }
        if (src && n_subset && !ids_on_device) LY_HIP(hipMemcpyAsync(fc.d_subset, src, n_subset * 8, hipMemcpyHostToDevice, st0));
        if (!direct) {  // subset ids -> row bitmask on the device (the reference's bitset, flat_mmap.rs:672-679)
            const uint64_t words = (std::max<uint64_t>(h->n, h->capacity) + 511) / 32 + 8;
            if (words > fc.mask_words) {
                if (fc.d_mask) (void)hipFree(fc.d_mask);
                fc.d_mask = nullptr;
                LY_HIP(hipMalloc(&fc.d_mask, words * 4));
                fc.mask_words = words;
            }
            LY_HIP(hipMemsetAsync(fc.d_mask, 0, fc.mask_words * 4, st0));
            if (bitset_words) {  // the caller's BitSet words ARE the mask (u64 LE = two u32 words); bits >= len never match a row
                const uint64_t bytes = std::min<uint64_t>(n_words * 8, (h->n + 63) / 64 * 8);
                LY_HIP(hipMemcpyAsync(fc.d_mask, bitset_words, bytes, hipMemcpyHostToDevice, st0));
            } else {
                const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_subset + 255) / 256, (uint64_t)h->num_cu * 8);
                hipLaunchKernelGGL(k_mask_build, dim3(std::max<uint32_t>(blocks, 1)), dim3(256), 0, st0, fc.d_subset, n_subset, h->n, fc.d_mask);
                LY_HIP(hipGetLastError());
            }
        }
        if (direct && n_subset) {  // compact copy of the listed shadow rows + norms + 32-bit ids
            if (n_subset > h->g_cap) {
                for (void* p : {(void*)h->g_rows16, (void*)h->g_vn2, (void*)h->g_vrinv, (void*)h->g_ids32})
                    if (p) (void)hipFree(p);
                h->g_rows16 = nullptr; h->g_vn2 = h->g_vrinv = nullptr; h->g_ids32 = nullptr; h->g_cap = 0;
                LY_HIP(hipMalloc(&h->g_rows16, (size_t)n_subset * h->ld16 * 2 + 256));
                LY_HIP(hipMalloc(&h->g_vn2, ((size_t)n_subset + 256) * 4));
                LY_HIP(hipMalloc(&h->g_vrinv, ((size_t)n_subset + 256) * 4));
                LY_HIP(hipMalloc(&h->g_ids32, ((size_t)n_subset + 256) * 4));
                LY_HIP(hipMemsetAsync(h->g_vn2, 0, ((size_t)n_subset + 256) * 4, st0));
                LY_HIP(hipMemsetAsync(h->g_vrinv, 0, ((size_t)n_subset + 256) * 4, st0));
                h->g_cap = n_subset;
            }
            const uint64_t pieces = n_subset * (h->ld16 / 8);
            LY_TRY(need_shadow());   // (the gathered-rows strategy runs under the writer lock)
            hipLaunchKernelGGL(k_gather_rows16, dim3((uint32_t)std::min<uint64_t>((pieces + 255) / 256, (uint64_t)h->num_cu * 32)), dim3(256), 0,
                               st0, h->rows16, h->ld16, fc.d_subset, n_subset, h->g_rows16);
            LY_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_gather_norms, dim3((uint32_t)std::min<uint64_t>((n_subset + 255) / 256, (uint64_t)h->num_cu * 8)), dim3(256), 0,
                               st0, h->vn2, h->vrinv, fc.d_subset, n_subset, h->g_vn2, h->g_vrinv, h->g_ids32);
            LY_HIP(hipGetLastError());
        }
        LY_HIP(hipStreamSynchronize(st0));  // `subset` / sorted_subset are caller / stack memory
        if (direct && n_subset == 0) {  // every listed row was out of range
            if (on_device) LY_HIP(hipMemsetAsync(out_counts, 0, nq * 4, st0));
            else memset(out_counts, 0, nq * 4);
            if (on_device) LY_HIP(hipStreamSynchronize(st0));
            return LYNSE_OK;
        }
    }
    const uint32_t* mask = (filtered && !direct) ? fc.d_mask : nullptr;
    // the gathered path scans the compact store through the ordinary pipeline: point the scan-side fields of the handle
    // at it for the duration of this call (the mutex is held); rescoring keeps using the f32 source rows by original id
    struct ViewGuard {
        lynse_hip_flat* h; bool on; _Float16* r16; uint64_t n; float *vn2, *vrinv;
        ~ViewGuard() { if (on) { h->rows16 = r16; h->n = n; h->vn2 = vn2; h->vrinv = vrinv; } }
    } view{h, direct, h->rows16, h->n, h->vn2, h->vrinv};
    if (direct) { h->rows16 = h->g_rows16; h->n = n_subset; h->vn2 = h->g_vn2; h->vrinv = h->g_vrinv; }

    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    size_t ev_used = 0;
    std::vector<std::pair<size_t, uint64_t>> scan_events;
    if (tl_prof) {
        LY_TRY(get_event(h, ev_used++, &ev_begin));
        LY_TRY(get_event(h, ev_used++, &ev_end));
        LY_HIP(hipEventRecord(ev_begin, st));
    }
    uint64_t fallback_queries = 0;
    bool i8c_attempted = false;

    // queries per pipeline pass: QCHUNK; a widened handle (the k-means assignment's centroid store: unfiltered float search of
    // a shard of <= cap rows) takes h->qchunk queries per pass, one k_scan_h16 launch scoring qchunk / 256 chunks (blockIdx.y)
    const bool wide = h->qchunk > QCHUNK && !binary && !filtered && scan_variant() == 3 && h->n <= h->cap;
    const uint32_t QC = wide ? w.qcap : QCHUNK;
    for (uint64_t q0 = 0; q0 < nq; q0 += QC) {
        const uint32_t nqc = (uint32_t)std::min<uint64_t>(QC, nq - q0);
        // stage the chunk's queries in the workspace
        if (binary) {
            if (packed_queries) {
                LY_HIP(hipMemcpyAsync(w.QW, (const uint64_t*)q_src + q0 * h->words, (size_t)nqc * h->words * 8, in_kind, st));
            } else {  // pack_binary_query (flat_mmap.rs:1292-1296)
                LY_HIP(hipMemcpyAsync(w.Qf, (const float*)q_src + q0 * h->dim, (size_t)nqc * h->dim * 4, in_kind, st));
                hipLaunchKernelGGL(k_pack_bits<float>, dim3((nqc + 3) / 4), dim3(256), 0, st, w.Qf, h->dim, h->dim, nqc, w.QW, h->words);
                LY_HIP(hipGetLastError());
            }
        } else if (!on_device) {  // (device queries are read in place: they stay valid for the whole call)
            LY_HIP(hipMemcpyAsync(w.Qf, (const float*)q_src + q0 * h->dim, (size_t)nqc * h->dim * 4, in_kind, st));
        }
        const float* qsrc = (on_device && !binary) ? (const float*)q_src + q0 * h->dim : nullptr;
        // certified int8 coarse pass: FLAT-IP batches of 33..256 queries over an f32 shard with finite values (auto: shards
        // of >= 64K rows; LYNSE_HIP_COARSE=i8 / f16 forces / disables it).  An overflow first retries the f16 coarse pass;
        // three such strikes turn the int8 pass off for this handle (margins too wide for this data).
        bool i8c = i8c_eligible(h, metric, filtered && direct, nqc, caller_holds_exclusive, filtered && !direct);
        if (i8c) {
            const bool msk = filtered && !direct;
            if (xlk.owns_lock()) LY_TRY(ensure_i8c_codes_locked(h, metric, nqc, msk));   // lazy build: exclusive path only
            else if (!i8c_codes_ready(h, metric, nqc, msk)) i8c = false;                    // (shared path: derived_ready() saw them built)
            if (i8c && !i8c_codes_finite(h, metric, nqc, msk)) { i8c_strike_counter(h, metric).store(-1); i8c = false; }
        }
        i8c_attempted = i8c_attempted || i8c;
        if (!binary && !i8c && !small_path_ok(h, nqc, kk, metric, filtered)) LY_TRY(need_shadow());   // this chunk scans the f16 shadow
        if (small_path_ok(h, nqc, kk, metric, filtered)) {
            // fused single-launch search: the last workgroup writes straight into the caller's device buffers, or into
            // pinned host memory (no copy kernels, one synchronisation); it cannot overflow
            const size_t rows_b = (size_t)nqc * k * 8, dists_b = (size_t)nqc * k * 4;
            uint64_t* r_dst = on_device ? out_rows + q0 * k : reinterpret_cast<uint64_t*>(w.h_out);
            float* d_dst = on_device ? out_dists + q0 * k : reinterpret_cast<float*>(w.h_out + rows_b);
            uint32_t* c_dst = on_device ? out_counts + q0 : w.h_hdr;
            LY_TRY(run_small(h, nqc, kk, k, metric, st, &ev_used, &scan_events, r_dst, d_dst, c_dst, w.h_hdr + w.qcap,
                             on_device ? (const float*)q_src + q0 * h->dim : w.Qf));  // device queries are read in place
            LY_HIP(hipStreamSynchronize(st));
            if (!on_device) {
                memcpy(out_rows + q0 * k, w.h_out, rows_b);
                memcpy(out_dists + q0 * k, w.h_out + rows_b, dists_b);
                memcpy(out_counts + q0, w.h_hdr, nqc * 4);
            }
            continue;
        }
        bool allow_sts = true;
        for (int level = 0; level < 3; ++level) {  // sampled plan -> contiguous plan -> exhaustive plan (make_plan)
            bool sampled = false, used_sts = false;
            // k_final writes the results where they belong — the caller's device arrays, or the pinned (device-visible)
            // staging buffer for small host-API results: no copy kernels behind the search; only large host results go
            // through the workspace + two device-to-host copies.  One synchronisation per chunk (counts + overflow flags in
            // one small pinned readback); a retry on the next plan level overwrites the outputs in stream order.
            const size_t rows_b = (size_t)nqc * k * 8, dists_b = (size_t)nqc * k * 4;
            const bool staged = !on_device && rows_b + dists_b <= H_OUT_BYTES;
            uint64_t* r_dst = on_device ? out_rows + q0 * k : (staged ? reinterpret_cast<uint64_t*>(w.h_out) : nullptr);
            float* d_dst = on_device ? out_dists + q0 * k : (staged ? reinterpret_cast<float*>(w.h_out + rows_b) : nullptr);
            LY_TRY(run_chunk(h, nqc, kk, k, metric, level, st, &ev_used, &scan_events, &sampled, mask, direct ? h->g_ids32 : nullptr, i8c,
                             r_dst, d_dst, on_device ? out_counts + q0 : nullptr, nullptr, qsrc, true, allow_sts, &used_sts));
            if (!on_device && !staged) {
                LY_HIP(hipMemcpyAsync(out_rows + q0 * k, w.out_rows, rows_b, out_kind, st));
                LY_HIP(hipMemcpyAsync(out_dists + q0 * k, w.out_dists, dists_b, out_kind, st));
            }
            LY_TRY(stream_wait(st));  // (counts + overflow flags: written into the pinned header by k_final)
            uint32_t nov = 0;
            for (uint32_t i = 0; i < nqc; ++i) nov += w.h_hdr[w.qcap + i] ? 1 : 0;
            if (nov == 0) {
                if (staged) { memcpy(out_rows + q0 * k, w.h_out, rows_b); memcpy(out_dists + q0 * k, w.h_out + rows_b, dists_b); }
                break;
            }
            if (level == 2) return set_error(LYNSE_ERR_INTERNAL, "candidate overflow on the exhaustive plan");
            fallback_queries += nov;
            if (used_sts) {  // the self-tightening scan overflowed (scores rising along every workgroup's chunk): the staged plan of the same coarse pass
                allow_sts = false;
                --level;
                continue;
            }
            if (i8c) {  // same plan level again with the f16 coarse pass
                i8c = false;
                i8c_add_strike(h, metric);
                LY_TRY(need_shadow());
                --level;
                continue;
            }
            if (level == 0 && !sampled) level = 1;  // level 1 would repeat the same contiguous plan
        }
        if (!on_device) memcpy(out_counts + q0, w.h_hdr, nqc * 4);
    }

    if (tl_prof) {
        LY_HIP(hipEventRecord(ev_end, st));
        LY_TRY(prof_accumulate(h, cur(h), ev_begin, ev_end, scan_events, binary, fallback_queries));
        if (i8c_attempted) {  // bit 6: the search STARTED on the certified int8 pass (an overflow may have sent it to the f16 pass)
            std::lock_guard<std::mutex> plk(h->prof_mu);
            h->prof.last_plan |= 64u;
        }
    }
    return LYNSE_OK;
}

// Diagnostics of the CERTIFICATE (tests/test_gpu_certificate.py): the coarse score of every row of a small shard (n <= cap) for up
// to 256 queries, exactly as the scan kernels compute it (one emit-all stage of the real pipeline: MFMA dot products, the kernels'
// float expressions), in the metric's own space (IP: score, L2 / cosine: distance), and the per-query bound E the prep kernel
// certified (the margin the pipeline keeps is 2E).  coarse = 0: the f16 shadow, 1: the certified int8 pass in the form a batch of
// this shape would take; *out_form: bit 0 int8, bit 1 augmented-L2 codes, bit 2 plain-code L2, bit 3 unit-row cosine codes.
extern "C" int lynse_hip_flat_coarse_scores(lynse_hip_flat* h, const float* queries, uint64_t nq, int metric, int coarse,
                                            float* out_scores, float* out_bound, int* out_form) {
    if (!h || !queries || !out_scores || !out_bound) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    if (metric < M_IP || metric > M_COS) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "float metrics only");
    if (nq == 0 || nq > QCHUNK) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "1..256 queries");
    LY_TRY(use_device(h));
    std::unique_lock<std::shared_mutex> xlk;
    LY_TRY(writer_lock(h, xlk));
    if (h->packed_only || h->n == 0 || h->n > h->cap) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "a float shard of 1..cap rows");
    LY_TRY(finalize_locked(h));
    LY_TRY(ensure_workspace(h, 1));
    const bool i8c = coarse != 0;
    if (!i8c) LY_TRY(ensure_shadow_locked(h));
    if (i8c) {
        LY_TRY(ensure_i8c_codes_locked(h, metric, nq));
        if (!i8c_codes_finite(h, metric, nq)) return set_error(LYNSE_ERR_UNSUPPORTED, "the shard holds non-finite values: no int8 pass");
    }
    Workspace& w = cur(h).ws;
    hipStream_t st = cur(h).stream;
    LY_HIP(hipMemcpyAsync(w.Qf, queries, (size_t)nq * h->dim * 4, hipMemcpyHostToDevice, st));
    size_t ev_used = 0;
    std::vector<std::pair<size_t, uint64_t>> scan_events;
    bool sampled = false;
    tl_coarse_dump = true;
    const int rc = run_chunk(h, (uint32_t)nq, 1, 1, metric, 1, st, &ev_used, &scan_events, &sampled, nullptr, nullptr, i8c);
    tl_coarse_dump = false;
    LY_TRY(rc);
    LY_HIP(hipStreamSynchronize(st));
    const bool l2n = i8c && metric == M_L2 && l2_plain(h, nq, false);
    const bool aug = i8c && metric == M_L2 && !l2n, cosq = i8c && metric == M_COS;
    if (out_form) *out_form = (i8c ? 1 : 0) | (aug ? 2 : 0) | (l2n ? 4 : 0) | (cosq ? 8 : 0);
    const bool key_asc = metric_ascending((aug || cosq) ? (int)M_IP : metric);
    std::vector<uint64_t> keys(h->n);
    std::vector<float> m2(nq);
    LY_HIP(hipMemcpy(m2.data(), w.marg2, nq * 4, hipMemcpyDeviceToHost));
    for (uint64_t q = 0; q < nq; ++q) {
        LY_HIP(hipMemcpy(keys.data(), w.cand + (size_t)q * w.cap, (size_t)h->n * 8, hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < h->n; ++r) {
            if (key_row(keys[r]) != (uint32_t)r) return set_error(LYNSE_ERR_INTERNAL, "the emit-all stage did not leave row r in slot r");
            const float sc = key_score(keys[r], key_asc);
            out_scores[q * h->n + r] = (aug || cosq) ? -sc : sc;   // (those forms scan the NEGATED distance as an inner product)
        }
        out_bound[q] = 0.5f * m2[q];
    }
    return LYNSE_OK;
}

// State of the coarse-pass selection of a shard (tests, diagnostics): strikes = overflows of the certified int8 pass so
// far (3 switch it off for the handle, -1 = switched off because the data is not finite), sq8_rows = rows covered by the
// SQ8 codes currently built.
extern "C" int lynse_hip_flat_coarse_state(lynse_hip_flat* h, int* out_strikes, uint64_t* out_sq8_rows) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    std::shared_lock<std::shared_mutex> lk(h->rw);
    if (out_strikes) *out_strikes = h->i8c_strikes.load();
    if (out_sq8_rows) *out_sq8_rows = h->sq8 ? h->n_sq8 : 0;
    return LYNSE_OK;
}

extern "C" uint64_t lynse_hip_flat_bpm_rows(const lynse_hip_flat* h) {
    if (!h) return 0;
    std::shared_lock<std::shared_mutex> lk(const_cast<lynse_hip_flat*>(h)->rw);   // (writers reallocate these fields)
    return h->bpm ? h->n_bpm : 0;
}

// Builds — now, not inside the first search that needs it — every derived copy a batch of `nq` queries of `metric` will read:
// row statistics + f16 shadow (float metrics), the SQ8 codes of the certified int8 pass (IP batches of 33..256 queries over
// >= 64K rows), the packed words (binary metrics from f32 rows) and the +-1 byte copy of batched Hamming (>= 96 queries).
// The reference builds its own derived data lazily too (ensure_binary / ensure_sq8, flat_mmap.rs:375-401); a server calls
// this after a bulk load so that the first query does not pay for it under the writer lock.
extern "C" int lynse_hip_flat_prepare(lynse_hip_flat* h, int metric, uint64_t nq) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    if (!metric_valid(metric)) return set_error(LYNSE_ERR_UNKNOWN_METRIC, "Unknown metric id");
    LY_WRITER(h, lk);
    LY_TRY(use_device(h));
    if (h->n == 0) return LYNSE_OK;
    const uint64_t nqc = std::min<uint64_t>(std::max<uint64_t>(nq, 1), QCHUNK);
    if (metric >= M_HAMMING) {
        LY_TRY(ensure_packed_locked(h));
        if (bin_mfma_eligible(h, metric, false, nqc)) LY_TRY(ensure_bpm_locked(h));
        return LYNSE_OK;
    }
    if (h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "store holds packed rows; float metrics unavailable");
    LY_TRY(finalize_locked(h));
    if (i8c_eligible(h, metric, false, nqc)) {
        LY_TRY(ensure_i8c_codes_locked(h, metric, nqc));
        if (!i8c_codes_finite(h, metric, nqc)) i8c_strike_counter(h, metric).store(-1);
    }
    // a batch that does not take the int8 pass scans the f16 shadow (unless the exact few-query kernel answers it)
    if (!i8c_eligible(h, metric, false, nqc) && !small_path_ok(h, nqc, 10, metric, false)) LY_TRY(ensure_shadow_locked(h));
    return LYNSE_OK;
}

// Bytes of HBM the shard holds: source rows + every derived copy built so far (search workspaces not included).
extern "C" uint64_t lynse_hip_flat_hbm_bytes(const lynse_hip_flat* h) {
    if (!h) return 0;
    std::shared_lock<std::shared_mutex> lk(const_cast<lynse_hip_flat*>(h)->rw);   // (writers reallocate these fields)
    uint64_t b = 0;
    if (h->rows) b += h->capacity * h->ld * 4ull;
    if (h->rows_h) b += h->capacity * h->ld16 * 2ull;
    if (h->rows16 && !h->shadow_alias) b += h->cap16 * h->ld16 * 2ull;
    if (h->vn2) b += 2ull * (h->stats_capacity + 256) * 4;
    if (h->packed) b += h->packed_capacity * h->words * 8ull;
    if (h->sq8) b += h->sq8_cap * h->ld8 + 2ull * (h->sq8_cap + 256) * 4;
    if (h->sq8a) b += h->sq8a_cap * h->ld8a;
    if (h->sq8c) b += h->sq8c_cap * h->ld8;
    if (h->bpm) b += h->bpm_cap * h->ld_bpm;
    return b;
}

// Large k: the shard is cut into row ranges of `cap` rows (any k <= range size is exact there: emit-all + select), every
// range is searched through a temporary VIEW of the handle (pointers advanced to the range, row map offset), and the sorted
// per-range lists are merged on the host in the canonical (distance, row) order (VectorStore::merge_results semantics,
// vector_store.rs:953-970 — the reference's own segments are merged the same way).  Exclusive: the view borrows the handle.
// With a subset filter (search_filtered, flat_mmap.rs:498-560: k.min(subset.len())) every range is searched with the ids of
// the subset that fall into it.
static int search_large_k(lynse_hip_flat* h, const void* q_src, bool packed_queries, uint64_t nq, uint32_t k, int metric,
                          uint64_t* out_rows, float* out_dists, uint32_t* out_counts, bool on_device, hipStream_t user_stream,
                          const uint64_t* subset, uint64_t n_subset, bool filtered, const uint64_t* bitset_words, uint64_t n_words) {
    LY_WRITER(h, xlk);
    LY_TRY(use_device(h));
    const bool binary = metric >= M_HAMMING;
    if (packed_queries && !binary) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "packed queries need a binary metric");
    if (!binary && h->packed_only) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "store holds packed rows; float metrics unavailable");
    if (binary) LY_TRY(ensure_packed_locked(h));
    else { LY_TRY(finalize_locked(h)); LY_TRY(ensure_shadow_locked(h)); }   // (the row-range views below scan the f16 shadow: built for the WHOLE shard first)
    const uint64_t n = h->n, R = h->cap;
    // the subset as a sorted set of valid row ids (a BitSet's to_vec; ids >= n are skipped, duplicates count once)
    std::vector<uint64_t> ids;
    if (filtered) {
        if (bitset_words) {
            const uint64_t nw = std::min<uint64_t>(n_words, (n + 63) / 64);
            for (uint64_t w = 0; w < nw; ++w)
                for (uint64_t bits = bitset_words[w]; bits; bits &= bits - 1) {
                    const uint64_t r = w * 64 + (uint64_t)__builtin_ctzll(bits);
                    if (r < n) ids.push_back(r);
                }
        } else {
            ids.assign(subset, subset + n_subset);
            std::sort(ids.begin(), ids.end());
            ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
            while (!ids.empty() && ids.back() >= n) ids.pop_back();
        }
    }
    const uint32_t kk = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(k, n), filtered ? ids.size() : n);
    const uint64_t n_ranges = (n + R - 1) / R;
    const size_t q_elems = packed_queries ? h->words : h->dim, q_bytes = q_elems * (packed_queries ? 8 : 4);
    if (kk == 0) {  // an empty subset: empty results
        if (on_device) LY_TRY(memset_done(out_counts, 0, nq * 4));
        else memset(out_counts, 0, nq * 4);
        return LYNSE_OK;
    }
    // queries on the host (the per-range searches go through the host-array path)
    std::vector<uint8_t> q_host;
    const uint8_t* qh = reinterpret_cast<const uint8_t*>(q_src);
    if (on_device) {
        q_host.resize((size_t)nq * q_bytes);
        LY_HIP(hipMemcpy(q_host.data(), q_src, q_host.size(), hipMemcpyDeviceToHost));
        qh = q_host.data();
    }
    struct View {  // the scan-side fields of the handle, advanced to one row range for the duration of a call
        lynse_hip_flat* h;
        float* rows; _Float16 *rows16, *rows_h; float *vn2, *vrinv; uint64_t* packed;
        uint64_t n, n_stats, n16, n_packed, row_offset;
        explicit View(lynse_hip_flat* hh) : h(hh), rows(hh->rows), rows16(hh->rows16), rows_h(hh->rows_h), vn2(hh->vn2), vrinv(hh->vrinv), packed(hh->packed),
                                            n(hh->n), n_stats(hh->n_stats), n16(hh->n16), n_packed(hh->n_packed), row_offset(hh->row_offset) {}
        void set(uint64_t r0, uint64_t r1) {
            h->rows = rows ? rows + r0 * h->ld : nullptr;
            h->rows_h = rows_h ? rows_h + r0 * h->ld16 : nullptr;
            h->rows16 = rows16 ? rows16 + r0 * h->ld16 : nullptr;
            h->vn2 = vn2 ? vn2 + r0 : nullptr;
            h->vrinv = vrinv ? vrinv + r0 : nullptr;
            h->packed = packed ? packed + r0 * h->words : nullptr;
            h->n = r1 - r0;
            h->n_stats = h->n; h->n16 = rows16 ? h->n : 0; h->n_packed = packed ? h->n : 0;
            h->row_offset = row_offset + r0 * h->row_stride;
        }
        ~View() {
            h->rows = rows; h->rows_h = rows_h; h->rows16 = rows16; h->vn2 = vn2; h->vrinv = vrinv; h->packed = packed;
            h->n = n; h->n_stats = n_stats; h->n16 = n16; h->n_packed = n_packed; h->row_offset = row_offset;
        }
    } view(h);
    // the IP accumulation form follows the size of the WHOLE shard (flat_mmap.rs:4852-4854), not of a range
    struct FormGuard { lynse_hip_flat* h; int f; ~FormGuard() { h->ip_form = f; } } form_guard{h, h->ip_form};
    if (h->ip_form == LYNSE_IPFORM_AUTO) h->ip_form = n < 4096 ? LYNSE_IPFORM_SINGLE : LYNSE_IPFORM_BATCH8;
    const uint64_t qb = std::max<uint64_t>(1, std::min<uint64_t>(QCHUNK, (64ull << 20) / std::max<uint64_t>(1, n_ranges * kk * 12)));  // queries per pass (<= 64 MB of lists)
    std::vector<uint64_t> l_rows((size_t)qb * n_ranges * kk), m_rows(kk);
    std::vector<float> l_dists((size_t)qb * n_ranges * kk), m_dists(kk);
    std::vector<uint32_t> l_counts((size_t)qb * n_ranges), r_counts(qb);
    std::vector<uint64_t> t_rows((size_t)qb * kk);
    std::vector<float> t_dists((size_t)qb * kk);
    std::vector<uint64_t> o_rows(on_device ? (size_t)nq * k : 0);
    std::vector<float> o_dists(on_device ? (size_t)nq * k : 0);
    std::vector<uint32_t> o_counts(on_device ? nq : 0);
    uint64_t* res_rows = on_device ? o_rows.data() : out_rows;
    float* res_dists = on_device ? o_dists.data() : out_dists;
    uint32_t* res_counts = on_device ? o_counts.data() : out_counts;
    for (uint64_t q0 = 0; q0 < nq; q0 += qb) {
        const uint64_t nb = std::min<uint64_t>(qb, nq - q0);
        for (uint64_t ri = 0; ri < n_ranges; ++ri) {
            const uint64_t r0 = ri * R, r1 = std::min<uint64_t>(n, r0 + R);
            uint32_t kr = (uint32_t)std::min<uint64_t>(kk, r1 - r0);
            std::vector<uint64_t> local;   // the subset ids of this range, range-local
            if (filtered) {
                const auto lo = std::lower_bound(ids.begin(), ids.end(), r0), hi = std::lower_bound(ids.begin(), ids.end(), r1);
                local.assign(lo, hi);
                for (uint64_t& v : local) v -= r0;
                kr = (uint32_t)std::min<uint64_t>(kr, local.size());
                if (local.empty()) {
                    for (uint64_t q = 0; q < nb; ++q) l_counts[q * n_ranges + ri] = 0;
                    continue;
                }
            }
            view.set(r0, r1);
            const int rc = search_impl(h, qh + q0 * q_bytes, packed_queries, nb, kr, metric, t_rows.data(), t_dists.data(), r_counts.data(),
                                       false, user_stream, filtered ? local.data() : nullptr, local.size(), filtered, nullptr, 0, true);
            if (rc != LYNSE_OK) return rc;
            for (uint64_t q = 0; q < nb; ++q) {  // [query][range][kk] lists for the merge
                const uint32_t c = r_counts[q];
                l_counts[q * n_ranges + ri] = c;
                memcpy(&l_rows[(q * n_ranges + ri) * kk], &t_rows[q * kr], (size_t)c * 8);
                memcpy(&l_dists[(q * n_ranges + ri) * kk], &t_dists[q * kr], (size_t)c * 4);
            }
        }
        for (uint64_t q = 0; q < nb; ++q) {
            uint32_t cnt = 0;
            LY_TRY(lynse_hip_merge_topk(&l_rows[q * n_ranges * kk], &l_dists[q * n_ranges * kk], &l_counts[q * n_ranges], (uint32_t)n_ranges, kk, kk,
                                        metric, m_rows.data(), m_dists.data(), &cnt));
            memcpy(res_rows + (q0 + q) * k, m_rows.data(), (size_t)cnt * 8);
            memcpy(res_dists + (q0 + q) * k, m_dists.data(), (size_t)cnt * 4);
            res_counts[q0 + q] = cnt;
        }
    }
    if (on_device) {
        LY_TRY(h2d_done(out_rows, o_rows.data(), o_rows.size() * 8));
        LY_TRY(h2d_done(out_dists, o_dists.data(), o_dists.size() * 4));
        LY_TRY(h2d_done(out_counts, o_counts.data(), o_counts.size() * 4));
    }
    return LYNSE_OK;
}

// FLAT-*-SQ8 with more pass-1 candidates than one pass holds (n_cand = 20 k, cosine 100 k, flat_mmap.rs:5883-5893; above
// cap / 4 over more than cap rows): pass 1 runs per row range of `cap` rows (exact top-n_cand of the integer code scores
// there), the per-range lists are merged on the host in the (score, row) order, and pass 2 — the exact f32 scores of the
// n_cand rows, (distance, row) order, top k — is the subset-filtered exact search over those rows (single-row kernels, as
// flat_mmap.rs:5899-5906).  One query at a time: a rare shape, not a tuned one.  Called with the writer lock held.
static int search_sq8_large(lynse_hip_flat* h, std::unique_lock<std::shared_mutex>& lk, const float* queries, uint64_t nq, uint32_t k, uint32_t kk,
                            uint32_t n_cand, int metric, uint64_t* out_rows, float* out_dists, uint32_t* out_counts) {
    const uint64_t n = h->n, R = h->cap, n_ranges = (n + R - 1) / R;
    const int m1 = metric == M_IP ? M_IP : M_L2;
    LY_TRY(ensure_workspace(h, (uint32_t)std::min<uint64_t>(n_cand, R)));
    Workspace& w = cur(h).ws;
    hipStream_t st = cur(h).stream;
    struct Sq8View {  // the pass-1 side of the handle advanced to one row range; rows come back as RAW shard rows
        lynse_hip_flat* h;
        float* rows; int8_t* sq8; int *sum, *sum2; uint64_t n, n_sq8, stride, offset;
        explicit Sq8View(lynse_hip_flat* hh) : h(hh), rows(hh->rows), sq8(hh->sq8), sum(hh->sq8_sum), sum2(hh->sq8_sum2), n(hh->n), n_sq8(hh->n_sq8),
                                               stride(hh->row_stride), offset(hh->row_offset) {}
        void set(uint64_t r0, uint64_t r1) {
            h->rows = rows + r0 * h->ld; h->sq8 = sq8 + r0 * h->ld8; h->sq8_sum = sum + r0; h->sq8_sum2 = sum2 + r0;
            h->n = r1 - r0; h->n_sq8 = h->n; h->row_stride = 1; h->row_offset = r0;
        }
        void restore() { h->rows = rows; h->sq8 = sq8; h->sq8_sum = sum; h->sq8_sum2 = sum2; h->n = n; h->n_sq8 = n_sq8; h->row_stride = stride; h->row_offset = offset; }
        ~Sq8View() { restore(); }
    };
    std::vector<std::vector<uint64_t>> cand(nq);   // per query: the top-n_cand rows of pass 1
    {
        Sq8View view(h);
        const uint64_t qb = std::max<uint64_t>(1, std::min<uint64_t>(QCHUNK, (64ull << 20) / std::max<uint64_t>(1, n_ranges * n_cand * 12)));
        std::vector<uint64_t> l_rows((size_t)qb * n_ranges * n_cand), m_rows(n_cand), t_rows;
        std::vector<float> l_dists((size_t)qb * n_ranges * n_cand), m_dists(n_cand), t_dists;
        std::vector<uint32_t> l_counts((size_t)qb * n_ranges);
        for (uint64_t q0 = 0; q0 < nq; q0 += qb) {
            const uint32_t nb = (uint32_t)std::min<uint64_t>(qb, nq - q0);
            LY_HIP(hipMemcpyAsync(w.Qf, queries + q0 * h->dim, (size_t)nb * h->dim * 4, hipMemcpyHostToDevice, st));
            for (uint64_t ri = 0; ri < n_ranges; ++ri) {
                const uint64_t r0 = ri * R, r1 = std::min<uint64_t>(n, r0 + R);
                const uint32_t kr = (uint32_t)std::min<uint64_t>(n_cand, r1 - r0);
                view.set(r0, r1);
                for (int level = 1; level < 3; ++level) {
                    LY_TRY(run_chunk_sq8(h, nb, kr, kr, kr, metric, level, st, true));
                    LY_HIP(hipMemcpyAsync(w.h_hdr, w.out_counts, (size_t)2 * w.qcap * 4, hipMemcpyDeviceToHost, st));
                    LY_HIP(hipStreamSynchronize(st));
                    uint32_t nov = 0;
                    for (uint32_t i = 0; i < nb; ++i) nov += w.h_hdr[w.qcap + i] ? 1 : 0;
                    if (nov == 0) break;
                    if (level == 2) return set_error(LYNSE_ERR_INTERNAL, "candidate overflow on the exhaustive plan");
                }
                t_rows.resize((size_t)nb * kr);
                t_dists.resize((size_t)nb * kr);
                LY_HIP(hipMemcpy(t_rows.data(), w.out_rows, t_rows.size() * 8, hipMemcpyDeviceToHost));
                LY_HIP(hipMemcpy(t_dists.data(), w.out_dists, t_dists.size() * 4, hipMemcpyDeviceToHost));
                for (uint32_t q = 0; q < nb; ++q) {
                    const uint32_t c = w.h_hdr[q];
                    l_counts[q * n_ranges + ri] = c;
                    memcpy(&l_rows[(q * n_ranges + ri) * n_cand], &t_rows[(size_t)q * kr], (size_t)c * 8);
                    memcpy(&l_dists[(q * n_ranges + ri) * n_cand], &t_dists[(size_t)q * kr], (size_t)c * 4);
                }
            }
            for (uint32_t q = 0; q < nb; ++q) {
                uint32_t cnt = 0;
                LY_TRY(lynse_hip_merge_topk(&l_rows[(size_t)q * n_ranges * n_cand], &l_dists[(size_t)q * n_ranges * n_cand], &l_counts[(size_t)q * n_ranges],
                                            (uint32_t)n_ranges, n_cand, n_cand, m1, m_rows.data(), m_dists.data(), &cnt));
                cand[q0 + q].assign(m_rows.begin(), m_rows.begin() + cnt);
            }
        }
    }
    // pass 2 through the public filtered path (it takes the locks itself; large k there goes through the row-range views)
    lk.unlock();
    for (uint64_t q = 0; q < nq; ++q) {
        LY_TRY(search_impl(h, queries + q * h->dim, false, 1, k, metric, out_rows + q * k, out_dists + q * k, out_counts + q, false, nullptr,
                           cand[q].data(), cand[q].size(), true));
        (void)kk;
    }
    return LYNSE_OK;
}

extern "C" int lynse_hip_flat_search_f32(lynse_hip_flat* h, const float* queries, uint64_t nq, uint32_t k, int metric,
                                         uint64_t* out_rows, float* out_dists, uint32_t* out_counts) {
    return search_impl(h, queries, false, nq, k, metric, out_rows, out_dists, out_counts, false, nullptr);
}

extern "C" int lynse_hip_flat_search_filtered_f32(lynse_hip_flat* h, const float* queries, uint64_t nq, uint32_t k, int metric,
                                                  const uint64_t* subset_rows, uint64_t n_subset, uint64_t* out_rows,
                                                  float* out_dists, uint32_t* out_counts) {
    return search_impl(h, queries, false, nq, k, metric, out_rows, out_dists, out_counts, false, nullptr, subset_rows, n_subset, true);
}

extern "C" int lynse_hip_flat_search_filtered_bitset_f32(lynse_hip_flat* h, const float* queries, uint64_t nq, uint32_t k, int metric,
                                                         const uint64_t* bitset_words, uint64_t n_words, uint64_t* out_rows,
                                                         float* out_dists, uint32_t* out_counts) {
    if (!h) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "handle is NULL");
    if (n_words && !bitset_words) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "bitset is NULL");
    const uint64_t n = lynse_hip_flat_len(h);
    uint64_t count = 0;  // BitSet::count() restricted to rows below len
    const uint64_t full = std::min<uint64_t>(n_words, n / 64);
    for (uint64_t w = 0; w < full; ++w) count += (uint64_t)__builtin_popcountll(bitset_words[w]);
    if (full < n_words && (n % 64)) count += (uint64_t)__builtin_popcountll(bitset_words[full] & ((1ull << (n % 64)) - 1ull));
    return search_impl(h, queries, false, nq, k, metric, out_rows, out_dists, out_counts, false, nullptr, nullptr, count, true,
                       bitset_words, n_words);
}

extern "C" int lynse_hip_flat_search_f32_device(lynse_hip_flat* h, const float* d_queries, uint64_t nq, uint32_t k,
                                                int metric, uint64_t* d_out_rows, float* d_out_dists,
                                                uint32_t* d_out_counts, void* stream) {
    return search_impl(h, d_queries, false, nq, k, metric, d_out_rows, d_out_dists, d_out_counts, true, (hipStream_t)stream);
}

extern "C" int lynse_hip_flat_search_packed_u64(lynse_hip_flat* h, const uint64_t* qw, uint64_t nq, uint32_t k, int metric,
                                                uint64_t* out_rows, float* out_dists, uint32_t* out_counts) {
    return search_impl(h, qw, true, nq, k, metric, out_rows, out_dists, out_counts, false, nullptr);
}

extern "C" int lynse_hip_flat_search_packed_u64_device(lynse_hip_flat* h, const uint64_t* d_qw, uint64_t nq, uint32_t k,
                                                       int metric, uint64_t* d_out_rows, float* d_out_dists,
                                                       uint32_t* d_out_counts, void* stream) {
    return search_impl(h, d_qw, true, nq, k, metric, d_out_rows, d_out_dists, d_out_counts, true, (hipStream_t)stream);
}

// --------------------------------------------------------------------------- stand-alone calls ----
extern "C" int lynse_hip_top_k_search(const float* query, const float* candidates, uint64_t n, uint32_t dim, uint32_t k,
                                      int metric, int device, uint32_t* out_idx, float* out_dist, uint32_t* out_count) {
    // distance::top_k_search (distance/mod.rs:373-422): single-row kernels for every candidate.
    if (!out_count) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "out_count is NULL");
    *out_count = 0;
    if (!metric_valid(metric)) return set_error(LYNSE_ERR_UNKNOWN_METRIC, "Unknown metric id");
    if (dim == 0) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "dimension must be greater than zero");
    if (n == 0 || k == 0) return LYNSE_OK;
    if (!query || !candidates || !out_idx || !out_dist) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    // A scratch shard per calling thread and (device, dim), emptied and refilled per call: py_top_k_search / py_compute_distance /
    // Collection::pending_search call this in loops, and a handle (stream, stats buffers, workspace) is not free to create.
    struct Scratch {
        lynse_hip_flat* h = nullptr; int device = -1; uint32_t dim = 0;
        ~Scratch() { if (h) lynse_hip_flat_destroy(h); }
    };
    static thread_local Scratch sc;
    if (!sc.h || sc.device != device || sc.dim != dim) {
        if (sc.h) { lynse_hip_flat_destroy(sc.h); sc.h = nullptr; }
        LY_TRY(lynse_hip_flat_create(dim, device, &sc.h));
        sc.device = device; sc.dim = dim;
        LY_TRY(lynse_hip_flat_set_ip_form(sc.h, LYNSE_IPFORM_SINGLE));
    }
    lynse_hip_flat* h = sc.h;
    {   // empty the shard (capacity and buffers stay)
        std::unique_lock<std::shared_mutex> lk(h->rw);
        h->n = 0; h->n_stats = 0; h->n16 = 0; h->n_packed = 0; h->n_sq8 = 0; h->n_sq8a = 0; h->n_sq8c = 0; h->n_bpm = 0; h->i8c_strikes.store(0); h->i8c_strikes_l2.store(0); h->i8c_strikes_cos.store(0);
        h->amax = h->vmax = h->vmin = 0.f; h->sv = 1.f; h->cos_degenerate = 0; h->rows_integer = 0; h->rows_nonneg = 0;
        const uint32_t init[5] = {0u, 0u, 0x7f800000u, 0u, 0u};
        LY_TRY(use_device(h));
        LY_TRY(h2d_done(h->d_stats, init, sizeof init));
    }
    int rc = lynse_hip_flat_append_f32(h, candidates, n);
    const uint32_t kk = (uint32_t)std::min<uint64_t>(k, n);
    std::vector<uint64_t> rows(kk);
    uint32_t cnt = 0;
    if (rc == LYNSE_OK) rc = lynse_hip_flat_search_f32(h, query, 1, kk, metric, rows.data(), out_dist, &cnt);
    if (rc != LYNSE_OK) return rc;
    for (uint32_t i = 0; i < cnt; ++i) out_idx[i] = (uint32_t)rows[i];
    *out_count = cnt;
    return LYNSE_OK;
}

extern "C" int lynse_hip_compute_distance(const float* a, const float* b, uint32_t dim, int metric, int device, float* out) {
    // distance::compute_distance_f32(a, b) (distance/mod.rs:193-213): a plays the query.
    if (!a || !b || !out) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    uint32_t idx = 0, cnt = 0;
    float d = 0.f;
    LY_TRY(lynse_hip_top_k_search(a, b, 1, dim, 1, metric, device, &idx, &d, &cnt));
    if (cnt != 1) return set_error(LYNSE_ERR_INTERNAL, "distance kernel returned no result");
    *out = d;
    return LYNSE_OK;
}

extern "C" int lynse_hip_pack_binary_f32(const float* rows, uint64_t n, uint32_t dim, int device, uint64_t* out_words) {
    if (n == 0) return LYNSE_OK;
    if (!rows || !out_words) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    lynse_hip_flat* h = nullptr;
    LY_TRY(lynse_hip_flat_create(dim, device, &h));
    int rc = lynse_hip_flat_append_f32(h, rows, n);
    if (rc == LYNSE_OK) rc = lynse_hip_flat_read_packed(h, 0, n, out_words);
    lynse_hip_flat_destroy(h);
    return rc;
}

extern "C" int lynse_hip_merge_topk(const uint64_t* ids, const float* dists, const uint32_t* counts, uint32_t n_lists,
                                    uint32_t stride, uint32_t k, int metric, uint64_t* out_ids, float* out_dists,
                                    uint32_t* out_count) {
    // VectorStore::merge_results (vector_store.rs:953-970): (distance in metric order, id ascending).
    if (!out_count) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "out_count is NULL");
    *out_count = 0;
    if (!metric_valid(metric)) return set_error(LYNSE_ERR_UNKNOWN_METRIC, "Unknown metric id");
    if (n_lists == 0 || k == 0) return LYNSE_OK;
    if (!ids || !dists || !counts || !out_ids || !out_dists) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    const bool asc = metric_ascending(metric);
    struct E { float d; uint64_t id; };
    std::vector<E> all;
    for (uint32_t l = 0; l < n_lists; ++l) {
        if (counts[l] > stride) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "count exceeds stride");
        for (uint32_t i = 0; i < counts[l]; ++i) all.push_back({dists[(size_t)l * stride + i], ids[(size_t)l * stride + i]});
    }
    auto before = [asc](const E& x, const E& y) {
        if (x.d < y.d) return asc;
        if (x.d > y.d) return !asc;
        return x.id < y.id;  // NaN compares Equal like partial_cmp().unwrap_or(Equal)
    };
    const size_t kk = std::min<size_t>(k, all.size());
    std::partial_sort(all.begin(), all.begin() + kk, all.end(), before);
    for (size_t i = 0; i < kk; ++i) { out_ids[i] = all[i].id; out_dists[i] = all[i].d; }
    *out_count = (uint32_t)kk;
    return LYNSE_OK;
}

static int merge_topk_device_impl(const void* d_blocks, uint64_t block_bytes, uint64_t rows_off,
                                  uint64_t dists_off, uint64_t counts_off, uint32_t n_lists, uint64_t nq,
                                  uint32_t k, int metric, uint64_t* d_out_rows, float* d_out_dists,
                                  uint32_t* d_out_counts, void* stream, uint64_t status_off, uint32_t* out_status) {
    if (!metric_valid(metric)) return set_error(LYNSE_ERR_UNKNOWN_METRIC, "Unknown metric id");
    if (nq == 0) return LYNSE_OK;
    if (!d_blocks || !d_out_counts || (k && (!d_out_rows || !d_out_dists))) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_lists == 0) return set_error(LYNSE_ERR_INVALID_ARGUMENT, "n_lists is zero");
    if (k == 0) {
        LY_HIP(hipMemsetAsync(d_out_counts, 0, nq * 4, (hipStream_t)stream));
        return LYNSE_OK;
    }
    const uint64_t total = (uint64_t)n_lists * k;
    if (total > 8192) return set_error(LYNSE_ERR_UNSUPPORTED, "n_lists * k exceeds the merge kernel capacity (8192)");
    uint32_t np2 = 2;
    while (np2 < total) np2 <<= 1;
    static std::atomic<bool> attr = false;
    if (!attr) {
        LY_TRY(set_max_lds(k_merge<256>, 8192 * 12));
        attr = true;
    }
    MergeArgs a{};
    a.blocks = (const char*)d_blocks; a.block_bytes = block_bytes; a.rows_off = rows_off; a.dists_off = dists_off;
    a.counts_off = counts_off; a.n_lists = n_lists; a.k = k; a.metric = metric;
    a.out_rows = d_out_rows; a.out_dists = d_out_dists; a.out_counts = d_out_counts;
    a.status_off = status_off; a.out_status = out_status;
    hipLaunchKernelGGL(k_merge<256>, dim3((uint32_t)nq), dim3(256), (size_t)np2 * 12, (hipStream_t)stream, a);
    LY_HIP(hipGetLastError());
    return LYNSE_OK;
}

extern "C" int lynse_hip_merge_topk_device(const void* d_blocks, uint64_t block_bytes, uint64_t rows_off,
                                           uint64_t dists_off, uint64_t counts_off, uint32_t n_lists, uint64_t nq,
                                           uint32_t k, int metric, uint64_t* d_out_rows, float* d_out_dists,
                                           uint32_t* d_out_counts, void* stream) {
    return merge_topk_device_impl(d_blocks, block_bytes, rows_off, dists_off, counts_off, n_lists, nq, k, metric, d_out_rows,
                                  d_out_dists, d_out_counts, stream, 0, nullptr);
}

// ---------------------------------------------------------------------------------------------- bounded waits ----
// Every host wait that ends in a COLLECTIVE (the all-gather of a sharded batch, the all-reduce of a Lloyd iteration) is bounded: a rank
// that died or returned early must not strand its peers inside hipStreamSynchronize for good.  The reference's coordinator retries a
// shard once and errors (src/cluster.rs:243-261).  lynse_hip_set_wait_timeout_ms / LYNSE_HIP_WAIT_TIMEOUT_MS; 0 = 30 s for waits behind a
// collective, unbounded for one shard alone.
static std::atomic<uint32_t> g_wait_timeout_ms{[]() { const char* e = getenv("LYNSE_HIP_WAIT_TIMEOUT_MS"); return e ? (uint32_t)atoll(e) : 0u; }()};
extern "C" int lynse_hip_set_wait_timeout_ms(uint32_t ms) { g_wait_timeout_ms.store(ms); return LYNSE_OK; }
static uint32_t collective_timeout_ms() { const uint32_t to = g_wait_timeout_ms.load(); return to ? to : 30000u; }

// hipEventSynchronize with an upper bound: polls the event (spinning first: a batch is a millisecond or two), sleeps in between
static int event_wait_bounded(hipEvent_t ev, uint32_t timeout_ms) {
    if (timeout_ms == 0) { LY_HIP(hipEventSynchronize(ev)); return LYNSE_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return LYNSE_OK;
        if (q != hipErrorNotReady) return set_error(LYNSE_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
        const auto el = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (el > (int64_t)timeout_ms * 1000) return set_error(LYNSE_ERR_TIMEOUT, "the batch did not finish within the wait timeout (a rank of the collective is gone?)");
        if (el > 2000) std::this_thread::sleep_for(std::chrono::microseconds(el > 100000 ? 1000 : 50));
    }
}
// ... of everything enqueued on `st` so far, through an event of the caller (created on first use)
static int stream_wait_bounded(hipStream_t st, hipEvent_t* ev, uint32_t timeout_ms) {
    if (!*ev) LY_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    LY_HIP(hipEventRecord(*ev, st));
    return event_wait_bounded(*ev, timeout_ms);
}
// Status word of a result block (comm_block_layout), OR-ed over the ranks by the merge.  The low byte asks for the batch to be
// answered again, on every rank: bit 0 = a query of that shard overflowed its candidate buffers (next plan level), bit 1 = an IVF query
// whose probed lists are all empty (k_or_word: the scan-every-list fallback of ivf.rs:258-265).  Any bit ABOVE the low byte = that rank
// FAILED while it built its block (the failing rank fills the word's bytes with 2: hipMemsetAsync) — it still takes part in the
// exchange (with an empty block), so that its peers learn of it from the merged word instead of from a timeout.
constexpr uint32_t STATUS_OVERFLOW = 1u, STATUS_EMPTY_LISTS = 2u, STATUS_REDO = STATUS_OVERFLOW | STATUS_EMPTY_LISTS;
static inline bool status_failed(uint32_t st) { return (st & 0xffffff00u) != 0u; }

#include "ivf_host.inc"
#include "shard_host.inc"
#include "comm_host.inc"
#include "async_host.inc"
#include "ivf_async.inc"
