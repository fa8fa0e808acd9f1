// scan_qh.h — k_scan_qh: the QUERY-STATIONARY tiling for the f16 shadow at LOW dimension (64 .. 256 columns: BASELINE config 3,
// SIFT-like 1M x 128, squared L2, k = 100).  Threshold stages of an unfiltered FLAT batch of 33..256 queries on the float path.
//
// Reference work on this path: FlatMmap::search -> exact_flat_search's chunked scan (src/storage/flat_mmap.rs:2179-2256, :2132-2176;
// the distance kernels src/distance/simd.rs:1529-1581) — every row of the shard scored against every query of the batch.
//
// What bounded the 256-row x 256-query tile of k_scan_h16<4,2,2,4> there: at 128 columns a tile is two slab steps, so the
// float epilogue (one value per (row, query) pair: as many VALU operations as the two steps have MFMA cycles) and the tile boundary —
// eight waves draining into the epilogue together — weigh as much as the MFMAs; nothing overlapped them.  Here:
//   * the query operand is REGISTER-resident for the whole launch, as in k_scan_qs (scan_qs.h): wave w owns queries [32 w, 32 w + 32)
//     and keeps their f16 image — the B operand of v_mfma_f32_32x32x16_f16 for every k-step, NSLAB x 16 registers (32 at 128 columns);
//   * a wave needs 128 registers and a workgroup 78 KB of LDS, so TWO workgroups share a CU (four waves per SIMD): the branchy float
//     epilogue of one runs under the MFMAs of the other, and twice as many waves fill each other's LDS / branch bubbles;
//   * inside a workgroup the two waves of a SIMD run half a step out of phase (the ping-pong roles of k_scan_qs): early waves
//     barrier -> MFMAs -> epilogue -> DMA issue, late waves  barrier -> epilogue of the PREVIOUS tile -> DMA issue -> MFMAs;
//   * the LDS holds only rows (a ring of NS whole-K stages of 64 rows, global_load_lds_dwordx4 in full 128-B lines, the XOR slot
//     swizzle of k_scan_h16 on the source address), for L2 / cosine a small ring of the tile's f32 row norms, and the waves' staging
//     regions of the deferred emission (below);
//   * level 1 of the epilogue is that of the DENSE float epilogue of k_scan_h16 (the same value against the same loosened threshold),
//     the exact coarse expression and the keys are k_scan_h16's — computed one staged group per lane between tiles, not by one or two
//     active lanes inside the tile epilogue.
// Byte layouts are those of k_scan_h16: f16 rows with a pitch of ld16 halves (whole 64-element slabs), the query image of
// k_prep_queries (layout 2: [slab][q][8 slots ^ ((q >> 1) & 7)][16 B]) — a 16-element k-step of v_mfma_f32_32x32x16_f16 is 32 bytes
// of a row, exactly the k-step of the int8 form.
// Measured and not kept: 64 queries per wave on 128-row tiles with one workgroup per CU (every fragment read feeds two
// MFMAs: 10-20 % slower), keys stored straight to global memory, keys staged one by one with the exact test inside the tile epilogue.
#pragma once

namespace lynse {

typedef float qh_f32x2 __attribute__((ext_vector_type(2)));

// QB = query blocks of 32 per wave: 1 (the template keeps the parameter of the measured 64-queries-per-wave form; static_assert below).
// One segment of candB per workgroup and query (nseg = grid).  debug_flags & 2 = no emission, & 64 = s_memtime phase sums (scripts/qh_phase_timing.py)
// SMP = 1: the THRESHOLD-ONLY SAMPLE STAGE of a staged plan on this tiling (it was one 256-row tile of k_scan_h16<.., EMIT = 2> per CU:
// 23 us of launch ramp, query image and ring fill for 65,536 rows).  a.ntiles tiles of 64 rows, tile t = rows (t / 4) * a.tile_stride +
// (t % 4) * 64 ... of the shard (the same sample rows); no threshold, no staging — every lane keeps the best TWO rows of those it sees for
// its query as PACKED key words (the score word of make_key rounded UP to a multiple of 1024 — towards 'worse': the key claims no better
// score than the row has — with the tile ordinal and the accumulator slot in the low 10 bits) and writes them to
// cand[query][(blockIdx.x * 2 + hi) * 2 + t]: 4 keys per workgroup and query.  Any k distinct real rows bound the k-th best score, so
// k_select's threshold-only rule turns the k-th best of these keys into a valid first threshold as it does with k_scan_qs<.., SMP>'s.
template <int NSLAB, int MET, int NS, int NBUF, int QB, int SMP = 0>
__global__ void __launch_bounds__(512, (QB == 1 ? 4 : 2)) k_scan_qh(ScanArgs a) {
    static_assert(QB == 1, "query blocks per wave: the 64-queries-per-wave form (QB = 2) measured 10-20 % slower and is not instantiated any more");
    constexpr int RT = 64 * QB;              // rows per tile
    constexpr int SB = NSLAB * RT * 128;     // bytes per ring stage (a whole tile, all K)
    constexpr int PP = SB / 1024;            // LDS-DMA instructions per stage (1 KiB each: 8 rows x 128 B)
    constexpr int PPW = PP / 8;              // ... per wave
    static_assert(PP % 8 == 0, "the pieces of a stage split evenly over the 8 waves");
    constexpr int NR = NSLAB * 4 * 2;        // fragment reads per wave and step (slab, k-step, row block); each feeds QB MFMAs
    static_assert(NR % NBUF == 0 && NBUF >= 2 && NBUF <= NR, "fragment ring");
    static_assert(NS >= 3, "ring depth");
    constexpr bool NORMS = MET != M_IP;
    constexpr bool ASC = MET != M_IP;
    constexpr int NRM = RT * 4;              // bytes of row norms per tile slot (one one-dword LDS-DMA of wave 0 per 64 rows)
    constexpr int NRM_SLOTS = NS + 1;        // (a slot is refilled two steps after its tile was computed: the late waves' epilogues are safe)
    constexpr int NRM_OFF = NS * SB;
    static_assert(NRM_OFF + (NORMS ? NRM_SLOTS * NRM : 0) <= (QB == 1 ? 80 : 160) * 1024, "LDS");
    // Deferred emission (round 5).  A key stored straight to global memory rides on vmcnt in front of the ring pieces of its step, and the
    // next counted wait in front of the barrier then waits for its write acknowledgement: at 128 columns a step is as short as that round
    // trip (~1.8 us) and with k = 100 every step of every workgroup holds a key (s_memtime: 148k of a stage's ticks with emission against
    // 78k without).  And the grouped slow path — norm re-read, the exact expression, the pass test, the key — ran ~100 instructions with one
    // or two of 64 lanes active, per hit about as long as the MFMAs of the step.  Here the tile epilogue only STAGES the groups of four
    // values whose level-1 maximum passes: the lane appends (4 accumulators, first row, query) to its wave's LDS region (appends ride on
    // lgkmcnt; one ballot per group).  The region is worked off BETWEEN tiles once it is half full and at the end of the launch, one ENTRY per lane:
    // norms from global memory, the exact expression of k_scan_h16, the pass test, the key, a slot of the workgroup's segment of that
    // query (one LDS counter per query), a plain store.
    constexpr int EW = QB == 1 ? (NS == 3 ? 144 : 64) : 128;   // groups per wave region (16 B of accumulators + first row + query)
    constexpr int STG_OFF = NRM_OFF + (NORMS ? NRM_SLOTS * NRM : 0);
    constexpr int QCNT_OFF = STG_OFF + 8 * EW * 24;   // u32[256]: keys of this workgroup per query
    static_assert(QCNT_OFF + 256 * 4 <= (QB == 1 ? 80 : 160) * 1024, "LDS");
    constexpr int NXW = NORMS ? QB : 0;      // LDS-DMAs wave 0 issues per step on top of its ring pieces
    constexpr int WAITN = (NS - 2) * PPW;    // ring pieces that may still be in flight at the barrier
    static_assert((NS - 1) * (PPW + NXW) <= 63, "vmcnt");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = QB == 2 ? (wave & 3) : wave, wr = QB == 2 ? (wave >> 2) : 0;
    const bool late = wave >= 4;
    const int l32 = lane & 31, hi = lane >> 5;
    const uint32_t ntiles = SMP != 0 ? a.ntiles : (a.row1 - a.row0 + RT - 1) / RT;
    if (blockIdx.x >= ntiles) return;
    auto tile_row0 = [&](uint32_t t) -> uint32_t {   // first row of tile t
        if constexpr (SMP != 0) return a.row0 + (t >> 2) * a.tile_stride + (t & 3u) * (uint32_t)RT;
        else return a.row0 + t * RT;
    };
    const uint32_t n_ord = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;   // this workgroup's tiles: blockIdx.x, + gridDim.x, ...
    auto tile_of = [&](uint32_t ord) -> uint32_t { return blockIdx.x + ord * gridDim.x; };
    const uint32_t ldb = a.ld16 * 2u;        // row pitch in bytes

    // ---- the wave's query blocks: B fragments of every k-step, register-resident for the whole launch
    const int swz = (l32 >> 1) & 7;
    qs_i32x4 bq[QB][NSLAB * 4];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const char* qimg = reinterpret_cast<const char*>(a.Q16) + (size_t)(wq * (32 * QB) + j * 32 + l32) * 128;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                bq[j][s * 4 + kk] = *reinterpret_cast<const qs_i32x4*>(qimg + (size_t)s * a.qpad * 128 + (((kk * 2 + hi) ^ swz) * 16));
    }
    // per-query constants of this lane's queries (column lane % 32 of each block)
    uint32_t qn[QB];
    bool ok[QB];
    float c_qinv[QB], c_extra[QB], c_thr[QB], c_lim[QB];   // c_lim: what the level-1 value is compared with (k_scan_h16, set_pre)
    const bool wave_live = (uint32_t)wq * (32u * QB) < a.nq;     // a wave whose queries all lie beyond the batch only feeds the ring
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        qn[j] = (uint32_t)(wq * (32 * QB) + j * 32 + l32);
        ok[j] = qn[j] < a.nq;
        const uint32_t n = ok[j] ? qn[j] : 0u;
        c_qinv[j] = ok[j] ? a.qinv[n] : 0.0f;
        c_extra[j] = MET == M_L2 ? (ok[j] ? a.qn2[n] : 0.0f) : (MET == M_COS ? (ok[j] ? a.qrinv[n] : 0.0f) : 0.0f);
        c_thr[j] = ok[j] ? a.thr[n] : 0.0f;
        float lim;
        if constexpr (MET == M_IP) {
            lim = c_thr[j];                                        // (IP: the level-1 value is the score itself)
        } else if constexpr (MET == M_L2) {
            const float qn2 = c_extra[j], vmax2 = a.vmax2;
            lim = (qn2 - c_thr[j]) - 2e-6f * (qn2 + fabsf(c_thr[j]) + vmax2);
            if (!(vmax2 > 0.0f)) lim = -LY_INF;
            if (!(fabsf(lim) < 3.0e38f)) lim = -LY_INF;            // inf / NaN (open threshold, overflow): no pre-filter
        } else {
            const float den = c_qinv[j] * c_extra[j];              // qinv / |q|
            lim = ((1.0f - c_thr[j]) - 1e-6f * (1.0f + fabsf(c_thr[j]))) / den;
            lim = lim - fabsf(lim) * 2e-6f;
            if (!(den > 0.0f)) lim = -LY_INF;
            if (!(fabsf(lim) < 3.0e38f)) lim = -LY_INF;
        }
        if (a.debug_flags & 2) lim = LY_INF;                       // (timing experiments: no emission)
        c_lim[j] = ok[j] ? lim : LY_INF;
    }
    asm volatile("" ::: "memory");   // (the fragment / constant loads stay in front of the ring pieces)

    // ---- row stream (LDS-DMA): step g of this workgroup = its g-th tile.  Piece p = wave * PPW + j of a stage: slab p / (RT / 8), rows
    // (p % (RT / 8)) * 8 .. +8 of the tile; lane l brings the 16 B of row (l >> 3), PHYSICAL slot (l & 7) = logical slot (l & 7) ^ ((row >> 1) & 7)
    uint32_t v_off[PPW];
    const char* v_base = nullptr;   // uniform: first row of the tile being issued
    uint32_t is_ord = 0, is_stage = 0, is_count = 0;
    auto enter_tile = [&]() {
        const uint32_t rbase = tile_row0(tile_of(is_ord));
        const uint32_t span = a.row1 - 1 - rbase;   // rows past the last one re-read it (masked in the epilogue)
        v_base = reinterpret_cast<const char*>(a.V16) + (size_t)rbase * ldb;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = wave * PPW + j;
            uint32_t r = (p % (RT / 8)) * 8 + (lane >> 3);
            const uint32_t col = (uint32_t)(p / (RT / 8)) * 128u + (((lane & 7) ^ ((r >> 1) & 7)) * 16);
            r = r < span ? r : span;
            v_off[j] = r * ldb + col;
        }
    };
    auto issue_pieces = [&]() {
#pragma unroll
        for (int j = 0; j < PPW; ++j) glds16<2>(v_base + v_off[j], smem + is_stage * SB + (wave * PPW + j) * 1024);
        if constexpr (NORMS) {   // wave 0: the f32 norms of the tile being issued (rows past the end: the arrays' slack of 256 floats)
            if (wave == 0) {
                const float* src = (MET == M_L2 ? a.vn2 : a.vrinv) + tile_row0(tile_of(is_ord)) + lane;
                char* dst = smem + NRM_OFF + (is_count % NRM_SLOTS) * NRM;
#pragma unroll
                for (int t = 0; t < QB; ++t)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * t), (__attribute__((address_space(3))) void*)(dst + 256 * t), 4, 0, 0);
            }
        }
    };
    auto advance = [&]() {   // past the end the last real tile is issued again (uniform DMA counts; its stage is never read)
        is_stage = is_stage + 1 == NS ? 0 : is_stage + 1;
        if (++is_count < n_ord) {
            ++is_ord;
            enter_tile();
        }
    };
    enter_tile();
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) {
        issue_pieces();
        advance();
    }
    // the query fragments and constants are in registers; this wave's (NS - 1) stages of ring pieces stay in flight
    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 1) * (PPW + NXW)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 1) * PPW) : "memory");

    // ---- fragment reads: row r of slab sl of a stage lives at sl * RT * 128 + r * 128, logical 16-B slot c at physical c ^ ((r >> 1) & 7)
    // lane (l32, hi) of the A fragment (sl, kk, rb): row wr * 64 + rb * 32 + l32, slot kk * 2 + hi.  Reads and waits are issued by hand
    // (hipcc answers every ds_read it can see with s_waitcnt vmcnt(0) while LDS-DMA is in flight); LDS reads return in order.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = lds0 + (uint32_t)(wr * 64 + l32) * 128u + (uint32_t)((hi ^ swz) * 16);
    qs_i32x4 af[NBUF];
    uint32_t ad_cur[4];
    auto read_frag = [&](qs_i32x4& dst, const uint32_t (&ad)[4], auto idxc) {   // read idx of a step: (sl, kk, rb) = (idx / 8, (idx / 2) % 4, idx % 2)
        constexpr int idx = decltype(idxc)::value;
        constexpr int sl = idx / 8, kk = (idx / 2) % 4, rb = idx % 2;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad[kk]), "n"(sl * (RT * 128) + rb * (32 * 128)));
    };
    f32x16 acc[2][QB];   // [row block][query block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    uint32_t e_cnt = 0;   // (uniform) groups staged in this wave's LDS region
    uint32_t c_stage = NS - 1;    // (incremented before use: the first step computes stage 0)
    typedef _Float16 qh_h8 __attribute__((ext_vector_type(8)));
    auto mfma_step = [&]() {
        ly_static_for<NBUF - 1>([&](auto ic) { read_frag(af[decltype(ic)::value], ad_cur, ic); });
        __builtin_amdgcn_sched_barrier(0);
        ly_static_for<NR>([&](auto ic) {
            constexpr int idx = decltype(ic)::value;
            constexpr int nxt = idx + NBUF - 1;
            if constexpr (nxt < NR) read_frag(af[nxt % NBUF], ad_cur, std::integral_constant<int, nxt>{});
            constexpr int outstanding = nxt < NR ? NBUF : NR - idx;   // reads for MFMAs idx .. min(nxt, last): the oldest is this one's
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[idx % NBUF]) : "n"(outstanding - 1));
            __builtin_amdgcn_sched_barrier(0);
            constexpr int sl = idx / 8, kk = (idx / 2) % 4, rb = idx % 2;
            constexpr int ks = sl * 4 + kk;
            const qh_h8 av = __builtin_bit_cast(qh_h8, af[idx % NBUF]);
            ly_static_for<QB>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (ks == 0) {   // first k-step of a tile: C = 0 (no accumulator clears)
                    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(qh_h8, bq[j][ks]), z, 0, 0, 0);
                } else {
                    acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(qh_h8, bq[j][ks]), acc[rb][j], 0, 0, 0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto wait_and_barrier = [&]() {
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (PPW + NXW)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        __builtin_amdgcn_s_barrier();
        c_stage = c_stage + 1 == NS ? 0 : c_stage + 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ad_cur[kk] = (a_lane ^ (uint32_t)(kk * 32)) + c_stage * SB;
    };

    const uint32_t e_base = lds0 + (uint32_t)STG_OFF + (uint32_t)wave * (uint32_t)(EW * 24);   // accumulators [EW] 16 B, then (first row, query) [EW] 8 B
    if (tid < 256) asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + (uint32_t)QCNT_OFF + (uint32_t)tid * 4u), "v"(0u) : "memory");   // (the first barrier of the main loop orders it before any flush)
    // Every load of the flush is inline assembly that waits for itself: a load the compiler can see leaves "a register may still be
    // written by a memory operation" on the rare path, and where that path joins the main loop again the compiler answers with
    // s_waitcnt vmcnt(0) in front of the next reuse of such a register — behind the ring pieces of EVERY step (measured: the DMA-issue
    // phase 7k -> 30k ticks per stage, the ring drained once per step).
    auto flush_groups = [&]() __attribute__((always_inline)) {   // (between tiles once the region is half full, and once at the end of the launch; its waits drain the ring)
        const uint32_t qcnt0 = lds0 + (uint32_t)QCNT_OFF;
        for (uint32_t e = (uint32_t)lane; e < e_cnt; e += 64u) {
            f32x4 x;
            uint64_t mq;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x), "=&v"(mq) : "v"(e_base + e * 16u), "v"(e_base + (uint32_t)(EW * 16) + e * 8u) : "memory");
            const uint32_t m0 = (uint32_t)mq, q = (uint32_t)(mq >> 32);
            float qi, thr, ex = 0.0f;
            f32x4 nv = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (MET == M_IP) {
                asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off\n\ts_waitcnt vmcnt(0)" : "=&v"(qi), "=&v"(thr) : "v"(a.qinv + q), "v"(a.thr + q) : "memory");
            } else {   // (m0 is a multiple of 4: one 16-B load of the four rows' norms; rows past the end: the arrays' slack)
                const float* exs = MET == M_L2 ? a.qn2 : a.qrinv;
                const float* nrm = MET == M_L2 ? a.vn2 : a.vrinv;
                asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %5, off\n\tglobal_load_dword %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(qi), "=&v"(thr), "=&v"(ex), "=&v"(nv) : "v"(a.qinv + q), "v"(a.thr + q), "v"(exs + q), "v"(nrm + m0) : "memory");
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float sc = x[t] * qi;   // the exact coarse expression of k_scan_h16 (score(): separate mul / add)
                if constexpr (MET == M_L2) sc = nv[t] - 2.0f * sc + ex;
                if constexpr (MET == M_COS) sc = 1.0f - sc * nv[t] * ex;
                const uint32_t m = m0 + (uint32_t)t;
                if (m < a.row1 && (ASC ? (sc <= thr) : (sc >= thr))) {
                    const uint64_t key = make_key(sc, m, ASC);
                    uint32_t slot;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(slot) : "v"(qcnt0 + q * 4u), "v"(1u) : "memory");
                    if (slot < a.seg) {
                        a.candB[((size_t)q * a.nseg + blockIdx.x) * a.seg + slot] = key;
                    } else {   // a full segment (massive ties): the shared region
                        uint32_t gs;
                        asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(gs) : "v"(a.count + q), "v"(1u) : "memory");
                        if (gs < a.cap) a.cand[(size_t)q * a.cap + gs] = key;
                    }
                }
            }
        }
        e_cnt = 0;
    };
    // ---- tile epilogue: this lane's 2 x 16 values per query block belong to ONE query each; rows wr * 64 + rb * 32 + (r & 3) + 8 (r >> 2) + 4 hi
    [[maybe_unused]] uint32_t smp1 = 0xffffffffu, smp2 = 0xffffffffu;   // SMP: the two smallest packed key words this lane has seen (smp1 <= smp2)
    auto epilogue = [&](uint32_t e_tile, [[maybe_unused]] uint32_t e_ord_) {
        if constexpr (SMP != 0) {
            static_assert(QB == 1, "sample stage: one query block per wave");
            const uint32_t nb = lds0 + NRM_OFF + (e_ord_ % NRM_SLOTS) * NRM + (uint32_t)hi * 16u;
            const uint32_t rb0 = tile_row0(e_tile) + 4u * (uint32_t)hi;
            const bool whole = tile_row0(e_tile) + (uint32_t)RT <= a.row1;   // (uniform) every row of the tile exists: all but the shard's last tile
            const uint32_t pos0 = (e_ord_ & 31u) << 5;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 nv[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
                if constexpr (NORMS) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(nv[g]) : "v"(nb + (uint32_t)(i * 32 + 8 * g) * 4u));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nv[0]), "+v"(nv[1]), "+v"(nv[2]), "+v"(nv[3]));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sc = acc[i][0][r] * c_qinv[0];   // the exact coarse expression of k_scan_h16 (score())
                    if constexpr (MET == M_L2) sc = nv[r >> 2][r & 3] - 2.0f * sc + c_extra[0];
                    if constexpr (MET == M_COS) sc = 1.0f - sc * nv[r >> 2][r & 3] * c_extra[0];
                    uint32_t o = (uint32_t)(make_key(sc, 0u, ASC) >> 32);
                    o = o > 0xfffffbffu ? 0xfffffc00u : ((o + 1023u) & ~1023u);   // rounded up to a multiple of 1024 (saturating)
                    uint32_t t = o | pos0 | (uint32_t)(i * 16 + r);
                    if (!whole) t = (rb0 + (uint32_t)(i * 32 + (r & 3) + 8 * (r >> 2))) < a.row1 ? t : 0xffffffffu;
                    const uint32_t hi3 = smp1 > t ? smp1 : t;
                    smp1 = smp1 < t ? smp1 : t;
                    smp2 = smp2 < hi3 ? smp2 : hi3;
                }
            }
            return;
        }
        const uint32_t nbase = lds0 + NRM_OFF + (e_ord_ % NRM_SLOTS) * NRM + (uint32_t)(wr * 64) * 4u + (uint32_t)hi * 16u;
        // the norms of the wave's 16 rows of row block i (broadcast reads: all lanes of a wave half hold the same rows)
        auto read_norms = [&](f32x4 (&nv)[4], int i) {
            if constexpr (NORMS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(nv[g]) : "v"(nbase + (uint32_t)(i * 32 + 8 * g) * 4u));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nv[0]), "+v"(nv[1]), "+v"(nv[2]), "+v"(nv[3]));
            }
        };
        // level 1: the maximum per group of four rows of 2 q.v - |v|^2 (L2) / q.v / |v| (cosine) / the score itself (IP)
        auto level1 = [&](const f32x4 (&nv)[4], int i, int j, int g) -> float {
            const float mm = MET == M_L2 ? -2.0f * c_qinv[j] : c_qinv[j];
            const qh_f32x2 m2 = {mm, mm};
            const qh_f32x2 x01 = {acc[i][j][4 * g], acc[i][j][4 * g + 1]}, x23 = {acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            qh_f32x2 p01, p23;
            if constexpr (MET == M_L2) {
                // (the NEGATED level-1 value |v|^2 - 2 q.v with its minimum: the norms enter the fma as they are — negating them cost 16 v_xor per tile)
                const qh_f32x2 n01 = {nv[g][0], nv[g][1]}, n23 = {nv[g][2], nv[g][3]};
                p01 = __builtin_elementwise_fma(x01, m2, n01);
                p23 = __builtin_elementwise_fma(x23, m2, n23);
                return -fminf(fminf(p01[0], p01[1]), fminf(p23[0], p23[1]));
            } else if constexpr (MET == M_COS) {
                const qh_f32x2 n01 = {nv[g][0], nv[g][1]}, n23 = {nv[g][2], nv[g][3]};
                p01 = x01 * n01;
                p23 = x23 * n23;
            } else {
                p01 = x01 * m2;
                p23 = x23 * m2;
            }
            return fmaxf(fmaxf(p01[0], p01[1]), fmaxf(p23[0], p23[1]));
        };
        float gm[2][QB][4];
        float best[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) best[j] = -LY_INF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 nv[4];
            read_norms(nv, i);
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float m = level1(nv, i, j, g);
                    gm[i][j][g] = m;
                    best[j] = fmaxf(best[j], m);
                }
        }
        bool any = false;
#pragma unroll
        for (int j = 0; j < QB; ++j) any = any || best[j] >= c_lim[j];
        if (__builtin_expect(__ballot(any) == 0ull, 1)) return;
        const uint32_t rbase = a.row0 + e_tile * RT + (uint32_t)(wr * 64) + 4u * (uint32_t)hi;
        const bool whole = a.row0 + e_tile * RT + (uint32_t)RT <= a.row1;   // (uniform) every row of the tile exists: all but the stage's last tile
        // The region is worked off BETWEEN tiles, where no accumulator is live (main loop: below half full after every tile).  A tile that
        // adds more than the free half in one go (hundreds of rows inside the threshold of one query block in 64 rows: massive ties) marks
        // the queries of the groups that do not fit as overflowed (count > cap: k_select flags them, the batch goes down the plan ladder,
        // whose levels run k_scan_h16).  A flush inside this walk would sit between live accumulators: its spill reloads are scratch loads,
        // and where that rare path joins the loop again the compiler waits for them with s_waitcnt vmcnt(0) — behind the ring pieces of
        // EVERY step (measured: the DMA-issue phase 7k -> 30k ticks per stage).
#pragma unroll
        for (int j = 0; j < QB; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // (rows past the end of the stage re-read its last row and their norm slots hold the arrays' slack: a group that lies
                    // wholly behind row1 is no hit whatever its values; a group that straddles it is filtered per row by the flush)
                    const bool hit = gm[i][j][g] >= c_lim[j] && (whole || rbase + (uint32_t)(i * 32 + 8 * g) < a.row1);
                    const uint64_t pm = __ballot(hit);
                    if (__builtin_expect(pm == 0ull, 1)) continue;   // (uniform)
                    // stage the group of the lanes that passed level 1: rows rbase + i 32 + 8 g + {0..3}, this lane's query
                    const uint32_t np = (uint32_t)__popcll(pm);
                    if (__builtin_expect(e_cnt + np > (uint32_t)EW, 0)) {
                        if (hit) __hip_atomic_fetch_add(&a.count[qn[j]], a.cap + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (no return value: nothing to wait for)
                        continue;
                    }
                    const uint32_t slot = e_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    if (hit) {
                        const f32x4 x = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        const uint32_t m0 = rbase + (uint32_t)(i * 32 + 8 * g);
                        asm volatile("ds_write_b128 %0, %1\n\tds_write_b64 %2, %3" ::"v"(e_base + slot * 16u), "v"(x), "v"(e_base + (uint32_t)(EW * 16) + slot * 8u),
                                     "v"((uint64_t)m0 | ((uint64_t)qn[j] << 32)) : "memory");
                    }
                    e_cnt += np;
                }
            }
        }
    };

    // ---- main loop (ping-pong roles, scan_qs.h): the early waves (0-3: one per SIMD) run
    // barrier -> MFMAs -> epilogue -> DMA issue, the late waves (4-7) barrier -> epilogue of the PREVIOUS tile -> DMA issue -> MFMAs.
    // ONE loop over half steps with ONE copy of the MFMA step and ONE of the epilogue (two inlined copies of either made the compiler
    // keep two sets of accumulators and copy 64 registers per step): an even half step opens with the barrier; the early waves compute
    // in the even half steps and finish their tile in the odd ones, the late waves the other way round.
    bool have = false;   // a computed tile is waiting for its epilogue
    uint32_t e_ord = 0;
    // debug_flags & 64: s_memtime sums per wave (a.dbg[(block * 8 + wave) * 4 ..]: DMA wait + barrier, MFMA step, epilogue, DMA issue)
    const bool timing = (a.debug_flags & 64) != 0 && a.dbg != nullptr;
    unsigned long long t_ph[4] = {0ull, 0ull, 0ull, 0ull}, tp = timing ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int b) {
        if (timing) {
            asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[1][QB - 1][15]));
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_ph[b] += t - tp;
            tp = t;
        }
    };
    for (uint32_t ph = 0; ph <= 2u * n_ord; ++ph) {
        const uint32_t g = ph >> 1;
        if (!(ph & 1u) && g < n_ord) { wait_and_barrier(); stamp(0); }
        if (((ph & 1u) != 0u) == late) {
            if (g < n_ord) {
                if (wave_live) mfma_step();
                have = true;
                stamp(1);
            }
        } else {
            if (have) {
                if (wave_live) epilogue(tile_of(e_ord), e_ord);
                ++e_ord;
                have = false;
                if constexpr (SMP == 0) { if (__builtin_expect(e_cnt > (uint32_t)(EW / 2), 0)) flush_groups(); }   // (uniform; no accumulator is live here)
                stamp(2);
            }
            // (the keys of the epilogue above were stored BEFORE these pieces: at the next counted wait they are older than everything
            // that may stay in flight)
            if (g < n_ord) {
                issue_pieces();
                advance();
                stamp(3);
            }
        }
    }
    if (timing && lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = t_ph[0]; o[1] = t_ph[1]; o[2] = t_ph[2]; o[3] = t_ph[3];
    }
    if constexpr (SMP != 0) {
        if (ok[0]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t slot = (blockIdx.x * 2u + (uint32_t)hi) * 2u + (uint32_t)t;
                const uint32_t pk = t == 0 ? smp1 : smp2;
                uint64_t key = KEY_SENTINEL;
                if (pk != 0xffffffffu) {
                    const uint32_t pos = pk & 1023u, r = pos & 15u;
                    const uint32_t m = tile_row0(tile_of(pos >> 5)) + ((pos >> 4) & 1u) * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)hi;
                    key = ((uint64_t)(pk & ~1023u) << 32) | (uint64_t)m;
                }
                if (slot < a.cap) a.cand[(size_t)qn[0] * a.cap + slot] = key;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    flush_groups();
    __syncthreads();   // every wave's groups are worked off: the workgroup's key counts per query are final
    if (a.seg && tid < 256 && (uint32_t)tid < a.nq) {
        uint32_t n;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(n) : "v"(lds0 + (uint32_t)QCNT_OFF + (uint32_t)tid * 4u) : "memory");
        a.segcnt[(size_t)tid * a.nseg + blockIdx.x] = (uint8_t)(n < a.seg ? n : a.seg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace lynse
