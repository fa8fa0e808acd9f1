// scan_qs.h — k_scan_qs: the QUERY-STATIONARY tiling of the certified int8 coarse pass (FLAT-IP / cosine, 129..256 queries).
//
// Replaces, for the threshold stages of a batch, the 256-row x 256-query tile of k_scan_h16<2,4,4,2,IP,…,I8Q=2> (kernels.h).
// Reference work on this path: FlatMmap::search -> exact_flat_search's chunked scan (src/storage/flat_mmap.rs:2179-2256,
// :4845-4982) — every row of the shard scored against every query of the batch.
//
// What bounded the 256 x 256 tile: per slab step a CU moved 32 KB of rows (HBM) AND 32 KB of query image
// (L2) through the LDS-DMA path, and the two rings were coupled by one barrier per step.  Here the query operand never
// crosses that path again:
//   * a workgroup is 8 waves; wave w OWNS queries [32 w, 32 w + 32) for the whole launch and keeps their int8 image — the
//     B operand of v_mfma_i32_32x32x32_i8 for every k-step, NSLAB x 4 fragments x 4 registers (96 for 768 dimensions) —
//     in registers, loaded once from the L2-resident image k_i8c_prep_queries wrote;
//   * the LDS holds ONLY rows: a ring of NS stages, one stage = RB x 32 rows x SL slabs of 128 B, filled by
//     global_load_lds_dwordx4 in full 128-B lines (8 rows x 128 B per instruction, the XOR slot swizzle of k_scan_h16 on the
//     per-lane source address);
//   * every wave reads EVERY row fragment of a stage (one ds_read_b128 per MFMA: 128 B / clk / CU of the 256 the LDS delivers)
//     and multiplies it with its own 32 queries; the accumulators of a tile are RB x 16 registers;
//   * a lane's 16 x RB accumulators all belong to ONE query (column lane % 32 of the wave's block), so the epilogue is a
//     v_max chain against ONE register-resident integer threshold (the integer image of the certified threshold, DESIGN.md
//     §3b), one ballot, and a rare grouped path that appends (score, row) keys to the lane's private segment;
//   * XPF: the first fragments of step g + 1 are read before barrier g + 1 (the wait in front of barrier g covers the data of
//     step g + 1 too), so no wave opens a step waiting for its first LDS round trip.
// Keys, segments and counts are the ones k_select gathers (ScanArgs::candB / segcnt; two segments per workgroup and query: the
// two wave halves of the owning wave).
//
// STS — SELF-TIGHTENING thresholds: the whole shard in ONE launch, no sample stage, no select between stages.  One query per
// lane makes a running threshold cheap: per query, dyn_ks (= k) maxima of the coarse integer dot product over DISJOINT row
// partitions (partition of a row = its tile index % dyn_ks) live in global memory; their minimum tau is reached by at least k
// distinct rows, so at ANY moment every row whose dot is below tau - M (M = the certified margin 2E in dot units) is out of the
// top-k — however stale the value a lane holds (tau only grows).  k_i8c_prep_queries seeds the maxima from a few sample rows
// (valid from the first tile on).  During the launch: a lane whose tile maximum beats its tau raises the tile's partition
// maximum (fire-and-forget atomic max); one wave per workgroup re-reads the partition maxima of "its" query every few steps
// and publishes their minimum; every lane re-reads its query's tau every few steps.  Both reads are 4-byte LDS-DMAs into a
// landing area behind the row ring (they ride on the ring's counted waits: no register is written asynchronously, nothing
// drains).  The first dyn_warm tiles of every workgroup only feed the maxima (no emission) and are scanned again at the end.
// The keys of all tiles go to k_select_final exactly as a last threshold stage's would.
#pragma once

namespace lynse {

typedef int qs_i32x4 __attribute__((ext_vector_type(4)));
typedef int qs_i32x16 __attribute__((ext_vector_type(16)));

constexpr int QS_STS_LDS = 8 * 256 + 8 * 256;   // landing areas behind the ring: per wave 64 thresholds + 64 partition maxima

// DBG (timing experiments): 1 no MFMA, 2 no LDS fragment reads, 8 no row DMA, 16 no epilogue, 32 s_memtime phase sums, 64 waves 4-7 do not compute
// MET: 0 = inner product / cosine (integer threshold image), 1 = squared L2 on the PLAIN codes: the int8 dot product feeds a float
// epilogue with the exact f32 row norms (coarse distance |v|^2 - 2 s_q dot + (|q|^2 - 2 B_q), k_i8c_prep_queries l2n; the norms of a
// tile ride in a ring of NS + 1 slots behind the row ring, one small LDS-DMA per step by wave 0).  All lanes of a wave half hold the
// SAME rows, so a lane reads the norms of its 16 RB rows as broadcast ds_read_b128s and scores them against ONE query's constants.
// SMP = 1: the THRESHOLD-ONLY SAMPLE STAGE of a staged plan on this tiling (it was the 256 x 256 tile of k_scan_h16: one tile per CU, 30 us
// of launch ramp, 196-KB query image and ring fill for 65,536 rows).  a.ntiles tiles of 64 rows: tile t = rows (t / 4) * a.tile_stride +
// (t % 4) * 64 ... of the shard (the same sample rows); no threshold, no segments — every lane keeps the best TWO (coarse score, row) of
// all the rows it sees for its query and writes them to cand[query][(blockIdx.x * 2 + hi) * 2 + t]: 4 keys per workgroup and query,
// gridDim.x * 4 per query.  Any k distinct real rows bound the k-th best score from below, so k_select's threshold-only rule turns the
// k-th best of these keys into a valid first threshold exactly as it does with the lane-max keys of the old sample tiles.
// MSK = 1: the MASKED threshold stages of a subset-filtered search (FlatMmap::search_filtered as a row bitmask): a key is
// emitted only for rows whose bit is set in a.mask, and the 256-row tiles of the emit-all sample stage (tile t at t * a.skip_stride,
// t < a.skip_tiles: their rows are candidates already) are computed but emit nothing.  Both checks live in the rare slow path / once per
// tile: the scan itself runs at the speed of the unfiltered one.
// F4 = 1: batched HAMMING as a +-1 GEMM: the operands are FP4 nibbles (+1.0 / -1.0 / 0, exact in E2M1; k_bits_to_fp4 rows, the
// query image of k_bpm_prep_queries) on v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales.  A 128-B line holds 256 elements = four k-steps
// of 64, so rings, fragments and the B-register layout are those of the int8 form; the f32 accumulators hold exact integers and are
// converted where the integer epilogue reads them.  The emit-all sample tiles of the plan (a.skip_stride) are computed but emit nothing.
template <int NSLAB, int RB, int SL, int NS, bool XPF, int NBUF, int DBG = 0, int PING = 0, int STS = 0, int MET = 0, int SMP = 0, int MSK = 0, int F4 = 0>
__global__ void __launch_bounds__(512, 2) k_scan_qs(ScanArgs a) {
    static_assert(NSLAB % SL == 0, "a tile is a whole number of steps");
    constexpr int TS = NSLAB / SL;          // steps per tile
    constexpr int RT = RB * 32;             // rows per tile
    constexpr int SB = SL * RT * 128;       // bytes per ring stage
    constexpr int PP = SB / 1024;           // LDS-DMA instructions per stage (1 KiB each: 8 rows x 128 B)
    static_assert(PP % 8 == 0, "the pieces of a stage split evenly over the 8 waves");
    constexpr int PPW = PP / 8;
    constexpr int NM = SL * 4 * RB;         // MFMAs per wave and step
    static_assert(NM % NBUF == 0 && NBUF >= 2 && NBUF <= NM, "fragment ring");
    static_assert(NS >= (XPF ? 4 : 3), "ring depth");
    constexpr int NRM = MET == 1 ? (RT == 64 ? 256 : 1024) : 0;   // bytes of row norms per tile slot (64 floats by one dword LDS-DMA, else 256 by one dwordx4)
    constexpr int NRM_SLOTS = NS + 1;                        // (a slot is refilled two steps after its tile was computed: late epilogues are safe)
    constexpr int NRM_OFF = NS * SB + (STS ? QS_STS_LDS : 0);
    static_assert(MET == 0 || (STS == 0 && TS == 1 && RT <= 256), "L2: whole-K stages; one dword / dwordx4 LDS-DMA brings a tile's norms");
    static_assert(NRM_OFF + NRM_SLOTS * NRM <= 160 * 1024, "LDS");
    static_assert(PING == 0 || TS == 1, "ping-pong: one step per tile");
    static_assert(STS == 0 || (!XPF && TS == 1), "self-tightening thresholds: whole-K stages, no cross-barrier prefetch");
    static_assert(SMP == 0 || (STS == 0 && MET == 0 && RT == 64 && TS == 1), "sample stage: 64-row tiles of the IP / cosine form");
    static_assert(MSK == 0 || (STS == 0 && MET == 0 && SMP == 0), "masked threshold stages: the IP / cosine form");
    static_assert(F4 == 0 || (STS == 0 && MET == 0 && SMP == 0 && MSK == 0 && DBG == 0), "FP4 Hamming: plain threshold stages");
    constexpr int WAITN = (XPF ? NS - 3 : NS - 2) * PPW;   // DMA instructions that may still be in flight at the barrier
    static_assert(WAITN <= 63, "vmcnt");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hi = lane >> 5;
    const uint32_t ntiles = SMP != 0 ? a.ntiles : (a.row1 - a.row0 + RT - 1) / RT;
    auto tile_row0 = [&](uint32_t t) -> uint32_t {   // first row of tile t
        if constexpr (SMP != 0) return a.row0 + (t >> 2) * a.tile_stride + (t & 3u) * (uint32_t)RT;
        else return a.row0 + t * RT;
    };
    // Tile order.  Staged plans: workgroup b takes tiles b, b + grid, ... (the chip sweeps the stage front to back).  STS: workgroup b
    // owns the CONTIGUOUS tiles [b * pitch, (b + 1) * pitch) — the tiles the chip works on at any moment are spread evenly over the
    // whole shard, so the running thresholds are representative of it whatever the insertion order (a shard sorted by score would
    // otherwise beat its own thresholds with every new tile, the way it overflows the contiguous staged plan); the pitch is coprime
    // to dyn_ks, so the tiles of one round fall into every partition.
    const uint32_t pitch = STS != 0 ? a.dyn_pitch : 0u;
    uint32_t my_tiles;
    if constexpr (STS != 0) {
        if ((uint64_t)blockIdx.x * pitch >= ntiles) return;
        my_tiles = ntiles - blockIdx.x * pitch < pitch ? ntiles - blockIdx.x * pitch : pitch;
    } else {
        if (blockIdx.x >= ntiles) return;
        my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    }
    // STS: ordinals 0 .. warm-1 are scanned twice — first without emission (they only feed the partition maxima), again at the end
    uint32_t warm = 0;
    if constexpr (STS != 0) warm = a.dyn_warm * 4u <= my_tiles ? a.dyn_warm : my_tiles / 4u;
    const uint32_t n_ord = my_tiles + warm;      // tile ordinals of this workgroup: ordinal j is its (j < my_tiles ? j : j - my_tiles)-th tile
    const uint32_t G = n_ord * TS;
    [[maybe_unused]] const unsigned long long t_kernel0 = (a.debug_flags & 64) ? __builtin_amdgcn_s_memtime() : 0ull;
    auto tile_of = [&](uint32_t ord) -> uint32_t {
        const uint32_t i = ord < my_tiles ? ord : ord - my_tiles;
        if constexpr (STS != 0) return blockIdx.x * pitch + i;
        else return blockIdx.x + i * gridDim.x;
    };

    // ---- the wave's query block: B fragments of every k-step, register-resident for the whole launch
    const int swz = (l32 >> 1) & 7;
    qs_i32x4 bq[NSLAB * 4];
    {
        const char* qimg = reinterpret_cast<const char*>(a.Q16) + (size_t)(wave * 32 + l32) * 128;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                bq[s * 4 + kk] = *reinterpret_cast<const qs_i32x4*>(qimg + (size_t)s * a.qpad * 128 + (((kk * 2 + hi) ^ swz) * 16));
    }
    // per-query constants of this lane's query (one query per lane: column lane % 32 of the wave's block)
    const uint32_t qn = wave * 32 + l32;
    const bool q_ok = qn < a.nq;
    // a wave whose 32 queries all lie beyond the batch (65..224 queries) only feeds the ring: no fragment reads, MFMAs or epilogue —
    // the scan is bound by the power the matrix pipe draws, idle MFMAs are not free
    const bool wave_live = (uint32_t)wave * 32u < a.nq;
    const float s_q = q_ok ? a.qinv[qn] : 0.0f, b_q = q_ok ? a.qn2[qn] : 0.0f;
    // (the threshold of this lane's query: loaded HERE, in front of the ring prologue — a load issued behind the LDS-DMA pieces returns
    // behind them, and the bisection below would wait out the first HBM round trip of the ring)
    [[maybe_unused]] const float thr_q = (q_ok && STS == 0 && SMP == 0) ? a.thr[qn] : 0.0f;
    // Round 5: the ring prologue is issued BEFORE anything waits for the fragments and constants above.  It used to follow an
    // s_waitcnt vmcnt(0): fragments + constants (1.8 us, L2), 12 LDS-DMA pieces per wave (1.6 us of issue) and the first HBM round trip
    // (2.4 us) ran one after the other in every launch (s_memtime stamps of the sample stage: 5.8 us before its first MFMA); now the
    // pieces are in flight while the fragments arrive and the integer threshold is bisected.  Loads return in order: the counted wait
    // behind the prologue leaves exactly this wave's pieces outstanding.
    asm volatile("" ::: "memory");   // (the fragment / constant loads stay in front of the pieces)
    // SMP + debug_flags & 64: phase stamps of the sample stage per workgroup (a.dbg[block * 8 ..]: entry, fragments + constants loaded,
    // ring primed [1] / fragments + constants in registers [2] (round 5: the ring is primed first), first stage landed, tiles done, keys written; [6] = s_memrealtime at entry: launch skew across the grid)
    [[maybe_unused]] unsigned long long smp_t[4] = {0ull, 0ull, 0ull, 0ull}, smp_rt = 0ull;
    [[maybe_unused]] const bool smp_stamps = SMP != 0 && (a.debug_flags & 64) != 0;
    if constexpr (SMP != 0) { if (smp_stamps) smp_rt = __builtin_amdgcn_s_memrealtime(); }

    // ---- row stream (LDS-DMA): step gi of this workgroup = tile ordinal gi / TS, slabs [(gi % TS) SL, +SL)
    // piece p = wave * PPW + j of a stage: slab p / (RB 4) of the step, rows (p % (RB 4)) * 8 .. +8 of the tile; lane l brings the 16 B
    // of row (l >> 3), PHYSICAL slot (l & 7) = logical slot (l & 7) ^ ((row >> 1) & 7)
    uint32_t v_off[PPW];
    const char* v_base = nullptr;   // uniform: first row of the tile being issued
    uint32_t is_ord = 0, is_sub = 0, is_stage = 0, is_count = 0;
    auto enter_tile = [&]() {
        const uint32_t rbase = tile_row0(tile_of(is_ord));
        const uint32_t span = a.row1 - 1 - rbase;   // rows past the last one re-read it (masked in the epilogue)
        v_base = reinterpret_cast<const char*>(a.V16) + (size_t)rbase * a.ld16;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = wave * PPW + j;
            uint32_t r = (p % (RB * 4)) * 8 + (lane >> 3);
            const uint32_t col = (uint32_t)(p / (RB * 4)) * 128u + (((lane & 7) ^ ((r >> 1) & 7)) * 16);
            r = r < span ? r : span;
            v_off[j] = r * a.ld16 + col;
        }
    };
    auto issue_piece = [&](int j) {
        if (DBG & 8) return;
        glds16<2>(v_base + (size_t)(is_sub * (SL * 128)) + v_off[j], smem + is_stage * SB + (wave * PPW + j) * 1024);
    };
    [[maybe_unused]] auto issue_norms = [&]() {   // wave 0, once per step: the f32 norms of the tile being issued (rows past the end: the array's slack)
        if constexpr (MET == 1) {
            if (wave == 0) {
                const uint32_t rbase = a.row0 + tile_of(is_ord) * RT;
                char* dst = smem + NRM_OFF + (is_count % NRM_SLOTS) * NRM;
                if constexpr (RT == 64) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.vn2 + rbase + lane), (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
                else glds16<0>(a.vn2 + rbase + lane * 4, dst);
            }
        }
    };
    auto advance = [&]() {   // past the end the last real step is issued again (uniform DMA counts; its stage is never read)
        is_stage = is_stage + 1 == NS ? 0 : is_stage + 1;
        if (++is_count < G) {
            if (++is_sub == TS) {
                is_sub = 0;
                ++is_ord;
                enter_tile();
            }
        }
    };
    enter_tile();
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue_piece(j);
        issue_norms();
        advance();
    }
    if constexpr (SMP != 0) { if (smp_stamps) smp_t[0] = __builtin_amdgcn_s_memtime(); }

    int T = 0x7fffffff;
    // L2: pass iff |v|^2 - 2 s_q dot + c_q <= thr.  Level 1 maximises 2 s_q dot - |v|^2 against pre = (c_q - thr) LOOSENED by more than the
    // rounding differences between that form and the exact expression (k_scan_h16, set_pre); the slow path evaluates the exact expression.
    [[maybe_unused]] float l2_thr = 0.0f, l2_pre = LY_INF, l2_2s = 0.0f;
    if constexpr (MET == 1) {
        l2_thr = q_ok ? thr_q : -LY_INF;
        l2_2s = 2.0f * s_q;
        const float vmax2 = a.vmax2;
        float pre = (b_q - l2_thr) - 2e-6f * (b_q + fabsf(l2_thr) + vmax2);
        if (!(vmax2 > 0.0f) || !(fabsf(pre) < 3.0e38f)) pre = -LY_INF;   // open threshold / overflow: no pre-filter
        l2_pre = q_ok ? pre : LY_INF;
#ifdef LYNSE_EXPERIMENTS
        if (a.debug_flags & 2) l2_pre = LY_INF;
#endif
    }
    if constexpr (STS == 0 && MET == 0 && SMP == 0) {
        // INTEGER image of the threshold (k_scan_h16, load_qc_thr): B_q + s_q * (float)dot is monotone non-decreasing in the
        // integer dot product, so "score >= thr" is exactly "dot >= T", T = the smallest passing dot (bisection over |dot| <= 2^29)
        const float th = thr_q;
        int lo = -(1 << 29), hi_ = 1 << 29;
#pragma unroll 1
        for (int it = 0; it < 31; ++it) {
            const int mid = lo + ((hi_ - lo) >> 1);
            const bool ge = (b_q + s_q * (float)mid) >= th;
            hi_ = ge ? mid : hi_;
            lo = ge ? lo : mid + 1;
        }
        T = q_ok ? lo : 0x7fffffff;
#ifdef LYNSE_EXPERIMENTS
        if (a.debug_flags & 2) T = 0x7fffffff;
#endif
    }
    // STS state: tau_c = the largest tau of its query this lane has seen, T = tau_c - margin; the landing areas in LDS
    [[maybe_unused]] int tau_c = -2147483647 - 1, dyn_m = 0, pub_last = -2147483647 - 1;
    [[maybe_unused]] int* const land_thr = reinterpret_cast<int*>(smem + NS * SB) + wave * 64 + lane;          // tau of query qn
    [[maybe_unused]] int* const land_slot = reinterpret_cast<int*>(smem + NS * SB + 8 * 256) + wave * 64 + lane;  // partition maximum lane % 32 of the helper query
    [[maybe_unused]] const int* thr_src = nullptr;
    [[maybe_unused]] const int* slot_src = nullptr;
    [[maybe_unused]] const uint32_t hq = blockIdx.x + (uint32_t)wave * gridDim.x;   // the query whose minimum this wave republishes (if < nq)
    [[maybe_unused]] const bool helper = STS != 0 && hq < a.nq;
    [[maybe_unused]] uint32_t sts_step = 0;
    [[maybe_unused]] int peek_thr = -2147483647 - 1, peek_slot = -2147483647 - 1;
    [[maybe_unused]] auto sts_T = [&]() -> int {
        if (!q_ok) return 0x7fffffff;
#ifdef LYNSE_EXPERIMENTS
        if (a.debug_flags & 2) return 0x7fffffff;
#endif
        return tau_c < -(1 << 30) ? -(1 << 30) - (1 << 29) : tau_c - dyn_m;   // (|dot| <= 2^29, margin <= 2^30)
    };
    if constexpr (STS != 0) {
        thr_src = a.dyn_thr + (q_ok ? qn : 0u);
        slot_src = a.dyn_slot + (size_t)(helper ? hq : 0u) * 32 + l32;
        dyn_m = q_ok ? a.dyn_marg[qn] : 0;
        tau_c = __hip_atomic_load(thr_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *land_thr = tau_c;
        *land_slot = -2147483647 - 1;   // (nothing is published before a real set of maxima has landed)
        T = sts_T();
    }
    // the query fragments and constants are in registers; this wave's (NS - 1) PPW ring pieces (wave 0, L2 form: + the norm pieces) stay in flight
    if constexpr (MET == 1) {
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 1) * PPW + (NS - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 1) * PPW) : "memory");
    } else if constexpr ((DBG & 8) != 0 || STS != 0) {   // (STS: its loads above were issued behind the pieces)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 1) * PPW) : "memory");
    }
    if constexpr (SMP != 0) { if (smp_stamps) smp_t[1] = __builtin_amdgcn_s_memtime(); }

    // ---- fragment reads: row r of slab sl of a stage lives at sl * RT * 128 + r * 128, logical 16-B slot c at physical c ^ ((r >> 1) & 7)
    // lane (l32, hi) of the A fragment (rb, kk): row rb * 32 + l32, slot kk * 2 + hi  ->  (l32 * 128 + ((hi ^ swz) * 16)) ^ (kk * 32)
    // The reads and their waits are issued by hand: hipcc answers every pending ds_read with s_waitcnt lgkmcnt(0) while LDS-DMA is in
    // flight, which drains the NBUF-deep fragment ring once per ring revolution; LDS reads return in order, so MFMA idx only needs
    // "at most NBUF - 1 reads outstanding" after the read for MFMA idx + NBUF - 1 has been issued.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = lds0 + (uint32_t)l32 * 128u + (uint32_t)((hi ^ swz) * 16);
    qs_i32x4 af[NBUF];
    auto read_frag = [&](qs_i32x4& dst, const uint32_t (&ad)[4], auto idxc) {   // MFMA idx of a step: (sl, kk, rb) = (idx / (4 RB), (idx / RB) % 4, idx % RB)
        constexpr int idx = decltype(idxc)::value;
        constexpr int sl = idx / (4 * RB), kk = (idx / RB) % 4, rb = idx % RB;
#ifdef LYNSE_QS_AF_AGPR   // experiment (round 5): the A fragments land in AGPRs (does the LDS return then stop blocking the matrix pipe?)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(ad[kk]), "n"(sl * (RT * 128) + rb * (32 * 128)));
#else
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad[kk]), "n"(sl * (RT * 128) + rb * (32 * 128)));
#endif
    };
    using acc_t = std::conditional_t<F4 != 0, f32x16, qs_i32x16>;
    acc_t acc[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    [[maybe_unused]] int f4_unit = 0x7f7f7f7f;   // E8M0 scale bytes 127 = 2^0 (ly_mfma_fp4: kept opaque in a VGPR across the loop)
    if constexpr (F4 != 0) asm volatile("" : "+v"(f4_unit));

    uint32_t cnt = 0;            // keys in this lane's private segment
    uint32_t c_ord = 0;          // tile ordinal being computed
    uint32_t c_stage = 0;
    uint32_t ad_cur[4], ad_nxt[4];   // fragment addresses (per kk) in the stage being computed / the next one
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ad_nxt[kk] = a_lane ^ (uint32_t)(kk * 32);
    if constexpr (XPF) {   // the first fragments of step 0 (every later step: read during the step before)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!(DBG & 2)) ly_static_for<NBUF - 1>([&](auto ic) { read_frag(af[decltype(ic)::value], ad_nxt, ic); });
    }
    constexpr bool TIMING = (DBG & 32) != 0;   // s_memtime sums per wave: DMA wait, barrier, MFMA loop, epilogue + DMA issue (a.dbg[512 + (block * 8 + wave) * 4 ..])
    [[maybe_unused]] unsigned long long t_wait = 0, t_bar = 0, t_loop = 0, t_epi = 0, tp = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
    [[maybe_unused]] auto stamp = [&](unsigned long long& bucket) {
        if constexpr (TIMING) { const unsigned long long t = __builtin_amdgcn_s_memtime(); bucket += t - tp; tp = t; }
    };
    // the MFMAs of one step (s = the step's position in its tile); DMA_IN: the refill pieces are issued between them
    auto mfma_step = [&](auto sc, auto dma_in) {
        constexpr int s = decltype(sc)::value;
        constexpr bool DMA_IN = decltype(dma_in)::value;
        if constexpr (!XPF) {
            if (!(DBG & 2)) ly_static_for<NBUF - 1>([&](auto ic) { read_frag(af[decltype(ic)::value], ad_cur, ic); });
        }
        __builtin_amdgcn_sched_barrier(0);
        ly_static_for<NM>([&](auto ic) {
            constexpr int idx = decltype(ic)::value;
            constexpr int nxt = idx + NBUF - 1;
            if (!(DBG & 2)) {
                if constexpr (nxt < NM) read_frag(af[nxt % NBUF], ad_cur, std::integral_constant<int, nxt>{});
                else if constexpr (XPF) read_frag(af[nxt % NBUF], ad_nxt, std::integral_constant<int, nxt - NM>{});
                // outstanding reads now: those for MFMAs idx .. min(nxt, last): the oldest one is this MFMA's
                constexpr int outstanding = (nxt < NM || XPF) ? NBUF : NM - idx;
#ifdef LYNSE_QS_AF_AGPR
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+a"(af[idx % NBUF]) : "n"(outstanding - 1));
#else
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[idx % NBUF]) : "n"(outstanding - 1));
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int sl = idx / (4 * RB), kk = (idx / RB) % 4, rb = idx % RB;
            constexpr int ks = (s * SL + sl) * 4 + kk;
            if constexpr ((DBG & 1) != 0) {
                asm volatile("" ::"v"(af[idx % NBUF]), "v"(bq[ks]));
            } else if constexpr ((DBG & 128) != 0) {
                // energy experiment (round 5): the same MACs as TWO v_mfma_i32_16x16x64_i8 (16K MACs each from the same 16 B per lane and
                // operand) — timing / power only, the accumulators do not hold the scan's dot products
                qs_i32x4 c0 = __builtin_shufflevector(acc[rb], acc[rb], 0, 1, 2, 3), c1 = __builtin_shufflevector(acc[rb], acc[rb], 4, 5, 6, 7);
                c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[idx % NBUF], bq[ks], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[idx % NBUF], bq[ks], c1, 0, 0, 0);
                acc[rb][0] = c0[0]; acc[rb][1] = c0[1]; acc[rb][2] = c0[2]; acc[rb][3] = c0[3];
                acc[rb][4] = c1[0]; acc[rb][5] = c1[1]; acc[rb][6] = c1[2]; acc[rb][7] = c1[3];
            } else if constexpr (F4 != 0) {
                if constexpr (ks == 0) {
                    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    acc[rb] = ly_mfma_fp4(af[idx % NBUF], bq[ks], z, f4_unit);
                } else {
                    acc[rb] = ly_mfma_fp4(af[idx % NBUF], bq[ks], acc[rb], f4_unit);
                }
            } else if constexpr (ks == 0) {   // first k-step of a tile: C = 0 (no accumulator clears)
                const qs_i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc[rb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[idx % NBUF], bq[ks], z, 0, 0, 0);
            } else {
                acc[rb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[idx % NBUF], bq[ks], acc[rb], 0, 0, 0);
            }
            if constexpr (DMA_IN) {   // the refill of the stage computed last (step g + NS - 1), spread behind the MFMAs
                ly_static_for<PPW>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (idx == (NM / PPW) * j + 1) issue_piece(j);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    [[maybe_unused]] uint32_t sts_extra = 0;   // 4-byte LDS-DMAs this wave issued behind the previous step's ring pieces (they may stay in flight too)
    auto wait_and_barrier = [&]() {
        if constexpr (MET == 1) {   // wave 0 carries one more LDS-DMA per step (the norms)
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN + (XPF ? NS - 3 : NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        } else if constexpr (STS != 0) {
            if (sts_extra == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
            else if (sts_extra == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN + 2) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        }
        stamp(t_wait);
        __builtin_amdgcn_s_barrier();
        stamp(t_bar);
        if constexpr (STS != 0) {
            // peek at the landing areas: issued in front of the step's fragment reads (LDS reads return in order, so the first counted
            // lgkmcnt of the MFMA loop covers them) — waiting for them behind the MFMAs cost the late waves ~150 exposed cycles per step
            asm volatile("ds_read_b32 %0, %1" : "=v"(peek_thr) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const int*)land_thr) : "memory");
            if (helper) asm volatile("ds_read_b32 %0, %1" : "=v"(peek_slot) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const int*)land_slot) : "memory");
        }
        c_stage = c_stage + 1 == NS ? 0 : c_stage + 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            ad_cur[kk] = ad_nxt[kk];
            ad_nxt[kk] = (a_lane ^ (uint32_t)(kk * 32)) + c_stage * SB;
        }
    };
    // STS: re-read tau (every lane) and the helper query's partition maxima — behind the step's ring pieces, every step at first
    [[maybe_unused]] auto sts_refresh = [&]() {
        const uint32_t s = sts_step++;
        sts_extra = 0;
#ifdef LYNSE_EXPERIMENTS
        if ((a.debug_flags & 256) && s >= 16u) return;   // (timing experiment: thresholds frozen after the first steps)
#endif
        if (s < 16u || (s & 3u) == 0u) {
            sts_extra = helper ? 2u : 1u;
            constexpr int AUX = STS == 1 ? 17 : (STS == 2 ? 16 : 0);   // sc0 sc1 (system scope: the only policy that is coherent across XCDs) / experiments: sc1, plain
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)thr_src,
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<int*>(smem + NS * SB) + wave * 64), 4, 0, AUX);
            if (helper)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)slot_src,
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<int*>(smem + NS * SB + 8 * 256) + wave * 64), 4, 0, AUX);
        }
    };
    // STS, behind a step's MFMAs (NOT in front of the barrier: every cycle there is a cycle of the whole workgroup): pick up whatever
    // has landed (a 4-byte LDS-DMA needs no wait to be READ: an early read sees the previous value, and any value tau ever held is a
    // valid threshold); the helper wave republishes the minimum of its query's partition maxima
    [[maybe_unused]] auto sts_pickup = [&]() {
        // (the peeks were issued by hand in front of the MFMA loop: a volatile load through the generic pointer becomes a FLAT load
        // behind s_waitcnt vmcnt(0) — the ring drained once per step, +9 % — and a plain LDS load the compiler can see is ordered behind
        // every LDS-DMA in flight the same way)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(peek_thr), "+v"(peek_slot)::"memory");
        const int seen = peek_thr;
        tau_c = seen > tau_c ? seen : tau_c;
        T = sts_T();
        if (helper && (sts_step < 20u || (sts_step & 3u) == 3u)) {   // (uniform per wave)
            int m = peek_slot;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { const int t2 = __shfl_xor(m, o, 64); m = t2 < m ? t2 : m; }
            if (lane == 0 && m > pub_last) { atomicMax(a.dyn_thr + hq, m); pub_last = m; }
        }
    };
    // SMP: the best two rows this lane has seen for its query, as PACKED integers (dot & ~1023) | (tile ordinal of the workgroup << 5 |
    // accumulator slot): smp1 >= smp2.  Round 5: the sample stage spent 40k of its 54k cycles in this epilogue (s_memtime stamps: 10k
    // per 64-row tile against 3.6k for a threshold-stage tile) — two sorted (float score, row) pairs per lane cost ~28 VALU operations
    // per accumulator; the packed form costs three (v_and_or, v_med3, v_max).  The low 10 bits of the dot product make room for the
    // position: the key's score is the coarse score of (dot rounded DOWN to a multiple of 1024) <= the row's true coarse score (B_q + s_q x
    // is monotone in x), so "k distinct real rows score at least tau" still holds for the k-th best key — a threshold looser by at most
    // 1023 s_q (0.7 % of one standard deviation of the scores at 768 dimensions), never an invalid one.  Needs <= 32 tiles per workgroup
    // (host: launch_scan_qs_sample's caller).
    [[maybe_unused]] int smp1 = -2147483647 - 1, smp2 = -2147483647 - 1;
    // ---- tile epilogue: this lane's RB x 16 dot products all belong to query qn
    auto epilogue = [&](uint32_t e_tile, bool emit, [[maybe_unused]] uint32_t e_ord_) {
        if constexpr (SMP != 0) {
            const uint32_t rbase = tile_row0(e_tile);
            const int pos0 = (int)((e_ord_ & 31u) << 5);
            const bool whole = rbase + RT <= a.row1;   // (uniform) every row of the tile exists: all but the shard's last tile
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int t = (acc[i][r] & ~1023) | (pos0 | (i * 16 + r));
                    if (!whole) {
                        const uint32_t m = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        t = m < a.row1 ? t : -2147483647 - 1;
                    }
                    // (smp1 >= smp2) the new second-best is the median of the three, the new best their maximum
                    const int lo3 = smp1 < t ? smp1 : t, hi3 = smp1 < t ? t : smp1;
                    smp2 = smp2 > lo3 ? smp2 : lo3;
                    smp1 = hi3;
                }
        } else if constexpr ((DBG & 16) != 0) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][r]));
        } else if constexpr (MET == 1) {
            // ---- squared L2: level 1 = max over the tile of 2 s_q dot - |v|^2 per group of four rows (their norms: one broadcast
            // ds_read_b128 per group, issued by hand — a load the compiler can see is ordered behind every LDS-DMA in flight with
            // s_waitcnt vmcnt(0)), then ONE ballot; groups with a candidate run the exact expression
            const uint32_t nbase = lds0 + NRM_OFF + (e_ord_ % NRM_SLOTS) * NRM + (uint32_t)hi * 16u;
            float gmf[RB][4];
            f32x4 nvb[2][4];   // double buffer: the norms of row block i + 1 are in flight while block i is scored
            // (plain unrolled loops: an asm operand cannot name a variable captured by a nested lambda)
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(nvb[0][g]) : "v"(nbase + (uint32_t)(8 * g) * 4u));
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (i + 1 < RB) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(nvb[(i + 1) & 1][g]) : "v"(nbase + (uint32_t)((i + 1) * 32 + 8 * g) * 4u));
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(nvb[i & 1][0]), "+v"(nvb[i & 1][1]), "+v"(nvb[i & 1][2]), "+v"(nvb[i & 1][3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nvb[i & 1][0]), "+v"(nvb[i & 1][1]), "+v"(nvb[i & 1][2]), "+v"(nvb[i & 1][3]));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float p[4];
#pragma unroll
                    // (round 5, measured: the packed form of this line — v_pk_fma_f32 on pairs, negated form — is SLOWER here: 10M x 768 L2 2.27-2.28
                    // against 2.10-2.11 ms, same box, alternating: 256 registers are in use and the pairs cost moves)
                    for (int e = 0; e < 4; ++e) p[e] = __fmaf_rn((float)acc[i][4 * g + e], l2_2s, -nvb[i & 1][g][e]);
                    gmf[i][g] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
                }
            }
            float best = gmf[0][0];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) best = fmaxf(best, gmf[i][g]);
            if (__builtin_expect(__ballot(best >= l2_pre) != 0ull, 0)) {
                const uint32_t rbase = a.row0 + e_tile * RT;
                uint64_t* segdst = a.candB + ((size_t)qn * a.nseg + (blockIdx.x * 2 + hi)) * a.seg;
#pragma unroll
                for (int i = 0; i < RB; ++i) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (__builtin_expect(__ballot(gmf[i][g] >= l2_pre) == 0ull, 1)) continue;
                        f32x4 nv;   // (the group's norms again: the fast path keeps only two row blocks of them)
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nv) : "v"(nbase + (uint32_t)(i * 32 + 8 * g) * 4u));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * g + e;
                            float sc = (float)acc[i][r] * s_q;            // the exact coarse expression of k_scan_h16<.., I8Q = 4>: separate mul / add
                            sc = nv[e] - 2.0f * sc + b_q;
                            const uint32_t m = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            if (q_ok && sc <= l2_thr && m < a.row1) {
                                const uint64_t key = make_key(sc, m, true);
                                if (cnt < a.seg) {
                                    segdst[cnt] = key;
                                    ++cnt;
                                } else {
                                    const uint32_t slot = atomicAdd(&a.count[qn], 1u);
                                    if (slot < a.cap) a.cand[(size_t)qn * a.cap + slot] = key;
                                }
                            }
                        }
                    }
                }
            }
        } else {
            // (F4: the f32 accumulators hold exact integers)
            int av[RB][16];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) av[i][r] = (int)acc[i][r];
            // maxima of the groups of four accumulators first (the slow path re-uses them), then their maximum
            int gm[RB][4];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m01 = av[i][4 * g] > av[i][4 * g + 1] ? av[i][4 * g] : av[i][4 * g + 1];
                    const int m23 = av[i][4 * g + 2] > av[i][4 * g + 3] ? av[i][4 * g + 2] : av[i][4 * g + 3];
                    gm[i][g] = m01 > m23 ? m01 : m23;
                }
            int mx = gm[0][0];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) mx = mx > gm[i][g] ? mx : gm[i][g];
            if (__builtin_expect(__ballot(mx >= T) != 0ull, 0)) {
                if constexpr (STS != 0) {
                    // a new maximum candidate of this tile's partition (~k ln(rows) times per query and launch): fire and forget
                    if (q_ok && mx > tau_c) atomicMax(a.dyn_slot + (size_t)qn * 32 + (e_tile % a.dyn_ks), mx);
                    if (!emit) return;   // (uniform) a warm-up tile: scanned again at the end of the launch
                }
                const uint32_t rbase = a.row0 + e_tile * RT;
                if constexpr (MSK != 0 || F4 != 0) {   // (uniform) a tile of the emit-all sample stage: its rows are candidates already
                    if (a.skip_stride && rbase % a.skip_stride < 256u && rbase / a.skip_stride < a.skip_tiles) return;
                }
                uint64_t* segdst = a.candB + ((size_t)qn * a.nseg + (blockIdx.x * 2 + hi)) * a.seg;
#pragma unroll
                for (int i = 0; i < RB; ++i) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // one wave-level branch per group of four (a branch per element cost more than everything it guards)
                        if (__builtin_expect(__ballot(gm[i][g] >= T) == 0ull, 1)) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * g + e;
                            const int v = av[i][r];
                            if (v >= T) {
                                const uint32_t m = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                                if (m < a.row1 && (MSK == 0 || ((a.mask[m >> 5] >> (m & 31)) & 1u))) {   // (the mask word is read only for rows that beat the threshold)
                                    const uint64_t key = make_key(b_q + s_q * (float)v, m, false);
                                    if (cnt < a.seg) {
                                        segdst[cnt] = key;
                                        ++cnt;
                                    } else {
                                        const uint32_t slot = atomicAdd(&a.count[qn], 1u);
                                        if (slot < a.cap) a.cand[(size_t)qn * a.cap + slot] = key;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    };
    if constexpr (PING == 0) {
        for (; c_ord < n_ord; ++c_ord) {
            ly_static_for<TS>([&](auto sc) {
                stamp(t_epi);
                wait_and_barrier();
                if constexpr (SMP != 0) { if (smp_stamps && c_ord == 0) smp_t[2] = __builtin_amdgcn_s_memtime(); }
                if (((DBG & 64) && wave >= 4) || !wave_live) {   // (experiment: one computing wave per SIMD) / no queries: only feed the ring
#pragma unroll
                    for (int j = 0; j < PPW; ++j) issue_piece(j);
                } else {
                    mfma_step(sc, std::true_type{});
                }
                if constexpr (STS != 0) sts_refresh();
                issue_norms();
                advance();
                if constexpr (TIMING) asm volatile("" ::"v"(acc[0][0]));
                stamp(t_loop);
            });
            if constexpr (STS != 0) sts_pickup();
            if (wave_live) epilogue(tile_of(c_ord), c_ord >= warm, c_ord);
        }
    } else {
        // PING-PONG (PING): the two waves of a SIMD share one matrix pipe, and with equal priority the older wave wins every slot — waves
        // 0-3 finished a step's MFMAs first and then waited at the barrier (s_memtime: 45 % of their time) while waves 4-7 ran their
        // MFMAs, their DMA issue and their tile epilogue one after the other.  Here the roles are explicit: the EARLY waves (0-3: one
        // per SIMD) run  barrier -> MFMAs -> epilogue -> DMA issue,  the LATE waves (4-7)  barrier -> epilogue of the PREVIOUS tile ->
        // DMA issue -> MFMAs:  each half's epilogue / issue work sits beside the other half's MFMAs.  (PING == 2: s_setprio 1 on the
        // early waves' MFMAs.)  Measured (scripts/qs_microbench.hip): within 1 % of the plain order either way — the scan is bound by
        // the power the MFMAs, the LDS reads and the HBM stream draw together, not by how the waves take turns.
        const bool late = wave >= 4;
        bool pend = false;   // late waves: the tile computed in the previous step still has to go through the epilogue
        uint32_t e_ord = 0;
        for (uint32_t g = 0; g <= n_ord; ++g) {
            if (g < n_ord) {
                stamp(t_epi);
                wait_and_barrier();
                if (!late) {
                    if constexpr (PING == 2) __builtin_amdgcn_s_setprio(1);
                    if (wave_live) mfma_step(std::integral_constant<int, 0>{}, std::false_type{});
                    if constexpr (PING == 2) __builtin_amdgcn_s_setprio(0);
                    if constexpr (TIMING) asm volatile("" ::"v"(acc[0][0]));
                    stamp(t_loop);
                    pend = true;
                }
            }
            if (pend) {
                if constexpr (STS != 0) sts_pickup();
                if (wave_live) epilogue(tile_of(e_ord), e_ord >= warm, e_ord);
                ++e_ord;
                pend = false;
            }
            if (g < n_ord) {
                // (the keys of the epilogue above were stored BEFORE these pieces: at the next counted wait they are older than
                // everything that may stay in flight.  Stores issued behind a step's pieces made that wait hold back a ring piece each.)
#pragma unroll
                for (int j = 0; j < PPW; ++j) issue_piece(j);
                if constexpr (STS != 0) sts_refresh();
                issue_norms();
                advance();
                if (late) {
                    stamp(t_epi);
                    if (wave_live) mfma_step(std::integral_constant<int, 0>{}, std::false_type{});
                    if constexpr (TIMING) asm volatile("" ::"v"(acc[0][0]));
                    stamp(t_loop);
                    pend = true;
                }
            }
        }
    }
    if constexpr (SMP != 0) {
        if (smp_stamps) smp_t[3] = __builtin_amdgcn_s_memtime();
        if (q_ok) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t slot = (blockIdx.x * 2 + hi) * 2 + t;
                const int pk = t == 0 ? smp1 : smp2;
                uint64_t key = KEY_SENTINEL;
                if (pk != -2147483647 - 1) {
                    const uint32_t pos = (uint32_t)pk & 1023u, r = pos & 15u;
                    const uint32_t m = tile_row0(tile_of(pos >> 5)) + ((pos >> 4) & 1u) * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)hi;
                    key = make_key(b_q + s_q * (float)(pk & ~1023), m, false);
                }
                if (slot < a.cap) a.cand[(size_t)qn * a.cap + slot] = key;
            }
        }
    }
    if (a.seg && q_ok) a.segcnt[(size_t)qn * a.nseg + (blockIdx.x * 2 + hi)] = (uint8_t)cnt;
    if constexpr (TIMING) {
        if (a.dbg && lane == 0 && blockIdx.x < 64) {
            unsigned long long* o = a.dbg + 2 * 256 + ((size_t)blockIdx.x * 8 + wave) * 4;
            o[0] = t_wait; o[1] = t_bar; o[2] = t_loop; o[3] = t_epi;
        }
    }
    if constexpr (SMP != 0) {
        if (smp_stamps && a.dbg && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long* o = a.dbg + (size_t)blockIdx.x * 8;
            o[0] = t_kernel0; o[1] = smp_t[0]; o[2] = smp_t[1]; o[3] = smp_t[2]; o[4] = smp_t[3]; o[5] = __builtin_amdgcn_s_memtime(); o[6] = smp_rt;
        }
    } else
    if ((a.debug_flags & 64) && a.dbg && tid == 0) {   // shader cycles of this workgroup (s_memtime ticks / wall time = the clock held)
        a.dbg[blockIdx.x * 2] = t_kernel0;
        a.dbg[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace lynse
