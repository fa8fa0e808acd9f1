// scan_qs2.h — k_scan_qs2: the query-stationary int8 scan (scan_qs.h) with ONE wave per SIMD and TWO 32-query blocks per wave.
//
// Round-5 energy experiment.  k_scan_qs runs 8 waves x 32 queries: every wave reads EVERY row fragment of a
// stage from LDS (one ds_read_b128 per MFMA), i.e. each 16-B row fragment crosses the LDS port eight times.  The stand-alone
// decomposition prices those fragment reads at ~190 of the 1808 us of a launch on a chip that is bound by socket power,
// not by any one pipe.  Here a workgroup is 4 waves x 64 queries: wave w keeps the B operand of BOTH of its 32-query blocks in
// registers (2 x NSLAB x 4 fragments x 4 registers = 192 at 768 columns; a lone wave of a SIMD owns 512 registers, VGPRs + AGPRs)
// and every A fragment it reads feeds TWO MFMAs — half the LDS fragment traffic per MFMA, the same MFMA count, the same row ring.
// What it gives up: a second wave on the SIMD that issues while this one waits (barrier, first fragment reads of a step, epilogue).
//
// IP / cosine threshold stages only (integer threshold image, lane-private key segments — the layout k_select gathers); the
// reference work is the same chunked scan as scan_qs.h (src/storage/flat_mmap.rs:2179-2256, :4845-4982).
#pragma once

namespace lynse {

// DBG (timing experiments): 1 no MFMA, 2 no LDS fragment reads, 8 no row DMA, 16 no epilogue, 32 s_memtime phase sums per wave (a.dbg[4096 + (block * 4 + wave) * 4 ..])
template <int NSLAB, int RB, int NS, int NBUF, int DBG = 0>
__global__ void __launch_bounds__(256, 1) k_scan_qs2(ScanArgs a) {
    constexpr int SL = NSLAB;               // whole-K stages: one step per tile
    constexpr int RT = RB * 32;             // rows per tile
    constexpr int SB = SL * RT * 128;       // bytes per ring stage
    constexpr int PP = SB / 1024;           // LDS-DMA instructions per stage (8 rows x 128 B each)
    constexpr int NW = 4;
    static_assert(PP % NW == 0, "the pieces of a stage split evenly over the 4 waves");
    constexpr int PPW = PP / NW;
    constexpr int NM = SL * 4 * RB;         // fragment reads per wave and step; each feeds two MFMAs
    static_assert(NM % NBUF == 0 && NBUF >= 2 && NBUF <= NM, "fragment ring");
    static_assert(NS >= 3, "ring depth");
    constexpr int WAITN = (NS - 2) * PPW;   // DMA instructions that may still be in flight at the barrier
    static_assert(WAITN <= 63, "vmcnt");
    static_assert(NS * SB <= 160 * 1024, "LDS");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hi = lane >> 5;
    const uint32_t ntiles = (a.row1 - a.row0 + RT - 1) / RT;
    if (blockIdx.x >= ntiles) return;
    const uint32_t my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    [[maybe_unused]] const unsigned long long t_kernel0 = (a.debug_flags & 64) ? __builtin_amdgcn_s_memtime() : 0ull;

    // ---- the wave's two query blocks: B fragments of every k-step
    const int swz = (l32 >> 1) & 7;
    qs_i32x4 bq[2][NSLAB * 4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const char* qimg = reinterpret_cast<const char*>(a.Q16) + (size_t)(wave * 64 + qb * 32 + l32) * 128;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                bq[qb][s * 4 + kk] = *reinterpret_cast<const qs_i32x4*>(qimg + (size_t)s * a.qpad * 128 + (((kk * 2 + hi) ^ swz) * 16));
    }
    // per-query constants: lane (l32, hi) owns queries wave * 64 + qb * 32 + l32
    uint32_t qn[2];
    bool q_ok[2];
    float s_q[2], b_q[2];
    int T[2];
    const bool wave_live = (uint32_t)wave * 64u < a.nq;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qn[qb] = wave * 64 + qb * 32 + l32;
        q_ok[qb] = qn[qb] < a.nq;
        s_q[qb] = q_ok[qb] ? a.qinv[qn[qb]] : 0.0f;
        b_q[qb] = q_ok[qb] ? a.qn2[qn[qb]] : 0.0f;
        // INTEGER image of the threshold (scan_qs.h): "B_q + s_q (float)dot >= thr" is exactly "dot >= T"
        const float th = q_ok[qb] ? a.thr[qn[qb]] : 0.0f;
        int lo = -(1 << 29), hi_ = 1 << 29;
#pragma unroll 1
        for (int it = 0; it < 31; ++it) {
            const int mid = lo + ((hi_ - lo) >> 1);
            const bool ge = (b_q[qb] + s_q[qb] * (float)mid) >= th;
            hi_ = ge ? mid : hi_;
            lo = ge ? lo : mid + 1;
        }
        T[qb] = q_ok[qb] ? lo : 0x7fffffff;
#ifdef LYNSE_EXPERIMENTS
        if (a.debug_flags & 2) T[qb] = 0x7fffffff;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- row stream (LDS-DMA), the ring of scan_qs.h with 4 issuing waves
    uint32_t v_off[PPW];
    const char* v_base = nullptr;
    uint32_t is_ord = 0, is_stage = 0, is_count = 0;
    auto enter_tile = [&]() {
        const uint32_t rbase = a.row0 + (blockIdx.x + is_ord * gridDim.x) * RT;
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
        glds16<2>(v_base + v_off[j], smem + is_stage * SB + (wave * PPW + j) * 1024);
    };
    auto advance = [&]() {   // past the end the last real step is issued again (uniform DMA counts; its stage is never read)
        is_stage = is_stage + 1 == NS ? 0 : is_stage + 1;
        if (++is_count < my_tiles) {
            ++is_ord;
            enter_tile();
        }
    };
    enter_tile();
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue_piece(j);
        advance();
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = lds0 + (uint32_t)l32 * 128u + (uint32_t)((hi ^ swz) * 16);
    qs_i32x4 af[NBUF];
    auto read_frag = [&](qs_i32x4& dst, const uint32_t (&ad)[4], auto idxc) {   // read idx of a step: (sl, kk, rb) = (idx / (4 RB), (idx / RB) % 4, idx % RB)
        constexpr int idx = decltype(idxc)::value;
        constexpr int sl = idx / (4 * RB), kk = (idx / RB) % 4, rb = idx % RB;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad[kk]), "n"(sl * (RT * 128) + rb * (32 * 128)));
    };
    qs_i32x16 acc[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][qb][r] = 0;

    uint32_t cnt[2] = {0u, 0u};   // keys in this lane's private segments
    uint32_t c_stage = 0;
    uint32_t ad_cur[4];
    auto mfma_step = [&]() {
        if (!(DBG & 2)) ly_static_for<NBUF - 1>([&](auto ic) { read_frag(af[decltype(ic)::value], ad_cur, ic); });
        __builtin_amdgcn_sched_barrier(0);
        ly_static_for<NM>([&](auto ic) {
            constexpr int idx = decltype(ic)::value;
            constexpr int nxt = idx + NBUF - 1;
            if (!(DBG & 2)) {
                if constexpr (nxt < NM) read_frag(af[nxt % NBUF], ad_cur, std::integral_constant<int, nxt>{});
                constexpr int outstanding = nxt < NM ? NBUF : NM - idx;
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[idx % NBUF]) : "n"(outstanding - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int sl = idx / (4 * RB), kk = (idx / RB) % 4, rb = idx % RB;
            constexpr int ks = sl * 4 + kk;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if constexpr ((DBG & 1) != 0) {
                    asm volatile("" ::"v"(af[idx % NBUF]), "v"(bq[qb][ks]));
                } else if constexpr (ks == 0) {   // first k-step of a tile: C = 0 (no accumulator clears)
                    const qs_i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    acc[rb][qb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[idx % NBUF], bq[qb][ks], z, 0, 0, 0);
                } else {
                    acc[rb][qb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[idx % NBUF], bq[qb][ks], acc[rb][qb], 0, 0, 0);
                }
            }
            // the refill of the stage computed last, spread behind the MFMAs
            ly_static_for<PPW>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (idx == (NM / PPW) * j + 1) issue_piece(j);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto epilogue = [&](uint32_t e_tile) {
        if constexpr ((DBG & 16) != 0) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][qb][r]));
        } else {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                int gm[RB][4];
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int m01 = acc[i][qb][4 * g] > acc[i][qb][4 * g + 1] ? acc[i][qb][4 * g] : acc[i][qb][4 * g + 1];
                        const int m23 = acc[i][qb][4 * g + 2] > acc[i][qb][4 * g + 3] ? acc[i][qb][4 * g + 2] : acc[i][qb][4 * g + 3];
                        gm[i][g] = m01 > m23 ? m01 : m23;
                    }
                int mx = gm[0][0];
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) mx = mx > gm[i][g] ? mx : gm[i][g];
                if (__builtin_expect(__ballot(mx >= T[qb]) != 0ull, 0)) {
                    const uint32_t rbase = a.row0 + e_tile * RT;
                    uint64_t* segdst = a.candB + ((size_t)qn[qb] * a.nseg + (blockIdx.x * 2 + hi)) * a.seg;
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (__builtin_expect(__ballot(gm[i][g] >= T[qb]) == 0ull, 1)) continue;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * g + e;
                                const int v = acc[i][qb][r];
                                if (v >= T[qb]) {
                                    const uint32_t m = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                                    if (m < a.row1) {
                                        const uint64_t key = make_key(b_q[qb] + s_q[qb] * (float)v, m, false);
                                        if (cnt[qb] < a.seg) {
                                            segdst[cnt[qb]] = key;
                                            ++cnt[qb];
                                        } else {
                                            const uint32_t slot = atomicAdd(&a.count[qn[qb]], 1u);
                                            if (slot < a.cap) a.cand[(size_t)qn[qb] * a.cap + slot] = key;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    };
    constexpr bool TIMING = (DBG & 32) != 0;
    [[maybe_unused]] unsigned long long t_wait = 0, t_bar = 0, t_loop = 0, t_epi = 0, tp = TIMING ? __builtin_amdgcn_s_memtime() : 0ull;
    [[maybe_unused]] auto stamp = [&](unsigned long long& bucket) {
        if constexpr (TIMING) { const unsigned long long t = __builtin_amdgcn_s_memtime(); bucket += t - tp; tp = t; }
    };
    for (uint32_t c_ord = 0; c_ord < my_tiles; ++c_ord) {
        stamp(t_epi);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        stamp(t_wait);
        __builtin_amdgcn_s_barrier();
        stamp(t_bar);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ad_cur[kk] = (a_lane ^ (uint32_t)(kk * 32)) + c_stage * SB;
        c_stage = c_stage + 1 == NS ? 0 : c_stage + 1;
        if (!wave_live) {   // no queries: only feed the ring
#pragma unroll
            for (int j = 0; j < PPW; ++j) issue_piece(j);
        } else {
            mfma_step();
        }
        advance();
        if constexpr (TIMING) asm volatile("" ::"v"(acc[0][0][0]));
        stamp(t_loop);
        if (wave_live) epilogue(blockIdx.x + c_ord * gridDim.x);
    }
    if constexpr (TIMING) {
        if (a.dbg && lane == 0 && blockIdx.x < 64) {
            unsigned long long* o = a.dbg + 4096 + ((size_t)blockIdx.x * 4 + wave) * 4;
            o[0] = t_wait; o[1] = t_bar; o[2] = t_loop; o[3] = t_epi;
        }
    }
    if (a.seg) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
            if (q_ok[qb]) a.segcnt[(size_t)qn[qb] * a.nseg + (blockIdx.x * 2 + hi)] = (uint8_t)cnt[qb];
    }
    if ((a.debug_flags & 64) && a.dbg && tid == 0) {   // shader cycles of this workgroup (s_memtime ticks / wall time = the clock held)
        a.dbg[blockIdx.x * 2] = t_kernel0;
        a.dbg[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace lynse
