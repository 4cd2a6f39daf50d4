// gemm_pp4.h — the persistent ping-pong GEMM (256 x 256 tiles, two wave groups alternating per half K-step): the kernel of the ViT and of the
// large language-model linears.  Instances: gemm.hip (plain bf16) and gemm_pp4_ext.hip (fp8 MFMA, folded LayerNorm consumer / producer).
#pragma once
#include "gemm_common.h"

namespace {

// ---- persistent ping-pong kernel ------------------------------------------------------------------------------
// One 512-thread workgroup per CU walks 256 x 256 tiles t = blockIdx.x, + gridDim.x, ... in the grouped XCD-aware order.
// The 8 waves are two groups of 4 that alternate per HALF K-step (K = 32): while one group issues its 12 fragment reads
// (and, every other half, its 8 LDS-DMA pieces) the other runs its 16 MFMAs at raised priority; raw s_barriers hand the
// MFMA pipe over.  What a per-tile launch pays once per tile — workgroup dispatch, the cold first DMA, the store tail of the
// epilogue — is overlapped: the first K-step of the NEXT tile is DMA'd into buffer 0 as soon as the K loop ends, while the
// epilogue stages through buffer 1, and the epilogue's stores drain under the next tile's first K-steps.
// The LDS-DMA moves whole K-steps of 64: every buffer_load ... lds fetches 8 rows x 128 B — full cache lines.  (Staging
// half K-steps as 16 rows x 64 B lands only 56-64 B/ns per CU, as long as the 16 MFMAs it should hide under; 128-byte rows
// land 97-146 B/ns: tools/probes/lds_dma_rate.hip.)  Two 64-KiB step buffers; step s + 1 is issued in the read phase of
// half 2s and waited for in the read phase of half 2s + 1.
// LN: 0 plain; 1 the A operand is a raw residual stream whose LayerNorm is folded into W / bias (GemmArgs::ln_rows, no residual input);
// 2 residual epilogue that also emits the row statistics of what it writes (GemmArgs::stat_out).
#ifndef EILEV_PP4_DEEP
#define EILEV_PP4_DEEP 1
#endif
// M16 (round 5): the K loop on v_mfma_f32_16x16x32_bf16 instead of 32x32x16 — same fragment bytes, same 128 accumulator registers per
// lane, 32 MFMAs of 16 cycles per phase instead of 16 of 32.  The chip is POWER-limited under this kernel (~1.0 "MFMA-GHz" whatever the
// schedule), and on random operands the 16 x 16 form costs less energy per flop: tools/probes/mfma_lds_ceiling.hip, the same ping-pong loop
// fed from LDS: 1818 vs 1684 TFLOP/s (profiles/r05_mfma_ceiling.log) — it is also what the vendor's kernel uses.  A lane then owns row
// l15 of every 16-row block and 4 consecutive columns (4 g4 ..) of every 16-column block: acc16[i][j][e] = C[16 i + l15][16 j + 4 g4 + e].
// Launches whose every tile can take the lean epilogue (the launcher checks: bf16 output, N % 128 == 0, ...): there is no general epilogue
// in this layout.  A last column tile with only 128 valid columns (N = 1408, 4224): see M16K below (profiles/r05_m16_ab_all_shapes.log,
// r05_m16_ht_ab.log).
template <int EPI, bool F8 = false, int LN = 0, int M16K = 0, bool A3 = false>
__global__ __launch_bounds__(512, 2) void gemm_pp4_kernel(const GemmArgs g) {
    // M16K: 0 = 32 x 32 x 16 MFMAs; 1 = 16 x 16 x 32, a half-valid last column tile runs as a WHOLE tile (its W rows past N are not staged:
    // stale, finite LDS bytes; the waves that own columns >= N skip the epilogue) — no half-tile code in the instance: the folded-LayerNorm
    // consumers, whose registers are tightest (qkv, N = 4224: 3 % padding, +2.1 %); 2 = 16 x 16 x 32 with the half-tile path (N = 1408: a
    // whole tile for the half column costs fc2 6 %)
    constexpr bool M16 = M16K != 0, HT16 = M16K == 2;
    static_assert(!M16 || !F8, "16 x 16 form: bf16 instances");
    constexpr int BM = 256, BN = 256, NWM = 2, NWN = 4, NW = 8;
    constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 32, TN = WN / 32;
    constexpr int STEP = (BM + BN) * 128;
    constexpr int PC = 8;  // 1-KiB LDS-DMA pieces (8 rows x 128 B) per wave and K-step
    // A3 (round 6; an own instance: plain 16 x 16 launches with K >= 5120 — the OPT fc2 of a prefill, the flan-t5 wo): the A operand in a ring of THREE K-steps,
    // W in two: A of step s + 2 is requested in the read phase of (s, half 0), two K-steps of flight instead of one.  LDS: A slots [0, 96 KiB) at
    // 32 KiB each, W slots [96, 160 KiB); the 32 KiB of epilogue staging OVERLAY A slot 2 — free while an epilogue runs (the next tile's steps
    // 0 and 1 are what is in flight then) and requested again only behind the barrier at the top of the next tile, which every wave reaches
    // after its epilogue.  Same-box A/B, bit-identical (profiles/r06_a3_ring_ab.log): OPT fc2 (M = 30 720, N = 2560, K = 10 240) +10 %; the ViT
    // shapes (K = 1408 / 6144) -0.3 ... -17 % and the other OPT linears -3 %, which is why it is not the default form.
    static_assert(!A3 || M16, "A3: lean epilogues only");
    constexpr int STG0 = A3 ? 2 * BM * 128 : 2 * STEP;  // first byte of the epilogue staging

    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto a_slot = [&](int st) { return smem + (A3 ? (st % 3) * (BM * 128) : (st & 1) * STEP); };
    auto w_slot = [&](int st) { return smem + (A3 ? 3 * (BM * 128) + (st & 1) * (BN * 128) : (st & 1) * STEP + BM * 128); };
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / NWN, wn = wid % NWN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int l15 = lane & 15, g4 = lane >> 4;
    constexpr int TM16 = WM / 16, TN16 = WN / 16;
    const int ns = g.K / 64;
    const bool late = wid >= NW / 2;
    const int prow = lane >> 3, pslot = lane & 7;

    // Who stages what (round 4; ADVICE r3: the round-3 schedule let a late wave's DMA land in rows another late wave might still be
    // reading, with only the DMA's latency in between).  The early group (waves 0-3) stages ALL of A: wave w the tile rows 64 w .. 64 w + 63.
    // The late group (waves 4-7) stages ALL of W: wave 4 + j the tile rows 64 j .. 64 j + 63 — exactly the W rows that wave reads itself
    // (wn = j) and that, besides it, only the early wave j reads, one barrier interval EARLIER.  So when a late wave issues step st + 2
    // at the end of its own reads of (st, half 1), nobody else can still be reading the rows it overwrites: no timing argument left.
    // (Half tiles re-split the reads 4 x 2, so there the late group issues after the barrier instead: kstep below.)
    // A through one descriptor PER TILE (base = the tile's first row, wave-uniform): the 32-bit offsets then span 256 rows, so an A
    // operand of 2 GiB or more (the ViT fc2 input of a bench launch: 279 616 x 6144 bf16 = 3.4 GB) needs no row chunking
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)g.W, 0, 0x7fffffff, 0x00020000);
    unsigned po[PC];  // per-lane byte offsets of this wave's pieces (chunk-swizzled source): into A for waves 0-3, into W for waves 4-7
    const int pw = late ? wid - NW / 2 : wid;  // which 64-row slab of its operand the wave stages
    auto set_tile = [&](int t, int &m0, int &n0) {
        int tm_i, tn_i;
        tile_coords(g, tiles_m, tiles_n, tm_i, tn_i, t);
        m0 = tm_i * BM;
        n0 = tn_i * BN;
        {
#ifdef EILEV_PP4_PROBE_A0  /* timing probe only (WRONG results): every tile reads the A rows of tile row 0 — same data statistics, no fabric traffic for A */
            const uint64_t base = (uint64_t)(g.A);
#else
            const uint64_t base = (uint64_t)(g.A + (int64_t)m0 * g.lda);
#endif
            const uint64_t ub = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);  // provably wave-uniform: no waterfall loop
            ra = __builtin_amdgcn_make_buffer_rsrc((void *)ub, 0, 0x7fffffff, 0x00020000);
        }
#pragma unroll
        for (int i = 0; i < PC; ++i) {
            const int row = (pw * PC + i) * 8 + prow;
            const unsigned sw = (unsigned)((pslot ^ ((row >> 1) & 7)) << 4);
            if (!late) {
                int gr = m0 + row;
                gr = gr < g.M ? gr : g.M - 1;
                po[i] = (unsigned)(gr - m0) * (unsigned)(g.lda * 2) + sw;
            } else {
                int gr = n0 + row;
                gr = gr < g.N ? gr : g.N - 1;
                po[i] = (unsigned)gr * (unsigned)(g.ldw * 2) + sw;
            }
        }
    };
    // part: 0 all 8 pieces, 1 pieces 0..3, 2 pieces 4..7 (EILEV_PP4_SPLIT: a wave's pieces of a K-step are issued in two read phases)
    auto stage_step = [&](int st, bool mine = true, int part = 0) {  // mine == false: a late wave whose W rows do not exist in a half tile
        char *sd = (late ? w_slot(st) : a_slot(st)) + (pw * PC) * 1024;
        if (!mine) return;
#ifndef EILEV_PP4_PROBE_SKIP
#define EILEV_PP4_PROBE_SKIP 0  /* timing probe only (WRONG results): 1 / 2 the early / late group issues half of its pieces, 4 / 8 none */
#endif
        constexpr int PCE = (EILEV_PP4_PROBE_SKIP & 4) ? 0 : (EILEV_PP4_PROBE_SKIP & 1) ? PC / 2 : PC;
        constexpr int PCL = (EILEV_PP4_PROBE_SKIP & 8) ? 0 : (EILEV_PP4_PROBE_SKIP & 2) ? PC / 2 : PC;
        if (late) {
#pragma unroll
            for (int i = 0; i < PCL; ++i)
                if (part == 0 || (part == 1) == (i < PC / 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(sd + i * 1024), 16, po[i], st * 128, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < PCE; ++i)
                if (part == 0 || (part == 1) == (i < PC / 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void *)(sd + i * 1024), 16, po[i], st * 128, 0, 0);
        }
    };
    f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
    bf16x8 af[2][TM], bfr[2][TN];
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    f32x4_t acc16[M16 ? TM16 : 1][M16 ? TN16 : 1];
    auto read_half = [&](int st, int h) {
        const char *sa = a_slot(st) + (wm * WM) * 128;
        const char *sb = w_slot(st) + (wn * WN) * 128;
        if constexpr (M16) {  // one k-slice of 32 per half step: chunk h * 4 + g4 of rows 16 i + l15 (af / bfr reused as flat arrays of 8 / 4)
            const int kc = h * 4 + g4;
#pragma unroll
            for (int j = 0; j < TN16; ++j) {
                const int row = j * 16 + l15;
                bfr[j >> 1][j & 1] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + swz(row, kc));
            }
#pragma unroll
            for (int i = 0; i < TM16; ++i) {
                const int row = i * 16 + l15;
                af[i >> 2][i & 3] = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + swz(row, kc));
            }
            return;
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kc = h * 4 + k2 * 2 + hi;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = j * 32 + l31;
                bfr[k2][j] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + swz(row, kc));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = i * 32 + l31;
                af[k2][i] = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + swz(row, kc));
            }
        }
    };
    // F8: the operands are e4m3 BYTES (the kernel is launched with K = bytes / 2 so every address below is unchanged): the 64 bytes a
    // lane group holds for a half K-step (its two 16-byte chunks of every row) are ONE 32x32x64 fp8 MFMA — twice the flops of the two
    // bf16 MFMAs they would be.  Which k a byte position stands for is irrelevant as long as A and W use the same assignment (they
    // do: same chunk indices), the products are summed over all 64.
    typedef int i32x8_t __attribute__((ext_vector_type(8)));
    struct Pair16 { bf16x8 lo, hi; };
    auto cat32 = [](const bf16x8 &lo, const bf16x8 &hi2) { return __builtin_bit_cast(i32x8_t, Pair16{lo, hi2}); };
    auto mma_half = [&]() {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat32(bfr[0][j], bfr[1][j]), cat32(af[0][i], af[1][i]), acc[i][j],
                                                                               0, 0, 0, 0, 0, 0);
        } else if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < TM16; ++i)
#pragma unroll
                for (int j = 0; j < TN16; ++j)
                    acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j >> 1][j & 1], af[i >> 2][i & 3], acc16[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[k2][j], af[k2][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // Half-empty last column tile (N % 256 <= 128, e.g. N = 1408 = 5.5 x 256): only columns [0, 128) of the tile exist.
    // The 8 waves re-split the valid 256 x 128 region as 4 x 2 blocks of 64 x 64 (half the fragment reads and MFMAs per
    // wave, same ping-pong schedule); waves 4..7 own W rows 128..255 of the tile and skip their W pieces.
    const int hm = wm * 2 + (wn >> 1), hn = wn & 1;
    auto read_half_ht = [&](int st, int h) {
        const char *sa = a_slot(st) + (hm * 64) * 128;
        const char *sb = w_slot(st) + (hn * 64) * 128;
        if constexpr (M16) {  // 64 x 64 per wave: 4 + 4 fragments of the one 32-wide k-slice
            const int kc = h * 4 + g4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = j * 16 + l15;
                bfr[j >> 1][j & 1] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + swz(row, kc));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 16 + l15;
                af[0][i] = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + swz(row, kc));
            }
            return;
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kc = h * 4 + k2 * 2 + hi;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = j * 32 + l31;
                bfr[k2][j] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + swz(row, kc));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = i * 32 + l31;
                af[k2][i] = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + swz(row, kc));
            }
        }
    };
    auto mma_half_ht = [&]() {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat32(bfr[0][j], bfr[1][j]), cat32(af[0][i], af[1][i]), acc[i][j],
                                                                               0, 0, 0, 0, 0, 0);
        } else if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j >> 1][j & 1], af[0][i], acc16[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[k2][j], af[k2][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // ---- lean epilogue (interior tiles, bf16 output, no column tail / patch remap / column scaling) -------------------------
    // 4 KiB of LDS staging per wave OUTSIDE the two step buffers, so the next tile's first TWO K-steps are already in flight
    // while it runs (the general epilogue below stages 69.6 KB through step buffer 1 and leaves a DMA-latency bubble at the top
    // of the next tile).  Units of 32 rows x 64 columns (one i block of the wave): activation / residual / bf16 into the staging
    // rows (16-byte chunk c of row r at chunk c ^ (r & 7)), read back as 128-byte row segments, buffer stores whose descriptor
    // drops rows past M.  Round 2 (tools/gemm_trace.py: 6.7 us of a 42 us bias-only tile, 8.5 us with GELU, all of it with the
    // MFMA pipe idle and most of it instruction issue): the residual variant is a compile-time copy (no per-cell branches, the
    // unit's 8 residual cells fetched from the staging rows in one batch, the next unit's rows in flight), and the GELU is the
    // degree-8 form of common.h.  (Folding the bias into the first MFMAs' C operand — no accumulator initialisation, no bias
    // arithmetic here — was built and makes hipcc spill 200-600 VGPRs in this 256-register kernel: not adopted.)
    char *const stg = smem + STG0 + wid * 4096;
    const unsigned stg_sw = (unsigned)(STG0 + wid * 4096 + l31 * 128 + hi * 8) ^ (unsigned)((l31 & 7) << 4);
    // 16 x 16 layout: the lane's 4 values of block (ib, j) of a unit = row 16 ib + l15, columns 16 j + 4 g4 ..: chunk 2 j + (g4 >> 1), bytes 8 (g4 & 1)
    const unsigned stg_sw16 = (unsigned)(STG0 + wid * 4096 + l15 * 128 + (g4 & 1) * 8) ^ (unsigned)((((g4 >> 1) ^ (l15 & 7)) << 4));
    const int srow = lane >> 3, schunk = lane & 7;
    auto is_lean = [&](int n0_) {
        return !(n0_ + 128 >= g.N && !(g.dbg & 524288)) && n0_ + BN <= g.N && !g.out_f32 && g.patch_group == 0 && g.scale_cols == 0 && !g.wscale && !g.ascale &&
               !(g.dbg & (1024 | 2048 | 1)) && !(g.dbg & 16777216);
    };
    // LN == 2 (proj / fc2, residual): the unit also emits the row statistics of what it writes (g.stat_out); LN == 1 (qkv / fc1, no residual):
    // (qkv / fc1) it finishes a folded LayerNorm (g.ln_rows / g.ln_csum): see GemmArgs.
    float ln_rs[LN == 1 ? (M16 ? TM16 : TM) : 1];  // LN == 1: rstd of the lane's row in each of its units (M16: 16-row blocks), fetched at the end of the K loop
    auto lean_epilogue = [&](int cm0, int cn0, auto res_c, auto ht_c) {
        constexpr bool RES = decltype(res_c)::value;
        constexpr bool HTE = decltype(ht_c)::value;  // M16 only: the half tile's 64 x 64 block of this wave (rows 64 hm .., columns 64 hn ..)
        constexpr int NU = HTE ? 2 : TM;              // units of 32 rows
        const int roff = HTE ? hm * 64 : wm * WM, coff = HTE ? hn * 64 : wn * WN;
        if constexpr (M16K == 1) {
            if (cn0 + coff >= g.N) return;  // (wave-uniform) this wave's 64 columns do not exist: the half-valid last column tile run whole
        }
        constexpr bool LNC = LN == 1 && !RES, LNP = LN == 2 && RES;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        const int r0 = cm0 + roff;
        const int rows = g.M - r0 < 0 ? 0 : (g.M - r0 < NU * 32 ? g.M - r0 : NU * 32);
        auto uniform_rsrc = [&](const void *ptr, int bytes) {
            const uint64_t base = (uint64_t)ptr;
            const uint64_t ub = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);  // provably wave-uniform: no waterfall loops
            return __builtin_amdgcn_make_buffer_rsrc((void *)ub, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t rc = uniform_rsrc(reinterpret_cast<bf16 *>(g.C) + (int64_t)r0 * g.ldc + cn0 + coff, rows * (int)(g.ldc * 2));
        const __amdgpu_buffer_rsrc_t rr = uniform_rsrc(RES ? g.resid + (int64_t)r0 * g.ldr + cn0 + coff : g.A, RES ? rows * (int)(g.ldr * 2) : 0);
        const unsigned st_voff = (unsigned)srow * (unsigned)(g.ldc * 2) + schunk * 16, rs_voff = (unsigned)srow * (unsigned)(g.ldr * 2) + schunk * 16;
        // scattered q|k|v output (GemmArgs::hm_tok; the folded-LayerNorm consumer without activation only): the lane's 16-byte chunk of token row t
        // goes to its head's block — the column part is fixed for the call and computed here (a table in memory put a dependent global load of
        // ~1.5 us in front of the first store of every tile: the q|k|v GEMM +45 us), the row part walks 8 token rows per store
#ifndef EILEV_HM_AUX
#define EILEV_HM_AUX EILEV_ST_AUX
#endif
        constexpr bool HMI = LN == 1 && !RES && EPI == 0 && M16K == 1;
        const bool hm = HMI && g.hm_tok != 0;  // (uniform)
        __amdgpu_buffer_rsrc_t rch = rc;
        unsigned hm_off = 0, hm_off_w = 0;  // byte offset of the lane's chunk in token row m0 = r0 + srow, and the same one frame step further
        int hm_str2 = 0, hm_t = 0;           // bytes per token row of the wave's block (uniform: a wave's 64 columns lie in one kind of block)
        if constexpr (HMI) {
            if (hm) {
                const int TOK = g.hm_tok, HD = g.hm_hd, NH = g.hm_heads, D3 = g.N / 3;
                const int n0 = cn0 + coff;  // the wave's 64 columns (uniform): all in [token][64] blocks or all in [token][HD - 64] blocks
                const int plane = __builtin_amdgcn_readfirstlane(n0 / D3), rr = n0 - plane * D3;
                int2 ent;  // (element offset inside the frame of the lane's chunk in token row 0, elements per token row of its block)
                if (rr < NH * 64) {
                    ent.x = (plane * NH + (rr >> 6)) * TOK * HD + schunk * 8;
                    ent.y = 64;
                } else {  // chunk jb of the second part: head jb / CPH, chunk jb % CPH of its (HD - 64)-wide rows
                    const int cph = (HD - 64) >> 3, jb = ((rr - NH * 64) >> 3) + schunk, hh = jb / cph;
                    ent.x = (plane * NH + hh) * TOK * HD + TOK * 64 + (jb - hh * cph) * 8;
                    ent.y = HD - 64;
                }
                const int fw = __builtin_amdgcn_readfirstlane(r0 / TOK);  // frame of the wave's first row: offsets below stay inside 32 bits
                const int m0 = r0 + srow, f0 = m0 / TOK;
                hm_t = m0 - f0 * TOK;
                hm_str2 = __builtin_amdgcn_readfirstlane(ent.y * 2);
                hm_off = (unsigned)(((f0 - fw) * TOK * g.N + ent.x + hm_t * ent.y) * 2);
                hm_off_w = hm_off + (unsigned)(TOK * g.N * 2) - (unsigned)(TOK * hm_str2);
                // rows past M lie in frames past the descriptor's range (M % TOK == 0): dropped without a test
                const int64_t left = ((int64_t)g.M / TOK - fw) * TOK * g.N * 2;
                rch = uniform_rsrc(reinterpret_cast<bf16 *>(g.C) + (int64_t)fw * TOK * g.N, (int)(unsigned)(left < 0xfffffff0ll ? left : 0xfffffff0ll));
            }
        }
        u32x4_t rv[4];
        auto res_load = [&](int u) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, rs_voff, (u * 32 + it * 8) * (int)(g.ldr * 2), 0);
        };
        if constexpr (RES) res_load(0);
        bf16x4 biasr[M16 ? 1 : TN][4];  // columns j*32 + q*8 + hi*4 + (0..3) of the wave's 64: the accumulator layout
        bf16x4 biasr16[M16 ? TN16 : 1];  // M16: columns 16 j + 4 g4 + (0..3)
        if constexpr (M16) {
#pragma unroll
            for (int j = 0; j < TN16; ++j) {
                if (g.bias) biasr16[j] = *reinterpret_cast<const bf16x4 *>(g.bias + cn0 + coff + j * 16 + g4 * 4);
                else biasr16[j] = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            }
        } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (g.bias) biasr[j][q] = *reinterpret_cast<const bf16x4 *>(g.bias + cn0 + wn * WN + j * 32 + q * 8 + hi * 4);
                else biasr[j][q] = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            }
        }
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        __amdgpu_buffer_rsrc_t rst = rr;
        if constexpr (LNP) rst = uniform_rsrc(g.stat_out + ((int64_t)((cn0 + coff) >> 6) * g.stat_ld + r0) * 2, rows * 8);
        static_for<NU>([&](auto u_c) {
            constexpr int U = decltype(u_c)::value;
            float st1 = 0.0f, st2 = 0.0f;
            bf16x4 rcell[M16 ? 2 : TN][4];  // (M16: [ib][j])
            if constexpr (RES) {  // the unit's residual rows -> staging (coalesced); every lane then fetches its 8 cells in one batch
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + srow;
                    *reinterpret_cast<u32x4_t *>(stg + row * 128 + ((schunk ^ (row & 7)) << 4)) = rv[it];
                }
                if constexpr (U + 1 < NU) res_load(U + 1);  // the next unit's rows arrive under this unit's arithmetic
                if constexpr (M16) {
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                        for (int j = 0; j < TN16; ++j) {
                            unsigned ca;
                            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ca) : "n"(((j * 2) << 4) | (ib << 11)), "v"(stg_sw16));
                            rcell[ib][j] = *reinterpret_cast<const bf16x4 *>(smem + ca);
                        }
                } else {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned ca;
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ca) : "n"((j * 4 + q) << 4), "v"(stg_sw));
                        rcell[j][q] = *reinterpret_cast<const bf16x4 *>(smem + ca);
                    }
                }
            }
            float st16[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};  // M16 + LNP: (sum, sum of squares) of the lane's 16 values of row 16 ib + l15
            if constexpr (M16) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int j = 0; j < TN16; ++j) {
                        float v[4];
                        if constexpr (LNC) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc16[2 * U + ib][j][e], ln_rs[2 * U + ib], (float)biasr16[j][e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc16[2 * U + ib][j][e] + (float)biasr16[j][e];
                        }
                        if constexpr (LN == 0 && EPI == 0 && !RES) {  // q pre-scaling of the fused q|k|v projection: (x W^T + b) * scale on the first
                            if (cn0 + coff + j * 16 < g.scale_cols) {   // scale_cols columns (a multiple of 16: whole blocks, wave-uniform)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] *= g.scale;
                            }
                        }
                        if constexpr (EPI == 2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                        }
                        if constexpr (EPI == 1) gelu_erf_n<4>(v);
                        if constexpr (RES) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rcell[ib][j][e];
                        }
                        if constexpr (LNP) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                st16[ib][0] += v[e];
                                st16[ib][1] = fmaf(v[e], v[e], st16[ib][1]);
                            }
                        }
                        unsigned ca;
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ca) : "n"(((j * 2) << 4) | (ib << 11)), "v"(stg_sw16));
                        *reinterpret_cast<bf16x4 *>(smem + ca) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    }
            } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
                    if constexpr (LNC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[U][j][q * 4 + e], ln_rs[U], (float)biasr[j][q][e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[U][j][q * 4 + e] + (float)biasr[j][q][e];
                    }
                    if constexpr (EPI == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    if constexpr (EPI == 1) gelu_erf_n<4>(v);
                    if constexpr (RES) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rcell[j][q][e];
                    }
                    if constexpr (LNP) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            st1 += v[e];
                            st2 = fmaf(v[e], v[e], st2);
                        }
                    }
                    unsigned ca;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ca) : "n"((j * 4 + q) << 4), "v"(stg_sw));
                    *reinterpret_cast<bf16x4 *>(smem + ca) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                }
            }
            if constexpr (LNP && M16) {  // the four lane quarters (g4) hold the four column quarters of a row: fixed summation order
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    float a1 = st16[ib][0], a2 = st16[ib][1];
                    a1 += __shfl_xor(a1, 16); a2 += __shfl_xor(a2, 16);
                    a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32);
                    const f32x2_t t = (f32x2_t){a1, a2};
                    if (g4 == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, t), rst, (U * 32 + ib * 16 + l15) * 8, 0, 0);
                }
            } else if constexpr (LNP) {  // the two lane halves hold the two column halves of a row
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                const f32x2_t t = (f32x2_t){st1 + __shfl_xor(st1, 32), st2 + __shfl_xor(st2, 32)};
                if (hi == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, t), rst, (U * 32 + l31) * 8, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // cells of a unit first, then its read-backs and stores; nothing of the next unit in between
            bf16x8 erb[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {  // read-backs and stores in the order R0 R1 S0 S1 R2 R3 S2 S3
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int row = (h2 * 2 + b) * 8 + srow;
                    erb[b] = *reinterpret_cast<const bf16x8 *>(stg + row * 128 + ((schunk ^ (row & 7)) << 4));
                }
                if (hm) {
                    if constexpr (HMI) {
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int RO = U * 32 + (h2 * 2 + b) * 8;  // row of this store relative to srow: < 128 < hm_tok, at most one frame step
                            // the row part RO * stride rides in the scalar offset (range-checked with the rest on gfx950: tools/probes/canary_rows_past_m.py)
                            const unsigned vo = hm_t + RO >= g.hm_tok ? hm_off_w : hm_off;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, erb[b]), rch, vo, RO * hm_str2, EILEV_HM_AUX);
                        }
                    }
                } else {
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, erb[b]), rc, st_voff, (U * 32 + (h2 * 2 + b) * 8) * (int)(g.ldc * 2), EILEV_ST_AUX);
                }
            }
            // Measured on gfx950 (round 2): with the next unit's arithmetic scheduled between these stores, a VALU write to the data
            // registers of a 128-bit buffer store issued the cycle before corrupted the first dword of the stored chunk (the "SGPR
            // soffset needs no wait state" exception of the GFX9 hazard table does not hold here).  Keep the scheduler out, and two
            // idle states between the last store and whatever reuses its registers.
            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        });
    };
// The waits in front of the hand-over barriers are BUILTINS, not inline asm (round 5): hipcc's wait-count pass cannot see into an asm
// statement, so behind an asm "s_waitcnt lgkmcnt(0)" it still believes the 12 fragment reads are outstanding and puts its own
// s_waitcnt lgkmcnt(9 / 8 / 7 / 6 / 3 / 2 / 1 / 0) between the 16 MFMAs of the phase — eight instructions that never wait and still take
// issue slots between back-to-back MFMAs (the trace of tools/gemm_itrace.py: an MFMA phase took 580-650 cycles, 16 x 32 = 512 ideal).
// simm16 of s_waitcnt on gfx9: vmcnt = [15:14 | 3:0], expcnt = [6:4], lgkmcnt = [11:8].
#define PP_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)      /* lgkmcnt(0) */
#define PP_WAIT_LGKM0_VM0() __builtin_amdgcn_s_waitcnt(0x0070)  /* vmcnt(0) lgkmcnt(0) */
#define PP_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)        /* vmcnt(0) */
#define PP_BARRIER()                       \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

    int t = blockIdx.x, m0, n0;
    if (t >= ntiles) return;
    set_tile(t, m0, n0);
    // half tile: only W rows 0..127 of the tile exist; the late waves 6 and 7 (rows 128..255) have nothing to stage
    auto w_piece_mine = [&](int n0_) { return !(n0_ + 128 >= g.N && !(g.dbg & 524288)) || wid < NW / 2 + 2; };
    stage_step(0, w_piece_mine(n0));
    bool pre1 = ns > 1 && (A3 || !(g.dbg & 16777216));  // step 1 of the coming tile is already staged (prologue / previous tile's tail)
    if (pre1) stage_step(1, w_piece_mine(n0));
    bool lean_cur = M16 || is_lean(n0);  // this tile runs the lean epilogue (M16: every tile does)
    // folded LayerNorm (LN == 1): C = rstd * (A . W^T - mean * csum) + bias.  The rank-1 term -mean[m] * csum[n] is one more K-slice on
    // the matrix cores: the tile's first MFMA of every 32 x 32 block multiplies (csum_hi, csum_lo, csum_hi, 0 ...) by (nm_hi, nm_hi,
    // nm_lo, 0 ...) with nm = -mean (v_mfma_f32_32x32x8_bf16_1k: half the cost of the K = 16 form) — two bf16 pieces each, the product is good to 2^-16 of |mean * csum|, far inside the bf16 output —
    // and starts the accumulators (C = 0); the epilogue multiplies by rstd.  8 short MFMAs per wave and tile (+0.6 % of the K loop), ~30
    // VALU operations, 6 four-byte loads per lane fetched one tile AHEAD next to the next tile's first DMA (rows past M / columns past N
    // read 0).  (Built and measured before this: accumulators initialised with v_mul from 32 csum registers per lane — 36 loads per
    // lane and tile through the texture addresser and 128 VALU operations in a read phase: fc1 +4.5 %.)
    float ln_nm[LN == 1 ? (M16 ? TM16 : TM) : 1], ln_cl[LN == 1 ? (M16 ? TN16 : TN) : 1];
    auto ln_fetch = [&](int m0_, int n0_) {
        if constexpr (LN == 1 && M16) {  // the lane's rows 16 i + l15 and columns 16 j + l15 (the operand layout of the 16 x 16 MFMA)
            const bool ht = HT16 && n0_ + 128 >= g.N;
            const int mb = m0_ + (ht ? hm * 64 : wm * WM) + l15, nb = n0_ + (ht ? hn * 64 : wn * WN) + l15;
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)g.ln_rows, 0, g.M * 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t rcs = __builtin_amdgcn_make_buffer_rsrc((void *)g.ln_csum, 0, g.N * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM16; ++i) ln_nm[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (mb + i * 16) * 8 + 4, 0, 0));
#pragma unroll
            for (int j = 0; j < TN16; ++j) ln_cl[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rcs, (nb + j * 16) * 4, 0, 0));
        } else if constexpr (LN == 1) {
            const bool ht = n0_ + 128 >= g.N && !(g.dbg & 524288);
            const int mb = m0_ + (ht ? hm * 64 : wm * WM) + l31, nb = n0_ + (ht ? hn * 64 : wn * WN) + l31;
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)g.ln_rows, 0, g.M * 8, 0x00020000);
            const __amdgpu_buffer_rsrc_t rcs = __builtin_amdgcn_make_buffer_rsrc((void *)g.ln_csum, 0, g.N * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i) ln_nm[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (mb + i * 32) * 8 + 4, 0, 0));
#pragma unroll
            for (int j = 0; j < TN; ++j) ln_cl[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rcs, (nb + j * 32) * 4, 0, 0));
        }
    };
    ln_fetch(m0, n0);
    int trace_i = 0;
    const bool tracer = g.trace != nullptr && (wid == 0 || wid == NW / 2) && lane == 0;
    // EILEV_PP4_ITRACE (probe build only, tools/gemm_itrace.py): instead of the per-tile phase stamps, the 8 slots of a (workgroup, wave group,
    // tile) record hold s_memtime at the 8 phase edges of ONE K-step (st == EILEV_PP4_ITRACE) of that tile: start of the read phase of half
    // 0 / its reads landed (before the barrier) / barrier released = first MFMA phase starts / its MFMAs issued / barrier released = read
    // phase of half 1 starts / reads (+ the late group's DMA wait and issue) done / barrier released / second MFMA phase issued.  The stamps
    // go through the wave's (idle) epilogue staging bytes and are copied out at the tile's end.
#ifdef EILEV_PP4_ITRACE
#define ITR(k)                                                                                                              \
    do {                                                                                                                    \
        if (st == EILEV_PP4_ITRACE && tracer) *reinterpret_cast<volatile unsigned long long *>(stg + (k) * 8) = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ITR(k) do { } while (0)
#endif
    auto stamp = [&](int k, bool core = false) {
#ifdef EILEV_PP4_ITRACE
        return;
#endif
        if (tracer && trace_i < g.trace_tiles)
            g.trace[(((size_t)blockIdx.x * 2 + (late ? 1 : 0)) * g.trace_tiles + trace_i) * 8 + k] =
                core ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memrealtime();
    };
    for (; t < ntiles; t += gridDim.x) {
        stamp(0);
        stamp(5, true);
        typedef __attribute__((ext_vector_type(4))) short s16x4_t;  // operand type of the K = 8 bf16 MFMA
        s16x4_t ln_a1[LN == 1 ? (M16 ? TM16 : TM) : 1], ln_w1[LN == 1 ? (M16 ? TN16 : TN) : 1];
        auto acc_prep = [&]() {  // the two fragments of the rank-1 K-slice (in a read phase)
            if constexpr (LN == 1 && M16) {  // v_mfma_f32_16x16x16_bf16: k-slots 4 g4 .. 4 g4 + 3; the values sit in the lanes g4 == 0
                const bf16 z = (bf16)0.0f;
#pragma unroll
                for (int i = 0; i < TM16; ++i) {
                    const float m = g4 ? 0.0f : ln_nm[i];
                    const bf16 mh = (bf16)m, ml = (bf16)(m - (float)mh);
                    ln_a1[i] = __builtin_bit_cast(s16x4_t, (bf16x4){mh, mh, ml, z});
                }
#pragma unroll
                for (int j = 0; j < TN16; ++j) {
                    const float c = g4 ? 0.0f : ln_cl[j];
                    const bf16 ch = (bf16)c, cl = (bf16)(c - (float)ch);
                    ln_w1[j] = __builtin_bit_cast(s16x4_t, (bf16x4){ch, cl, ch, z});
                }
            } else if constexpr (LN == 1) {
                const bf16 z = (bf16)0.0f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float m = hi ? 0.0f : ln_nm[i];  // k-slots 0..3 belong to lanes 0..31
                    const bf16 mh = (bf16)m, ml = (bf16)(m - (float)mh);
                    ln_a1[i] = __builtin_bit_cast(s16x4_t, (bf16x4){mh, mh, ml, z});
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float c = hi ? 0.0f : ln_cl[j];
                    const bf16 ch = (bf16)c, cl = (bf16)(c - (float)ch);
                    ln_w1[j] = __builtin_bit_cast(s16x4_t, (bf16x4){ch, cl, ch, z});
                }
            }
        };
        auto acc_init = [&]() {
            if constexpr (M16) {
                const f32x4_t zero4 = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < TM16; ++i)
#pragma unroll
                    for (int j = 0; j < TN16; ++j) {
                        if constexpr (LN == 1) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ln_w1[j], ln_a1[i], zero4, 0, 0, 0);
                        else acc16[i][j] = zero4;
                    }
            } else if constexpr (LN == 1) {
                f32x16 zero;
#pragma unroll
                for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ln_w1[j], ln_a1[i], zero, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            }
        };
        if constexpr (LN != 1) acc_init();
        // step 0 of this tile was issued by the prologue above or by the previous tile's tail; the wait also covers the
        // previous epilogue's stores, and the barrier its LDS staging reads (which overlay step buffer 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_BARRIER();
        if (late) PP_BARRIER();
        stamp(1);
        const int nsd = (g.dbg & 2) ? 1 : ns;
        const bool half_tile = n0 + 128 >= g.N && !(g.dbg & 524288);
        // one K-step = two half-steps; FIRST (compile-time) marks the tile's first K-step, whose first 16 MFMAs take C = bias / 0
        // DMA schedule (round 3).  A step buffer is free once BOTH wave groups have read its second half; the early group (E) gets
        // there one barrier interval before the late group (L).  r2 issued step st + 1 in the read phase of half 0 of step st and
        // waited for it in the read phase of half 1: 3 intervals (~1500 shader clocks, 0.9 us) between issue and wait, less than a
        // loaded L2 miss takes.  Now both groups get 4 intervals (a whole K-step): E issues as before but waits at the END of its
        // second MFMA phase; L issues step st + 2 at the end of its read phase of (st, half 1) — the buffer of step st is free for
        // it then — and waits for it a whole K-step later at the same place.
        auto kstep = [&](int st, auto first_c, auto ht_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            constexpr bool HT = decltype(ht_c)::value;
            const bool w_mine = !HT || wid < NW / 2 + 2;
            ITR(0);
            if constexpr (HT) read_half_ht(st, 0); else read_half(st, 0);
#if EILEV_PP4_DEEP
            // Round 5: a wave's 8 pieces of a K-step are issued in TWO read phases (whole tiles).  The interval trace (tools/gemm_itrace.py)
            // shows the read phases that carry a group's 32 pieces as the long ones (the CU's LDS-DMA path takes ~17 cycles per 1-KiB
            // piece: a 32-piece burst is longer than the other group's 16 MFMAs), the read phases without pieces as the short ones.
            // Early group: pieces 0..3 of step st + 1 in the read phase of half 0 (as before), pieces 4..7 in the read phase of half 1
            // (the buffer has been free since the previous barrier; waited for at the end of the second MFMA phase, as before).  Late
            // group: pieces 0..3 of step st + 2 at the end of its read phase of (st, half 1) (as before), pieces 4..7 one phase pair
            // later, in its read phase of (st + 1, half 0) — still only rows this wave and its early twin read, both done — waited for
            // at the end of the read phase of (st + 1, half 1), as before.  Same-box A/B (profiles/r05_dma_split_ab.log): fc2 +2.3 %,
            // fc2 + statistics +1.9 %, fc1 +0.8...1.4 %, proj +0.9 %, bit-identical; the folded-LayerNorm qkv instance loses 1.1 % and keeps
            // the unsplit schedule.
#ifndef EILEV_PP4_SPLIT
#define EILEV_PP4_SPLIT 3
#endif
#ifndef EILEV_PP4_SPLIT_QKV
#define EILEV_PP4_SPLIT_QKV 0
#endif
            constexpr bool SPLIT_OK = !HT && (EILEV_PP4_SPLIT_QKV || !(LN == 1 && EPI == 0));
            constexpr bool SPLIT_E = (EILEV_PP4_SPLIT & 1) && SPLIT_OK, SPLIT_L = (EILEV_PP4_SPLIT & 2) && SPLIT_OK;
            if (!late) {
                if constexpr (A3) {  // steps 0 and 1 came with the tile's head; A of step st + 2 goes into the slot both groups left at the last barrier
                    if (st + 2 < ns) stage_step(st + 2, w_mine, SPLIT_E ? 1 : 0);
                } else if (st + 1 < ns && !(FIRST && pre1)) stage_step(st + 1, w_mine, SPLIT_E ? 1 : 0);
            } else if (FIRST && !pre1 && ns > 1) stage_step(1, w_mine);
            else if (SPLIT_L && !FIRST && st + 1 < ns) stage_step(st + 1, w_mine, 2);  // its first half: the end of this wave's read phase of (st - 1, half 1)
#else
            if (st + 1 < ns && !(FIRST && pre1)) stage_step(st + 1, w_mine);
#endif
            if constexpr (FIRST && LN == 1) acc_prep();
            PP_WAIT_LGKM0();
            ITR(1);
            PP_BARRIER();
            ITR(2);
            if constexpr (FIRST && LN == 1) acc_init();
            if constexpr (HT) mma_half_ht(); else mma_half();
            ITR(3);
            PP_BARRIER();
            ITR(4);
            if constexpr (HT) read_half_ht(st, 1); else read_half(st, 1);
#if EILEV_PP4_DEEP
            PP_WAIT_LGKM0();  // unconditional and in straight-line code: a wait inside the branch below is not credited at the join
            if (late) {
                PP_WAIT_VM0();
                // full tile: the W rows this wave stages are read by itself (done: lgkmcnt(0) above) and by its early twin (done one
                // barrier ago) only.  Half tile: the reads are re-split 4 x 2, two late waves share W rows -> issue after the barrier.
                if constexpr (!HT) if (st + 2 < ns) stage_step(st + 2, w_mine, SPLIT_L ? 1 : 0);
            } else if (A3) {
                if (SPLIT_E && st + 2 < ns) stage_step(st + 2, w_mine, 2);
            } else if (SPLIT_E && st + 1 < ns && !(FIRST && pre1)) stage_step(st + 1, w_mine, 2);
            ITR(5);
            PP_BARRIER();
            ITR(6);
            if constexpr (HT) {
                if (late && st + 2 < ns) stage_step(st + 2, w_mine);
                mma_half_ht();
            } else mma_half();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (A3) {  // step st + 1 has landed; the 8 pieces of step st + 2 (this wave's most recent loads) stay in flight
                if (!late) {
                    if (st + 2 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else if (!late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ITR(7);
            PP_BARRIER();
#else
            PP_WAIT_LGKM0_VM0();
            PP_BARRIER();
            if constexpr (HT) mma_half_ht(); else mma_half();
            PP_BARRIER();
#endif
        };
        if (half_tile && M16K != 1) {
            if constexpr (M16K != 1) {
                kstep(0, std::true_type{}, std::true_type{});
                for (int st = 1; st < nsd; ++st) kstep(st, std::false_type{}, std::true_type{});
            }
        } else {
            kstep(0, std::true_type{}, std::false_type{});
            for (int st = 1; st < nsd; ++st) kstep(st, std::false_type{}, std::false_type{});
        }
        stamp(2);
#ifdef EILEV_PP4_ITRACE
        if (tracer && trace_i < g.trace_tiles) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                g.trace[(((size_t)blockIdx.x * 2 + (late ? 1 : 0)) * g.trace_tiles + trace_i) * 8 + k] = *reinterpret_cast<volatile unsigned long long *>(stg + k * 8);
        }
#endif  // (before the epilogue reuses the staging bytes)
        if (!late) PP_BARRIER();
        if constexpr (LN == 1) {  // rstd of the lane's rows for the epilogue: issued BEFORE the next tile's DMA (retire in order)
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)g.ln_rows, 0, g.M * 8, 0x00020000);
            if constexpr (M16) {
#pragma unroll
                for (int i = 0; i < TM16; ++i) ln_rs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (m0 + ((HT16 && half_tile) ? hm * 64 : wm * WM) + i * 16 + l15) * 8, 0, 0));
            } else {
#pragma unroll
            for (int u = 0; u < TM; ++u) ln_rs[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (m0 + wm * WM + u * 32 + l31) * 8, 0, 0));
            }
        }
        // every wave has finished reading both step buffers.  Lean tiles: both first K-steps of the next tile are staged now and
        // land under the epilogue; otherwise only step 0 (the general epilogue stages through buffer 1).
        const int cm0 = m0, cn0 = n0, tn = t + gridDim.x;
        const bool lean = lean_cur;
        pre1 = false;
        lean_cur = false;
        if (tn < ntiles) {
            set_tile(tn, m0, n0);
            stage_step(0, w_piece_mine(n0));
            pre1 = lean && ns > 1;
            if (pre1) stage_step(1, w_piece_mine(n0));
            lean_cur = M16 || is_lean(n0);
            ln_fetch(m0, n0);
        }
        stamp(3);
        if (HT16 && half_tile) {  // the wave's 64 x 64 block of the half tile, same lean epilogue
            if constexpr (HT16) {
                if constexpr (LN == 1) lean_epilogue(cm0, cn0, std::false_type{}, std::true_type{});
                else if constexpr (LN == 2) lean_epilogue(cm0, cn0, std::true_type{}, std::true_type{});
                else if (g.resid != nullptr) lean_epilogue(cm0, cn0, std::true_type{}, std::true_type{});
                else lean_epilogue(cm0, cn0, std::false_type{}, std::true_type{});
            }
        } else if (lean || M16) {  // LN kernels: the launcher guarantees ln_rows and no residual (1), stat_out and a residual (2); M16: lean tiles only
            if constexpr (LN == 1) lean_epilogue(cm0, cn0, std::false_type{}, std::false_type{});
            else if constexpr (LN == 2) lean_epilogue(cm0, cn0, std::true_type{}, std::false_type{});
            else if (g.resid != nullptr) lean_epilogue(cm0, cn0, std::true_type{}, std::false_type{});
            else lean_epilogue(cm0, cn0, std::false_type{}, std::false_type{});
        } else if constexpr (!M16) {
            if (half_tile) {
                gemm_epilogue<64, 64, EPI, 0, 2, LN, true>(g, reinterpret_cast<f32x16(&)[2][2]>(acc), smem + STEP, cm0, cn0, hm, hn, wid, lane);
            } else {
                gemm_epilogue<WM, WN, EPI, 0, TM / 2, LN, true>(g, acc, smem + STEP, cm0, cn0, wm, wn, wid, lane);
                gemm_epilogue<WM, WN, EPI, TM / 2, TM, LN, true>(g, acc, smem + STEP, cm0, cn0, wm, wn, wid, lane);
            }
        }
        stamp(4);
        stamp(6, true);
        ++trace_i;
    }
#undef PP_BARRIER
}

constexpr int PP4_SMEM = 2 * 65536 + 8 * 4096;  // two step buffers + 4 KiB of lean-epilogue staging per wave (the general epilogue
                                                // stages 69.6 KB from step buffer 1 on: 65536 + 69632 < 163840)

}  // namespace
