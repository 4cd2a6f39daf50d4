// gemm_w6.h — one wave per SIMD, 256 x 128 tiles, continuous K-step stream: the kernel of launches with few tiles (dispatch: gemm.hip).
#pragma once
#include "gemm_common.h"

namespace {

// ---- w6: one wave per SIMD, continuous K-step stream, lean chunked epilogue ---------------------------------------------
// 256 x 128 tile, 4 waves of 128 x 64, 3 LDS stages of one K-step of 64 (48 KiB each) + 4 KiB of output staging per wave.
// Per sub-step of 16 a wave issues 8 MFMAs and, slotted between them, the 6 fragment reads of the next sub-step and its
// share of the LDS-DMA two K-steps ahead (8 rows x 128 B pieces); ONE workgroup barrier per K-step.  The K-steps of
// consecutive tiles form one stream: the last steps of a tile already stage and read the next tile's first steps, so the
// epilogue runs while the next tile's operands land.  Operands are addressed through one buffer descriptor per tile (rows
// past M / N read zeros, no 2 GiB limit, nothing per-lane recomputed at a tile switch); the bias is folded into the
// accumulator init (C operand of the tile's first MFMAs, fetched with scalar loads); the finished accumulators move to a
// second register set (128 spare AGPRs) and are converted in chunks over units of 32 rows x 64 columns: activation, bf16,
// 4 KiB of LDS staging per wave, 128-byte-row buffer stores whose descriptor drops rows past M.
// Against the ping-pong kernel: main loop 1260 vs 1350 TFLOP/s (1.5x the DMA bytes per flop) but epilogue + tile switch
// cost 7 % instead of 16 %: +3-4 % on bias-only epilogues, equal on the GELU one.
__device__ __forceinline__ void w6_dma(__amdgpu_buffer_rsrc_t r, char *dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)dst, 16, voff, soff, 0, 0);
}

typedef int w6_i32x16 __attribute__((ext_vector_type(16)));

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_w6_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 128, WM = 128, WN = 64, TM = 4, TN = 2, NF = TM + TN;
    constexpr int STEP = (BM + BN) * 128, NST = 3;
    constexpr int CPC = EPI == 1 ? 4 + EILEV_GELU_DEG : 4;  // epilogue chunks per cell pair: prepare x 2, (GELU: one chunk per Horner step,) finish x 2
    constexpr int NCH = 4 * CPC + 8;        // chunks per 32-row unit: 4 cell pairs, 4 read-backs, 4 stores

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ns = g.K / 64;

    // Operand access: one buffer descriptor per tile and operand (base = first row of the tile, size = its valid rows, so rows
    // past M / N read zeros) + tile-independent per-lane offsets: nothing per-lane is recomputed at a tile switch
    unsigned pv[12];  // 8 pieces of A (8 rows x 128 B each), 4 of W: row * ld * 2 + swizzled 16-byte chunk
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wid * 8 + i) * 8 + (lane >> 3);
        pv[i] = (unsigned)row * (unsigned)(g.lda * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 8 + (lane >> 3);
        pv[8 + i] = (unsigned)row * (unsigned)(g.ldw * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    auto rsrc_a = [&](int m0) {
        const int rows = g.M - m0 < BM ? g.M - m0 : BM;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(g.A + (int64_t)m0 * g.lda), 0, rows * (int)(g.lda * 2), 0x00020000);
    };
    auto rsrc_w = [&](int n0) {
        const int rows = g.N - n0 < BN ? g.N - n0 : BN;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(g.W + (int64_t)n0 * g.ldw), 0, rows * (int)(g.ldw * 2), 0x00020000);
    };
    auto tile_origin = [&](int t, int &m0, int &n0) {
        int tm_i, tn_i;
        tile_coords(g, tiles_m, tiles_n, tm_i, tn_i, t);
        m0 = tm_i * BM;
        n0 = tn_i * BN;
    };
    auto piece = [&](__amdgpu_buffer_rsrc_t r_a, __amdgpu_buffer_rsrc_t r_w, int idx, int st, int soff) {
        if (idx < 8) w6_dma(r_a, smem + soff + (wid * 8 + idx) * 1024, pv[idx], st * 128);
        else w6_dma(r_w, smem + soff + BM * 128 + (wid * 4 + idx - 8) * 1024, pv[idx], st * 128);
    };
    f32x16 acc[TM][TN], accp[TM][TN];
    f32x16 cinit[TN];  // bias in the accumulator layout: the first MFMA of a tile takes it as its C operand
    bf16x8 f[2][NF];   // fragment sets: [.][0..3] activation rows (i), [.][4..5] weight rows (j)
    const int xo = (l31 >> 1) & 7;
    const int a_lane = (wm * WM + l31) * 128, b_lane = BM * 128 + (wn * WN + l31) * 128;

    // bias of the tile's 64 columns of this wave through scalar loads (not a vector-memory load: a vector load in the steady
    // state would make the compiler drain the LDS-DMA queue with vmcnt(0) before its first use)
    auto load_cinit = [&](int n0) {
        if (g.bias) {
            const bf16 *bp = g.bias + n0 + wn * WN;
            w6_i32x16 b0, b1;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=s"(b0), "=s"(b1) : "s"(bp) : "memory");
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // column j * 32 + 8 * (r >> 2) + 4 * hi + (r & 3): dword (col >> 1) of the 32, low / high half
                    const int d0 = (r >> 2) * 4 + ((r & 3) >> 1), d1 = d0 + 2;  // hi = 0 / hi = 1
                    const int lo = j == 0 ? b0[d0] : b1[d0], hv = j == 0 ? b0[d1] : b1[d1];
                    const unsigned w = (unsigned)(hi ? hv : lo);
                    cinit[j][r] = __builtin_bit_cast(float, (r & 1) ? (w & 0xffff0000u) : (w << 16));
                }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) cinit[j][r] = 0.0f;
        }
    };

    // ---- epilogue of the drained tile (accp), in chunks ----
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, 0, 0x00020000);
    f32x2 ex[4], eu[4], et[4], ep[4];
    bf16x8 erb[2];
    char *const stg = smem + NST * STEP + wid * 4096;  // 32 rows x 128 B, 16-byte chunk c of row r at chunk c ^ (r & 7)
    // cell (row l31, 16-byte chunk c, half hi) of the staging unit: stg_sw ^ (c << 4); the xor is an opaque asm so that the
    // compiler does not hoist the 8 per-cell addresses into 8 loop-invariant registers (it spilled them)
    const unsigned stg_sw = (unsigned)(NST * STEP + wid * 4096 + l31 * 128 + hi * 8) ^ (unsigned)((l31 & 7) << 4);
    const int srow = lane >> 3, schunk = lane & 7;
    const unsigned st_voff = (unsigned)srow * (unsigned)(g.ldc * 2) + schunk * 16;
    auto rsrc_c = [&](int m0, int n0) {  // this wave's 128 x 64 block of the output tile; rows past M are out of range: dropped
        const int r0 = m0 + wm * WM;
        const int rows = g.M - r0 < 0 ? 0 : (g.M - r0 < WM ? g.M - r0 : WM);
        const uint64_t base = (uint64_t)(reinterpret_cast<bf16 *>(g.C) + (int64_t)r0 * g.ldc + n0 + wn * WN);
        const uint64_t ub = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);  // provably wave-uniform: no waterfall loops
        return __builtin_amdgcn_make_buffer_rsrc((void *)ub, 0, __builtin_amdgcn_readfirstlane(rows * (int)(g.ldc * 2)), 0x00020000);
    };
    // residual (fc2 / proj): the unit's 32 rows x 64 columns go through the same staging rows first (coalesced 128-byte row
    // segments, same swizzle), each lane then adds its 8-byte cell in place.  (Vector loads: the compiler waits vmcnt(0) at
    // their first use, i.e. also for the next tile's first K-steps already in flight — they are due within a K-step anyway.)
    const bool has_res = g.resid != nullptr;
    __amdgpu_buffer_rsrc_t rr = rc;
    const unsigned rs_voff = (unsigned)srow * (unsigned)(g.ldr * 2) + schunk * 16;
    auto rsrc_r = [&](int m0, int n0) {
        const int r0 = m0 + wm * WM;
        const int rows = g.M - r0 < 0 ? 0 : (g.M - r0 < WM ? g.M - r0 : WM);
        const uint64_t base = (uint64_t)(g.resid + (int64_t)r0 * g.ldr + n0 + wn * WN);
        const uint64_t ub = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
        return __builtin_amdgcn_make_buffer_rsrc((void *)ub, 0, __builtin_amdgcn_readfirstlane(rows * (int)(g.ldr * 2)), 0x00020000);
    };
    auto stage_resid = [&](auto unit_c) {
        constexpr int U = decltype(unit_c)::value;
        typedef __attribute__((ext_vector_type(4))) unsigned w6_u32x4;
        w6_u32x4 rv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, rs_voff, (U * 32 + it * 8) * (int)(g.ldr * 2), 0);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + srow;
            *reinterpret_cast<w6_u32x4 *>(stg + row * 128 + ((schunk ^ (row & 7)) << 4)) = rv[it];
        }
    };
    auto epi_chunk = [&](auto unit_c, auto ch_c) {
        constexpr int U = decltype(unit_c)::value, CH = decltype(ch_c)::value;
        constexpr float gc[EILEV_GELU_DEG + 1] = EILEV_GELU_COEFFS;
        if constexpr (CH < 4 * CPC) {
            constexpr int CP = CH / CPC, SUB = CH % CPC, J = CP >> 1, QP = CP & 1;
            if constexpr (SUB < 2) {  // prepare half h = SUB (ReLU / GELU argument reduction)
                constexpr int h = SUB;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    f32x2 v = {accp[U][J][(2 * QP + h) * 4 + 2 * e2], accp[U][J][(2 * QP + h) * 4 + 2 * e2 + 1]};
                    if (EPI == 2) v = (f32x2){fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f)};
                    if (EPI == 1) {
                        eu[h * 2 + e2] = (f32x2){fminf(fabsf(v.x), EILEV_GELU_UMAX), fminf(fabsf(v.y), EILEV_GELU_UMAX)};
                        et[h * 2 + e2] = eu[h * 2 + e2] * (2.0f / EILEV_GELU_UMAX) + (-1.0f);
                        ep[h * 2 + e2] = (f32x2){gc[EILEV_GELU_DEG], gc[EILEV_GELU_DEG]};
                        v = (f32x2){fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f)};
                    }
                    ex[h * 2 + e2] = v;
                }
            }
            if constexpr (EPI == 1 && SUB >= 2 && SUB < 2 + EILEV_GELU_DEG) {
                constexpr int kk = EILEV_GELU_DEG + 1 - SUB;
#pragma unroll
                for (int n = 0; n < 4; ++n) ep[n] = ep[n] * et[n] + gc[kk];
            }
            if constexpr (SUB >= CPC - 2) {  // finish half h: (GELU: relu(x) - u p(t),) bf16, one 8-byte cell into the staging rows
                constexpr int h = SUB - (CPC - 2);
                f32x2 y0 = ex[h * 2], y1 = ex[h * 2 + 1];
                if (EPI == 1) {
                    y0 = y0 - ep[h * 2];
                    y1 = y1 - ep[h * 2 + 1];
                }
                constexpr int c = J * 4 + 2 * QP + h;
                unsigned ca;
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ca) : "n"(c << 4), "v"(stg_sw));
                if (has_res) {
                    const bf16x4 r4 = *reinterpret_cast<const bf16x4 *>(smem + ca);
                    y0 = y0 + (f32x2){(float)r4[0], (float)r4[1]};
                    y1 = y1 + (f32x2){(float)r4[2], (float)r4[3]};
                }
                *reinterpret_cast<bf16x4 *>(smem + ca) = (bf16x4){(bf16)y0.x, (bf16)y0.y, (bf16)y1.x, (bf16)y1.y};
            }
        } else {
            // read-backs and stores in the order R0 R1 S0 S1 R2 R3 S2 S3 (two buffers)
            constexpr int X = CH - 4 * CPC, IT = (X >> 2) * 2 + (X & 1);
            if constexpr ((X & 2) == 0) {
                const int row = IT * 8 + srow;
                erb[IT & 1] = *reinterpret_cast<const bf16x8 *>(stg + row * 128 + ((schunk ^ (row & 7)) << 4));
            } else {
                typedef __attribute__((ext_vector_type(4))) unsigned w6_u32x4;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w6_u32x4, erb[IT & 1]), rc, st_voff, (U * 32 + IT * 8) * (int)(g.ldc * 2), 0);
            }
        }
    };
    // One sub-step: the 8 MFMAs out of fragment set MSET (MMA = 2: first sub-step of a tile, C = bias); between them the 6 reads
    // of sub-step RSUB (stage offset rs) into the other set and NDMA pieces (first index DFIRST) of K-step dst_st into stage
    // offset ds
    auto phase = [&](auto mset_c, auto rsub_c, auto dfirst_c, auto ndma_c, auto mma_c, auto rd_c, auto dma_c, int rs,
                     __amdgpu_buffer_rsrc_t r_a, __amdgpu_buffer_rsrc_t r_w, int dst_st, int ds) {
        constexpr int MSET = decltype(mset_c)::value, RSUB = decltype(rsub_c)::value, DFIRST = decltype(dfirst_c)::value,
                      NDMA = decltype(ndma_c)::value;
        constexpr int MMA = decltype(mma_c)::value;  // 0: no MFMAs, 1: accumulate, 2: first sub-step of a tile (C = bias)
        constexpr bool RD = decltype(rd_c)::value, DMA = decltype(dma_c)::value;
        constexpr int RSET = MSET ^ 1;
        constexpr int RORD[NF] = {TM + 0, 0, TM + 1, 1, 2, 3};  // read order = consumption order of the MFMAs (i-major)
        const int co = ((RSUB * 2 + hi) ^ xo) << 4;
        const char *pa_ = smem + rs + a_lane + co, *pb_ = smem + rs + b_lane + co;
        static_for<TM * TN>([&](auto q_c) {
            constexpr int q = decltype(q_c)::value, i = q / TN, j = q % TN;
            if constexpr (MMA != 0) {
                if constexpr (MMA == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[MSET][TM + j], f[MSET][i], cinit[j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[MSET][TM + j], f[MSET][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (RD && q < NF) {
                constexpr int fi = RORD[q];
                f[RSET][fi] = *reinterpret_cast<const bf16x8 *>((fi < TM ? pa_ + fi * 4096 : pb_ + (fi - TM) * 4096));
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DMA && q >= TM * TN - NDMA) {
                piece(r_a, r_w, DFIRST + q - (TM * TN - NDMA), dst_st, ds);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I6 = std::integral_constant<int, 6>;
    using I9 = std::integral_constant<int, 9>;
    using T = std::true_type;
    using F = std::false_type;
#define W6_BARRIER()                       \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

    int t = blockIdx.x, m0, n0, m1 = 0, n1 = 0;
    if (t >= ntiles) return;
    tile_origin(t, m0, n0);
    __amdgpu_buffer_rsrc_t ra = rsrc_a(m0), rw = rsrc_w(n0), ra1 = ra, rw1 = rw;
    int so = 0;  // stage offset of the current K-step (rotates through the 3 buffers across tiles)
#pragma unroll
    for (int i = 0; i < 12; ++i) piece(ra, rw, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) piece(ra, rw, i, 1, STEP);
#pragma unroll
    for (int i = 0; i < 3; ++i) piece(ra, rw, i, 2, 2 * STEP);
    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    W6_BARRIER();
    phase(I1{}, I0{}, I0{}, I0{}, I0{}, T{}, F{}, 0, ra, rw, 0, 0);  // fragments of (step 0, sub-step 0) -> set 0
    // One K-step s.  (r2a, r2w) / k2: descriptors and K-step index of the step two ahead (its pieces 3..11 are issued here), r3* /
    // k3: three ahead (pieces 0..2); D2 / D3: those steps exist; N1: the next step exists (its first fragments are read here);
    // Z: first step of a tile.  The barrier X sits between sub-steps 2 and 3: my pieces of step s + 1 have landed (the 12 of
    // step s + 2 may still fly), every wave has read all of step s (its buffer takes step s + 3).
    auto kstep = [&](auto d2_c, auto d3_c, auto n1_c, auto z_c, __amdgpu_buffer_rsrc_t r2a, __amdgpu_buffer_rsrc_t r2w, int k2,
                     __amdgpu_buffer_rsrc_t r3a, __amdgpu_buffer_rsrc_t r3w, int k3) {
        constexpr bool D2 = decltype(d2_c)::value;
        using M0 = std::integral_constant<int, decltype(z_c)::value ? 2 : 1>;
        const int so1 = so + STEP >= NST * STEP ? so + STEP - NST * STEP : so + STEP;
        const int so2 = so1 + STEP >= NST * STEP ? so1 + STEP - NST * STEP : so1 + STEP;
        phase(I0{}, I1{}, I3{}, I3{}, M0{}, T{}, d2_c, so, r2a, r2w, k2, so2);
        phase(I1{}, I2{}, I6{}, I3{}, I1{}, T{}, d2_c, so, r2a, r2w, k2, so2);
        phase(I0{}, I3{}, I9{}, I3{}, I1{}, T{}, d2_c, so, r2a, r2w, k2, so2);
        if (D2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W6_BARRIER();
        phase(I1{}, I0{}, I0{}, I3{}, I1{}, n1_c, d3_c, so1, r3a, r3w, k3, so);
        so = so1;
    };
    for (;;) {
        const int tn = t + gridDim.x;
        const bool has_next = tn < ntiles;
        load_cinit(n0);  // (kept out of the previous tile's tail: 32 more live registers there spill)
        __builtin_amdgcn_sched_barrier(0);
        kstep(T{}, T{}, T{}, T{}, ra, rw, 2, ra, rw, 3);
        int st = 1;
        for (; st < ns - 3; ++st) kstep(T{}, T{}, T{}, F{}, ra, rw, st + 2, ra, rw, st + 3);
        if (has_next) {
            tile_origin(tn, m1, n1);
            ra1 = rsrc_a(m1);
            rw1 = rsrc_w(n1);
            kstep(T{}, T{}, T{}, F{}, ra, rw, ns - 1, ra1, rw1, 0);
            kstep(T{}, T{}, T{}, F{}, ra1, rw1, 0, ra1, rw1, 1);
            kstep(T{}, T{}, T{}, F{}, ra1, rw1, 1, ra1, rw1, 2);
        } else {
            kstep(T{}, F{}, T{}, F{}, ra, rw, ns - 1, ra, rw, 0);
            kstep(F{}, F{}, T{}, F{}, ra, rw, 0, ra, rw, 0);
            kstep(F{}, F{}, F{}, F{}, ra, rw, 0, ra, rw, 0);
        }
        // hand the finished accumulators to the drain set (the next tile's first MFMAs do not wait for the conversion below)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) accp[i][j] = acc[i][j];
        rc = rsrc_c(m0, n0);
        if (has_res) rr = rsrc_r(m0, n0);
        static_for<4>([&](auto u_c) {
            if (has_res) stage_resid(u_c);
            static_for<NCH>([&](auto c_c) { epi_chunk(u_c, c_c); });
        });
        if (!has_next) break;
        ra = ra1;
        rw = rw1;
        m0 = m1;
        n0 = n1;
        t = tn;
    }
#undef W6_BARRIER
}

int launch_w6(const GemmArgs &g, hipStream_t s) {
    static bool attr_set = false;
    static int num_cu = 0;
    constexpr int smem = 3 * 49152 + 4 * 4096;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_w6_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_w6_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_w6_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int dev = 0;
        EILEV_HIP_CHECK(hipGetDevice(&dev));
        EILEV_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_set = true;
    }
    const int tiles = ((g.M + 255) / 256) * ((g.N + 127) / 128);
    const int ncu = eilev_grid_cus() < num_cu ? eilev_grid_cus() : num_cu;
    const int grid = tiles < ncu ? tiles : ncu / 8 * 8;
    if (g.epi == 1) hipLaunchKernelGGL(gemm_w6_kernel<1>, dim3(grid), dim3(256), smem, s, g);
    else if (g.epi == 2) hipLaunchKernelGGL(gemm_w6_kernel<2>, dim3(grid), dim3(256), smem, s, g);
    else hipLaunchKernelGGL(gemm_w6_kernel<0>, dim3(grid), dim3(256), smem, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}


}  // namespace
