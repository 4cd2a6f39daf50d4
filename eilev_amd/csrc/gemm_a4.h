// gemm_a4.h — persistent 256 x 256-tile GEMM with ONE wave per SIMD (4 waves x 128 x 128) and a hand-scheduled K loop.
// Included by gemm.hip when EILEV_GEMM_PART == 3 (object gemm_a4.o), inside its anonymous namespace, after gemm_epilogue.
//
// Why (round 3): rocprofv3 PMC of the vendor library's kernel next to gemm_pp4_kernel on the ViT shapes (profiles/r03_yardstick_pmc.txt)
// shows the same shader clock (1.6-1.75 GHz) and the same fabric traffic, but the matrix pipe 71-80 % busy against 62-65 %.  pp4's 8 waves
// of 128 x 64 read 192 KB of fragments per K-step from LDS and land 64 KB of LDS-DMA beside them; waves of 128 x 128 read 128 KB, and
// holding a whole K-step of fragments in registers frees an LDS buffer a quarter into the step, so the DMA of step s + 2 starts
// there: ~1.2 K-steps between issue and first use with two 64-KiB buffers.  hipcc cannot allocate 256 accumulator AGPRs + 128 fragment
// VGPRs without spilling, so the K loop is inline asm with fixed registers (gen_a4_loop.py -> gemm_a4_loop.inc); tile
// walk, operand descriptors and the epilogues stay C++.
//
// LDS: two K-step buffers of 64 KiB (A rows 0..255 then W rows 0..255, 128 B per row, 16-byte chunk c of row r at c ^ ((r >> 1) & 7))
// + 8 KiB of epilogue staging per wave = 160 KiB.  Both first K-steps of the NEXT tile are staged before the epilogue starts.
#include "gemm_a4_loop.inc"

// EPI: 0 none, 1 GELU (fast form), 2 ReLU.  VAR: schedule variant of the K loop (probe).  LN: 0 plain; 2 residual epilogue that also emits the
// row statistics of what it writes (GemmArgs::stat_out: per 64-column slot (sum, sum of squares), the LayerNorm-folding producer of proj / fc2).
// Requires N % 128 == 0 (a last column tile is whole or exactly its first 128 columns) and K % 64 == 0, K >= 192.
// RES: the launch has a residual (compile-time: each epilogue form the kernel carries costs registers across the K loop).
template <int EPI, int VAR, int LN, bool RES>
__global__ __launch_bounds__(256, 1) void gemm_a4_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, WM = 128, WN = 128, TM = 4;
    constexpr int STEP = 65536;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ns = g.K / 64;
    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)smem;

    // LDS-DMA: piece i (0..7) of this wave = rows wid * 64 + i * 8 .. + 7 of the A (W) tile, 8 rows x 128 B; lane -> (row prow, chunk slot
    // pslot); the source chunk is swizzled so that LDS holds chunk c of row r at slot c ^ ((r >> 1) & 7): (r >> 1) & 7 = (i & 1) * 4 + (prow >> 1)
    const unsigned va_e = (unsigned)(wid * 64 + prow) * (unsigned)(g.lda * 2) + ((pslot ^ (prow >> 1)) << 4);
    const unsigned va_o = (unsigned)(wid * 64 + prow) * (unsigned)(g.lda * 2) + ((pslot ^ (4 + (prow >> 1))) << 4);
    const unsigned vw_e = (unsigned)(wid * 64 + prow) * (unsigned)(g.ldw * 2) + ((pslot ^ (prow >> 1)) << 4);
    const unsigned vw_o = (unsigned)(wid * 64 + prow) * (unsigned)(g.ldw * 2) + ((pslot ^ (4 + (prow >> 1))) << 4);
    const int stride_a = __builtin_amdgcn_readfirstlane((int)(g.lda * 16)), stride_w = __builtin_amdgcn_readfirstlane((int)(g.ldw * 16));
    const int dst0 = __builtin_amdgcn_readfirstlane((int)lds0 + wid * 8192);
    // fragment reads: lane (l31, hi) reads row l31 of a 32-row block, 16-byte chunk (k-slice * 2 + hi) ^ swizzle: k-slice k = XOR with k << 5
    const unsigned va_rd = lds0 + (unsigned)((wm * WM + l31) * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4));
    const unsigned vw_rd = lds0 + (unsigned)(BM * 128 + (wn * WN + l31) * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4));
    // half tiles (only columns 0..127 of the tile exist): every wave owns 128 rows x 64 columns, W rows wn * 64 ..
    const unsigned vw_rd_ht = lds0 + (unsigned)(BM * 128 + (wn * 64 + l31) * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4));

    auto uniform_rsrc = [&](const void *ptr, int bytes) {
        const uint64_t base = (uint64_t)ptr;
        const uint64_t ub = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);  // provably wave-uniform: no waterfall loops
        return __builtin_amdgcn_make_buffer_rsrc((void *)ub, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    // one descriptor per tile and operand: base = first row of the tile, size = its valid rows (rows past M / N read zeros; no 2 GiB limit)
    auto rsrc_a = [&](int m0) {
        const int rows = g.M - m0 < BM ? g.M - m0 : BM;
        return uniform_rsrc(g.A + (int64_t)m0 * g.lda, rows * (int)(g.lda * 2));
    };
    auto rsrc_w = [&](int n0) {
        const int rows = g.N - n0 < BN ? g.N - n0 : BN;
        return uniform_rsrc(g.W + (int64_t)n0 * g.ldw, rows * (int)(g.ldw * 2));
    };
    auto tile_origin = [&](int t, int &m0, int &n0) {
        int tm_i, tn_i;
        tile_coords(g, tiles_m, tiles_n, tm_i, tn_i, t);
        m0 = tm_i * BM;
        n0 = tn_i * BN;
    };
    auto stage = [&](__amdgpu_buffer_rsrc_t r_a, __amdgpu_buffer_rsrc_t r_w, int st, int buf) {
        char *da = smem + buf * STEP + wid * 8192, *dw = da + BM * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (lds_void *)(da + i * 1024), 16, ((i & 1) ? va_o : va_e) + (unsigned)(i * stride_a), st * 128, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void *)(dw + i * 1024), 16, ((i & 1) ? vw_o : vw_e) + (unsigned)(i * stride_w), st * 128, 0, 0);
        }
    };

    f32x16 acc[2][TM][2];  // [column half jh][i][j]; asm operand i * 4 + jh * 2 + j
    // ---- epilogues ----------------------------------------------------------------------------------------------------------------
    // acc[jh][i][j][r] = C[m = i*32 + (lane & 31)][n = jh*64 + j*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5)] of the wave's 128 x 128 block.
    // Lean epilogue (interior column tiles; rows past M are dropped by the output descriptor): the design of gemm_pp4_kernel's — units
    // of 32 rows x 64 columns through 4 KiB of private LDS staging OUTSIDE the step buffers (16-byte chunk c of row r at c ^ (r & 7)),
    // bias / activation / residual in registers, read back as 128-byte row segments, 16-byte buffer stores — here 8 units per wave.
    char *const stg = smem + 2 * STEP + wid * 8192;  // two staging units of 4 KiB: unit X uses X & 1
    const unsigned stg_sw = (unsigned)(2 * STEP + wid * 8192 + l31 * 128 + hi * 8) ^ (unsigned)((l31 & 7) << 4);
    const int srow = lane >> 3, schunk = lane & 7;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    // 8 units per wave (X = jh * 4 + U: column half jh, rows U * 32 ..), SOFTWARE-PIPELINED over the two staging units: the cells of unit
    // X + 1 are computed and written while the row segments of unit X come back from LDS and leave — with one wave per SIMD nothing else
    // would cover the LDS write -> read -> store chain of a unit.
    auto lean_epilogue = [&](int cm0, int cn0, auto ht_c) {
        constexpr bool HT = decltype(ht_c)::value;  // half tile: 4 units (jh = 0), the wave's columns are wn * 64 ..
        constexpr int NU = HT ? 4 : 8;
        constexpr bool LNP = LN == 2 && RES;
        const int r0 = cm0 + wm * WM, c0 = cn0 + wn * (HT ? 64 : WN);
        const int rows = g.M - r0 < 0 ? 0 : (g.M - r0 < WM ? g.M - r0 : WM);
        const __amdgpu_buffer_rsrc_t rc = uniform_rsrc(reinterpret_cast<bf16 *>(g.C) + (int64_t)r0 * g.ldc + c0, rows * (int)(g.ldc * 2));
        const __amdgpu_buffer_rsrc_t rr = uniform_rsrc(RES ? g.resid + (int64_t)r0 * g.ldr + c0 : g.A, RES ? rows * (int)(g.ldr * 2) : 0);
        const unsigned st_voff = (unsigned)srow * (unsigned)(g.ldc * 2) + schunk * 16, rs_voff = (unsigned)srow * (unsigned)(g.ldr * 2) + schunk * 16;
        // residual rows of two units in flight.  (Measured alternatives, both slower through register pressure: all eight units requested
        // before the next tile's LDS-DMA pieces (128 VGPRs, 220 spills: fc2 1072 TFLOP/s), four up front + four as they free (64 VGPRs, 155
        // spills: 1141) against 1152 for this form — whose loads queue behind the 32 pieces: vector-memory returns are in order.  The
        // residual epilogue is where this kernel loses to the ping-pong kernel, whose second wave per SIMD covers that wait.)
        u32x4_t rv[2][4];
        auto res_load = [&](auto x_c) {
            constexpr int X = decltype(x_c)::value, JH = X >> 2, U = X & 3;
#pragma unroll
            for (int it = 0; it < 4; ++it)
                rv[X & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(rr, rs_voff + JH * 128, (U * 32 + it * 8) * (int)(g.ldr * 2), 0);
        };
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
        __amdgpu_buffer_rsrc_t rst[2] = {rc, rc};  // LN == 2: one statistics slot per 64 columns, rows past M dropped
        if constexpr (LNP) {
            rst[0] = uniform_rsrc(g.stat_out + ((int64_t)(c0 >> 6) * g.stat_ld + r0) * 2, rows * 8);
            rst[1] = uniform_rsrc(g.stat_out + ((int64_t)((c0 >> 6) + 1) * g.stat_ld + r0) * 2, rows * 8);
        }
        bf16x4 biasr[2][2][4];  // [jh][j][q]: columns jh*64 + j*32 + q*8 + hi*4 + (0..3): the accumulator layout
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (g.bias && !(HT && jh == 1)) biasr[jh][j][q] = *reinterpret_cast<const bf16x4 *>(g.bias + c0 + jh * 64 + j * 32 + q * 8 + hi * 4);
                    else biasr[jh][j][q] = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
                }
        auto cells = [&](auto x_c) {
            constexpr int X = decltype(x_c)::value, JH = X >> 2, U = X & 3, SB = (X & 1) * 4096;
            bf16x4 rcell[2][4];
            if constexpr (RES) {  // the unit's residual rows -> staging (coalesced); every lane then fetches its 8 cells in one batch
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + srow;
                    *reinterpret_cast<u32x4_t *>(stg + SB + row * 128 + ((schunk ^ (row & 7)) << 4)) = rv[X & 1][it];
                }
                if constexpr (X + 2 < NU) res_load(std::integral_constant<int, X + 2>{});
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned ca;
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ca) : "n"(((j * 4 + q) << 4) + SB), "v"(stg_sw));
                        rcell[j][q] = *reinterpret_cast<const bf16x4 *>(smem + ca);
                    }
            }
            float st1 = 0.0f, st2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[JH][U][j][q * 4 + e] + (float)biasr[JH][j][q][e];
                    if constexpr (EPI == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    if constexpr (EPI == 1) gelu_erf_n<4>(v);
                    if constexpr (RES) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rcell[j][q][e];
                    }
                    if constexpr (LNP) {  // statistics of the fp32 values (gemm_pp4_kernel's producer: same values, same order)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            st1 += v[e];
                            st2 = fmaf(v[e], v[e], st2);
                        }
                    }
                    unsigned ca;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ca) : "n"(((j * 4 + q) << 4) + SB), "v"(stg_sw));
                    *reinterpret_cast<bf16x4 *>(smem + ca) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                }
            if constexpr (LNP) {  // the two lane halves hold the two column halves of a row
                const f32x2_t t2 = (f32x2_t){st1 + __shfl_xor(st1, 32), st2 + __shfl_xor(st2, 32)};
                if (hi == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, t2), rst[JH], (U * 32 + l31) * 8, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto flush = [&](auto x_c) {
            constexpr int X = decltype(x_c)::value, JH = X >> 2, U = X & 3, SB = (X & 1) * 4096;
            bf16x8 erb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int row = b * 8 + srow;
                erb[b] = *reinterpret_cast<const bf16x8 *>(stg + SB + row * 128 + ((schunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, erb[b]), rc, st_voff + JH * 128, (U * 32 + b * 8) * (int)(g.ldc * 2), EILEV_ST_AUX);
            // (gfx950, r2: a VALU write to the data registers of a 128-bit buffer store issued the cycle before corrupted the stored chunk:
            // keep the scheduler out, two idle states between the last store and whatever reuses its registers — see gemm_pp4_kernel)
            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (RES) {
            res_load(std::integral_constant<int, 0>{});
            res_load(std::integral_constant<int, 1>{});
        }
        cells(std::integral_constant<int, 0>{});
        static_for<NU>([&](auto x_c) {
            constexpr int X = decltype(x_c)::value;
            if constexpr (X + 1 < NU) cells(std::integral_constant<int, X + 1>{});
            flush(x_c);
        });
    };
    int t = blockIdx.x, m0, n0;
    if (t >= ntiles) return;
    tile_origin(t, m0, n0);
    __amdgpu_buffer_rsrc_t ra = rsrc_a(m0), rw = rsrc_w(n0);
    stage(ra, rw, 0, 0);
    stage(ra, rw, 1, 1);
    for (; t < ntiles; t += gridDim.x) {
#define A4_ASM(TEXT)                                                                                                                              \
    asm volatile(TEXT                                                                                                                             \
                 : "=a"(acc[0][0][0]), "=a"(acc[0][0][1]), "=a"(acc[1][0][0]), "=a"(acc[1][0][1]), "=a"(acc[0][1][0]), "=a"(acc[0][1][1]),       \
                   "=a"(acc[1][1][0]), "=a"(acc[1][1][1]), "=a"(acc[0][2][0]), "=a"(acc[0][2][1]), "=a"(acc[1][2][0]), "=a"(acc[1][2][1]),     \
                   "=a"(acc[0][3][0]), "=a"(acc[0][3][1]), "=a"(acc[1][3][0]), "=a"(acc[1][3][1])                                               \
                 : "v"(va_rd), "v"(vw_sel), "v"(va_e), "v"(va_o), "v"(vw_e), "v"(vw_o), "s"(ra), "s"(rw), "s"(stride_a), "s"(stride_w),         \
                   "s"(ns), "s"(dst0), "s"(form)                                                                                                  \
                 : A4_LOOP_CLOBBERS)
        const bool half = n0 + 128 >= g.N;  // N % 128 == 0: the last column tile is whole or exactly its first half
        // half tile: every wave 128 x 64 (operands i*4 + j, j < 2 = acc[0]); waves 2, 3 own W rows that do not exist: no W pieces
        const int form = half ? (wid >= 2 ? 2 : 1) : 0;
        const unsigned vw_sel = half ? vw_rd_ht : vw_rd;
        if constexpr (VAR == 0) A4_ASM(A4_LOOP_V0);
        else A4_ASM(A4_LOOP_V1);
#undef A4_ASM
        // the loop ends with every wave's own fragment reads complete; the barrier makes that true of all four before a buffer is re-staged
        __builtin_amdgcn_s_barrier();
        const int cm0 = m0, cn0 = n0, tn = t + gridDim.x;
        if (tn < ntiles) {  // both first K-steps of the next tile land under the epilogue (its staging is outside the step buffers)
            tile_origin(tn, m0, n0);
            ra = rsrc_a(m0);
            rw = rsrc_w(n0);
            stage(ra, rw, 0, 0);
            stage(ra, rw, 1, 1);
        }
        if (g.dbg & 1024) {  // probe: no epilogue at all (keep the accumulators alive)
            float keep = 0.0f;
#pragma unroll
            for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) keep += acc[jh][i][j][0] + acc[jh][i][j][7] + acc[jh][i][j][15];
            if (keep == 123.456f) reinterpret_cast<float *>(g.C)[0] = keep;
        } else if (half) lean_epilogue(cm0, cn0, std::true_type{});
        else lean_epilogue(cm0, cn0, std::false_type{});
    }
}

template <int EPI, int VAR, int LN, bool RES>
static int launch_a4_i(const GemmArgs &g, int grid, hipStream_t s) {
    constexpr int smem = 2 * 65536 + 4 * 8192;
    static bool attr_set = false;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_a4_kernel<EPI, VAR, LN, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_a4_kernel<EPI, VAR, LN, RES>), dim3(grid), dim3(256), smem, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
