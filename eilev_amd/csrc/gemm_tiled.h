// gemm_tiled.h — per-tile GEMM kernels: the register-staged reference kernel (any K % 8 == 0) and the LDS-DMA kernel (K % 64 == 0) with its
// tile shapes; what shapes with few tiles, odd K or fp32 / patch-remap epilogues run (dispatch: gemm.hip).
#pragma once
#include "gemm_common.h"

namespace {

template <int BM, int BN, int NWM, int NWN, int EPI>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_nt_kernel(const GemmArgs g) {
    constexpr int NW = NWM * NWN, NT = 64 * NW;
    constexpr int WM = BM / NWM, WN = BN / NWN;
    constexpr int TM = WM / 32, TN = WN / 32;              // 32x32 MFMA tiles per wave
    constexpr int A_CH = BM * 8 / NT, B_CH = BN * 8 / NT;  // 16-byte chunks per thread per K-step
    constexpr int STAGE = (BM + BN) * 128;                 // bytes per LDS stage
    static_assert(A_CH >= 1 && B_CH >= 1, "tile too small for the workgroup");
    static_assert(WN == 64, "the epilogue stores 128-byte row segments per wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    int tm_i, tn_i;
    tile_coords(g, (g.M + BM - 1) / BM, (g.N + BN - 1) / BN, tm_i, tn_i);
    const int m0 = tm_i * BM, n0 = tn_i * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / NWN, wn = wid % NWN;
    const int l31 = lane & 31, hi = lane >> 5;

    bf16x8 ra[A_CH], rb[B_CH];
    const int nk = (g.K + BK - 1) / BK;

    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int id = tid + i * NT, row = id >> 3, c = id & 7;
            int gr = m0 + row;
            gr = gr < g.M ? gr : g.M - 1;
            const int k = kt * BK + c * 8;
            ra[i] = (k < g.K) ? *reinterpret_cast<const bf16x8 *>(g.A + (int64_t)gr * g.lda + k) : zero8();
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int id = tid + i * NT, row = id >> 3, c = id & 7;
            int gr = n0 + row;
            gr = gr < g.N ? gr : g.N - 1;
            const int k = kt * BK + c * 8;
            rb[i] = (k < g.K) ? *reinterpret_cast<const bf16x8 *>(g.W + (int64_t)gr * g.ldw + k) : zero8();
        }
    };
    auto swrite = [&](int buf) {
        char *sa = smem + buf * STAGE, *sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int id = tid + i * NT, row = id >> 3, c = id & 7;
            *reinterpret_cast<bf16x8 *>(sa + row * 128 + swz(row, c)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int id = tid + i * NT, row = id >> 3, c = id & 7;
            *reinterpret_cast<bf16x8 *>(sb + row * 128 + swz(row, c)) = rb[i];
        }
    };

    // acc[i][j][reg] = C[m = i*32 + (lane & 31)][n = j*32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)]
    // (operands are swapped — MFMA rows are weight rows — so a lane owns runs of 4 consecutive n)
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    gload(0);
    swrite(0);
    __syncthreads();

    for (int kt = 0; kt < ((g.dbg & 2) ? 1 : nk); ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);  // in flight while the MFMAs below run
        const char *sa = smem + cur * STAGE + (wm * WM) * 128;
        const char *sb = smem + cur * STAGE + BM * 128 + (wn * WN) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // 4 x (k = 16) per 64-wide K-step
            bf16x8 bfr[TN];
            const int kc = ks * 2 + hi;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = j * 32 + l31;
                bfr[j] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + swz(row, kc));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = i * 32 + l31;
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + swz(row, kc));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af, acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) swrite(cur ^ 1);  // the other stage was last read before the previous barrier
        __syncthreads();
    }

    gemm_epilogue<WM, WN, EPI>(g, acc, smem, m0, n0, wm, wn, wid, lane);
}

// ---- fast path: direct-to-LDS staging (K % 64 == 0) -------------------------------------------------
// Same tile geometry, but the K-step tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA): no
// staging VGPRs and no ds_write pass.  The DMA writes LDS linearly (wave-uniform base + lane * 16), so
// the XOR swizzle is applied to the per-lane SOURCE address instead (lane p of an 8-row x 128-byte piece
// fetches chunk (p & 7) ^ f(row)); fragment reads use the same involution.  Rows past M / N are clamped
// to the last valid row (their products are never stored).
typedef __attribute__((address_space(1))) const void glb_void;

// One 1-KiB LDS-DMA piece through a buffer descriptor: lane i fetches 16 bytes at base + voff + soff and the wave
// writes 64 x 16 bytes linearly at `dst` (wave-uniform).  Non-template helper on purpose: with ROCm 7.2 the host
// pass silently drops the stub of a kernel TEMPLATE that calls this builtin in a dependent context.
__device__ __forceinline__ void lds_dma16(const void *base, char *dst, unsigned voff, int soff) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)dst, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int NWM, int NWN, int EPI, int NSTAGE, int MINW, int PRIO>
__global__ __launch_bounds__(64 * NWM * NWN, MINW) void gemm_glds_kernel(const GemmArgs g) {
    constexpr int NW = NWM * NWN;
    constexpr int WM = BM / NWM, WN = BN / NWN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_PC = BM / 8 / NW, B_PC = BN / 8 / NW;  // 1-KiB pieces (8 rows) per wave per K-step
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(A_PC >= 1 && B_PC >= 1 && WN == 64, "unsupported geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    int tm_i, tn_i;
    tile_coords(g, (g.M + BM - 1) / BM, (g.N + BN - 1) / BN, tm_i, tn_i);
    const int m0 = tm_i * BM, n0 = tn_i * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / NWN, wn = wid % NWN;
    const int l31 = lane & 31, hi = lane >> 5;
    // split-K (weight gradients: few output tiles, tens of thousands of rows to contract): slice blockIdx.y owns K-steps
    // [kbase, kbase + nk) and adds its partial sums to the zero-initialised f32 output
    const int kbase = g.k_slice > 0 ? (int)blockIdx.y * g.k_slice : 0;
    const int nk = g.k_slice > 0 ? min(g.k_slice, g.K / BK - kbase) : g.K / BK;

    // per-lane byte offsets of this wave's pieces; the LDS-DMA goes through buffer descriptors (one s_mov m0 +
    // one buffer_load ... lds per piece, K advance in the scalar offset: no 64-bit VALU address arithmetic)
    unsigned pa[A_PC], pb[B_PC];
    const int prow = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int i = 0; i < A_PC; ++i) {
        const int row = (wid * A_PC + i) * 8 + prow;  // tile row of this lane's LDS slot
        int gr = m0 + row;
        gr = gr < g.M ? gr : g.M - 1;
        pa[i] = (unsigned)gr * (unsigned)(g.lda * 2) + ((pslot ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < B_PC; ++i) {
        const int row = (wid * B_PC + i) * 8 + prow;
        int gr = n0 + row;
        gr = gr < g.N ? gr : g.N - 1;
        pb[i] = (unsigned)gr * (unsigned)(g.ldw * 2) + ((pslot ^ ((row >> 1) & 7)) << 4);
    }
    auto stage_in = [&](int buf, int kt) {
        char *sa = smem + buf * STAGE + (wid * A_PC) * 1024;
        char *sb = smem + buf * STAGE + BM * 128 + (wid * B_PC) * 1024;
#pragma unroll
        for (int i = 0; i < A_PC; ++i)
            lds_dma16(g.A, sa + i * 1024, pa[i], (kbase + kt) * (BK * 2));
#pragma unroll
        for (int i = 0; i < B_PC; ++i)
            lds_dma16(g.W, sb + i * 1024, pb[i], (kbase + kt) * (BK * 2));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto compute = [&](int buf) {
        const char *sa = smem + buf * STAGE + (wm * WM) * 128;
        const char *sb = smem + buf * STAGE + BM * 128 + (wn * WN) * 128;
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 bfr[TN];
            const int kc = ks * 2 + hi;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = j * 32 + l31;
                bfr[j] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + swz(row, kc));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = i * 32 + l31;
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + swz(row, kc));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af, acc[i][j], 0, 0, 0);
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    const int nkd = (g.dbg & 2) ? 1 : nk;
    // Half-empty last column tile (N % 256 <= 128, e.g. N = 1408 = 5.5 x 256): only columns [0, 128) of the tile exist.
    // Instead of letting the waves of the two right-hand column blocks multiply padding, the 8 waves re-split the valid
    // 256 x 128 region as 4 x 2 blocks of 64 x 64: half the MFMAs per wave, no DMA for the missing W rows.
    constexpr bool HALF_OK = BM == 256 && BN == 256 && NWM == 2 && NWN == 4 && NSTAGE == 2;
    const bool half_tile = HALF_OK && n0 + 128 >= g.N && !(g.dbg & 524288);
    if (HALF_OK && half_tile) {
        const int hm = wm * 2 + (wn >> 1), hn = wn & 1;  // 64-row block, 64-column block of this wave
        auto stage_half_tile = [&](int buf, int kt) {
            char *sa = smem + buf * STAGE + (wid * A_PC) * 1024;
            char *sb = smem + buf * STAGE + BM * 128 + (wid * B_PC) * 1024;
#pragma unroll
            for (int i = 0; i < A_PC; ++i) lds_dma16(g.A, sa + i * 1024, pa[i], (kbase + kt) * (BK * 2));
            if (wid < NW / 2) {  // W rows 128..255 of the tile are beyond N
#pragma unroll
                for (int i = 0; i < B_PC; ++i) lds_dma16(g.W, sb + i * 1024, pb[i], (kbase + kt) * (BK * 2));
            }
        };
        auto compute_half = [&](int buf) {
            const char *sa = smem + buf * STAGE + (hm * 64) * 128;
            const char *sb = smem + buf * STAGE + BM * 128 + (hn * 64) * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 bfr[2];
                const int kc = ks * 2 + hi;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = j * 32 + l31;
                    bfr[j] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + swz(hn * 64 + row, kc));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = i * 32 + l31;
                    const bf16x8 af = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + swz(hm * 64 + row, kc));
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af, acc[i][j], 0, 0, 0);
                }
            }
        };
        stage_half_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nkd; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage_half_tile(cur ^ 1, kt + 1);
            compute_half(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        gemm_epilogue<64, 64, EPI>(g, reinterpret_cast<f32x16(&)[2][2]>(acc), smem, m0, n0, hm, hn, wid, lane);
        return;
    }
    if (NSTAGE == 2) {
        stage_in(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nkd; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage_in(cur ^ 1, kt + 1);  // DMA runs under the MFMAs below
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // single LDS stage (two workgroups per CU hide each other's load / epilogue phases)
        for (int kt = 0; kt < nkd; ++kt) {
            stage_in(0, kt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }

    gemm_epilogue<WM, WN, EPI>(g, acc, smem, m0, n0, wm, wn, wid, lane);
}

template <int BM, int BN, int NWM, int NWN, int EPI, int NSTAGE, int MINW, int PRIO = 0>
int launch_tiled_e(const GemmArgs &g, hipStream_t s) {
    static bool attr_set = false;
    constexpr int stages = NSTAGE * (BM + BN) * 128, epi = NWM * NWN * (BM / NWM) * ((BN / NWN) * 2 + 8);
    constexpr int smem = stages > epi ? stages : epi;
    constexpr int smem_nt = 2 * (BM + BN) * 128 > epi ? 2 * (BM + BN) * 128 : epi;
    const bool fast = (g.K % BK) == 0 && !(g.dbg & 4) && (int64_t)g.M * g.lda * 2 < 0x7fff0000ll && (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_glds_kernel<BM, BN, NWM, NWN, EPI, NSTAGE, MINW, PRIO>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_nt_kernel<BM, BN, NWM, NWN, EPI>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, smem_nt));
        attr_set = true;
    }
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    if (fast) hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, NWM, NWN, EPI, NSTAGE, MINW, PRIO>), dim3(tiles), dim3(64 * NWM * NWN), smem, s, g);
    else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, NWM, NWN, EPI>), dim3(tiles), dim3(64 * NWM * NWN), smem_nt, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

template <int BM, int BN, int NWM, int NWN, int NSTAGE, int MINW, int PRIO = 0>
int launch_tiled(const GemmArgs &g, hipStream_t s) {
    if (g.epi == 1) return launch_tiled_e<BM, BN, NWM, NWN, 1, NSTAGE, MINW, PRIO>(g, s);
    if (g.epi == 2) return launch_tiled_e<BM, BN, NWM, NWN, 2, NSTAGE, MINW, PRIO>(g, s);
    return launch_tiled_e<BM, BN, NWM, NWN, 0, NSTAGE, MINW, PRIO>(g, s);
}


}  // namespace
