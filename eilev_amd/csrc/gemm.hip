// gemm.hip — bf16 MFMA GEMM family for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias) (+ resid)
//
// Replaces every nn.Linear on the path (hf modeling_blip_2.py:328,351,366-368,584-586,618,660,674;
// hf modeling_opt.py:151-179,239-247,512; ref:eilev/model/v2.py:308).  Weights stay in the checkpoint's
// [out,in] layout: both MFMA operands then read 8 consecutive k per lane (16-byte loads), no transposes.
//
// Tiled kernel (M > 16): BMxBNx64 tile, 64-lane waves each owning a (BM/NWM)x(BN/NWN) sub-tile made of
// 16x16x32 MFMAs (v_mfma_f32_16x16x32_bf16, fp32 accumulate).  Global->register->LDS staging with the
// next tile's loads issued before the current tile's MFMAs (register prefetch) and a 2-deep LDS ring,
// one barrier per K-step.  LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-swizzled with
// (row & 7) so the ds_read_b128 fragment reads of 16 consecutive rows spread over all bank groups.
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles.
//
// Skinny kernel (M <= 16, the decode step): one MFMA row-block of 16 weight rows per workgroup, the K
// range split over the 8 waves (and over gridDim.y when N is small) — HBM-bound weight streaming.
#include "common.h"

namespace {

constexpr int BK = 64;  // bf16 elements per K-step = one 128-byte LDS row

template <int BM, int BN, int NWM, int NWN, int EPI>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_nt_kernel(const GemmArgs g) {
    constexpr int NT = 64 * NWM * NWN;
    constexpr int WM = BM / NWM, WN = BN / NWN;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int A_CH = BM * 8 / NT, B_CH = BN * 8 / NT;  // 16-byte chunks per thread per K-step
    constexpr int STAGE = (BM + BN) * 128;                 // bytes per LDS stage
    static_assert(A_CH >= 1 && B_CH >= 1, "tile too small for the workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- tile id: XCD-aware remap (block b runs on XCD b % 8; give each XCD a contiguous tile range)
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // n fastest: consecutive workgroups of one XCD share the A row-panel, W streams through L2/MALL
    const int tm_i = bid / tiles_n, tn_i = bid % tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / NWN, wn = wid % NWN;
    const int l15 = lane & 15, lg = lane >> 4;

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
            *reinterpret_cast<bf16x8 *>(sa + row * 128 + ((c ^ (row & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int id = tid + i * NT, row = id >> 3, c = id & 7;
            *reinterpret_cast<bf16x8 *>(sb + row * 128 + ((c ^ (row & 7)) << 4)) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    gload(0);
    swrite(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);  // in flight while the MFMAs below run
        const char *sa = smem + cur * STAGE + (wm * WM) * 128;
        const char *sb = smem + cur * STAGE + BM * 128 + (wn * WN) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bfr[TN];
            const int kc = ks * 4 + lg;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = j * 16 + l15;
                bfr[j] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + ((kc ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = i * 16 + l15;
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + ((kc ^ (row & 7)) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) swrite(cur ^ 1);  // the other stage was last read before the previous barrier
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
    // (Every index into acc[][] must be a compile-time constant or the array is demoted to scratch.)
    const int ecol0 = n0 + wn * WN + l15, erow0 = m0 + wm * WM + lg * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = ecol0 + j * 16;
        const bool cok = col < g.N;
        const int ccol = cok ? col : 0;
        const float bv = g.bias ? (float)g.bias[ccol] : 0.0f;
        const float sc = ccol < g.scale_cols ? g.scale : 1.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = erow0 + i * 16 + r;
                const bool ok = cok && row < g.M;
                float v = (acc[i][j][r] + bv) * sc;
                if (EPI == 1) v = gelu_erf(v);
                else if (EPI == 2) v = fmaxf(v, 0.0f);
                if (ok) {
                    int64_t orow = row;
                    if (g.patch_group > 0) {
                        // patch-embedding mode: GEMM row m = frame * group + patch; the output has one extra
                        // (CLS) row in front of every frame and `resid` is the position table [1+group, N].
                        const int f = row / g.patch_group, p = row - f * g.patch_group;
                        orow = (int64_t)f * (g.patch_group + 1) + 1 + p;
                        v += (float)g.resid[(int64_t)(1 + p) * g.ldr + col];
                    } else if (g.resid) {
                        v += (float)g.resid[(int64_t)row * g.ldr + col];
                    }
                    if (g.out_f32) reinterpret_cast<float *>(g.C)[orow * g.ldc + col] = v;
                    else reinterpret_cast<bf16 *>(g.C)[orow * g.ldc + col] = (bf16)v;
                }
            }
        }
    }
}

// ---- skinny GEMM (M <= 16): weight-streaming, one 16-row block of W per workgroup ----------------
// grid = (ceil(N/16), KS).  Each of the 8 waves owns a contiguous slice of this workgroup's K range;
// per K-step of 32 a lane loads 16 B of W (row n0 + lane%16, k-group lane/16) and 16 B of A (batch row
// lane%16, zero beyond M) and issues one 16x16x32 MFMA.  Wave partials are summed through LDS in a
// fixed order (deterministic).  KS == 1: epilogue applied here; KS > 1: fp32 partials to `part`
// ([KS][16][N]) for skinny_reduce_kernel.
struct SkinnyArgs {
    GemmArgs g;
    float *part;
    int ks;
};

__device__ __forceinline__ void skinny_epilogue(const GemmArgs &g, int row, int col, float v) {
    if (g.bias) v += (float)g.bias[col];
    if (col < g.scale_cols) v *= g.scale;
    if (g.epi == 1) v = gelu_erf(v);
    else if (g.epi == 2) v = fmaxf(v, 0.0f);
    if (g.resid) v += (float)g.resid[(int64_t)row * g.ldr + col];
    if (g.out_f32) reinterpret_cast<float *>(g.C)[(int64_t)row * g.ldc + col] = v;
    else reinterpret_cast<bf16 *>(g.C)[(int64_t)row * g.ldc + col] = (bf16)v;
}

__global__ __launch_bounds__(512) void gemm_skinny_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ float red[8][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    int wrow = n0 + l15;
    wrow = wrow < g.N ? wrow : g.N - 1;
    // K range of this workgroup, then of this wave, in units of 32
    const int ksteps = (g.K + 31) / 32;
    const int per_wg = (ksteps + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ksteps, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 7) / 8;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);

    const bf16 *wp = g.W + (int64_t)wrow * g.ldw + lg * 8;
    const bf16 *ap = g.A + (int64_t)(l15 < g.M ? l15 : 0) * g.lda + lg * 8;
    const bool arow = l15 < g.M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = beg; s < end; ++s) {
        const int k = s * 32;
        const bool kin = (k + lg * 8) < g.K;
        bf16x8 wv = kin ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(wp + k)) : zero8();
        bf16x8 av = (kin && arow) ? *reinterpret_cast<const bf16x8 *>(ap + k) : zero8();
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][lane][r] = acc[r];
    __syncthreads();
    if (wid == 0) {
        const int col = n0 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w][lane][r];
            const int row = lg * 4 + r;
            if (row < g.M && col < g.N) {
                if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * 16 + row) * g.N + col] = v;
            }
        }
    }
}

__global__ void skinny_reduce_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.M * g.N) return;
    const int row = idx / g.N, col = idx - row * g.N;
    float v = 0.0f;
    for (int s = 0; s < a.ks; ++s) v += a.part[((int64_t)s * 16 + row) * g.N + col];
    skinny_epilogue(g, row, col, v);
}

template <int BM, int BN, int NWM, int NWN, int EPI>
int launch_tiled_e(const GemmArgs &g, hipStream_t s) {
    static bool attr_set = false;
    constexpr int smem = 2 * (BM + BN) * 128;
    auto kern = gemm_nt_kernel<BM, BN, NWM, NWN, EPI>;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * NWM * NWN), smem, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

template <int BM, int BN, int NWM, int NWN>
int launch_tiled(const GemmArgs &g, hipStream_t s) {
    if (g.epi == 1) return launch_tiled_e<BM, BN, NWM, NWN, 1>(g, s);
    if (g.epi == 2) return launch_tiled_e<BM, BN, NWM, NWN, 2>(g, s);
    return launch_tiled_e<BM, BN, NWM, NWN, 0>(g, s);
}

}  // namespace

int launch_gemm(const GemmArgs &g, int prof_kind, hipStream_t s) {
    if (g.M <= 0) return EILEV_OK;
    if (!g.A || !g.W || !g.C || g.N <= 0 || g.K <= 0) return EILEV_E_BADARG;
    if ((g.K & 7) || (g.lda & 7) || (g.ldw & 7) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return EILEV_E_UNSUPPORTED;
    int rc;
    if (g.M <= 16 && g.patch_group == 0) {
        SkinnyArgs a;
        a.g = g;
        const int nb = (g.N + 15) / 16;
        int ks = nb >= 512 ? 1 : (512 + nb - 1) / nb;
        const int ksteps = (g.K + 31) / 32;
        if (ks > ksteps / 8) ks = ksteps / 8 > 0 ? ksteps / 8 : 1;
        if (ks > 1 && (!g.scratch || (size_t)ks * 16 * g.N * sizeof(float) > g.scratch_bytes)) ks = 1;
        a.ks = ks;
        a.part = g.scratch;
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3(nb, ks), dim3(512), 0, s, a);
        EILEV_LAUNCH_CHECK();
        if (ks > 1) {
            const int total = g.M * g.N;
            hipLaunchKernelGGL(skinny_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
            EILEV_LAUNCH_CHECK();
        }
        return EILEV_OK;
    }
    const double flops = 2.0 * g.M * (double)g.N * g.K;
    if (prof_kind >= 0) prof_begin(prof_kind, flops, s);
    const int64_t t256 = ceil_div64(g.M, 256) * ceil_div64(g.N, 256);
    const bool n_fits_256 = (g.N % 256 == 0) || g.N >= 2048;
    if (t256 >= 256 && n_fits_256) rc = launch_tiled<256, 256, 2, 4>(g, s);
    else if (ceil_div64(g.M, 256) * ceil_div64(g.N, 128) >= 256) rc = launch_tiled<256, 128, 4, 2>(g, s);
    else rc = launch_tiled<128, 128, 2, 2>(g, s);
    if (prof_kind >= 0) prof_end(s);
    return rc;
}
