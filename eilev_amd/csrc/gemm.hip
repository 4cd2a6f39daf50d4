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
// Cache policy of the persistent kernel's output stores: 2 = nt (streaming).  A launch writes 0.8-3.4 GB through eight 4-MiB L2s whose
// job is to keep the A / W panels of the ~32 tiles in flight; nothing re-reads the output from L2.  Same-box A/B at the bench shapes
// (tools/gemm_ab.py, 2 runs): +0.2 ... +0.7 % on all five GEMMs; sc1 / sc1+nt: +-0.
#ifndef EILEV_ST_AUX
#define EILEV_ST_AUX 2
#endif
#include <type_traits>
#include <utility>

// This file is compiled as two objects so that its ~5 minutes of hipcc run side by side (build.py): EILEV_GEMM_PART 1 = everything
// except the fp8-MFMA and LayerNorm-folding instances of the persistent kernel, 2 = only those (launch_pp4_ext); 0 (default) = one object.
#ifndef EILEV_GEMM_PART
#define EILEV_GEMM_PART 0
#endif
#if EILEV_GEMM_PART < 2
int g_gemm_debug = 0;  // probe-only switches (tools/gemm_probe.py): 1 = skip stores, 2 = skip main loop
extern "C" int eilev_debug_gemm_flags(int f) { g_gemm_debug = f; return 0; }
int g_skinny_nb_default = 1;  // weight blocks per workgroup of the weight-streaming GEMV (set after measurement; see launch_gemm)
unsigned long long *g_gemm_trace = nullptr;  // probe-only: see GemmArgs::trace
int g_gemm_trace_tiles = 0;
extern "C" int eilev_debug_gemm_trace(void *buf, int tiles) { g_gemm_trace = (unsigned long long *)buf; g_gemm_trace_tiles = tiles; return 0; }
#endif

namespace {

constexpr int BK = 64;  // bf16 elements per K-step = one 128-byte LDS row

// LDS swizzle: 16-byte chunk c of tile row r lives at chunk c ^ ((r >> 1) & 7).  With it the 16 lanes
// that one ds_read_b128 services together (MI355X_MICROARCH.md §LDS) always hit 16 distinct 16-byte
// slots of the 256-byte bank row, for the 32-row fragment pattern of the 32x32x16 MFMA.
__device__ __forceinline__ int swz(int row, int c) { return (c ^ ((row >> 1) & 7)) << 4; }

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Workgroup -> tile map.  (1) XCD-aware: block b runs on XCD b % 8, so each XCD (private 4 MiB L2) gets a
// contiguous range of the tile order.  (2) Grouped order: consecutive tiles walk down GROUP_M tile rows before
// moving to the next tile column, so the ~32-64 tiles an XCD runs concurrently form a compact 2-D block and
// share A row-panels and W column-panels through its L2 (the K-slices they stream are in step).
__device__ __forceinline__ void tile_coords(const GemmArgs &g, int tiles_m, int tiles_n, int &tm, int &tn, int t = blockIdx.x) {
    const int nwg = tiles_m * tiles_n;
    const int xcd = t & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    if (g.dbg & 256) {  // probe: plain row-major order
        tm = t / tiles_n;
        tn = t % tiles_n;
        return;
    }
    // rows per group: 8 x 4 tiles per XCD round; narrow N (fc2, proj: 5.5 column tiles) shares better with 4 rows
    // (measured on fc2: 2 / 4 / 8 / 16 rows = 1122 / 1132 / 1093 / 1045 TFLOP/s; r3, proj on three boxes: 4 rows +1.8 ... +2.2 %; wide N:
    // 8 and 16 equal, 4 rows -1.5 % on fc1 and +0.1 ... +1.6 % on qkv, 32 worse)
    // The half tiles of the last column (N = 1408: 5.5 columns) stay MIXED into this order.  r2, same-box: all full tiles first and the
    // half tiles last (every XCD in step on equal work) = fc2 1090 -> 983, proj 930 -> 880, qkv 1095 -> 1084 TFLOP/s — 256 half tiles
    // at once are fabric-bound (an A panel per 128 output columns); two half tiles as one unit = fc2 1123 -> 906-1003 (a 1.5-tile unit
    // per ~11 doubles the imbalance of the static stride).
    const int gsel = (g.dbg >> 22) & 3;  // probe override: 1 -> 4 rows, 2 -> 8 rows, 3 -> 16 rows
    const int GROUP_M = gsel == 1 ? 4 : gsel == 2 ? 8 : gsel == 3 ? 16 : (tiles_n <= 8 ? 4 : 8);
    const int width = GROUP_M * tiles_n, group = t / width, first = group * GROUP_M;
    const int gsz = min(tiles_m - first, GROUP_M), in = t - group * width;
    tm = first + in % gsz;
    tn = in / gsz;
}

// Epilogue shared by every tiled kernel.  acc[i][j][reg] = C[m = i*32 + (lane & 31)][n = j*32 + (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)].
// bf16 output: each wave stages its rows [IBEG*32, IEND*32) x WN in a private LDS region so that HBM sees whole
// 128-byte row segments, 16 bytes per lane (2-byte stores straight from the MFMA layout cost as much as the K
// loop): (a) residual rows -> LDS (coalesced); (b) acc + bias, activation, + residual -> bf16 in place;
// (c) rows -> HBM.  Only the owning wave touches its region: no workgroup barrier.  The rare variants (fp32
// logits, q pre-scaling, patch-embedding row remap, tile tails) are wave-uniform branches around the hot path.
// FASTG: the degree-8 GELU of the persistent ViT kernel (common.h); every other kernel evaluates the degree-12 form.
template <int WM, int WN, int EPI, int IBEG = 0, int IEND = WM / 32, int LN = 0, bool FASTG = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, f32x16 (&acc)[WM / 32][WN / 32], char *smem, int m0, int n0,
                                              int wm, int wn, int wid, int lane) {
    constexpr int TN = WN / 32;
    constexpr int RS = WN * 2 + 8;  // staging row stride (bytes)
    constexpr int ROWS = (IEND - IBEG) * 32;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wrow0 = m0 + wm * WM, wcol0 = n0 + wn * WN;
    if (g.out_f32) {
        // fp32 output (logits): a lane's 4 consecutive n are one 16-byte store
#pragma unroll
        for (int i = IBEG; i < IEND; ++i) {
            const int row = wrow0 + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wcol0 + j * 32 + q * 8 + hi * 4;
                    if (row < g.M && col < g.N && !((g.dbg & 1) && row > 0)) {
                        float *dst = reinterpret_cast<float *>(g.C) + (int64_t)row * g.ldc + col;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[i][j][q * 4 + e];
                            if (g.ascale) v *= g.ascale[row];
                            if (g.wscale && col + e < g.N) v *= g.wscale[col + e];
                            if (g.bias) v += (float)g.bias[col + e];
                            if (col + e < g.scale_cols) v *= g.scale;
                            if (EPI == 1) v = gelu_erf(v);
                            else if (EPI == 2) v = fmaxf(v, 0.0f);
                            if (g.resid) v += (float)g.resid[(int64_t)row * g.ldr + col + e];
                            if (col + e < g.N) {
                                if (g.k_slice > 0) atomicAdd(dst + e, v);
                                else dst[e] = v;
                            }
                        }
                    }
                }
        }
        return;
    }
    if (g.dbg & 1024) {  // probe: no epilogue at all (keep acc alive)
        float keep = 0.0f;
#pragma unroll
        for (int i = IBEG; i < IEND; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) keep += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
        if (keep == 123.456f) reinterpret_cast<float *>(g.C)[0] = keep;
        return;
    }
    float st1[LN == 2 ? IEND - IBEG : 1], st2[LN == 2 ? IEND - IBEG : 1];  // stat_out: this wave's (sum, sum of squares) per row over its 64 columns
    if constexpr (LN == 2) {
#pragma unroll
        for (int i = 0; i < IEND - IBEG; ++i) st1[i] = st2[i] = 0.0f;
    }
    char *reg = smem + wid * (ROWS * RS);
    const int wrow1 = wrow0 + IBEG * 32;            // first global row of this pass
    const int srow = lane >> 3, schunk = lane & 7;  // row-major phases: 8 lanes per 128-byte row segment
    const bool patch = g.patch_group > 0;
    const bool interior = wrow1 + ROWS <= g.M && wcol0 + WN <= g.N && !patch && !(g.dbg & 1);
    const bool has_res = g.resid != nullptr;
    const bool has_scale = g.scale_cols > 0;
    // (a) residual rows -> LDS
    if (has_res) {
        if (interior) {
            const bf16 *rp = g.resid + (int64_t)(wrow1 + srow) * g.ldr + wcol0 + schunk * 8;
            char *dp = reg + srow * RS + schunk * 16;
#pragma unroll 4
            for (int it = 0; it < ROWS / 8; ++it) {
                const bf16x8 v = *reinterpret_cast<const bf16x8 *>(rp);
                bf16x4 *d = reinterpret_cast<bf16x4 *>(dp);
                d[0] = (bf16x4){v[0], v[1], v[2], v[3]};
                d[1] = (bf16x4){v[4], v[5], v[6], v[7]};
                rp += 8 * g.ldr;
                dp += 8 * RS;
            }
        } else {
#pragma unroll 2
            for (int it = 0; it < ROWS / 8; ++it) {
                const int lr = it * 8 + srow, row = wrow1 + lr, col = wcol0 + schunk * 8;
                bf16x8 v = zero8();
                if (row < g.M && col < g.N) {
                    const int64_t rrow = patch ? 1 + (row % g.patch_group) : row;
                    v = *reinterpret_cast<const bf16x8 *>(g.resid + rrow * g.ldr + col);
                }
                bf16x4 *d = reinterpret_cast<bf16x4 *>(reg + lr * RS + schunk * 16);
                d[0] = (bf16x4){v[0], v[1], v[2], v[3]};
                d[1] = (bf16x4){v[4], v[5], v[6], v[7]};
            }
        }
    }
    // (b) acc -> bf16 (+ bias, activation, residual) at [row][col] of the staging region
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lc = j * 32 + q * 8 + hi * 4, col = wcol0 + lc;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (g.bias && col < g.N) {
                const bf16x4 b4 = *reinterpret_cast<const bf16x4 *>(g.bias + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (float)b4[e];
            }
            float sc[4] = {1.f, 1.f, 1.f, 1.f};
            if (has_scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[e] = (col + e) < g.scale_cols ? g.scale : 1.0f;
            }
            char *cp = reg + l31 * RS + lc * 2;
            constexpr int NI = IEND - IBEG;
            float v[NI][4];
            if (g.wscale) {  // fp8 weights: per-output-channel scale before the bias
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ws = col + e < g.N ? g.wscale[col + e] : 1.0f;
#pragma unroll
                    for (int i = 0; i < NI; ++i) acc[IBEG + i][j][q * 4 + e] *= ws;
                }
            }
            if (g.ascale) {  // fp8 activations: per-row (token) scale
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int row = wrow0 + (IBEG + i) * 32 + l31;
                    const float as = row < g.M ? g.ascale[row] : 1.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[IBEG + i][j][q * 4 + e] *= as;
                }
            }
            if (LN == 1) {  // folded LayerNorm: the accumulators started from -mean[m] * csum[n]; what is left is rstd[m]
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int row = wrow0 + (IBEG + i) * 32 + l31;
                    const float la = row < g.M ? g.ln_rows[2 * (int64_t)row] : 1.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[IBEG + i][j][q * 4 + e] *= la;
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[i][e] = acc[IBEG + i][j][q * 4 + e] + bv[e];
                    if (has_scale) v[i][e] *= sc[e];
                    if (EPI == 2) v[i][e] = fmaxf(v[i][e], 0.0f);
                }
            if (EPI == 1) {
                f32x2 x[2 * NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    x[2 * i] = (f32x2){v[i][0], v[i][1]};
                    x[2 * i + 1] = (f32x2){v[i][2], v[i][3]};
                }
                if constexpr (FASTG) {
                    float y[4 * NI];
#pragma unroll
                    for (int i = 0; i < 2 * NI; ++i) { y[2 * i] = x[i].x; y[2 * i + 1] = x[i].y; }
                    gelu_erf_n<4 * NI>(y);
#pragma unroll
                    for (int i = 0; i < 2 * NI; ++i) x[i] = (f32x2){y[2 * i], y[2 * i + 1]};
                } else gelu_erf_pk<2 * NI>(x);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    v[i][0] = x[2 * i].x; v[i][1] = x[2 * i].y; v[i][2] = x[2 * i + 1].x; v[i][3] = x[2 * i + 1].y;
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                bf16x4 *cell = reinterpret_cast<bf16x4 *>(cp + i * 32 * RS);
                if (has_res) {
                    const bf16x4 r4 = *cell;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][e] += (float)r4[e];
                }
                if (LN == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < g.N) {
                            st1[LN == 2 ? i : 0] += v[i][e];
                            st2[LN == 2 ? i : 0] = fmaf(v[i][e], v[i][e], st2[LN == 2 ? i : 0]);
                        }
                }
                *cell = (bf16x4){(bf16)v[i][0], (bf16)v[i][1], (bf16)v[i][2], (bf16)v[i][3]};
            }
        }
    }
    if (LN == 2 && wcol0 < g.N) {
        static_assert(WN == 64, "one statistics slot per 64 columns");
#pragma unroll
        for (int i = 0; i < IEND - IBEG; ++i) {
            const float t1 = st1[LN == 2 ? i : 0] + __shfl_xor(st1[LN == 2 ? i : 0], 32), t2 = st2[LN == 2 ? i : 0] + __shfl_xor(st2[LN == 2 ? i : 0], 32);
            const int row = wrow0 + (IBEG + i) * 32 + l31;
            if (hi == 0 && row < g.M)
                *reinterpret_cast<float2 *>(g.stat_out + ((int64_t)(wcol0 >> 6) * g.stat_ld + row) * 2) = make_float2(t1, t2);
        }
    }
    if (g.dbg & 2048) return;  // probe: no store phase
    // (c) rows -> HBM
    if (interior) {
        bf16 *dp = reinterpret_cast<bf16 *>(g.C) + (int64_t)(wrow1 + srow) * g.ldc + wcol0 + schunk * 8;
        const char *sp0 = reg + srow * RS + schunk * 16;
#pragma unroll 4
        for (int it = 0; it < ROWS / 8; ++it) {
            const bf16x4 *sp = reinterpret_cast<const bf16x4 *>(sp0);
            const bf16x4 lo = sp[0], hi4 = sp[1];
            *reinterpret_cast<bf16x8 *>(dp) = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            dp += 8 * g.ldc;
            sp0 += 8 * RS;
        }
        return;
    }
#pragma unroll 2
    for (int it = 0; it < ROWS / 8; ++it) {
        const int lr = it * 8 + srow, row = wrow1 + lr, col = wcol0 + schunk * 8;
        if (row < g.M && col < g.N && !((g.dbg & 1) && row > 0)) {
            const bf16x4 *sp = reinterpret_cast<const bf16x4 *>(reg + lr * RS + schunk * 16);
            const bf16x4 lo = sp[0], hi4 = sp[1];
            int64_t orow = row;
            if (patch) {
                // patch-embedding mode: GEMM row m = frame * group + patch; the output has one extra (CLS)
                // row in front of every frame (and `resid` above was the position table [1 + group, N]).
                const int f = row / g.patch_group;
                orow = row + f + 1;
            }
            bf16 *dst = reinterpret_cast<bf16 *>(g.C) + orow * g.ldc + col;
            if (col + 8 <= g.N) {
                *reinterpret_cast<bf16x8 *>(dst) = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e < g.N) dst[e] = lo[e];
                    if (col + 4 + e < g.N) dst[4 + e] = hi4[e];
                }
            }
        }
    }
}

typedef __attribute__((address_space(3))) void lds_void;

#if EILEV_GEMM_PART == 3
// ---- object 3 (gemm_a4.hip): the one-wave-per-SIMD 128 x 128 kernel with the hand-scheduled K loop ----------------------------------
#include "gemm_a4.h"
}  // namespace

int launch_a4(const GemmArgs &g, hipStream_t s) {
    static int num_cu = 0, var = -1;
    if (!num_cu) {
        int dev = 0;
        EILEV_HIP_CHECK(hipGetDevice(&dev));
        EILEV_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    if (var < 0) {  // probe-only: schedule variant of the K loop (gen_a4_loop.py VARIANTS; bias-only launches)
        const char *e = getenv("EILEV_A4_VAR");
        var = e ? atoi(e) & 1 : 0;
    }
    if (g.K % 64 || g.K < 192 || g.N % 128 || g.out_f32 || g.patch_group || g.A8 || g.W8 || g.wscale || g.ascale || g.scale_cols || g.ln_rows || (g.ldc & 7) ||
        ((uintptr_t)g.C & 15) || (g.resid && ((g.ldr & 7) || ((uintptr_t)g.resid & 15))) || (g.bias && ((uintptr_t)g.bias & 7)) ||
        (g.stat_out && (!g.resid || g.epi != 0 || g.stat_ld < g.M || ((uintptr_t)g.stat_out & 7))))
        return EILEV_E_UNSUPPORTED;
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    const int grid = tiles < num_cu ? tiles : num_cu / 8 * 8;
    if (g.resid && g.epi != 0) return EILEV_E_UNSUPPORTED;  // (no caller: activation + residual)
    if (g.stat_out) return launch_a4_i<0, 0, 2, true>(g, grid, s);
    if (g.resid) return launch_a4_i<0, 0, 0, true>(g, grid, s);
    if (g.epi == 1) return launch_a4_i<1, 0, 0, false>(g, grid, s);
    if (g.epi == 2) return launch_a4_i<2, 0, 0, false>(g, grid, s);
    if (var == 1) return launch_a4_i<0, 1, 0, false>(g, grid, s);
    return launch_a4_i<0, 0, 0, false>(g, grid, s);
}
#else  // EILEV_GEMM_PART != 3: everything else
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
template <int EPI, bool F8 = false, int LN = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp4_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, NWM = 2, NWN = 4, NW = 8;
    constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 32, TN = WN / 32;
    constexpr int STEP = (BM + BN) * 128;
    constexpr int PC = 8;  // 1-KiB LDS-DMA pieces (8 rows x 128 B) per wave and K-step

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / NWN, wn = wid % NWN;
    const int l31 = lane & 31, hi = lane >> 5;
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
        char *sd = smem + (st & 1) * STEP + (late ? BM * 128 : 0) + (pw * PC) * 1024;
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
    f32x16 acc[TM][TN];
    bf16x8 af[2][TM], bfr[2][TN];
    auto read_half = [&](int st, int h) {
        const char *sa = smem + (st & 1) * STEP + (wm * WM) * 128;
        const char *sb = smem + (st & 1) * STEP + BM * 128 + (wn * WN) * 128;
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
        const char *sa = smem + (st & 1) * STEP + (hm * 64) * 128;
        const char *sb = smem + (st & 1) * STEP + BM * 128 + (hn * 64) * 128;
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
    char *const stg = smem + 2 * STEP + wid * 4096;
    const unsigned stg_sw = (unsigned)(2 * STEP + wid * 4096 + l31 * 128 + hi * 8) ^ (unsigned)((l31 & 7) << 4);
    const int srow = lane >> 3, schunk = lane & 7;
    auto is_lean = [&](int n0_) {
        return !(n0_ + 128 >= g.N && !(g.dbg & 524288)) && n0_ + BN <= g.N && !g.out_f32 && g.patch_group == 0 && g.scale_cols == 0 && !g.wscale && !g.ascale &&
               !(g.dbg & (1024 | 2048 | 1)) && !(g.dbg & 16777216);
    };
    // LN == 2 (proj / fc2, residual): the unit also emits the row statistics of what it writes (g.stat_out); LN == 1 (qkv / fc1, no residual):
    // (qkv / fc1) it finishes a folded LayerNorm (g.ln_rows / g.ln_csum): see GemmArgs.
    float ln_rs[LN == 1 ? TM : 1];  // LN == 1: rstd of the lane's row in each of its units, fetched at the end of the K loop
    auto lean_epilogue = [&](int cm0, int cn0, auto res_c) {
        constexpr bool RES = decltype(res_c)::value;
        constexpr bool LNC = LN == 1 && !RES, LNP = LN == 2 && RES;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        const int r0 = cm0 + wm * WM;
        const int rows = g.M - r0 < 0 ? 0 : (g.M - r0 < WM ? g.M - r0 : WM);
        auto uniform_rsrc = [&](const void *ptr, int bytes) {
            const uint64_t base = (uint64_t)ptr;
            const uint64_t ub = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);  // provably wave-uniform: no waterfall loops
            return __builtin_amdgcn_make_buffer_rsrc((void *)ub, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t rc = uniform_rsrc(reinterpret_cast<bf16 *>(g.C) + (int64_t)r0 * g.ldc + cn0 + wn * WN, rows * (int)(g.ldc * 2));
        const __amdgpu_buffer_rsrc_t rr = uniform_rsrc(RES ? g.resid + (int64_t)r0 * g.ldr + cn0 + wn * WN : g.A, RES ? rows * (int)(g.ldr * 2) : 0);
        const unsigned st_voff = (unsigned)srow * (unsigned)(g.ldc * 2) + schunk * 16, rs_voff = (unsigned)srow * (unsigned)(g.ldr * 2) + schunk * 16;
        u32x4_t rv[4];
        auto res_load = [&](int u) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, rs_voff, (u * 32 + it * 8) * (int)(g.ldr * 2), 0);
        };
        if constexpr (RES) res_load(0);
        bf16x4 biasr[TN][4];  // columns j*32 + q*8 + hi*4 + (0..3) of the wave's 64: the accumulator layout
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (g.bias) biasr[j][q] = *reinterpret_cast<const bf16x4 *>(g.bias + cn0 + wn * WN + j * 32 + q * 8 + hi * 4);
                else biasr[j][q] = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            }
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        __amdgpu_buffer_rsrc_t rst = rr;
        if constexpr (LNP) rst = uniform_rsrc(g.stat_out + ((int64_t)((cn0 + wn * WN) >> 6) * g.stat_ld + r0) * 2, rows * 8);
        static_for<TM>([&](auto u_c) {
            constexpr int U = decltype(u_c)::value;
            float st1 = 0.0f, st2 = 0.0f;
            bf16x4 rcell[TN][4];
            if constexpr (RES) {  // the unit's residual rows -> staging (coalesced); every lane then fetches its 8 cells in one batch
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + srow;
                    *reinterpret_cast<u32x4_t *>(stg + row * 128 + ((schunk ^ (row & 7)) << 4)) = rv[it];
                }
                if constexpr (U + 1 < TM) res_load(U + 1);  // the next unit's rows arrive under this unit's arithmetic
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned ca;
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ca) : "n"((j * 4 + q) << 4), "v"(stg_sw));
                        rcell[j][q] = *reinterpret_cast<const bf16x4 *>(smem + ca);
                    }
            }
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
            if constexpr (LNP) {  // the two lane halves hold the two column halves of a row
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
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, erb[b]), rc, st_voff, (U * 32 + (h2 * 2 + b) * 8) * (int)(g.ldc * 2), EILEV_ST_AUX);
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
    bool pre1 = ns > 1 && !(g.dbg & 16777216);  // step 1 of the coming tile is already staged (prologue / previous tile's tail)
    if (pre1) stage_step(1, w_piece_mine(n0));
    bool lean_cur = is_lean(n0);  // this tile runs the lean epilogue: its accumulators start from the bias
    // folded LayerNorm (LN == 1): C = rstd * (A . W^T - mean * csum) + bias.  The rank-1 term -mean[m] * csum[n] is one more K-slice on
    // the matrix cores: the tile's first MFMA of every 32 x 32 block multiplies (csum_hi, csum_lo, csum_hi, 0 ...) by (nm_hi, nm_hi,
    // nm_lo, 0 ...) with nm = -mean (v_mfma_f32_32x32x8_bf16_1k: half the cost of the K = 16 form) — two bf16 pieces each, the product is good to 2^-16 of |mean * csum|, far inside the bf16 output —
    // and starts the accumulators (C = 0); the epilogue multiplies by rstd.  8 short MFMAs per wave and tile (+0.6 % of the K loop), ~30
    // VALU operations, 6 four-byte loads per lane fetched one tile AHEAD next to the next tile's first DMA (rows past M / columns past N
    // read 0).  (Built and measured before this: accumulators initialised with v_mul from 32 csum registers per lane — 36 loads per
    // lane and tile through the texture addresser and 128 VALU operations in a read phase: fc1 +4.5 %.)
    float ln_nm[LN == 1 ? TM : 1], ln_cl[LN == 1 ? TN : 1];
    auto ln_fetch = [&](int m0_, int n0_) {
        if constexpr (LN == 1) {
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
        s16x4_t ln_a1[LN == 1 ? TM : 1], ln_w1[LN == 1 ? TN : 1];
        auto acc_prep = [&]() {  // the two fragments of the rank-1 K-slice (in a read phase)
            if constexpr (LN == 1) {
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
            if constexpr (LN == 1) {
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
            constexpr bool SPLIT_OK = !HT && !(LN == 1 && EPI == 0);
            constexpr bool SPLIT_E = (EILEV_PP4_SPLIT & 1) && SPLIT_OK, SPLIT_L = (EILEV_PP4_SPLIT & 2) && SPLIT_OK;
            if (!late) {
                if (st + 1 < ns && !(FIRST && pre1)) stage_step(st + 1, w_mine, SPLIT_E ? 1 : 0);
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
            } else if (SPLIT_E && st + 1 < ns && !(FIRST && pre1)) stage_step(st + 1, w_mine, 2);
            ITR(5);
            PP_BARRIER();
            ITR(6);
            if constexpr (HT) {
                if (late && st + 2 < ns) stage_step(st + 2, w_mine);
                mma_half_ht();
            } else mma_half();
            __builtin_amdgcn_sched_barrier(0);
            if (!late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ITR(7);
            PP_BARRIER();
#else
            PP_WAIT_LGKM0_VM0();
            PP_BARRIER();
            if constexpr (HT) mma_half_ht(); else mma_half();
            PP_BARRIER();
#endif
        };
        if (half_tile) {
            kstep(0, std::true_type{}, std::true_type{});
            for (int st = 1; st < nsd; ++st) kstep(st, std::false_type{}, std::true_type{});
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
#pragma unroll
            for (int u = 0; u < TM; ++u) ln_rs[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (m0 + wm * WM + u * 32 + l31) * 8, 0, 0));
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
            lean_cur = is_lean(n0);
            ln_fetch(m0, n0);
        }
        stamp(3);
        if (lean) {  // LN kernels: the launcher guarantees ln_rows and no residual (1), stat_out and a residual (2)
            if constexpr (LN == 1) lean_epilogue(cm0, cn0, std::false_type{});
            else if constexpr (LN == 2) lean_epilogue(cm0, cn0, std::true_type{});
            else if (g.resid != nullptr) lean_epilogue(cm0, cn0, std::true_type{});
            else lean_epilogue(cm0, cn0, std::false_type{});
        } else if (half_tile) {
            gemm_epilogue<64, 64, EPI, 0, 2, LN, true>(g, reinterpret_cast<f32x16(&)[2][2]>(acc), smem + STEP, cm0, cn0, hm, hn, wid, lane);
        } else {
            gemm_epilogue<WM, WN, EPI, 0, TM / 2, LN, true>(g, acc, smem + STEP, cm0, cn0, wm, wn, wid, lane);
            gemm_epilogue<WM, WN, EPI, TM / 2, TM, LN, true>(g, acc, smem + STEP, cm0, cn0, wm, wn, wid, lane);
        }
        stamp(4);
        stamp(6, true);
        ++trace_i;
    }
#undef PP_BARRIER
}

constexpr int PP4_SMEM = 2 * 65536 + 8 * 4096;  // two step buffers + 4 KiB of lean-epilogue staging per wave (the general epilogue
                                                // stages 69.6 KB from step buffer 1 on: 65536 + 69632 < 163840)
#if EILEV_GEMM_PART != 2
int launch_pp4(const GemmArgs &g, hipStream_t s) {
    static bool attr_set = false;
    static int num_cu = 0;
    constexpr int smem = PP4_SMEM;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int dev = 0;
        EILEV_HIP_CHECK(hipGetDevice(&dev));
        EILEV_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_set = true;
    }
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    const int grid = tiles < num_cu ? tiles : num_cu / 8 * 8;
    if (g.A8 || g.ln_rows || g.stat_out) return launch_pp4_ext(g, grid, s);  // fp8 MFMA / LayerNorm-folding instances (the other object)
    if (g.epi == 1) hipLaunchKernelGGL(gemm_pp4_kernel<1>, dim3(grid), dim3(512), smem, s, g);
    else if (g.epi == 2) hipLaunchKernelGGL(gemm_pp4_kernel<2>, dim3(grid), dim3(512), smem, s, g);
    else hipLaunchKernelGGL(gemm_pp4_kernel<0>, dim3(grid), dim3(512), smem, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
#endif

#if EILEV_GEMM_PART == 2
}  // namespace
#else
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
    const int grid = tiles < num_cu ? tiles : num_cu / 8 * 8;
    if (g.epi == 1) hipLaunchKernelGGL(gemm_w6_kernel<1>, dim3(grid), dim3(256), smem, s, g);
    else if (g.epi == 2) hipLaunchKernelGGL(gemm_w6_kernel<2>, dim3(grid), dim3(256), smem, s, g);
    else hipLaunchKernelGGL(gemm_w6_kernel<0>, dim3(grid), dim3(256), smem, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

// ---- skinny GEMM (M <= 16): weight-streaming, one 16-row block of W per workgroup ----------------
// grid = (ceil(N/16), KS).  Each of the 4 waves owns a contiguous slice of this workgroup's K range;
// per K-step of 32 a lane loads 16 B of W (row n0 + lane%16, k-group lane/16) and 16 B of A (batch row
// lane%16, zero beyond M) and issues one 16x16x32 MFMA; loads are issued 8 deep.  Wave partials are summed through LDS in a
// fixed order (deterministic).  KS == 1: epilogue applied here; KS > 1: fp32 partials to `part`
// ([KS][16][N]) for skinny_reduce_kernel.
struct SkinnyArgs {
    GemmArgs g;
    float *part;
    int ks;
    int mr;  // rows per split-K partial: 16 (M <= 16) or 32
};

__device__ __forceinline__ void skinny_epilogue(const GemmArgs &g, int row, int col, float v) {
    if (g.wscale) v *= g.wscale[col];
    if (g.bias) v += (float)g.bias[col];
    if (col < g.scale_cols) v *= g.scale;
    if (g.epi == 1) v = gelu_erf(v);
    else if (g.epi == 2) v = fmaxf(v, 0.0f);
    if (g.resid) v += (float)g.resid[(int64_t)row * g.ldr + col];
    if (g.out_f32) reinterpret_cast<float *>(g.C)[(int64_t)row * g.ldc + col] = v;
    else reinterpret_cast<bf16 *>(g.C)[(int64_t)row * g.ldc + col] = (bf16)v;
}

__global__ __launch_bounds__(256) void gemm_skinny_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ float red[4][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    int wrow = n0 + l15;
    wrow = wrow < g.N ? wrow : g.N - 1;
    // K range of this workgroup, then of this wave, in units of 32
    const int ksteps = (g.K + 31) / 32;
    const int per_wg = (ksteps + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ksteps, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 3) / 4;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);

    const bf16 *wp = g.W + (int64_t)wrow * g.ldw + lg * 8;
    const bf16 *ap = g.A + (int64_t)(l15 < g.M ? l15 : 0) * g.lda + lg * 8;
    const bool arow = l15 < g.M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = beg;
    // 8 independent 16-byte weight loads in flight per lane (the K tail of the matrix never lands here:
    // K % 256 == 0 for every decode shape; the remainder loop below handles the general case)
    for (; s + 8 <= end && (s + 8) * 32 <= g.K; s += 8) {
        bf16x8 wv[8], av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(wp + (s + u) * 32));
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = arow ? *reinterpret_cast<const bf16x8 *>(ap + (s + u) * 32) : zero8();
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u], wv[u], acc, 0, 0, 0);
    }
    for (; s < end; ++s) {
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
            for (int w = 0; w < 4; ++w) v += red[w][lane][r];
            const int row = lg * 4 + r;
            if (row < g.M && col < g.N) {
                if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * 16 + row) * g.N + col] = v;
            }
        }
    }
}

// Skinny kernel, DMA-staged variant (K % 256 == 0): the 16-row weight block is streamed through per-wave LDS
// buffers with global_load_lds_dwordx4 so that every load instruction covers two whole 512-byte row segments
// (fully coalesced; the direct MFMA-layout loads above touch 64 separate 16-byte pieces per instruction).
// Each wave runs its own 2-deep pipeline on a private 2 x 8 KiB region: DMA(t+1) is issued before the
// counted s_waitcnt vmcnt(8) that retires DMA(t); no workgroup barrier in the K loop.  16-byte chunk c of
// row r is stored at chunk c ^ (r & 15) (swizzle applied on the source address) so the ds_read_b128 fragment
// reads of 16 rows x 512-byte stride are bank-conflict-free.
// MB = 1: M <= 16; MB = 2: M <= 32 (two 16-row activation tiles share every weight fragment: the weight stream, which
// bounds the kernel, is read once for twice the rows)
// PRE: a wave has at most 3 K-tiles (every decode shape): ALL its activation fragments are loaded up front (one exposed L2 round trip
// instead of one per tile: the per-tile loads were 40 % of the kernel) and the tile loop is three static iterations.  Same summation
// order as the rolled form.
template <int MB, bool PRE>
__global__ __launch_bounds__(256) void gemm_skinny_dma_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ __attribute__((aligned(16))) char wbuf[4][2][8192];
    __shared__ float red[4][MB][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    // K range of this workgroup / wave in units of 256 (= one 16 x 256 tile = 8 MFMA steps)
    const int ktiles = g.K / 256;
    const int per_wg = (ktiles + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ktiles, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 3) / 4;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);

    // DMA source: piece i (rows 2i, 2i+1): lane p -> row 2i + p/32, LDS slot p%32 <- global chunk slot ^ (row & 15)
    const int prow = lane >> 5, pslot = lane & 31;
    const bf16 *src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 2 * i + prow;
        int gr = n0 + row;
        gr = gr < g.N ? gr : g.N - 1;
        src[i] = g.W + (int64_t)gr * g.ldw + ((pslot ^ (row & 15)) << 3);
    }
    auto stage_in = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_void *)(src[i] + t * 256), (lds_void *)(&wbuf[wid][buf][i * 1024]), 16, 0, 0);
    };
    const bf16 *ap[MB];
    bool arow[MB];
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int r = mb * 16 + l15;
        arow[mb] = r < g.M;
        ap[mb] = g.A + (int64_t)(arow[mb] ? r : 0) * g.lda + lg * 8;
        acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (PRE) {
        bf16x8 av[MB][24];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
            if (beg + tt < end) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        av[mb][tt * 8 + u] = arow[mb] ? *reinterpret_cast<const bf16x8 *>(ap[mb] + (beg + tt) * 256 + u * 32) : zero8();
            }
        if (beg < end) stage_in(0, beg);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int t = beg + tt;
            if (t < end) {
                if (t + 1 < end) {
                    stage_in((tt & 1) ^ 1, t + 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile t and every activation fragment (older) have landed
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                const char *wb = &wbuf[wid][tt & 1][0] + l15 * 512;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wb + (((u * 4 + lg) ^ l15) << 4));
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][tt * 8 + u], wv, acc[mb], 0, 0, 0);
                }
            }
        }
    } else {
    if (beg < end) stage_in(0, beg);
        for (int t = beg; t < end; ++t) {
            const int cur = (t - beg) & 1;
            bf16x8 av[MB][8];
    #pragma unroll
            for (int mb = 0; mb < MB; ++mb)
    #pragma unroll
                for (int u = 0; u < 8; ++u) av[mb][u] = arow[mb] ? *reinterpret_cast<const bf16x8 *>(ap[mb] + t * 256 + u * 32) : zero8();
            if (t + 1 < end) {
                stage_in(cur ^ 1, t + 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the 8 pieces of tile t (older than the 8 just issued)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const char *wb = &wbuf[wid][cur][0] + l15 * 512;
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wb + (((u * 4 + lg) ^ l15) << 4));
    #pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][u], wv, acc[mb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid][mb][lane][r] = acc[mb][r];
    __syncthreads();
    if (wid < MB) {  // wave mb finishes row tile mb
        const int col = n0 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[w][wid][lane][r];
            const int row = wid * 16 + lg * 4 + r;
            if (row < g.M && col < g.N) {
                if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * (16 * MB) + row) * g.N + col] = v;
            }
        }
    }
}

// Skinny kernel for fp8 (OCP e4m3) weights: the same per-wave 2-deep LDS-DMA pipeline over 16-row x 256-K tiles, which are now
// 4 KiB (half the bytes of the bound stream).  Piece i = rows 4i .. 4i+3 x 256 B; 16-byte chunk c of row r is stored at chunk
// c ^ (r & 15).  A lane's 8 weights of an MFMA k-step are 8 bytes: v_cvt_pk_f32_fp8 + one v_perm_b32 per pair make the bf16
// fragment (every e4m3 value is exactly a bf16 value); the per-channel scale is applied to the fp32 sum in the epilogue.
// ---- weight streaming with the activations held in registers across several weight blocks (round 2) ---------------------------
// gemm_skinny_dma_kernel<MB, true> loads a wave's activation fragments (MB x 16 rows x its K slice, up to 192 VGPRs) and then streams
// ONE 16-row weight block (at most 3 tiles of 8 KB per wave): every workgroup pays 2 x its weight bytes in activation loads from L2
// and never reaches a steady stream (measured at batch 32: 2.1 TB/s).  Here a workgroup keeps the SAME activation fragments for NB
// consecutive weight blocks: the loads are paid once per NB blocks and each wave streams NB x its tiles through a 3-deep LDS-DMA
// ring (24 KB in flight per wave).  Partial sums of the 4 waves (K quarters) meet in LDS per block, in a fixed order.
template <int MB, int NB, int ST = 3>
__global__ __launch_bounds__(256) void gemm_skinny_nb_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    extern __shared__ __attribute__((aligned(16))) char smem_nb[];
    char(*wbuf)[ST][8192] = reinterpret_cast<char(*)[ST][8192]>(smem_nb);                                  // [4 waves][ST stages][8 KB]
    float(*red)[4][MB][64][4] = reinterpret_cast<float(*)[4][MB][64][4]>(smem_nb + 4 * ST * 8192);       // [2 (ping-pong)][4 waves][MB][64][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int nblocks = (g.N + 15) / 16;
    const int b0 = blockIdx.x * NB;
    const int ktiles = g.K / 256;
    const int per_wg = (ktiles + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ktiles, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 3) / 4;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);
    const int nt = max(end - beg, 0);  // tiles of this wave per weight block (<= 3)

    const int prow = lane >> 5, pslot = lane & 31;
    // piece i of block j: rows 2i, 2i+1 of the block; lane p -> row 2i + p/32, LDS slot p%32 <- global chunk slot ^ (row & 15)
    auto src = [&](int j, int i) {
        const int row = 2 * i + prow;
        int gr = (b0 + j) * 16 + row;
        gr = gr < g.N ? gr : g.N - 1;
        return g.W + (int64_t)gr * g.ldw + ((pslot ^ (row & 15)) << 3);
    };
    auto stage_in = [&](int stage, int j, int t) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_void *)(src(j, i) + t * 256), (lds_void *)(&wbuf[wid][stage][i * 1024]), 16, 0, 0);
    };
    // activation fragments of this wave's K slice, once
    bf16x8 av[MB][24];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
        if (tt < nt) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int r = mb * 16 + l15;
                const bf16 *ap = g.A + (int64_t)(r < g.M ? r : 0) * g.lda + lg * 8 + (beg + tt) * 256;
#pragma unroll
                for (int u = 0; u < 8; ++u) av[mb][tt * 8 + u] = r < g.M ? *reinterpret_cast<const bf16x8 *>(ap + u * 32) : zero8();
            }
        }
    const int nbl = min(NB, nblocks - b0);  // weight blocks of this workgroup
    const int total = nbl * nt;             // tiles this wave streams: flat index f = j * nt + tt
    // prologue: ST - 1 tiles in flight
    if (total > 0) stage_in(0, 0, beg);
    if (ST > 2 && total > 1) stage_in(1, 1 / nt, beg + 1 % nt);
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int f = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (j < nbl) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                if (tt < nt) {
                    // issue tile f + ST - 1, then wait until tile f has landed: the 8 pieces of each younger tile stay in flight
                    constexpr int AH = ST - 1;
                    if (f + AH < total) {
                        stage_in((f + AH) % ST, (f + AH) / nt, beg + (f + AH) % nt);
                        if (AH == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    } else if (AH == 2 && f + 1 < total) {
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const char *wb = &wbuf[wid][f % ST][0] + l15 * 512;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wb + (((u * 4 + lg) ^ l15) << 4));
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][tt * 8 + u], wv, acc[mb], 0, 0, 0);
                    }
                    ++f;
                }
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[j & 1][wid][mb][lane][r] = acc[mb][r];
            acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // one barrier per weight block; the partials ping-pong between two LDS regions, so the waves that finish block j (below) are
        // done before anybody writes region j & 1 again (after the barrier of block j + 1)
        __syncthreads();
        if (j < nbl) {  // (round 4) the MB x 4 (row tile, register) pairs of this block dealt over the four waves: the 4 K-quarter partials in a fixed order
            const int col = (b0 + j) * 16 + l15;
#pragma unroll
            for (int cb = 0; cb < MB; ++cb) {
                const int mb = (wid * MB + cb) >> 2, r = (wid * MB + cb) & 3;
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[j & 1][w][mb][lane][r];
                const int row = mb * 16 + lg * 4 + r;
                if (row < g.M && col < g.N) {
                    if (a.ks == 1) skinny_epilogue(g, row, col, v);
                    else a.part[((int64_t)blockIdx.y * (16 * MB) + row) * g.N + col] = v;
                }
            }
        }
    }
}

// (round 4, measured and removed: gemm_skinny5_kernel — one 16-row weight block x 5 waves x 2 K tiles per workgroup, every weight tile and
// activation fragment requested up front, no ring: 24.2 us per q|k|v / fc1 / fc2 launch against 19.8 us for the kernel above, 5.53 vs 5.40
// ms/token at batch 32 (profiles/r04_skinny5_rejected_*).  The ring depth was not the limit: each workgroup re-reads the 160 KB of
// activations from L2, and with one block per workgroup that is 2-3x the weight bytes entering every CU.)
// ---- round 4: 17..32 rows, ONE workgroup per CU, the activations loaded once per CU -----------------------------------------------------
// What bounds the kernels above at batch 32 is not the weight stream but the activations: every workgroup re-reads the 32 x K rows from
// L2 (160 KB at K = 2560) for 16-32 weight rows (80-160 KB) — 2-3x the weight bytes enter each CU (measured twice: 2 / 4 / 8 blocks per
// workgroup in round 2, and round 4's one-block-per-workgroup variant with every load up front, which was SLOWER: 24 vs 20 us).  Here the
// grid is the CUs.  A 512-thread workgroup splits K over its 8 waves (wave w: K / (8 ks) columns = KS k-steps of 32): its slice of the
// 32 rows is 80 VGPRs of MFMA A fragments, loaded ONCE; the CU then walks its share of the 16-row weight blocks, every wave streaming
// its K slice of a block straight into registers as B fragments (16 rows x 64 B per instruction, non-temporal) through a ring of RB
// blocks (30 KB per wave, 240 KB per CU in flight), 2 KS MFMAs per block, the 8 K-slice partials summed through LDS in a fixed order
// (ping-pong buffers, one barrier per block).  blockIdx.y = K split across CUs where the 8-wave slice would not fit the registers (fc2:
// K = 10240 -> 4) or the matrix has fewer blocks than CUs (out_proj): partials + reduce_ln_kernel as before.
template <int MB, int KS, int RB>
__global__ __launch_bounds__(512) void gemm_rows32_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ float red[2][8][MB][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int nb = (g.N + 15) / 16;
    const int G = gridDim.x;
    // this CU's blocks of K split blockIdx.y: [b0, b1)
    const int per = nb / G, rem = nb % G;
    const int b0 = blockIdx.x * per + min((int)blockIdx.x, rem), b1 = b0 + per + ((int)blockIdx.x < rem ? 1 : 0);
    const int nblk = b1 - b0;
    if (nblk <= 0) return;  // (uniform per workgroup)
    const int k0 = (blockIdx.y * 8 + wid) * (KS * 32);
    bf16x8 wv[RB][KS];
    float bv[RB];  // bias of the lane's output column, requested WITH the block's weights: a load in the epilogue put one global round trip
                   // (~1.5 us) on the critical path of every block (measured without any operand loads: 7.5 us per q|k|v launch, 20.7 for the lm_head)
    const bool plain_epi = a.ks == 1 && !g.wscale && !g.resid;
    auto load_block = [&](int j, auto buf_c) {
        constexpr int B = decltype(buf_c)::value;
        int gr = (b0 + j) * 16 + l15;
        gr = gr < g.N ? gr : g.N - 1;
        bv[B] = (plain_epi && g.bias) ? (float)g.bias[gr] : 0.0f;
        const bf16 *wp = g.W + (int64_t)gr * g.ldw + k0 + lg * 8;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
#ifdef ROWS32_NOW
            wv[B][u] = zero8();
#else
            wv[B][u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(wp + u * 32));
#endif
        }
    };
    bf16x8 av[MB][KS];
    auto load_x = [&]() {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int r = mb * 16 + l15;
            const bf16 *ap = g.A + (int64_t)(r < g.M ? r : 0) * g.lda + k0 + lg * 8;
#pragma unroll
            for (int u = 0; u < KS; ++u) {
#ifdef ROWS32_NOX
                av[mb][u] = zero8();
#else
                av[mb][u] = r < g.M ? *reinterpret_cast<const bf16x8 *>(ap + u * 32) : zero8();
#endif
            }
        }
    };
    auto consume = [&](int j, auto buf_c) {
        constexpr int B = decltype(buf_c)::value;
        f32x4 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][u], wv[B][u], acc[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[j & 1][wid][mb][lane][r] = acc[mb][r];
        __syncthreads();  // the partials of block j are complete; region (j + 1) & 1 is free again (its readers passed the previous barrier's successor)
        // every wave finishes ONE (row tile, accumulator register) pair of the block — 8 LDS reads and one store each — instead of waves
        // 0 .. MB - 1 finishing four: the next block's barrier waits for the finishers (same-box step 4.95 -> 4.79 ms/token).  (All of a CU's
        // blocks behind ONE barrier — MFMAs of every block first, then every reduction — was measured too: 5.31, the longer code spills.)
        if ((wid >> 2) < MB) {
            const int mb = wid >> 2, r = wid & 3, col = (b0 + j) * 16 + l15;
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[j & 1][w][mb][lane][r];
            const int row = mb * 16 + lg * 4 + r;
            if (row < g.M && col < g.N) {
                if (plain_epi) {  // bias (prefetched) + activation + store: no load here
                    v += bv[B];
                    if (col < g.scale_cols) v *= g.scale;
                    if (g.epi == 1) v = gelu_erf(v);
                    else if (g.epi == 2) v = fmaxf(v, 0.0f);
                    if (g.out_f32) reinterpret_cast<float *>(g.C)[(int64_t)row * g.ldc + col] = v;
                    else reinterpret_cast<bf16 *>(g.C)[(int64_t)row * g.ldc + col] = (bf16)v;
                } else if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * (16 * MB) + row) * g.N + col] = v;
            }
        }
    };
    if (nblk >= 2 * RB) {  // long: the lm_head (12 blocks per CU) — ring with a branch-free steady loop
        static_for<RB>([&](auto j_c) { load_block(decltype(j_c)::value, j_c); });
        load_x();
        int base = 0;
        for (; base + 2 * RB <= nblk; base += RB)
            static_for<RB>([&](auto j_c) {
                consume(base + decltype(j_c)::value, j_c);
                load_block(base + decltype(j_c)::value + RB, j_c);
            });
        static_for<RB>([&](auto j_c) {
            consume(base + decltype(j_c)::value, j_c);
            if (base + decltype(j_c)::value + RB < nblk) load_block(base + decltype(j_c)::value + RB, j_c);
        });
        base += RB;
        static_for<RB>([&](auto j_c) {
            if (base + decltype(j_c)::value < nblk) consume(base + decltype(j_c)::value, j_c);
        });
    } else {  // short: the block matrices (1-3 blocks per CU): everything requested up front
        static_for<RB>([&](auto j_c) {
            if (decltype(j_c)::value < nblk) load_block(decltype(j_c)::value, j_c);
        });
        load_x();
        static_for<RB>([&](auto j_c) {
            constexpr int J = decltype(j_c)::value;
            if (J < nblk) {
                consume(J, j_c);
                if (J + RB < nblk) load_block(J + RB, j_c);
            }
        });
        static_for<RB>([&](auto j_c) {
            if (decltype(j_c)::value + RB < nblk) consume(decltype(j_c)::value + RB, j_c);
        });
    }
}

static int skinny_n_cu() { return eilev_num_cu(); }

// K split of gemm_rows32_kernel: the 8-wave K slice must be 5 or 10 k-steps of 32; more splits when the matrix has fewer blocks than CUs
static bool rows32_plan(const GemmArgs &g, int nb, int n_cu, SkinnyArgs &a, int &ks, int &ksteps) {
    if (g.K % 256) return false;
    const int per_wave = g.K / 256;  // k-steps of 32 per wave without a split
    int k5 = 0;
    for (int c = 1; c <= 8; c *= 2)
        if (per_wave % c == 0 && (per_wave / c == 10 || per_wave / c == 5)) {
            k5 = c;
            if (nb * c >= n_cu || per_wave / c == 5 || !(g.dbg & 134217728)) break;
        }
    if (!k5) return false;
    // fewer blocks than CUs (out_proj: 160): the unsplit form leaves a third of the chip idle and needs a separate LayerNorm launch after it
    // (9.4 us: the split-K reduce of the round-2 kernel produces the LayerNorm for free) — those shapes keep the round-2 / round-3 kernels
    if (k5 == 1 && nb < n_cu && g.ln_out && !(g.dbg & 134217728)) return false;
    if (k5 > 1 && !(g.dbg & 134217728)) return false;  // measured: the split-K forms (out_proj, fc2) lose to the round-3 kernels in the step; probe flag 1 << 27 enables them
    if (k5 > 1 && (!g.scratch || (size_t)k5 * a.mr * g.N * sizeof(float) > g.scratch_bytes)) return false;
    ks = k5;
    a.ks = k5;
    ksteps = per_wave / k5;
    return true;
}

template <int MB, bool PRE>
__global__ __launch_bounds__(256) void gemm_skinny_w8_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ __attribute__((aligned(16))) char wbuf[4][2][4096];
    __shared__ float red[4][MB][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int ktiles = g.K / 256;
    const int per_wg = (ktiles + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ktiles, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 3) / 4;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);

    const int prow = lane >> 4, pslot = lane & 15;
    const uint8_t *src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + prow;
        int gr = n0 + row;
        gr = gr < g.N ? gr : g.N - 1;
        src[i] = g.W8 + (int64_t)gr * g.ldw + ((pslot ^ (row & 15)) << 4);
    }
    auto stage_in = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void *)(src[i] + t * 256), (lds_void *)(&wbuf[wid][buf][i * 1024]), 16, 0, 0);
    };
    const bf16 *ap[MB];
    bool arow[MB];
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int r = mb * 16 + l15;
        arow[mb] = r < g.M;
        ap[mb] = g.A + (int64_t)(arow[mb] ? r : 0) * g.lda + lg * 8;
        acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (PRE) {
        bf16x8 av[MB][24];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
            if (beg + tt < end) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        av[mb][tt * 8 + u] = arow[mb] ? *reinterpret_cast<const bf16x8 *>(ap[mb] + (beg + tt) * 256 + u * 32) : zero8();
            }
        if (beg < end) stage_in(0, beg);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int t = beg + tt;
            if (t < end) {
                if (t + 1 < end) {
                    stage_in((tt & 1) ^ 1, t + 1);
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                const char *wb = &wbuf[wid][tt & 1][0] + l15 * 256 + (lg & 1) * 8;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                    const u32x2_t q = *reinterpret_cast<const u32x2_t *>(wb + (((u * 2 + (lg >> 1)) ^ l15) << 4));
                    u32x4_t wbits;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], true);
                        const float l0 = lo.x, l1 = lo.y, h0 = hi2.x, h1 = hi2.y;
                        wbits[2 * h] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
                        wbits[2 * h + 1] = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(h0), 0x07060302u);
                    }
                    const bf16x8 wv = __builtin_bit_cast(bf16x8, wbits);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][tt * 8 + u], wv, acc[mb], 0, 0, 0);
                }
            }
        }
    } else {
    if (beg < end) stage_in(0, beg);
        for (int t = beg; t < end; ++t) {
            const int cur = (t - beg) & 1;
            bf16x8 av[MB][8];
    #pragma unroll
            for (int mb = 0; mb < MB; ++mb)
    #pragma unroll
                for (int u = 0; u < 8; ++u) av[mb][u] = arow[mb] ? *reinterpret_cast<const bf16x8 *>(ap[mb] + t * 256 + u * 32) : zero8();
            if (t + 1 < end) {
                stage_in(cur ^ 1, t + 1);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // the 4 pieces of tile t (older than the 4 just issued)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const char *wb = &wbuf[wid][cur][0] + l15 * 256 + (lg & 1) * 8;
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                const u32x2_t q = *reinterpret_cast<const u32x2_t *>(wb + (((u * 2 + (lg >> 1)) ^ l15) << 4));
                u32x4_t wbits;
    #pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // (element reads through float variables: __builtin_bit_cast of a vector subscript picks element 0 twice here)
                    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], true);
                    const float l0 = lo.x, l1 = lo.y, h0 = hi2.x, h1 = hi2.y;
                    // bf16 pair = high halves of the two floats (exact)
                    wbits[2 * h] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
                    wbits[2 * h + 1] = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(h0), 0x07060302u);
                }
                const bf16x8 wv = __builtin_bit_cast(bf16x8, wbits);
    #pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][u], wv, acc[mb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid][mb][lane][r] = acc[mb][r];
    __syncthreads();
    if (wid < MB) {
        const int col = n0 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[w][wid][lane][r];
            const int row = wid * 16 + lg * 4 + r;
            if (row < g.M && col < g.N) {
                if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * (16 * MB) + row) * g.N + col] = v;
            }
        }
    }
}

// e4m3 bytes -> bf16 (exact), 16 bytes per thread
__global__ void w8_expand_kernel(const uint8_t *__restrict__ src, bf16 *__restrict__ dst, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const u32x4_t q = *reinterpret_cast<const u32x4_t *>(src + i * 16);
    u32x4_t o[2];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], true);
        const float l0 = lo.x, l1 = lo.y, h0 = hi2.x, h1 = hi2.y;
        o[h >> 1][(h & 1) * 2] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
        o[h >> 1][(h & 1) * 2 + 1] = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(h0), 0x07060302u);
    }
    *reinterpret_cast<u32x4_t *>(dst + i * 16) = o[0];
    *reinterpret_cast<u32x4_t *>(dst + i * 16 + 8) = o[1];
}

__global__ void skinny_reduce_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.M * g.N) return;
    const int row = idx / g.N, col = idx - row * g.N;
    float v = 0.0f;
    for (int s = 0; s < a.ks; ++s) v += a.part[((int64_t)s * a.mr + row) * g.N + col];
    skinny_epilogue(g, row, col, v);
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

static int launch_gemm_core(const GemmArgs &g_in, int prof_kind, hipStream_t s, bool *ln_done);

// GemmArgs::ln_out: the LayerNorm of the output rows is either produced by the split-K reduction of the decode GEMV (launch_gemm_core
// sets ln_done) or by a LayerNorm launch here.
int launch_gemm(const GemmArgs &g, int prof_kind, hipStream_t s) {
    bool ln_done = false;
    const int rc = launch_gemm_core(g, prof_kind, s, &ln_done);
    if (rc != EILEV_OK || !g.ln_out || ln_done || g.M <= 0) return rc;
    if (g.out_f32 || g.patch_group) return EILEV_E_UNSUPPORTED;
    return launch_layernorm(reinterpret_cast<const bf16 *>(g.C), g.ldc, g.ln_gamma, g.ln_beta, g.ln_out, g.N, g.M, g.N, g.ln_eps, s);
}

static int launch_gemm_core(const GemmArgs &g_in, int prof_kind, hipStream_t s, bool *ln_done) {
    GemmArgs g = g_in;
    g.dbg = g_gemm_debug;
    g.trace = g_gemm_trace;
    g.trace_tiles = g_gemm_trace_tiles;
    if (g.dbg & 4096) g.lda = 0;   // probe: every A row aliases row 0 (cache-resident operand)
    if (g.dbg & 8192) g.ldw = 0;   // probe: every W row aliases row 0
    if (g.dbg & 131072) g.ldc = 0;  // probe: every output row aliases row 0 (stores stay in L2)
    if (g.M <= 0) return EILEV_OK;
    if (g.A8) {
        // fp8 activations x fp8 weights (eilev_linear_a8w8): the persistent ping-pong kernel on the fp8 MFMA, general epilogue
        // with the row and column scales
        if (!g.W8 || !g.C || !g.wscale || !g.ascale || g.N <= 0 || g.K <= 0) return EILEV_E_BADARG;
        if ((g.K % 128) || g.lda != g.K || g.ldw != g.K || ((uintptr_t)g.A8 & 15) || ((uintptr_t)g.W8 & 15) || g.patch_group != 0 ||
            (int64_t)g.M * g.K >= 0x7fff0000ll || (int64_t)g.N * g.K >= 0x7fff0000ll)
            return EILEV_E_UNSUPPORTED;
        if (!g.out_f32 && ((g.ldc & 7) || (g.N & 3) || ((uintptr_t)g.C & 15) || (g.resid && ((g.ldr & 7) || ((uintptr_t)g.resid & 15))))) return EILEV_E_UNSUPPORTED;
        if (prof_kind >= 0) prof_begin(prof_kind, 2.0 * g.M * (double)g.N * g.K, s);
        const int rc8 = launch_pp4(g, s);
        if (prof_kind >= 0) prof_end(s);
        return rc8;
    }
    if (g.W8) {
        // fp8 weights.  M <= 32 (decode): streamed as bytes by the fp8 skinny kernel.  Larger M (prefill): expanded to bf16 in the
        // caller's scratch (exact), then the bf16 kernels with the per-channel scale in their epilogue.
        if (!g.A || !g.C || !g.wscale || g.N <= 0 || g.K <= 0 || g.ldw != g.K || ((uintptr_t)g.W8 & 15) || (g.K & 15)) return EILEV_E_BADARG;
        if (!(g.M <= 32 && g.K % 256 == 0 && g.patch_group == 0)) {
            if (!g.w8_scratch) return EILEV_E_WORKSPACE;
            const int64_t n16 = (int64_t)g.N * g.K / 16;
            hipLaunchKernelGGL(w8_expand_kernel, dim3((unsigned)ceil_div64(n16, 256)), dim3(256), 0, s, g.W8, g.w8_scratch, n16);
            EILEV_LAUNCH_CHECK();
            GemmArgs e = g_in;
            e.W = g.w8_scratch;
            e.W8 = nullptr;
            return launch_gemm_core(e, prof_kind, s, ln_done);
        }
        g.W = reinterpret_cast<const bf16 *>(g.W8);  // (never dereferenced as bf16: the skinny fp8 kernel reads W8)
    }
    if (!g.A || !g.W || !g.C || g.N <= 0 || g.K <= 0) return EILEV_E_BADARG;
    if ((g.K & 7) || (g.lda & 7) || (g.ldw & 7) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return EILEV_E_UNSUPPORTED;
    const bool dma_ok = g.K % 256 == 0 && (!(g.dbg & 8) || g.W8);
    const bool ln_fold = g.ln_rows != nullptr || g.stat_out != nullptr;  // LayerNorm-folding variants: the persistent kernel only
    if (ln_fold && (g.W8 || g.wscale || g.out_f32 || g.patch_group || g.scale_cols || g.K % BK || (int64_t)g.N * g.ldw * 2 >= 0x7fff0000ll)) return EILEV_E_UNSUPPORTED;
    const bool skinny = (g.M <= 16 || (g.M <= 32 && dma_ok)) && g.patch_group == 0 && !ln_fold;
    if (!skinny && !g.out_f32 && ((g.ldc & 7) || (g.N & 3) || ((uintptr_t)g.C & 15) || (g.resid && ((g.ldr & 7) || ((uintptr_t)g.resid & 15))) ||
                                   (g.bias && ((uintptr_t)g.bias & 7))))
        return EILEV_E_UNSUPPORTED;
    int rc;
    if (skinny) {
        SkinnyArgs a;
        a.g = g;
        a.mr = g.M <= 16 ? 16 : 32;
        const int nb = (g.N + 15) / 16;
        int ks = nb >= 384 ? 1 : (512 + nb - 1) / nb;  // >= 1.5 workgroups per CU: no split (and no reduce launch); (1024: decode 5.03 -> 5.18 ms/token)
        const int ksteps = (g.K + 31) / 32;
        if (ks > ksteps / 32) ks = ksteps / 32 > 0 ? ksteps / 32 : 1;  // >= 8 K-steps of 32 per wave
        if (ks > 1 && (!g.scratch || (size_t)ks * a.mr * g.N * sizeof(float) > g.scratch_bytes)) ks = 1;
        a.ks = ks;
        a.part = g.scratch;
        // tiles of 256 per wave: ceil(ceil(K / 256 / ks) / 4); up to 3 (every decode shape) the activations are preloaded
        const int per_w = ((g.K / 256 + ks - 1) / ks + 3) / 4;
        const bool pre = per_w <= 3 && !(g.dbg & 128);
        // weight blocks per workgroup (activation fragments reused): probe override (dbg >> 26) & 7 = 1 / 2 / 4; default by shape below
        int ks32 = 0;
        int nbsel = (g.dbg >> 26) & 7;
        // measured at M = 32 (tools/skinny_sweep.py, 2 LDS stages so that two workgroups share a CU): lm_head (3142 blocks) 2.82 -> 3.45 /
        // 3.76 / 4.20 TB/s with 2 / 4 / 8 blocks per workgroup, qkv (480) 2.25 -> 2.46 with 2 (1.71 with 4: 120 workgroups leave CUs idle),
        // fc1 (640) 2.17 -> 2.30 with 2; the 2560-row matrices and M <= 16 are best with one block: keep >= 240 workgroups
        if (nbsel == 0) {
            nbsel = g_skinny_nb_default;
            // (workgroups = blocks x K splits: fc2 of OPT-2.7B has 160 blocks x 4 splits; probe flag 1 << 30: count blocks only, as before)
            const int wgs = (g.dbg & 1073741824) ? nb : nb * ks;
            if (g.M > 16) nbsel = wgs >= 8 * 240 ? 8 : (wgs >= 4 * 240 ? 4 : (wgs >= 2 * 240 ? 2 : 1));
        }
        if (nbsel == 7) nbsel = 8;  // probe encoding
        if (nbsel != 2 && nbsel != 4 && nbsel != 8) nbsel = 1;
        if (nbsel == 8 && g.M <= 16) nbsel = 4;
        if (g.W8) nbsel = 1;
        if (g.W8 && g.M > 16) {
            if (pre) hipLaunchKernelGGL((gemm_skinny_w8_kernel<2, true>), dim3(nb, ks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_skinny_w8_kernel<2, false>), dim3(nb, ks), dim3(256), 0, s, a);
        } else if (g.W8) {
            if (pre) hipLaunchKernelGGL((gemm_skinny_w8_kernel<1, true>), dim3(nb, ks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_skinny_w8_kernel<1, false>), dim3(nb, ks), dim3(256), 0, s, a);
        } else if (g.M > 16 && !(g.dbg & 268435456) && rows32_plan(g, nb, skinny_n_cu(), a, ks, ks32)) {
            // round 4 (gemm_rows32_kernel): one workgroup per CU, the 32 rows loaded once per CU.  probe flag 1 << 28: the kernels below
            const int grid_x = nb < skinny_n_cu() ? nb : skinny_n_cu();
            if (ks32 == 10) hipLaunchKernelGGL((gemm_rows32_kernel<2, 10, 3>), dim3(grid_x, ks), dim3(512), 0, s, a);
            else hipLaunchKernelGGL((gemm_rows32_kernel<2, 5, 4>), dim3(grid_x, ks), dim3(512), 0, s, a);
        } else if (dma_ok && pre && nbsel > 1) {
            // activations held across NB weight blocks per workgroup (see gemm_skinny_nb_kernel): 2 LDS-DMA stages + ping-pong partials =
            // 64 KB + 2 x MB x 4 KB, so two workgroups share a CU
            const int mbk = g.M > 16 ? 2 : 1;
            const int grid_x = (nb + nbsel - 1) / nbsel;
            const size_t sm = 4 * 2 * 8192 + (size_t)2 * 4 * mbk * 64 * 4 * 4;
            static bool attr_nb = false;
            if (!attr_nb) {
                const int mx = 4 * 2 * 8192 + 2 * 4 * 2 * 64 * 4 * 4;
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<2, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<2, 8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<1, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_nb_kernel<1, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, mx));
                attr_nb = true;
            }
            if (mbk == 2 && nbsel == 8) hipLaunchKernelGGL((gemm_skinny_nb_kernel<2, 8, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
            else if (mbk == 2 && nbsel == 4) hipLaunchKernelGGL((gemm_skinny_nb_kernel<2, 4, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
            else if (mbk == 2) hipLaunchKernelGGL((gemm_skinny_nb_kernel<2, 2, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
            else if (nbsel == 4) hipLaunchKernelGGL((gemm_skinny_nb_kernel<1, 4, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
            else hipLaunchKernelGGL((gemm_skinny_nb_kernel<1, 2, 2>), dim3(grid_x, ks), dim3(256), sm, s, a);
        } else if (dma_ok && g.M > 16) {
            if (pre) hipLaunchKernelGGL((gemm_skinny_dma_kernel<2, true>), dim3(nb, ks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_skinny_dma_kernel<2, false>), dim3(nb, ks), dim3(256), 0, s, a);
        } else if (dma_ok) {
            if (pre) hipLaunchKernelGGL((gemm_skinny_dma_kernel<1, true>), dim3(nb, ks), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_skinny_dma_kernel<1, false>), dim3(nb, ks), dim3(256), 0, s, a);
        }
        else hipLaunchKernelGGL(gemm_skinny_kernel, dim3(nb, ks), dim3(256), 0, s, a);
        EILEV_LAUNCH_CHECK();
        if (ks > 1) {
            if (g.ln_out && !(g.dbg & 536870912) && !g.out_f32 && g.epi == 0 && g.scale_cols == 0 && (g.N & 7) == 0 && g.N <= 4096 && (g.ldc & 7) == 0 && (!g.resid || (g.ldr & 7) == 0)) {
                // split-K partials -> row (+ bias + residual) -> its LayerNorm in one launch (norm.hip)
                const int rc_ln = launch_reduce_ln(a.part, ks, a.mr, g.M, g.N, g.wscale, g.bias, g.resid, g.ldr, reinterpret_cast<bf16 *>(g.C), g.ldc,
                                                   g.ln_gamma, g.ln_beta, g.ln_out, g.ln_eps, s);
                if (rc_ln != EILEV_OK) return rc_ln;
                *ln_done = true;
                return EILEV_OK;
            }
            const int total = g.M * g.N;
            hipLaunchKernelGGL(skinny_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
            EILEV_LAUNCH_CHECK();
        }
        return EILEV_OK;
    }
    // One wave per SIMD, 128 x 128 per wave, hand-scheduled K loop (gemm_a4.h; per-tile descriptors: no 2 GiB limit, so before the chunking
    // below).  OPT-IN (probe flag 10 << 4; tests/test_gemm_a4.py): same-box against the ping-pong kernel at the ViT launch shapes (round 3,
    // profiles/r03_a4_vs_pp4.log) its K loop is 5 % faster (1400-1450 vs 1340-1370 TFLOP/s) and bias-only GEMMs gain 1.7-3.5 % (fc1 without
    // GELU 1239 vs 1218, qkv 1207 vs 1178), but every residual epilogue loses (fc2 1141-1152 vs 1162, proj 954 vs 1020): with one wave per
    // SIMD nothing covers the residual loads that queue behind the next tile's LDS-DMA pieces.  The bench path runs the LayerNorm-folded
    // forms (consumer variant not built for this kernel), so nothing dispatches here by default.
    {
        const int force0 = (g.dbg >> 4) & 15;
        const bool a4_ok = g.K % BK == 0 && g.K >= 192 && g.N % 128 == 0 && !g.out_f32 && g.patch_group == 0 && g.scale_cols == 0 && !g.wscale && !g.ascale &&
                           !g.W8 && !g.ln_rows && !g.ln_out && g.k_slice == 0 && g.M > 32 && (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0 &&
                           (!g.resid || ((g.ldr & 7) == 0 && ((uintptr_t)g.resid & 15) == 0)) && (!g.bias || ((uintptr_t)g.bias & 7) == 0) &&
                           (!g.stat_out || g.resid) && (!g.resid || g.epi == 0) && (int64_t)256 * g.lda * 2 < 0x7fff0000ll && !(g.dbg & (4 | 1 | 2048 | 2));
        if (a4_ok && force0 == 10) {
            if (prof_kind >= 0) prof_begin(prof_kind, 2.0 * g.M * (double)g.N * g.K, s);
            const int rc_a4 = launch_a4(g, s);
            if (prof_kind >= 0) prof_end(s);
            return rc_a4;
        }
    }
    // The LDS-DMA kernels address A through a 32-bit buffer offset: an A operand of 2 GiB or more (the Q-Former k|v
    // projection of a whole step: 1.1 M rows x 1408) is processed as row chunks that fit, each with the fast kernels
    const int64_t a_bytes = (int64_t)g.M * g.lda * 2;
    // (the persistent ping-pong kernel addresses A per tile: shapes it takes — >= 1024 tiles of 256 x 256 — are not chunked)
    const int64_t t256_pre = ceil_div64(g.M, 256) * ceil_div64(g.N, 256);
    const bool pp4_takes = g.K % BK == 0 && g.patch_group == 0 && t256_pre >= 1024 && (g.N >= 2048 || ceil_div64(g.M, 256) * ceil_div64(g.N, 128) >= 512) &&
                           (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll && !g.dbg;
    if (a_bytes >= 0x7fff0000ll && g.K % BK == 0 && g.patch_group == 0 && !g.dbg && !pp4_takes) {
        const int64_t rows_per = (0x7fff0000ll / (g.lda * 2)) / 256 * 256;
        if (rows_per >= 256) {
            for (int64_t r0 = 0; r0 < g.M; r0 += rows_per) {
                GemmArgs c = g_in;
                c.M = (int)((g.M - r0) < rows_per ? (g.M - r0) : rows_per);
                c.A = g.A + r0 * g.lda;
                if (g.resid) c.resid = g.resid + r0 * g.ldr;
                if (g.stat_out) c.stat_out = g.stat_out + r0 * 2;  // stat_ld stays the row count of the whole matrix
                if (g.ln_rows) c.ln_rows = g.ln_rows + r0 * 2;
                c.C = g.out_f32 ? (void *)(reinterpret_cast<float *>(g.C) + r0 * g.ldc) : (void *)(reinterpret_cast<bf16 *>(g.C) + r0 * g.ldc);
                c.ln_out = nullptr;  // (the caller normalises the whole matrix once)
                const int rc_chunk = launch_gemm(c, prof_kind, s);
                if (rc_chunk != 0) return rc_chunk;
            }
            return EILEV_OK;
        }
    }
    const double flops = 2.0 * g.M * (double)g.N * g.K;
    if (ln_fold) {
        if (prof_kind >= 0) prof_begin(prof_kind, flops, s);
        const int rc_ln = launch_pp4(g, s);
        if (prof_kind >= 0) prof_end(s);
        return rc_ln;
    }
    if (prof_kind >= 0) prof_begin(prof_kind, flops, s);
    const int force = (g.dbg >> 4) & 15;  // probe-only override of the tile choice
    const int64_t tm256 = ceil_div64(g.M, 256);
    int cfg;
    if (tm256 * ceil_div64(g.N, 256) >= 256 && g.N >= 2048) cfg = 1;        // 256x256, 2 LDS stages, 1 WG/CU
    else if (tm256 * ceil_div64(g.N, 128) >= 512) cfg = 3;                   // 256x128, 1 stage, 2 WG/CU (N = 1408 / 1536)
    else if (tm256 * ceil_div64(g.N, 128) >= 192) cfg = 2;                   // 256x128, 2 stages
    else cfg = 4;                                                            // 128x128
    bool wide_tiles = false;
    const bool w6_ok = g.K % 64 == 0 && g.K >= 256 && g.N % 128 == 0 && (!g.resid || (g.epi == 0 && (g.ldr & 7) == 0)) && !g.out_f32 && g.patch_group == 0 && g.scale_cols == 0 && !g.wscale &&
                       (int64_t)g.M * g.lda * 2 < 0x7fff0000ll && (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll && (g.ldc & 7) == 0;
    // N = 1408 / 1536 with >= 512 column-half tiles: the persistent kernels win at every row count measured (fc2 at 34 952 rows:
    // per-tile 696 us, ping-pong 641, one-wave-per-SIMD 582; at 279 616 rows the ping-pong kernel despite its N padding).  With
    // fewer tiles (192-511 halves: 17-34 frames) the one-wave-per-SIMD kernel alone wins (fc2 at 4369 rows: 104 -> 77 us; pp4 124)
    if (cfg == 3 && g.K % BK == 0 && !(g.dbg & 16384)) { cfg = 1; wide_tiles = true; }
    if (cfg == 2 && w6_ok && force == 0 && !(g.dbg & (16384 | 2097152 | 4))) { cfg = 1; wide_tiles = true; }
    if (force == 9 || force == 10) cfg = 1;  // probe: persistent kernel regardless of the shape
    if (force == 13 || force == 14 || force == 15) cfg = 4;  // probe: 64x128 / 128x128 tiles / split-K
    else if (force >= 1 && force <= 4) cfg = force;
    // one-wave-per-SIMD continuous-stream kernel (256 x 128 tiles): its smaller tiles balance better when there are fewer than
    // 4 rounds of 256 x 256 tiles (M = 7680 prefill GEMMs: +28 %); with more tiles the ping-pong kernel with the lean epilogue wins
    // (qkv +5 %, OPT out_proj +3 %)
    // Round 5 (profiles/r05_w6_vs_pp4_rows.log: rows swept 3840 .. 30 720 at N = 2048 / 2560 / 6144 / 7680 / 10 240): which of the two wins is
    // the wave quantisation of its tile count over the CUs — 256 x 256 tiles fill ceil(t / CUs) rounds, the 256 x 128 tiles of w6 twice as
    // many half-sized ones — times the ping-pong kernel's ~6 % higher rate at equal fill (e.g. flan-t5-xl wo / o at 30 720 rows: 960 tiles =
    // 3.75 rounds, ping-pong 1144 / 995 TFLOP/s against 1071 / 864; OPT qkv at 15 360 rows: 1800 tiles = 7.03 rounds, w6 1166 against 1101).
    // The old rule (w6 below 1024 tiles) stays for N that is not a whole number of 256-column tiles.
    const int64_t tiles256 = tm256 * ceil_div64(g.N, 256);
    const int n_cu_q = skinny_n_cu() / 8 * 8;
    auto fill = [&](int64_t t) { return (double)t / (double)(ceil_div64(t, n_cu_q) * n_cu_q); };
    const bool w6_pick = cfg == 1 && (g.N % 256 == 0 ? 1.06 * fill(tiles256) < fill(tm256 * ceil_div64(g.N, 128)) : tiles256 < 1024);
    if ((force == 12 || (force == 0 && w6_pick && !(g.dbg & (2097152 | 4)))) && w6_ok)
        rc = launch_w6(g, s);
    else if (cfg == 1 && !(wide_tiles && (g.dbg & 1048576)) && (force == 0 || force == 9) && !(g.dbg & 4) && g.K % BK == 0 && (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll)
        rc = launch_pp4(g, s);  // persistent ping-pong kernel
    else if (cfg == 1) rc = launch_tiled<256, 256, 2, 4, 2, 2>(g, s);
    else if (cfg == 3) rc = launch_tiled<256, 128, 4, 2, 1, 4, 1>(g, s);
    else if (cfg == 2) rc = launch_tiled<256, 128, 4, 2, 2, 2>(g, s);
    else if (cfg == 4 && g.out_f32 && !g.bias && !g.resid && g.epi == 0 && !g.wscale && g.scale_cols == 0 && g.patch_group == 0 && g.K % BK == 0 &&
             g.K >= 8192 && ceil_div64(g.M, 128) * ceil_div64(g.N, 128) <= 128 && (force == 0 || force == 15) && !(g.dbg & 4) &&
             (int64_t)g.M * g.lda * 2 < 0x7fff0000ll && (int64_t)g.N * g.ldw * 2 < 0x7fff0000ll) {
        // weight-gradient shape (dW = dY^T X: a few output tiles, K = rows of the step): split K over enough slices to fill the CUs
        const int tiles = (int)(ceil_div64(g.M, 128) * ceil_div64(g.N, 128)), nk = g.K / BK;
        int slices = 512 / tiles;
        slices = slices < 2 ? 2 : (slices > 16 ? 16 : slices);
        if (slices > nk / 8) slices = nk / 8 > 1 ? nk / 8 : 1;
        GemmArgs gs = g;
        gs.k_slice = (nk + slices - 1) / slices;
        slices = (nk + gs.k_slice - 1) / gs.k_slice;
        static bool attr_set = false;
        constexpr int smem = 2 * (128 + 128) * 128;
        if (!attr_set) {
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_glds_kernel<128, 128, 2, 2, 0, 2, 2, 0>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set = true;
        }
        EILEV_HIP_CHECK(hipMemsetAsync(g.C, 0, (size_t)g.M * g.ldc * sizeof(float), s));
        hipLaunchKernelGGL((gemm_glds_kernel<128, 128, 2, 2, 0, 2, 2, 0>), dim3(tiles, slices), dim3(256), smem, s, gs);
        const hipError_t le = hipGetLastError();
        rc = le == hipSuccess ? EILEV_OK : (int)le;
    }
    else if ((force == 0 || force == 13) && cfg == 4 && (force == 13 || ceil_div64(g.M, 128) * ceil_div64(g.N, 128) < 96) && g.M > 64 && !(g.dbg & 4))
        rc = launch_tiled<64, 128, 1, 2, 2, 3>(g, s);  // a handful of 128x128 tiles (Q-Former graph: 544 rows): 64x128, 2 waves, 3 WG/CU (+13 % at 544 x 768 x 768; slower from ~160 tiles on)
    else rc = launch_tiled<128, 128, 2, 2, 2, 2>(g, s);
    if (prof_kind >= 0) prof_end(s);
    return rc;
}
#endif  // EILEV_GEMM_PART != 2

#if EILEV_GEMM_PART != 1
// The fp8-MFMA and LayerNorm-folding instances of the persistent kernel (called by launch_pp4 with its grid).
int launch_pp4_ext(const GemmArgs &g, int grid, hipStream_t s) {
    constexpr int smem = PP4_SMEM;
    if (g.A8) {  // fp8 x fp8 on the fp8 MFMA: byte operands, K halved so that the kernel's 2-byte strides are byte strides
        static bool attr8 = false;
        if (!attr8) {
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr8 = true;
        }
        if (g.epi == 1) return EILEV_E_UNSUPPORTED;
        GemmArgs h = g;
        h.A = reinterpret_cast<const bf16 *>(g.A8);
        h.W = reinterpret_cast<const bf16 *>(g.W8);
        h.K = g.K / 2; h.lda = g.lda / 2; h.ldw = g.ldw / 2;
        if (g.epi == 2) hipLaunchKernelGGL((gemm_pp4_kernel<2, true>), dim3(grid), dim3(512), smem, s, h);
        else hipLaunchKernelGGL((gemm_pp4_kernel<0, true>), dim3(grid), dim3(512), smem, s, h);
        EILEV_LAUNCH_CHECK();
        return EILEV_OK;
    }
    // LayerNorm-folding variants: consumer (qkv, fc1 + GELU) / producer (proj, fc2 with the residual)
    static bool attr_ln = false;
    if (!attr_ln) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<1, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pp4_kernel<0, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_ln = true;
    }
    if (g.epi == 2 || g.out_f32 || (g.ln_rows && (g.resid || g.stat_out || !g.ln_csum || ((uintptr_t)g.ln_csum & 15) || ((uintptr_t)g.ln_rows & 7))) ||
        (g.stat_out && (!g.resid || g.epi != 0 || g.stat_ld < g.M || ((uintptr_t)g.stat_out & 7))))
        return EILEV_E_UNSUPPORTED;
    if (g.stat_out) hipLaunchKernelGGL((gemm_pp4_kernel<0, false, 2>), dim3(grid), dim3(512), smem, s, g);
    else if (g.epi == 1) hipLaunchKernelGGL((gemm_pp4_kernel<1, false, 1>), dim3(grid), dim3(512), smem, s, g);
    else hipLaunchKernelGGL((gemm_pp4_kernel<0, false, 1>), dim3(grid), dim3(512), smem, s, g);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
#endif
#endif  // EILEV_GEMM_PART != 3
