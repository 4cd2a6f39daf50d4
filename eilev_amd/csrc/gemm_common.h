// gemm_common.h — what the bf16 GEMM kernel families of this directory share: the K-step geometry, the LDS swizzle, the workgroup -> tile
// map (XCD-aware, grouped) and the general epilogue (bias / activation / residual / LayerNorm statistics / fp32 / patch remap).  Included by
// gemm.hip (dispatch + the per-tile, one-wave-per-SIMD and weight-streaming families) and gemm_pp4_ext.hip (the fp8 / LayerNorm-folding
// instances of the persistent ping-pong kernel).
#pragma once
#include "common.h"
// Cache policy of the persistent kernel's output stores: 2 = nt (streaming).  A launch writes 0.8-3.4 GB through eight 4-MiB L2s whose
// job is to keep the A / W panels of the ~32 tiles in flight; nothing re-reads the output from L2.  Same-box A/B at the bench shapes
// (tools/gemm_ab.py, 2 runs): +0.2 ... +0.7 % on all five GEMMs; sc1 / sc1+nt: +-0.
#ifndef EILEV_ST_AUX
#define EILEV_ST_AUX 2
#endif
#include <type_traits>
#include <utility>

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


}  // namespace
