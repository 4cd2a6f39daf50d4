// gemv.hip — the language-model decode step at SMALL batch (M <= 8 rows: latency mode, beam search, one sample per GPU when a
// global step is strong-scaled over 8 GPUs): nn.Linear as row dot products, one output row per wave pass, with the work of the
// kernels AROUND the linear folded into its prologue / epilogue so that an OPT block is 5 launches instead of 8-9.
//
// Replaces, for one decode step of hf OPTDecoderLayer (modeling_opt.py:151-179, 226-247) via GenerationMixin._sample
// (ref:eilev/model/v2.py:318-322):
//   self_attn_layer_norm + q|k|v         -> gemv_rows_kernel<PRO_LN>            (LayerNorm recomputed per workgroup: M x 2560 values)
//   flash-decoding merge + out_proj + x  -> gemv_rows_kernel<PRO_MERGE / PRO_X> (no split-K, no reduce launch)
//   final_layer_norm + fc1 + ReLU        -> gemv_rows_kernel<PRO_LN>
//   fc2 + residual                       -> gemv_rows_kernel<PRO_X>
//   final_layer_norm + lm_head           -> gemv_rows_kernel<PRO_LN>, fp32 logits
// Round 2 measured the batch-1 step at 2.64 ms per token: per block 43.6 us of weight streaming and 40 us in seven small kernels, each
// at its 4.5-6 us launch floor (profiles/HISTORY.md §5 item 3).  The MFMA weight-streaming kernels (gemm.hip) stay for 9 <= M <= 32.
//
// Arithmetic: weights [N, K] row-major bf16; a wave owns output rows; lane l reads 16-byte chunks l, l + 64, ... of the row (1 KiB per
// load instruction, fully coalesced) and the matching chunks of the M activation rows from LDS (staged once per workgroup, bf16 — the
// values the separate LayerNorm / merge kernels would have stored); v_dot2c_f32_bf16 accumulates in fp32; 6 xor-shuffle steps reduce
// the 64 lane partials; the epilogue (bias, q scaling, ReLU, residual) runs on lane 0..M-1.
// HBM-bound: the weight bytes are read exactly once; the activations (M x K x 2 B) come from L2 once per workgroup.
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

enum { PRO_X = 0, PRO_LN = 1, PRO_MERGE = 2 };

struct GemvArgs {
    const bf16 *x;  // PRO_X / PRO_LN: (M, K) rows, leading dimension ldx
    int64_t ldx;
    const bf16 *gamma, *beta;  // PRO_LN
    float eps;
    const float *part;  // PRO_MERGE: flash-decoding partials (M, heads, nsplit, hd + 2): [max, sum, o[hd]]  (K = heads * hd)
    int heads, hd, nsplit;
    const bf16 *W;  // (N, K) row-major
    const bf16 *bias, *resid;
    int64_t ldr;
    void *out;  // bf16 or f32 (M, N), leading dimension ldo
    int64_t ldo;
    int out_f32;
    int M, N, K, KB;  // (KB = K: the whole rows are staged)
    int rows_per_wave;
    int epi;  // 0 none, 2 ReLU
    float scale;  // columns [0, scale_cols) are multiplied by scale after the bias (q pre-scaling)
    int scale_cols;
};

__device__ __forceinline__ float dot8(const u32x4_t &w, const u32x4_t &x, float acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // (copies first: __builtin_bit_cast applied directly to the vector element w[e] read element 0 for every e under hipcc 7.2)
        const unsigned we = w[e], xe = x[e];
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, we), __builtin_bit_cast(bf16x2_t, xe), acc, false);
    }
    return acc;
}

template <typename F, int... I>
__device__ __forceinline__ void static_for_i_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for_i(F &&f) {
    static_for_i_impl(f, std::make_integer_sequence<int, N>{});
}

// MR: the number of rows M, compile-time (1..8: a run-time bound made hipcc branch around every row's dot products).  The rows [M][K] are staged ONCE in LDS as bf16 (host: M * K * 2 <= 150 KB), the
// wave then streams its output rows' weights one row after the other in sub-blocks of up to 8 chunks (8 KiB per wave) through a register
// double buffer: the next sub-block's loads are in flight while the current one is multiplied.  The first sub-block is requested BEFORE
// the prologue (the weights do not depend on the activations), so LayerNorm / merge latency hides under the first loads.
// CPS: chunks (of 512 elements) per sub-block — compile-time, so that the loads and dot products of a sub-block are straight-line code
// (with a run-time count hipcc branched around every load and drained the queue between them); K / 512 is a multiple of CPS.
template <int MR, int PRO, int CPS>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16 *xs = reinterpret_cast<bf16 *>(smem);  // [M][K]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int M = MR;
    const int K = a.K, nch = K >> 9;  // 512-element chunks per row
    const int nsb = nch / CPS;                 // sub-blocks of CPS chunks per row
    const int rpw = a.rows_per_wave;
    const int n_first = (blockIdx.x * 4 + wid) * rpw;
    const int items = rpw * nsb;
    u32x4_t wv[2][CPS];
    auto load_item = [&](int it, auto buf_c) {
        constexpr int B = decltype(buf_c)::value;
        const int r = it / nsb, sb = it - r * nsb;
        const int n = n_first + r;
        const bf16 *wrow = a.W + (int64_t)(n < a.N ? n : a.N - 1) * K + sb * (CPS * 512) + lane * 8;
#pragma unroll
        for (int c = 0; c < CPS; ++c) wv[B][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(wrow + c * 512));
    };
    load_item(0, std::integral_constant<int, 0>{});

    // ---- prologue: the activation rows -> LDS (bf16) ------------------------------------------------------------------------------
    if constexpr (PRO == PRO_LN) {
        // one wave per row (rows wid, wid + 4), the row held in registers: mean, then the centred second moment, like layernorm_kernel
        for (int m = wid; m < M; m += 4) {
            const bf16 *row = a.x + (int64_t)m * a.ldx;
            float f[8][8];  // K <= 4096: up to 8 chunks of 8 per lane
            float s1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < nch) {
                    unpack8(*reinterpret_cast<const bf16x8 *>(row + c * 512 + lane * 8), f[c]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s1 += f[c][e];
                }
            const float mean = wave_sum(s1) / (float)K;
            float s2 = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < nch) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s2 = fmaf(f[c][e] - mean, f[c][e] - mean, s2);
                }
            const float rstd = rsqrtf(wave_sum(s2) / (float)K + a.eps);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < nch) {
                    float gm[8], bt[8];
                    unpack8(*reinterpret_cast<const bf16x8 *>(a.gamma + c * 512 + lane * 8), gm);
                    unpack8(*reinterpret_cast<const bf16x8 *>(a.beta + c * 512 + lane * 8), bt);
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)((f[c][e] - mean) * rstd * gm[e] + bt[e]);
                    *reinterpret_cast<bf16x8 *>(xs + (int64_t)m * K + c * 512 + lane * 8) = v;
                }
        }
    } else if constexpr (PRO == PRO_MERGE) {
        // merged attention rows: o = sum_s w_s o_s / sum_s w_s l_s, w_s = exp(max_s - max) (attn_decode_merge_kernel's arithmetic)
        for (int idx = tid * 8; idx < M * K; idx += 2048) {
            const int m = idx / K, k = idx - m * K;
            const int h = k / a.hd, t0 = k - h * a.hd;
            const float *pp = a.part + ((int64_t)m * a.heads + h) * a.nsplit * (a.hd + 2);
            float mx = -1e30f;
            for (int s = 0; s < a.nsplit; ++s) mx = fmaxf(mx, pp[s * (a.hd + 2)]);
            float l = 0.0f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < a.nsplit; ++s) {
                const float *ps = pp + s * (a.hd + 2);
                const float wgt = ps[1] > 0.0f ? __expf(ps[0] - mx) : 0.0f;
                l += wgt * ps[1];
                if (wgt > 0.0f) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += wgt * ps[2 + t0 + e];
                }
            }
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)(l > 0.0f ? o[e] / l : 0.0f);
            *reinterpret_cast<bf16x8 *>(xs + idx) = v;
        }
    } else {
        for (int idx = tid * 8; idx < M * K; idx += 2048) {
            const int m = idx / K, k = idx - m * K;
            *reinterpret_cast<bf16x8 *>(xs + idx) = *reinterpret_cast<const bf16x8 *>(a.x + (int64_t)m * a.ldx + k);
        }
    }
    __syncthreads();

    // ---- main loop: item = (row of this wave, sub-block of the row) ------------------------------------------------------------------
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.0f;
    auto compute_item = [&](int it, auto buf_c) {
        constexpr int B = decltype(buf_c)::value;
        const int r = it / nsb, sb = it - r * nsb;
        const bf16 *xb = xs + sb * (CPS * 512) + lane * 8;
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
                acc[m] = dot8(wv[B][c], *reinterpret_cast<const u32x4_t *>(xb + m * K + c * 512), acc[m]);
        }
        if (sb + 1 == nsb) {  // the row is complete: reduce over lanes, epilogue on lane m
            const int n = n_first + r;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const float t = wave_sum(acc[m]);
                acc[m] = 0.0f;
                if (lane == m && n < a.N) {
                    float v = t;
                    if (a.bias) v += (float)a.bias[n];
                    if (n < a.scale_cols) v *= a.scale;
                    if (a.epi == 2) v = fmaxf(v, 0.0f);
                    if (a.resid) v += (float)a.resid[(int64_t)m * a.ldr + n];
                    if (a.out_f32) reinterpret_cast<float *>(a.out)[(int64_t)m * a.ldo + n] = v;
                    else reinterpret_cast<bf16 *>(a.out)[(int64_t)m * a.ldo + n] = (bf16)v;
                }
            }
        }
    };
    for (int it = 0; it < items; it += 2) {
        if (it + 1 < items) load_item(it + 1, std::integral_constant<int, 1>{});
        compute_item(it, std::integral_constant<int, 0>{});
        if (it + 1 < items) {
            if (it + 2 < items) load_item(it + 2, std::integral_constant<int, 0>{});
            compute_item(it + 1, std::integral_constant<int, 1>{});
        }
    }
}

template <int MR, int PRO, int CPS>
int launch_rows_c(const GemvArgs &a, int grid, size_t smem, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemv_rows_kernel<MR, PRO, CPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL((gemv_rows_kernel<MR, PRO, CPS>), dim3(grid), dim3(256), smem, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

template <int MR, int PRO>
int launch_rows_m(const GemvArgs &a, int grid, size_t smem, hipStream_t s) {
    const int nch = a.K >> 9;  // sub-block = 8 or 5 chunks, whichever divides the row (2560: 5, 10240: 4 x 5, 4096: 8, 16384: 4 x 8); else 1
    if (nch % 8 == 0) return launch_rows_c<MR, PRO, 8>(a, grid, smem, s);
    if (nch % 5 == 0) return launch_rows_c<MR, PRO, 5>(a, grid, smem, s);
    return launch_rows_c<MR, PRO, 1>(a, grid, smem, s);
}

template <int PRO>
int launch_rows_p(const GemvArgs &a, int grid, size_t smem, hipStream_t s) {
    switch (a.M) {
        case 1: return launch_rows_m<1, PRO>(a, grid, smem, s);
        case 2: return launch_rows_m<2, PRO>(a, grid, smem, s);
        case 3: return launch_rows_m<3, PRO>(a, grid, smem, s);
        case 4: return launch_rows_m<4, PRO>(a, grid, smem, s);
        case 5: return launch_rows_m<5, PRO>(a, grid, smem, s);
        case 6: return launch_rows_m<6, PRO>(a, grid, smem, s);
        case 7: return launch_rows_m<7, PRO>(a, grid, smem, s);
        default: return launch_rows_m<8, PRO>(a, grid, smem, s);
    }
}


// lane reduction as DPP adds (VALU) instead of ds_bpermute round trips; the same tree in gemv1_kernel and gemvm_kernel, so a row's bits do
// not depend on how many rows share the launch
__device__ __forceinline__ float dpp_wave_sum(float v) {  // total of the 64 lanes, returned wave-uniform
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror: 16-lane sums
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)); // row_bcast15 -> rows 1, 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false)); // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}


// ---- M = 1 (round 4): activations in REGISTERS, no LDS, no barrier, a register ring of weight sub-blocks ----------------------------------
// Per-kernel durations of the batch-1 step (profiles/r04_decode_b1_kernel_stats.md): q|k|v 13 us for 39 MB, out_proj 11.3 us for 13 MB,
// fc1 15.4 / fc2 13.3 us for 52 MB each — 3-4 TB/s inside a kernel, ~2.2 TB/s over the block — where a launch-per-op batch-1 layer of
// this size streams at ~4 TB/s end to end on this part (MI355X_MICROARCH.md, launches-baseline: 121.6 MB in 30.7 us).  What the kernel
// above spends around its stream: the rows staged through LDS behind a __syncthreads with ONE sub-block of weights (5 KB per wave) in
// flight, a grid that covers the CUs unevenly (480 / 320 / 640 workgroups over 256 CUs), the merge of the attention partials repeated by
// every workgroup of out_proj.  Here, for one row:
//   * grid = one 320-thread workgroup per CU (5 waves: 1280 waves divide N = 2560 / 7680 / 10240 evenly), wave w owns contiguous output
//     rows: every CU streams the same number of bytes and the launch ends everywhere at once;
//   * lane l keeps ITS 8 elements of every 512-element chunk of x in registers (K = 10240: 80 VGPRs), loaded straight from L2 (PRO_X) or
//     produced by a per-wave LayerNorm (PRO_LN: the row is 5 KB; every wave recomputes mean / variance from registers) — no LDS, no barrier;
//   * RB sub-blocks of SB chunks (K = 2560: 8 x 5 KB = every byte of a wave's q|k|v or fc1 rows) are requested before anything else;
//     waves with more items than ring slots (the lm_head: 40 rows) run a branch-free steady loop — consume slot, refill slot — so that
//     hipcc's wait-count pass sees straight-line code and emits counted vmcnt waits (with a branch around a refill it joins the paths
//     with vmcnt(0): one sub-block in flight; inline-asm loads with hand-counted waits were tried first and are not safe: the compiler
//     may COPY an asm output register before the hand-written wait);
//   * bias / residual of the wave's rows come through the SCALAR cache (wave-uniform addresses): a vector load behind the prefetched
//     weights would return after them (returns are in order) and stall the ring once per row;
//   * results stay in lane (row - first row) until the end: one store per wave.
template <int NCH, int PRO, int SB, int RB>
__global__ __launch_bounds__(320) void gemv1_kernel(const GemvArgs a) {
    constexpr int IPR = NCH / SB;  // items (sub-blocks) per output row
    static_assert(NCH % SB == 0 && RB % IPR == 0, "ring / row geometry");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 5 + (threadIdx.x >> 6)));
    // rows [r0, r1) of this wave: N / W each, the first N % W waves one more (quotient and remainder from the host: a 64-bit division
    // here costs some two hundred scalar instructions before the first load)
    const int r0 = wave * a.rows_per_wave + (wave < a.KB ? wave : a.KB), r1 = r0 + a.rows_per_wave + (wave < a.KB ? 1 : 0);
    const int total = (r1 - r0) * IPR;
    constexpr int K = NCH * 512;
    u32x4_t wv[RB][SB];
    const bf16 *wbase = a.W + (int64_t)r0 * K + lane * 8;
    auto load_item = [&](int it, auto buf_c) {  // item it = (row it / IPR, sub-block it % IPR): SB * 512 consecutive weights
        constexpr int B = decltype(buf_c)::value;
        const bf16 *wrow = wbase + (int64_t)it * (SB * 512);
#pragma unroll
        for (int c = 0; c < SB; ++c) wv[B][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(wrow + c * 512));
    };
    u32x4_t xr[NCH];
    auto load_x = [&]() {  // the row of activations: lane l holds elements [c * 512 + l * 8, + 8) of every chunk c
        if constexpr (PRO == PRO_LN) {
            // the row stays bf16 in registers (NCH x 4 VGPRs) and is unpacked in each of the three passes: an fp32 copy (8 NCH registers)
            // beside the ring made hipcc spill
            bf16x8 xb[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) xb[c] = *reinterpret_cast<const bf16x8 *>(a.x + c * 512 + lane * 8);
            float s1 = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float f[8];
                unpack8(xb[c], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s1 += f[e];
            }
            const float mean = wave_sum(s1) / (float)K;
            float s2 = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float f[8];
                unpack8(xb[c], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s2 = fmaf(f[e] - mean, f[e] - mean, s2);
            }
            const float rstd = rsqrtf(wave_sum(s2) / (float)K + a.eps);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float f[8], gm[8], bt[8];
                unpack8(xb[c], f);
                unpack8(*reinterpret_cast<const bf16x8 *>(a.gamma + c * 512 + lane * 8), gm);
                unpack8(*reinterpret_cast<const bf16x8 *>(a.beta + c * 512 + lane * 8), bt);
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)((f[e] - mean) * rstd * gm[e] + bt[e]);
                xr[c] = __builtin_bit_cast(u32x4_t, v);
            }
        } else if constexpr (PRO == PRO_MERGE) {
            // x = the attention row, merged HERE from the flash-decoding partials of attn_decode_part_kernel (o = sum_s w_s o_s / sum_s w_s l_s,
            // w_s = exp(max_s - max): attn_decode_merge_kernel's arithmetic).  Wave w of the workgroup merges chunk w — its lane's 8 elements
            // belong to one head (hd % 8 == 0), all 8 x 5 loads of the (up to) 8 splits in flight at once — and the five chunks meet in LDS:
            // one L2 round trip and one barrier.  (Every wave merging all five chunks in a run-time loop over the splits: 31.7 us per launch.)
            static_assert(NCH == 5, "one chunk per wave of the 320-thread workgroup");
            __shared__ __attribute__((aligned(16))) bf16 xm[NCH * 512];
            const int wv_ = threadIdx.x >> 6, ns = a.nsplit;
            {
                const int k = wv_ * 512 + lane * 8, hh = k / a.hd, t0 = k - hh * a.hd;
                const float *pp = a.part + (int64_t)hh * ns * (a.hd + 2);
                float2 ml[8], ov[8][4];
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) {
                    const float *pq = pp + (sp < ns ? sp : 0) * (a.hd + 2);
                    ml[sp] = *reinterpret_cast<const float2 *>(pq);
                    const float2 *po2 = reinterpret_cast<const float2 *>(pq + 2 + t0);  // (hd + 2 and t0 are even: 8-byte aligned)
#pragma unroll
                    for (int q = 0; q < 4; ++q) ov[sp][q] = po2[q];
                }
                float mx = -1e30f;
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) mx = fmaxf(mx, sp < ns ? ml[sp].x : -1e30f);
                float l = 0.0f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) {
                    const float wgt = (sp < ns && ml[sp].y > 0.0f) ? __expf(ml[sp].x - mx) : 0.0f;
                    l += wgt * ml[sp].y;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[2 * q] += wgt * ov[sp][q].x;
                        o[2 * q + 1] += wgt * ov[sp][q].y;
                    }
                }
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)(l > 0.0f ? o[e] / l : 0.0f);
                *reinterpret_cast<bf16x8 *>(xm + k) = v;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NCH; ++c) xr[c] = *reinterpret_cast<const u32x4_t *>(xm + c * 512 + lane * 8);
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c) xr[c] = *reinterpret_cast<const u32x4_t *>(a.x + c * 512 + lane * 8);
        }
    };
    float acc = 0.0f, mine = 0.0f;  // mine: the finished value of row r0 + lane (mod 64)
    const unsigned *bias32 = reinterpret_cast<const unsigned *>(a.bias), *res32 = reinterpret_cast<const unsigned *>(a.resid);
    auto finish_row = [&](int r) {  // r wave-uniform
        const int n = r0 + r;
        float v = dpp_wave_sum(acc);
        acc = 0.0f;
        if (bias32) {  // wave-uniform address: one scalar load, no entry in the vector-memory queue
            const unsigned w2 = bias32[n >> 1];
            v += __builtin_bit_cast(float, (n & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
        }
        if (n < a.scale_cols) v *= a.scale;
        if (a.epi == 2) v = fmaxf(v, 0.0f);
        if (res32) {
            const unsigned w2 = res32[n >> 1];
            v += __builtin_bit_cast(float, (n & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
        }
        if (lane == (r & 63)) mine = v;
        if ((r & 63) == 63 || r + 1 == r1 - r0) {  // 64 rows collected or the wave's last row: one store
            const int rr = (r & ~63) + lane;
            if (rr <= r) {
                if (a.out_f32) reinterpret_cast<float *>(a.out)[r0 + rr] = mine;
                else reinterpret_cast<bf16 *>(a.out)[r0 + rr] = (bf16)mine;
            }
        }
    };
    auto consume = [&](int it, auto j_c) {  // slot J holds item it; it % IPR == J % IPR (every base is a multiple of RB, RB % IPR == 0)
        constexpr int J = decltype(j_c)::value, SBI = J % IPR;
#pragma unroll
        for (int c = 0; c < SB; ++c) acc = dot8(wv[J][c], xr[SBI * SB + c], acc);
        if constexpr (SBI == IPR - 1) finish_row(it / IPR);
    };
    if (total >= 2 * RB) {
        // long waves (the lm_head): RB items in flight, branch-free steady loop
        static_for_i<RB>([&](auto j_c) { load_item(decltype(j_c)::value, j_c); });
        load_x();
        int base = 0;
        for (; base + 2 * RB <= total; base += RB)
            static_for_i<RB>([&](auto j_c) {
                consume(base + decltype(j_c)::value, j_c);
                load_item(base + decltype(j_c)::value + RB, j_c);
            });
        static_for_i<RB>([&](auto j_c) {
            consume(base + decltype(j_c)::value, j_c);
            if (base + decltype(j_c)::value + RB < total) load_item(base + decltype(j_c)::value + RB, j_c);
        });
        base += RB;
        static_for_i<RB>([&](auto j_c) {
            if (base + decltype(j_c)::value < total) consume(base + decltype(j_c)::value, j_c);
        });
    } else {
        // short waves (every block matrix at K = 2560: 6 / 2 / 8 items): all of the wave's weights are requested up front
        static_for_i<RB>([&](auto j_c) {
            if (decltype(j_c)::value < total) load_item(decltype(j_c)::value, j_c);
        });
        load_x();
        static_for_i<RB>([&](auto j_c) {
            constexpr int J = decltype(j_c)::value;
            if (J < total) {
                consume(J, j_c);
                if (J + RB < total) load_item(J + RB, j_c);
            }
        });
        static_for_i<RB>([&](auto j_c) {
            if (decltype(j_c)::value + RB < total) consume(decltype(j_c)::value + RB, j_c);
        });
    }
}

template <int NCH, int PRO>
int launch_gemv1_c(const GemvArgs &a, int grid, hipStream_t s) {
    constexpr int SB = NCH % 5 == 0 ? 5 : 8;
    constexpr int IPR = NCH / SB;
    // registers: ring RB * SB * 4 + x NCH * 4 (K = 2560: 160 + 20; K = 10240: 80 + 80); the LayerNorm prologue holds the row as 8 NCH floats
    // beside the ring (RB = 8 spilled 204 VGPRs there)
    constexpr int RB = IPR == 1 ? (SB == 5 ? (PRO == PRO_LN ? 4 : 8) : (PRO == PRO_LN ? 2 : 4)) : 4;
    hipLaunchKernelGGL((gemv1_kernel<NCH, PRO, SB, RB>), dim3(grid), dim3(320), 0, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}


// ---- 2 <= M <= 8 rows with the geometry of gemv1_kernel (round 4) -----------------------------------------------------------------------
// Beam search (5 beams of one sample: the sample script's default), two / four samples per GPU of a strong-scaled step.  The rows cannot
// live in registers (M x K / 64 values per lane), so they are staged ONCE per workgroup in LDS as bf16 — behind the weight ring's first
// RB sub-blocks, which are requested before the staging starts — and read back per dot product (M ds_read_b128 per 1 KiB of weights: LDS
// bandwidth is not the limit at these M).  Everything else is gemv1_kernel: one 320-thread workgroup per CU, contiguous equal row ranges
// per wave, all of a short wave's weights requested up front, branch-free steady loop for long waves (the lm_head), bias / residual
// through the scalar cache, one store per row of x at the end.  The lane reduction is DPP adds (VALU) instead of ds_bpermute round trips:
// with M accumulators per output row the shuffles were the longest chain in the kernel.
template <int MR, int NCH, int PRO, int SB, int RB, bool LONG>
__global__ __launch_bounds__(320) void gemvm_kernel(const GemvArgs a) {
    constexpr int IPR = NCH / SB, K = NCH * 512;
    static_assert(NCH % SB == 0 && RB % IPR == 0, "ring / row geometry");
    extern __shared__ __attribute__((aligned(16))) char smem_m[];
    bf16 *xs = reinterpret_cast<bf16 *>(smem_m);  // [MR][K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = blockIdx.x * 5 + wid;
    const int r0 = wave * a.rows_per_wave + (wave < a.KB ? wave : a.KB), r1 = r0 + a.rows_per_wave + (wave < a.KB ? 1 : 0);
    const int total = (r1 - r0) * IPR;
    u32x4_t wv[RB][SB];
    const bf16 *wbase = a.W + (int64_t)r0 * K + lane * 8;
    auto load_item = [&](int it, auto buf_c) {
        constexpr int B = decltype(buf_c)::value;
        const bf16 *wrow = wbase + (int64_t)it * (SB * 512);
#pragma unroll
        for (int c = 0; c < SB; ++c) wv[B][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(wrow + c * 512));
    };
    auto stage_x = [&]() {
        if constexpr (PRO == PRO_LN) {
            for (int m = wid; m < MR; m += 5) {  // one wave per row, the row in registers (bf16), mean then centred second moment
                const bf16 *row = a.x + (int64_t)m * a.ldx;
                bf16x8 xb[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) xb[c] = *reinterpret_cast<const bf16x8 *>(row + c * 512 + lane * 8);
                float s1 = 0.0f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    float f[8];
                    unpack8(xb[c], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s1 += f[e];
                }
                const float mean = wave_sum(s1) / (float)K;
                float s2 = 0.0f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    float f[8];
                    unpack8(xb[c], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s2 = fmaf(f[e] - mean, f[e] - mean, s2);
                }
                const float rstd = rsqrtf(wave_sum(s2) / (float)K + a.eps);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    float f[8], gm[8], bt[8];
                    unpack8(xb[c], f);
                    unpack8(*reinterpret_cast<const bf16x8 *>(a.gamma + c * 512 + lane * 8), gm);
                    unpack8(*reinterpret_cast<const bf16x8 *>(a.beta + c * 512 + lane * 8), bt);
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)((f[e] - mean) * rstd * gm[e] + bt[e]);
                    *reinterpret_cast<bf16x8 *>(xs + m * K + c * 512 + lane * 8) = v;
                }
            }
        } else {
            for (int idx = tid * 8; idx < MR * K; idx += 320 * 8) {
                const int m = idx / K, k = idx - m * K;
                *reinterpret_cast<bf16x8 *>(xs + idx) = *reinterpret_cast<const bf16x8 *>(a.x + (int64_t)m * a.ldx + k);
            }
        }
        __syncthreads();
    };
    float acc[MR], mine[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = mine[m] = 0.0f;
    const unsigned *bias32 = reinterpret_cast<const unsigned *>(a.bias);
    auto bf_at = [](const unsigned *p32, int64_t i) {  // element i of a bf16 array through a (wave-uniform) 4-byte scalar load
        const unsigned w2 = p32[i >> 1];
        return __builtin_bit_cast(float, (i & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
    };
    auto finish_row = [&](int r) {
        const int n = r0 + r;
        const float bv = bias32 ? bf_at(bias32, n) : 0.0f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float v = dpp_wave_sum(acc[m]) + bv;
            acc[m] = 0.0f;
            if (n < a.scale_cols) v *= a.scale;
            if (a.epi == 2) v = fmaxf(v, 0.0f);
            if (a.resid) v += bf_at(reinterpret_cast<const unsigned *>(a.resid), (int64_t)m * a.ldr + n);
            if (lane == (r & 63)) mine[m] = v;
        }
        if ((r & 63) == 63 || r + 1 == r1 - r0) {
            const int rr = (r & ~63) + lane;
            if (rr <= r) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    if (a.out_f32) reinterpret_cast<float *>(a.out)[(int64_t)m * a.ldo + r0 + rr] = mine[m];
                    else reinterpret_cast<bf16 *>(a.out)[(int64_t)m * a.ldo + r0 + rr] = (bf16)mine[m];
                }
            }
        }
    };
    auto consume = [&](int it, auto j_c) {
        constexpr int J = decltype(j_c)::value, SBI = J % IPR;
        const bf16 *xb = xs + SBI * (SB * 512) + lane * 8;
#pragma unroll
        for (int c = 0; c < SB; ++c) {
            __builtin_amdgcn_sched_barrier(0);  // one chunk's LDS reads at a time: left alone, hipcc hoists the reads of every slot and spills
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[m] = dot8(wv[J][c], *reinterpret_cast<const u32x4_t *>(xb + m * K + c * 512), acc[m]);
        }
        if constexpr (SBI == IPR - 1) finish_row(it / IPR);
    };
    if constexpr (LONG) {  // (host: every wave has at least 2 RB items)
        static_for_i<RB>([&](auto j_c) { load_item(decltype(j_c)::value, j_c); });
        stage_x();
        int base = 0;
        for (; base + 2 * RB <= total; base += RB)
            static_for_i<RB>([&](auto j_c) {
                consume(base + decltype(j_c)::value, j_c);
                load_item(base + decltype(j_c)::value + RB, j_c);
            });
        static_for_i<RB>([&](auto j_c) {
            consume(base + decltype(j_c)::value, j_c);
            if (base + decltype(j_c)::value + RB < total) load_item(base + decltype(j_c)::value + RB, j_c);
        });
        base += RB;
        static_for_i<RB>([&](auto j_c) {
            if (base + decltype(j_c)::value < total) consume(base + decltype(j_c)::value, j_c);
        });
    } else {
        static_for_i<RB>([&](auto j_c) {
            if (decltype(j_c)::value < total) load_item(decltype(j_c)::value, j_c);
        });
        stage_x();
        static_for_i<RB>([&](auto j_c) {
            constexpr int J = decltype(j_c)::value;
            if (J < total) {
                consume(J, j_c);
                if (J + RB < total) load_item(J + RB, j_c);
            }
        });
        static_for_i<RB>([&](auto j_c) {
            if (decltype(j_c)::value + RB < total) consume(decltype(j_c)::value + RB, j_c);
        });
    }
}

template <int MR, int NCH, int PRO>
int launch_gemvm_c(const GemvArgs &a, int grid, hipStream_t s) {
    constexpr int SB = 5, IPR = NCH / SB;
    // ring slots: 8 (K = 2560: every byte of a wave's block rows up front) where the registers allow; with 3+ rows hipcc hoists the LDS reads
    // of all slots and spills (MR = 5, 8 slots: 862 VGPRs spilled), so 4 there
    constexpr int RB = (PRO == PRO_LN || MR >= 3) ? 4 : 8;
    const size_t smem = (size_t)MR * NCH * 512 * sizeof(bf16);
    static bool attr = false;
    if (!attr) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemvm_kernel<MR, NCH, PRO, SB, RB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemvm_kernel<MR, NCH, PRO, SB, RB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // one code path per instance (both in one kernel doubled the register pressure): long waves only when EVERY wave has 2 RB items
    if (a.rows_per_wave * IPR >= 2 * RB) hipLaunchKernelGGL((gemvm_kernel<MR, NCH, PRO, SB, RB, true>), dim3(grid), dim3(320), smem, s, a);
    else hipLaunchKernelGGL((gemvm_kernel<MR, NCH, PRO, SB, RB, false>), dim3(grid), dim3(320), smem, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
template <int NCH, int PRO>
int launch_gemvm_m(const GemvArgs &a, int grid, hipStream_t s) {
    switch (a.M) {
        case 2: return launch_gemvm_c<2, NCH, PRO>(a, grid, s);
        case 3: return launch_gemvm_c<3, NCH, PRO>(a, grid, s);
        case 4: return launch_gemvm_c<4, NCH, PRO>(a, grid, s);
        case 5: return launch_gemvm_c<5, NCH, PRO>(a, grid, s);
        case 6: return launch_gemvm_c<6, NCH, PRO>(a, grid, s);
        case 7: return launch_gemvm_c<7, NCH, PRO>(a, grid, s);
        default: return launch_gemvm_c<8, NCH, PRO>(a, grid, s);
    }
}

}  // namespace

// M rows of K bf16 must fit the LDS staging (150 KB): M = 8 with K = 10240 (160 KB) does not — those shapes keep the MFMA kernels
bool gemv_rows_ok(int M, int N, int K) { return M >= 1 && M <= 8 && K % 512 == 0 && K >= 512 && N >= 1 && (int64_t)M * K * 2 <= 150 * 1024; }

// C[M, N] = epi((pro(x) . W^T + bias) [* scale on the first scale_cols columns]) (+ resid), M <= 8.  pro: 0 = x as given, 1 = LayerNorm
// (gamma, beta, eps) of x, 2 = merge of the flash-decoding partials `part` (heads x nsplit x (hd + 2) floats per row).
int launch_gemv_rows(int pro, const bf16 *x, int64_t ldx, const bf16 *gamma, const bf16 *beta, float eps, const float *part, int heads, int hd,
                     int nsplit, const bf16 *W, const bf16 *bias, const bf16 *resid, int64_t ldr, void *out, int64_t ldo, int out_f32, int M, int N,
                     int K, int epi, float scale, int scale_cols, hipStream_t s) {
    if (!gemv_rows_ok(M, N, K) || !W || !out) return EILEV_E_UNSUPPORTED;
    if ((pro != PRO_MERGE && (!x || (ldx & 7) || ((uintptr_t)x & 15))) || (pro == PRO_LN && (!gamma || !beta)) ||
        (pro == PRO_MERGE && (!part || heads * hd != K || (hd & 7))) || ((uintptr_t)W & 15))
        return EILEV_E_BADARG;
    GemvArgs a;
    a.x = x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.eps = eps; a.part = part; a.heads = heads; a.hd = hd; a.nsplit = nsplit;
    a.W = W; a.bias = bias; a.resid = resid; a.ldr = ldr; a.out = out; a.ldo = ldo; a.out_f32 = out_f32; a.M = M; a.N = N; a.K = K;
    a.epi = epi; a.scale = scale; a.scale_cols = scale_cols;
    if (pro == PRO_LN && K > 4096) return EILEV_E_UNSUPPORTED;  // the LayerNorm prologue holds a row in registers: 8 chunks of 512
    a.KB = K;
    // rows per wave: enough workgroups to cover the CUs a few times over, two rows in flight per wave pass
    const int waves_target = 256 * 4 * 2;
    int rpw = (N + waves_target - 1) / waves_target;
    rpw = rpw < 2 ? 2 : rpw;
    a.rows_per_wave = rpw;
    const int grid = (N + 4 * rpw - 1) / (4 * rpw);
    const size_t smem = (size_t)M * K * sizeof(bf16);
    if (pro == PRO_LN) return launch_rows_p<PRO_LN>(a, grid, smem, s);
    if (pro == PRO_MERGE) {  // (every workgroup repeats the merge: M <= 2 only)
        if (M == 1) return launch_rows_m<1, PRO_MERGE>(a, grid, smem, s);
        if (M == 2) return launch_rows_m<2, PRO_MERGE>(a, grid, smem, s);
        return EILEV_E_UNSUPPORTED;
    }
    return launch_rows_p<PRO_X>(a, grid, smem, s);
}


// M = 1 fast path (gemv1_kernel): x (K), K / 512 in {5, 8, 20}; pro 0 = x as given, 1 = LayerNorm(x).  n_cu workgroups of 5 waves.
bool gemv1_ok(int N, int K, int pro) {
    const int nch = K >> 9;
    if (K % 512 || N < 1) return false;
    if (pro == PRO_LN) return nch == 5;  // (K = 4096 with the LayerNorm prologue spills 60 VGPRs: the LDS kernel above)
    return pro == PRO_X && (nch == 5 || nch == 8 || nch == 20);  // (K = 16384: 128 registers of x alone — the LDS kernel above)
}
int launch_gemv1(int pro, const bf16 *x, const bf16 *gamma, const bf16 *beta, float eps, const bf16 *W, const bf16 *bias, const bf16 *resid, void *out,
                 int out_f32, int N, int K, int epi, float scale, int scale_cols, hipStream_t s, const float *part, int heads, int hd, int nsplit) {
    if (pro == PRO_MERGE) {
        if (nsplit < 1 || nsplit > 8) return EILEV_E_UNSUPPORTED;
        if (!part || heads * hd != K || (hd & 7) || (K >> 9) != 5 || K % 512 || !W || !out || ((uintptr_t)W & 15) || ((uintptr_t)part & 7)) return EILEV_E_UNSUPPORTED;
    } else if (!gemv1_ok(N, K, pro) || !x || !W || !out || ((uintptr_t)x & 15) || ((uintptr_t)W & 15)) return EILEV_E_UNSUPPORTED;
    if ((bias && ((uintptr_t)bias & 3)) || (resid && ((uintptr_t)resid & 3)) || (pro == PRO_LN && (!gamma || !beta))) return EILEV_E_BADARG;
    const int n_cu = eilev_num_cu();
    GemvArgs a = {};
    a.x = x; a.ldx = K; a.gamma = gamma; a.beta = beta; a.eps = eps; a.W = W; a.bias = bias; a.resid = resid; a.ldr = N; a.out = out; a.ldo = N;
    a.out_f32 = out_f32; a.M = 1; a.N = N; a.K = K; a.epi = epi; a.scale = scale; a.scale_cols = scale_cols;
    a.part = part; a.heads = heads; a.hd = hd; a.nsplit = nsplit;
    int grid = n_cu;
    if ((int64_t)grid * 5 > N) grid = (N + 4) / 5;
    a.rows_per_wave = N / (grid * 5);
    a.KB = N % (grid * 5);  // (field reused: the first KB waves own one row more)
    const int nch = K >> 9;
    if (pro == PRO_LN) return launch_gemv1_c<5, PRO_LN>(a, grid, s);
    if (pro == PRO_MERGE) return launch_gemv1_c<5, PRO_MERGE>(a, grid, s);
    switch (nch) {
        case 5: return launch_gemv1_c<5, PRO_X>(a, grid, s);
        case 8: return launch_gemv1_c<8, PRO_X>(a, grid, s);
        default: return launch_gemv1_c<20, PRO_X>(a, grid, s);
    }
}


// 2 <= M <= 8 rows, K = 2560 (plain or LayerNorm): gemvm_kernel
bool gemvm_ok(int M, int N, int K, int pro) {
    const int nch = K >> 9;
    if (M < 2 || M > 8 || K % 512 || N < 1 || (int64_t)M * K * 2 > 150 * 1024) return false;
    return nch == 5 && (pro == PRO_X || pro == PRO_LN);  // (K = 10240: every variant tried spills hundreds of VGPRs — fc2 keeps gemv_rows_kernel)
}
int launch_gemvm(int pro, const bf16 *x, int64_t ldx, const bf16 *gamma, const bf16 *beta, float eps, const bf16 *W, const bf16 *bias, const bf16 *resid,
                 int64_t ldr, void *out, int64_t ldo, int out_f32, int M, int N, int K, int epi, float scale, int scale_cols, hipStream_t s) {
    if (!gemvm_ok(M, N, K, pro) || !x || !W || !out || ((uintptr_t)x & 15) || (ldx & 7) || ((uintptr_t)W & 15)) return EILEV_E_UNSUPPORTED;
    if ((bias && ((uintptr_t)bias & 3)) || (resid && (((uintptr_t)resid & 3) || (ldr & 1))) || (pro == PRO_LN && (!gamma || !beta))) return EILEV_E_BADARG;
    const int n_cu = eilev_num_cu();
    GemvArgs a = {};
    a.x = x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.eps = eps; a.W = W; a.bias = bias; a.resid = resid; a.ldr = ldr; a.out = out; a.ldo = ldo;
    a.out_f32 = out_f32; a.M = M; a.N = N; a.K = K; a.epi = epi; a.scale = scale; a.scale_cols = scale_cols;
    int grid = n_cu;
    if ((int64_t)grid * 5 > N) grid = (N + 4) / 5;
    a.rows_per_wave = N / (grid * 5);
    a.KB = N % (grid * 5);
    if (pro == PRO_LN) return launch_gemvm_m<5, PRO_LN>(a, grid, s);
    return launch_gemvm_m<5, PRO_X>(a, grid, s);
}
