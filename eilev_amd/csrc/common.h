// common.h — shared device helpers for the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/eilev.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define EILEV_HIP_CHECK(expr)                          \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

// CUs of the current device (cached; 256 when the query fails).  hipDeviceGetAttribute, not hipGetDeviceProperties: the property STRUCT
// differs between the HIP runtime this library was built against and the one the process mapped first (PyTorch ships its own copy), and
// a failed query must not leave its error behind for the next launch check's hipGetLastError().
static inline int eilev_num_cu() {
    static int n = 0;
    if (!n) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) n = cu;
        else {
            (void)hipGetLastError();
            n = 256;
        }
    }
    return n;
}

// CUs the persistent kernels (one workgroup per CU: GEMM, frame attention) size their grids for: all of them, or fewer when the caller runs
// them on a CU-masked stream next to another stream (probe switch eilev_debug_grid_cus; tools/overlap_probe.py).
extern int g_eilev_grid_cus;
static inline int eilev_grid_cus() { return g_eilev_grid_cus > 0 ? g_eilev_grid_cus : eilev_num_cu(); }

#define EILEV_LAUNCH_CHECK()                           \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

// Counter-based dropout mask (splitmix64 finaliser): the same function in the forward and the backward kernels and in the oracle,
// so a mask is never stored.  keep iff hash >= p * 2^32.
__host__ __device__ __forceinline__ uint32_t eilev_hash32(uint32_t seed, uint64_t idx) {
    uint64_t z = idx + 0x9E3779B97F4A7C15ull * ((uint64_t)seed + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
__host__ __forceinline__ uint32_t eilev_drop_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
}

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

// 16-byte vector of 8 bf16 <-> 8 floats
__device__ __forceinline__ void unpack8(const bf16x8 &v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    return v;
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)0.0f;
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Exact-erf GELU (hf ACT2FN["gelu"], Blip2MLP) = max(x, 0) - r(|x|) with r(u) = u * Phi(-u), a smooth bump (max 0.17 at u = 0.75).
// Two polynomial forms, no transcendental, no division, Horner in fp32:
//  * gelu_erf / gelu_erf_pk (everything but the two persistent kernels: Q-Former, training forward, small-tile launches, decode): Phi(-u) on
//    [0, 5] as a degree-12 polynomial in t = 0.4 u - 1 (weighted minimax fit): max abs error of the GELU 1.5e-6, i.e. below the bf16
//    rounding of the stored activation everywhere; r < 1.5e-6 beyond u = 5 (x < -5 returns -0.0-ish, approaching 0 like the exact form).
//  * gelu_erf_n<NP> = the FAST form of the persistent kernels' epilogues (gemm_pp4_kernel, gemm_w6_kernel: with a GELU only
//    the ViT fc1 reaches them — >= 192 tiles of 256 x 128): r itself on [0, 4] as a degree-8
//    polynomial in t = u / 2 - 1: max abs error 1.1e-4 (relative 2.2e-3 where |y| > 0.05; a constant -1.3e-4 for x < -4).  The stored
//    activation is bf16 (2^-9 relative) and the reference's own bf16 run evaluates GELU on a pre-activation that was itself rounded
//    to bf16 (an input error of 4e-3 |x|), so this is an order of magnitude inside the reference's own noise; 12 VALU operations per
//    element instead of 16: the epilogue runs with the MFMA pipe idle, the difference was worth +2.2 % on the fc1 GEMM (r2 A/B).
//    End-to-end effect on the logits: profiles/parity_r03.json `gelu_deg8_vs_deg12`.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define EILEV_GELU12_COEFFS                                                                                                 \
    {6.210111547e-03f, -4.381819814e-02f, 1.368895024e-01f, -2.397004068e-01f, 2.326955497e-01f, -6.489974260e-02f,       \
     -1.311938316e-01f, 1.613862813e-01f, -2.371504903e-02f, -7.562928647e-02f, 3.900733590e-02f, 1.263864804e-02f,        \
     -9.870870970e-03f}
__device__ __forceinline__ float gelu_erf(float x) {
    constexpr float c[13] = EILEV_GELU12_COEFFS;
    const float u = fminf(fabsf(x), 5.0f);
    const float t = fmaf(u, 0.4f, -1.0f);
    float p = c[12];
#pragma unroll
    for (int k = 11; k >= 0; --k) p = fmaf(p, t, c[k]);
    return fmaf(-u, p, fmaxf(x, 0.0f));
}
// 2 * NP elements at a time: independent Horner chains interleaved step by step (a dependent FMA waits ~4 cycles for its predecessor)
__device__ __forceinline__ float gelu_erf_n1(float x);  // (the fast form on one element: below)
template <int NP>
__device__ __forceinline__ void gelu_erf_pk(f32x2 (&x)[NP]) {
    constexpr float c[13] = EILEV_GELU12_COEFFS;
    float v[2 * NP], u[2 * NP], t[2 * NP], p[2 * NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) { v[2 * n] = x[n].x; v[2 * n + 1] = x[n].y; }
#pragma unroll
    for (int n = 0; n < 2 * NP; ++n) {
        u[n] = fminf(fabsf(v[n]), 5.0f);
        t[n] = fmaf(u[n], 0.4f, -1.0f);
        p[n] = c[12];
    }
#pragma unroll
    for (int k = 11; k >= 0; --k)
#pragma unroll
        for (int n = 0; n < 2 * NP; ++n) p[n] = fmaf(p[n], t[n], c[k]);
#pragma unroll
    for (int n = 0; n < NP; ++n) x[n] = (f32x2){fmaf(-u[2 * n], p[2 * n], fmaxf(v[2 * n], 0.0f)), fmaf(-u[2 * n + 1], p[2 * n + 1], fmaxf(v[2 * n + 1], 0.0f))};
}
#define EILEV_GELU_DEG 8
#define EILEV_GELU_UMAX 4.0f
#define EILEV_GELU_COEFFS                                                                                                   \
    {4.543383146e-02f, -1.712931058e-01f, 2.188764488e-01f, 1.158597774e-02f, -3.094834958e-01f, 2.450403219e-01f,         \
     3.056864685e-02f, -8.538186866e-02f, 1.466789586e-02f}
// Round 6 (EILEV_GELU_PHI, the default): the same GELU as x * Phi(x) with Phi(x) = 1 / 2 + xc q(xc^2), xc = x clamped to +-3.95, q of degree 6
// (minimax with Phi(3.95) = 1 pinned, so that x > 3.95 returns x and x < -3.95 returns 0): max abs error 1.67e-4 (relative 2.6e-3 where
// |y| > 0.05) against 1.1e-4 / 2.2e-3 of the form above — both an order of magnitude inside the bf16 rounding of the pre-activation the
// reference evaluates GELU on.  What it buys is VALU issue slots, the one thing the epilogue costs (tools/probes/valu_shadow.hip: a SIMD of
// gfx950 does not run VALU and MFMA instructions concurrently, §5 of DESIGN.md): one unpacked operation (v_med3) and nine that hipcc packs
// into v_pk_mul / v_pk_fma_f32, against three unpacked (|x| min, max, the bias of the clamp) and nine packed.
#ifndef EILEV_GELU_PHI
#define EILEV_GELU_PHI 1
#endif
#define EILEV_GELU_PHI_C 3.95f
#define EILEV_GELU_PHI_COEFFS                                                                                               \
    {3.979867044e-01f, -6.472275670e-02f, 8.842919395e-03f, -8.290393928e-04f, 4.955192888e-05f, -1.681151575e-06f, 2.443575809e-08f}
// NP independent elements at a time: NP Horner chains interleaved step by step.
template <int NP>
__device__ __forceinline__ void gelu_erf_n(float (&x)[NP]) {
#if EILEV_GELU_PHI
    constexpr float q[7] = EILEV_GELU_PHI_COEFFS;
    if constexpr (NP % 2 == 0) {  // explicit pairs: v_pk_mul_f32 / v_pk_fma_f32 (left to itself hipcc emits one v_fmaak_f32 per element and step)
        f32x2 xv[NP / 2], xc[NP / 2], s[NP / 2], p[NP / 2];
#pragma unroll
        for (int n = 0; n < NP / 2; ++n) {
            xv[n] = (f32x2){x[2 * n], x[2 * n + 1]};
            xc[n] = (f32x2){__builtin_amdgcn_fmed3f(x[2 * n], -EILEV_GELU_PHI_C, EILEV_GELU_PHI_C), __builtin_amdgcn_fmed3f(x[2 * n + 1], -EILEV_GELU_PHI_C, EILEV_GELU_PHI_C)};
            s[n] = xc[n] * xc[n];
            p[n] = (f32x2){q[6], q[6]};
        }
#pragma unroll
        for (int k = 5; k >= 0; --k)
#pragma unroll
            for (int n = 0; n < NP / 2; ++n) p[n] = __builtin_elementwise_fma(p[n], s[n], (f32x2){q[k], q[k]});
#pragma unroll
        for (int n = 0; n < NP / 2; ++n) {
            const f32x2 y = xv[n] * __builtin_elementwise_fma(xc[n], p[n], (f32x2){0.5f, 0.5f});
            x[2 * n] = y.x;
            x[2 * n + 1] = y.y;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            const float xc = __builtin_amdgcn_fmed3f(x[n], -EILEV_GELU_PHI_C, EILEV_GELU_PHI_C), s = xc * xc;
            float p = q[6];
#pragma unroll
            for (int k = 5; k >= 0; --k) p = fmaf(p, s, q[k]);
            x[n] = x[n] * fmaf(xc, p, 0.5f);
        }
    }
#else
    constexpr float c[EILEV_GELU_DEG + 1] = EILEV_GELU_COEFFS;
    float t[NP], p[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        t[n] = fmaf(fminf(fabsf(x[n]), EILEV_GELU_UMAX), 2.0f / EILEV_GELU_UMAX, -1.0f);
        p[n] = c[EILEV_GELU_DEG];
    }
#pragma unroll
    for (int k = EILEV_GELU_DEG - 1; k >= 0; --k)
#pragma unroll
        for (int n = 0; n < NP; ++n) p[n] = fmaf(p[n], t[n], c[k]);
#pragma unroll
    for (int n = 0; n < NP; ++n) x[n] = fmaxf(x[n], 0.0f) - p[n];
#endif
}

__device__ __forceinline__ float gelu_erf_n1(float x) {
    float v[1] = {x};
    gelu_erf_n<1>(v);
    return v[0];
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- host-side launchers shared between translation units -------------------------------------
struct GemmArgs {
    const bf16 *A;      // [M, K] row-major, leading dimension lda (elements, multiple of 8)
    int64_t lda;
    const bf16 *W;      // [N, K] row-major (checkpoint layout), leading dimension ldw
    int64_t ldw;
    const bf16 *Wp = nullptr;  // optional: the same matrix in the stream layout of gemm_rows32_kernel (gemm_skinny.h; eilev_stream_layout_pack)
    // Row-block layout of the <= 32 activation rows of a decode step (round 5; frag32_index below): only gemm_rows32_kernel reads (a_frag) or
    // writes (c_frag: plain bf16 epilogue; ln_frag: the LayerNorm rows that ride on a split-K reduce) it — any other kernel family refuses
    int a_frag = 0, c_frag = 0, ln_frag = 0;
    // Scattered output of the ViT q|k|v projection (round 5): C is not [M][N].  With the weight rows reordered so that, per third of N, the output
    // columns are [head h dims 0..63] for all hm_heads heads and then [head h dims 64..hm_hd-1] for all heads (engine: ensure_vit_fold), the
    // epilogue gives every (frame, q | k | v, head) one block of hm_tok * hm_hd elements: [token][64] followed by [token][hm_hd - 64].  The frame
    // attention stages a head's 45-KB image from two contiguous runs (row-major rows: 257 segments of 176 bytes, 8448 bytes apart, each
    // straddling 2-3 cache lines that the neighbouring heads fetch again), and a wave's store of 8 token rows x 64 columns of a [token][64]
    // block is one contiguous kilobyte (a plain head-major [token][88] layout made every store ~12 partial lines: profiles/r05_head_major_*).
    // Only the folded-LayerNorm 16 x 16 instance of the persistent kernel writes it (hm_takes below); everything else refuses.
    int hm_tok = 0, hm_heads = 0, hm_hd = 0;
    const bf16 *bias;   // [N] or null
    const bf16 *resid;  // [M, N] (ldr) or null; in patch mode: position table [1+group, N]
    int64_t ldr;
    void *C;            // bf16 or f32 [M(+CLS rows), N] with leading dimension ldc
    int64_t ldc;
    int M, N, K;
    int epi;            // 0 none, 1 GELU(erf), 2 ReLU
    int out_f32;
    float scale;        // columns [0, scale_cols) are multiplied by scale after the bias
    int scale_cols;
    int patch_group;    // > 0: patch-embedding row remap (see gemm.hip)
    int dbg;            // probe-only
    float *scratch;     // optional fp32 scratch for the skinny kernel's split-K partials
    size_t scratch_bytes;
    // fp8 (OCP e4m3) weights with one fp32 scale per output channel: C = (A . Wq^T) * wscale + bias (eilev_linear_w8)
    const float *wscale = nullptr;  // [N] or null
    const uint8_t *W8 = nullptr;    // [N, K] e4m3 bytes (ldw = K) when the weights are still quantised (skinny kernel)
    bf16 *w8_scratch = nullptr;     // [N, K] bf16: where a large-M call expands W8 to (exact: every e4m3 value is a bf16 value)
    // fp8 (e4m3) ACTIVATIONS with one fp32 scale per row (eilev_linear_a8w8): together with W8 the product runs on the fp8 MFMA
    // (v_mfma_f32_32x32x64_f8f6f4, twice the bf16 rate): C = (Aq . Wq^T) * ascale[m] * wscale[n] + bias.  lda / ldw count BYTES = k.
    const uint8_t *A8 = nullptr;    // [M, K] e4m3 bytes
    const float *ascale = nullptr;  // [M]
    // LayerNorm folded into the GEMM that consumes it (eilev_linear_lnfold; ViT ln1 -> qkv, ln2 -> fc1): A holds the RAW residual stream,
    // W = gamma (.) W_orig, bias = b + W_orig . beta, and C = rstd[m] * (A . W^T - mean[m] * csum[n]) + bias[n]: the accumulators start
    // from -mean[m] * csum[n], the epilogue multiplies by rstd[m].  ln_rows = (rstd, -mean) per row, csum[n] = sum_k W[n, k].
    const float *ln_rows = nullptr;  // [M][2] or null
    const float *ln_csum = nullptr;  // [N]
    // ... and the GEMM that PRODUCES the residual stream (proj / fc2 with the residual epilogue) emits, per 64-column slot and row, the
    // partial (sum, sum of squares) of what it writes: stat_out[(slot * stat_ld + row) * 2 + {0, 1}], slot = column / 64 (fixed order
    // of summation downstream: deterministic).  stat_ld = rows of the whole matrix (row-chunked launches shift stat_out, not stat_ld).
    float *stat_out = nullptr;
    int64_t stat_ld = 0;
    // LayerNorm of the finished output rows, written next to them: ln_out[m] = LN(C[m]) * ln_gamma + ln_beta over the bf16 rows of C.
    // Decode (M <= 32, split-K): done by the kernel that sums the partials (one launch instead of reduce + LayerNorm, bit-identical);
    // every other shape: a LayerNorm launch after the GEMM.
    const bf16 *ln_gamma = nullptr, *ln_beta = nullptr;
    bf16 *ln_out = nullptr;  // [M, N], leading dimension N
    float ln_eps = 0.0f;
    int k_slice = 0;                // > 0: split-K launch (gridDim.y slices of k_slice K-steps, f32 output accumulated atomically)
    // probe-only (tools/gemm_trace.py): per-tile phase timestamps of the persistent ping-pong kernel, 8 u64 per (workgroup, wave
    // group, tile): s_memrealtime at loop top / K-loop start / K-loop end / epilogue start / epilogue end, s_memtime at top / end
    unsigned long long *trace = nullptr;
    int trace_tiles = 0;
};

// Row-block ("fragment order") layout of up to 32 activation rows x K: element (row, col) at (col / 32) * 1024 + row * 32 + col % 32 — the 16
// rows x 32 columns an MFMA operand load of the decode GEMVs reads are ONE contiguous kilobyte (row-major gives it 16 segments of 64 bytes
// a row apart: 4.04 -> 3.85 ms / token at batch 32).  A buffer in this layout holds 32 rows whatever M is.  8-element chunks stay contiguous.
__host__ __device__ static inline int64_t frag32_index(int row, int col) { return (int64_t)(col >> 5) * 1024 + row * 32 + (col & 31); }

int launch_gemm(const GemmArgs &g, int prof_kind, hipStream_t s);
int launch_pp4_ext(const GemmArgs &g, int grid, hipStream_t s);  // gemm_pp4_ext.hip
// every tile of a persistent-kernel launch can take the lean epilogue (what the 16 x 16 MFMA instances need: gemm_pp4.h M16)
// Probe build: the per-tile stamps of tools/gemm_trace.py / gemm_timeline.py run on the product's 16 x 16 instances.  (In rounds 2-5 a trace
// buffer — like any probe flag — made a launch fall back to the 32 x 32 instances, whose half tiles cost 0.85-1.0 of a whole tile instead of
// 0.72-0.77: a traced or flagged arm of an A/B measured another kernel than the product's.  Probe flags still do: an A/B of two flag values
// compares like with like, an A/B against flags = 0 does not.)
#ifdef EILEV_PROBES
#define EILEV_TRACE_M16 true
#else
#define EILEV_TRACE_M16 false
#endif
static inline bool pp4_all_lean(const GemmArgs &g) {
    // (column scaling — the q part of a fused q|k|v projection — only on plain bias-only launches, in whole 16-column blocks)
    const bool scale_ok = g.scale_cols == 0 || (g.scale_cols % 16 == 0 && g.epi == 0 && !g.resid && !g.ln_rows && !g.stat_out);
    return g.N % 128 == 0 && !g.out_f32 && g.patch_group == 0 && scale_ok && !g.wscale && !g.ascale && !g.dbg && (!g.trace || EILEV_TRACE_M16);
}
// the launch can write the head-major q|k|v layout (GemmArgs::hm_tok)
static inline bool hm_takes(const GemmArgs &g) {
    return g.hm_tok > 0 && g.hm_heads > 0 && g.hm_hd > 64 && g.hm_hd % 8 == 0 && g.N == 3 * g.hm_heads * g.hm_hd && g.ln_rows && g.epi == 0 && !g.resid &&
           !g.stat_out && !g.scale_cols && pp4_all_lean(g) && g.N % 64 == 0 && g.M % g.hm_tok == 0 && (int64_t)g.M * g.N * 2 < 0xfffffff0ll;
}
// gemv.hip: nn.Linear on M <= 8 rows as row dot products with the LayerNorm / flash-decoding merge in its prologue
bool gemv_rows_ok(int M, int N, int K);
int launch_gemv_rows(int pro, const bf16 *x, int64_t ldx, const bf16 *gamma, const bf16 *beta, float eps, const float *part, int heads, int hd,
                     int nsplit, const bf16 *W, const bf16 *bias, const bf16 *resid, int64_t ldr, void *out, int64_t ldo, int out_f32, int M, int N,
                     int K, int epi, float scale, int scale_cols, hipStream_t s);
// gemv.hip, M = 1: activations in registers, a 4-deep ring of weight sub-blocks, one workgroup per CU (round 4)
bool gemv1_ok(int N, int K, int pro);
int launch_gemv1(int pro, const bf16 *x, const bf16 *gamma, const bf16 *beta, float eps, const bf16 *W, const bf16 *bias, const bf16 *resid, void *out,
                 int out_f32, int N, int K, int epi, float scale, int scale_cols, hipStream_t s, const float *part = nullptr, int heads = 0, int hd = 0,
                 int nsplit = 0);  // pro = 2: x = the merge of `part` (flash-decoding partials of one row: heads x nsplit x (hd + 2) floats)
bool gemvm_ok(int M, int N, int K, int pro);  // 2 <= M <= 8 rows with the same geometry, rows staged in LDS (round 4)
int launch_gemvm(int pro, const bf16 *x, int64_t ldx, const bf16 *gamma, const bf16 *beta, float eps, const bf16 *W, const bf16 *bias, const bf16 *resid,
                 int64_t ldr, void *out, int64_t ldo, int out_f32, int M, int N, int K, int epi, float scale, int scale_cols, hipStream_t s);
// misc.hip: single-query attention at small batch, one workgroup per (row, head), merged output (round 4)
bool attn_decode1_ok(int batch, int cap, int hd);
int launch_beam_advance(const float *row_lp, const int32_t *row_tok, int batch, int beams, int keep, int max_new, const int32_t *state,
                        const int64_t *eos_ids, int n_eos, const float *len_pow, int recip, int early, int64_t *run_seq, float *run_score,
                        int64_t *fin_seq, float *fin_score, int64_t *fin_len, uint8_t *finished, uint8_t *can_improve, int64_t *tokens, int32_t *anc,
                        int gen_cap, int64_t *scratch, hipStream_t s);
int launch_topk_logprob(const float *logits, const float *row_score, int rows, int vocab, int keep, float *out_val, int32_t *out_idx, hipStream_t s);
int launch_attn_decode1(const bf16 *qkv, bf16 *kc, bf16 *vc, bf16 *out, const int32_t *attn_mask, const int32_t *state, int batch, int seq_len,
                        int cap, int heads, int hd, hipStream_t s);
int attn_decode_part_splits(int cap);  // misc.hip: the one-pass loading scheme over 128-key ranges, partials for gemv1_kernel's merge prologue (round 4)
int launch_attn_decode_part(const bf16 *qkv, bf16 *kc, bf16 *vc, float *part, size_t part_bytes, const int32_t *attn_mask, const int32_t *state, int batch,
                            int seq_len, int cap, int heads, int hd, hipStream_t s);
int launch_layernorm(const bf16 *x, int64_t ldx, const bf16 *g, const bf16 *b, bf16 *y, int64_t ldy, int64_t rows,
                     int cols, float eps, hipStream_t s);
int launch_fold_layernorm(const bf16 *w, const bf16 *gamma, const bf16 *beta, const bf16 *bias, int N, int K, bf16 *wf, float *csum, bf16 *bf,
                          hipStream_t s);
int launch_reduce_ln(const float *part, int ks, int mr, int M, int N, const float *wscale, const bf16 *bias, const bf16 *resid, int64_t ldr, bf16 *C,
                     int64_t ldc, const bf16 *gamma, const bf16 *beta, bf16 *ln_out, float eps, hipStream_t s, int ln_frag = 0);
int launch_ln_finalize(const float *part, int slots, int64_t rows, int cols, float eps, float *out, hipStream_t s);

struct AttnArgs {
    const bf16 *q, *k, *v;
    bf16 *o;
    int64_t q_bs, k_bs, v_bs, o_bs;  // batch strides (elements)
    int64_t q_hs, k_hs, v_hs, o_hs;  // head strides (elements)
    int64_t ldq, ldk, ldv, ldo;      // row strides (elements)
    int batch, heads, sq, skv, hd;
    float scale;
    int causal;
    const int32_t *key_mask;         // (batch, mask_ld) or null
    int64_t mask_ld;
    int dbg;                         // probe-only
    // T5 relative position bias (additive, before the softmax): bias(h, i, j) = rel_tab[h * rel_hs + (j - i - (skv - sq)) +
    // rel_off], a per-head table over the relative distance (built by launch_t5_rel_table); null = none
    const float *rel_tab = nullptr;
    int64_t rel_hs = 0;
    int rel_off = 0, rel_n = 0;
    // ViT frame attention on the scattered q|k|v of GemmArgs::hm_tok (round 5): q / k / v point at the first (frame, head) block of their
    // plane, *_bs = frame stride, *_hs = block stride (S * hd); a block is [S][64] followed by [S][hd - 64] elements.  ld* are ignored.
    int hm = 0;
    // dropout on the attention probabilities (training graph): element (b, h, i, j) is kept iff eilev_hash32(drop_seed, its linear
    // index) >= drop_thr, kept probabilities are scaled by drop_scale = 1 / (1 - p); drop_thr = 0: none
    uint32_t drop_thr = 0, drop_seed = 0;
    float drop_scale = 1.0f;
};
int launch_attention(const AttnArgs &a, hipStream_t s);
int launch_rmsnorm(const bf16 *x, int64_t ldx, const bf16 *g, bf16 *y, int64_t ldy, int64_t rows, int cols, float eps, hipStream_t s);

void prof_begin(int kind, double flops, hipStream_t s);
void prof_end(hipStream_t s);
