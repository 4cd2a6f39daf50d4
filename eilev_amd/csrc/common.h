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

// Exact-erf GELU (hf ACT2FN["gelu"], Blip2MLP) = max(x, 0) - r(|x|) with r(u) = u * Phi(-u), a smooth bump that is
// < 1.5e-6 beyond u = 5.  Phi(-u) on [0, 5] is a degree-12 polynomial in t = 0.4 u - 1 (weighted minimax fit, Horner in
// fp32: max abs error of the GELU 1.5e-6, i.e. 1.4e-3 relative to max(|y|, 1e-3) — below the bf16 rounding of the stored
// activation).  No transcendental and no division: 12 FMAs that pack two elements per v_pk_fma_f32; the previous
// rcp + exp2 form (A&S 7.1.26) cost 11 % of the fc1 GEMM in its epilogue.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define EILEV_GELU_COEFFS                                                                                                   \
    {6.210111547e-03f, -4.381819814e-02f, 1.368895024e-01f, -2.397004068e-01f, 2.326955497e-01f, -6.489974260e-02f,       \
     -1.311938316e-01f, 1.613862813e-01f, -2.371504903e-02f, -7.562928647e-02f, 3.900733590e-02f, 1.263864804e-02f,        \
     -9.870870970e-03f}
__device__ __forceinline__ float gelu_erf(float x) {
    constexpr float c[13] = EILEV_GELU_COEFFS;
    const float u = fminf(fabsf(x), 5.0f);
    const float t = fmaf(u, 0.4f, -1.0f);
    float p = c[12];
#pragma unroll
    for (int k = 11; k >= 0; --k) p = fmaf(p, t, c[k]);
    return fmaf(-u, p, fmaxf(x, 0.0f));
}
// 2 * NP elements at a time: NP independent Horner chains of v_pk_fma_f32, interleaved step by step (a dependent
// packed FMA needs a wait state; one chain alone runs at a fraction of the VALU rate)
template <int NP>
__device__ __forceinline__ void gelu_erf_pk(f32x2 (&x)[NP]) {
    constexpr float c[13] = EILEV_GELU_COEFFS;
    f32x2 u[NP], t[NP], p[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        u[n] = (f32x2){fminf(fabsf(x[n].x), 5.0f), fminf(fabsf(x[n].y), 5.0f)};
        t[n] = u[n] * 0.4f + (-1.0f);
        p[n] = (f32x2){c[12], c[12]};
    }
#pragma unroll
    for (int k = 11; k >= 0; --k)
#pragma unroll
        for (int n = 0; n < NP; ++n) p[n] = p[n] * t[n] + c[k];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const f32x2 relu = {fmaxf(x[n].x, 0.0f), fmaxf(x[n].y, 0.0f)};
        x[n] = relu - u[n] * p[n];
    }
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- host-side launchers shared between translation units -------------------------------------
struct GemmArgs {
    const bf16 *A;      // [M, K] row-major, leading dimension lda (elements, multiple of 8)
    int64_t lda;
    const bf16 *W;      // [N, K] row-major (checkpoint layout), leading dimension ldw
    int64_t ldw;
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
    int k_slice = 0;                // > 0: split-K launch (gridDim.y slices of k_slice K-steps, f32 output accumulated atomically)
    // probe-only (tools/gemm_trace.py): per-tile phase timestamps of the persistent ping-pong kernel, 8 u64 per (workgroup, wave
    // group, tile): s_memrealtime at loop top / K-loop start / K-loop end / epilogue start / epilogue end, s_memtime at top / end
    unsigned long long *trace = nullptr;
    int trace_tiles = 0;
};

int launch_gemm(const GemmArgs &g, int prof_kind, hipStream_t s);
int launch_layernorm(const bf16 *x, int64_t ldx, const bf16 *g, const bf16 *b, bf16 *y, int64_t ldy, int64_t rows,
                     int cols, float eps, hipStream_t s);

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
    // dropout on the attention probabilities (training graph): element (b, h, i, j) is kept iff eilev_hash32(drop_seed, its linear
    // index) >= drop_thr, kept probabilities are scaled by drop_scale = 1 / (1 - p); drop_thr = 0: none
    uint32_t drop_thr = 0, drop_seed = 0;
    float drop_scale = 1.0f;
};
int launch_attention(const AttnArgs &a, hipStream_t s);
int launch_rmsnorm(const bf16 *x, int64_t ldx, const bf16 *g, bf16 *y, int64_t ldy, int64_t rows, int cols, float eps, hipStream_t s);

void prof_begin(int kind, double flops, hipStream_t s);
void prof_end(hipStream_t s);
