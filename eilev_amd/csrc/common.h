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

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below
// the bf16 rounding of the stored activation), rearranged as max(x, 0) - |x| * t * P(t) * exp(-x^2 / 2)
// with t = 1 / (1 + p |x| / sqrt 2): 2 transcendentals + 10 VALU ops, no libm call (libm's erff in a
// 128-accumulator epilogue spills to scratch).
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float p = 0.5f * 1.061405429f;
    p = fmaf(p, t, -0.5f * 1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, -0.5f * 0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2 / 2)
    return fmaf(-(p * t), ax * e, fmaxf(x, 0.0f));
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
};
int launch_attention(const AttnArgs &a, hipStream_t s);

void prof_begin(int kind, double flops, hipStream_t s);
void prof_end(hipStream_t s);
