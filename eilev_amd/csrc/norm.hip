// norm.hip — LayerNorm over rows of bf16 (fp32 statistics, two-pass in registers).
// Replaces nn.LayerNorm at hf modeling_blip_2.py:390,397,522 (ViT), :619,:675,:913 (Q-Former) and
// hf modeling_opt.py:215,226,387 (OPT).  One 64-lane wave per row, 16-byte loads/stores, the row is
// held in registers between the mean and the variance pass (matches torch's biased variance).
// HBM-bound: reads and writes each element once.
#include "common.h"

namespace {

template <int MAXC>  // max 16-byte chunks per lane: cols <= MAXC * 512
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16 *__restrict__ x, int64_t ldx,
                                                        const bf16 *__restrict__ gamma,
                                                        const bf16 *__restrict__ beta, bf16 *__restrict__ y,
                                                        int64_t ldy, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = cols >> 3;
    const bf16 *xr = x + row * ldx;
    float v[MAXC][8];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            bf16x8 t = *reinterpret_cast<const bf16x8 *>(xr + c * 8);
            unpack8(t, v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[i][e];
        }
    }
    const float mean = wave_sum(sum) / (float)cols;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)cols + eps);
    bf16 *yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float gv[8], bv[8], o[8];
            unpack8(*reinterpret_cast<const bf16x8 *>(gamma + c * 8), gv);
            unpack8(*reinterpret_cast<const bf16x8 *>(beta + c * 8), bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * gv[e] + bv[e];
            *reinterpret_cast<bf16x8 *>(yr + c * 8) = pack8(o);
        }
    }
}

}  // namespace

int launch_layernorm(const bf16 *x, int64_t ldx, const bf16 *g, const bf16 *b, bf16 *y, int64_t ldy, int64_t rows,
                     int cols, float eps, hipStream_t s) {
    if (rows <= 0) return EILEV_OK;
    if (!x || !g || !b || !y) return EILEV_E_BADARG;
    if ((cols & 7) || (ldx & 7) || (ldy & 7) || cols > 8 * 512) return EILEV_E_UNSUPPORTED;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (cols <= 3 * 512) hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, x, ldx, g, b, y, ldy, rows, cols, eps);
    else if (cols <= 5 * 512) hipLaunchKernelGGL(layernorm_kernel<5>, grid, block, 0, s, x, ldx, g, b, y, ldy, rows, cols, eps);
    else hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, ldx, g, b, y, ldy, rows, cols, eps);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

// ---- T5LayerNorm (hf models/t5/modeling_t5.py:50-72): y = w * bf16(x * rsqrt(mean(x^2) + eps)), no mean subtraction, no bias.
// One wave per row, the row stays in registers (cols <= 8 * 512).
namespace {
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16 *__restrict__ x, int64_t ldx, const bf16 *__restrict__ g,
                                                      bf16 *__restrict__ y, int64_t ldy, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int chunks = cols >> 3;
    const bf16 *xr = x + row * ldx;
    float v[MAXC][8];
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 64 + lane;
        if (c < chunks) {
            const bf16x8 t = *reinterpret_cast<const bf16x8 *>(xr + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = (float)t[e];
                ss += v[i][e] * v[i][e];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rs = rsqrtf(ss / (float)cols + eps);
    bf16 *yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 64 + lane;
        if (c < chunks) {
            const bf16x8 gw = *reinterpret_cast<const bf16x8 *>(g + c * 8);
            bf16x8 out;
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (bf16)((float)gw[e] * (float)(bf16)(v[i][e] * rs));
            *reinterpret_cast<bf16x8 *>(yr + c * 8) = out;
        }
    }
}
}  // namespace

int launch_rmsnorm(const bf16 *x, int64_t ldx, const bf16 *g, bf16 *y, int64_t ldy, int64_t rows, int cols, float eps, hipStream_t s) {
    if (rows <= 0) return EILEV_OK;
    if (!x || !g || !y) return EILEV_E_BADARG;
    if ((cols & 7) || (ldx & 7) || (ldy & 7) || cols > 8 * 512) return EILEV_E_UNSUPPORTED;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (cols <= 3 * 512) hipLaunchKernelGGL(rmsnorm_kernel<3>, grid, block, 0, s, x, ldx, g, y, ldy, rows, cols, eps);
    else if (cols <= 5 * 512) hipLaunchKernelGGL(rmsnorm_kernel<5>, grid, block, 0, s, x, ldx, g, y, ldy, rows, cols, eps);
    else hipLaunchKernelGGL(rmsnorm_kernel<8>, grid, block, 0, s, x, ldx, g, y, ldy, rows, cols, eps);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

// ---- LayerNorm folded into the consuming GEMM (see GemmArgs::ln_rows in common.h) ------------------------------------------
// y = LN(x) . W^T + b  =  rstd * (x . (gamma (.) W)^T - mean * csum) + (b + W . beta),  csum[n] = sum_k gamma[k] W[n, k].
// eilev_fold_layernorm prepares the right-hand sides once per weight; ln_finalize_kernel turns the per-slot partial sums the producing
// GEMM wrote into (rstd, -mean) per row.  The reference computes the same product with the LayerNorm output rounded to
// bf16 in between (hf modeling_blip_2.py:383-402 under bf16 weights); here the bf16 rounding sits on gamma (.) W instead.
__global__ __launch_bounds__(256) void fold_ln_kernel(const bf16 *__restrict__ w, const bf16 *__restrict__ gamma, const bf16 *__restrict__ beta,
                                                      const bf16 *__restrict__ bias, int N, int K, bf16 *__restrict__ wf,
                                                      float *__restrict__ csum, bf16 *__restrict__ bf) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // one wave per output channel
    if (n >= N) return;
    const bf16 *wr = w + (int64_t)n * K;
    bf16 *wo = wf + (int64_t)n * K;
    float cs = 0.0f, bs = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const float wv = (float)wr[k];
        const bf16 f = (bf16)(wv * (float)gamma[k]);
        wo[k] = f;
        cs += (float)f;  // csum of what the MFMA will really multiply by
        bs = fmaf(wv, (float)beta[k], bs);
    }
    cs = wave_sum(cs);
    bs = wave_sum(bs);
    if (lane == 0) {
        csum[n] = cs;
        bf[n] = (bf16)(bs + (bias ? (float)bias[n] : 0.0f));
    }
}

__global__ __launch_bounds__(256) void ln_finalize_kernel(const float *__restrict__ part, int slots, int64_t rows, int cols, float eps,
                                                          float *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float s1 = 0.0f, s2 = 0.0f;
    for (int sl = 0; sl < slots; ++sl) {  // fixed order: the same bits on every run
        const float2 p = *reinterpret_cast<const float2 *>(part + ((int64_t)sl * rows + r) * 2);
        s1 += p.x;
        s2 += p.y;
    }
    const float mean = s1 / (float)cols;
    const float var = fmaxf(s2 / (float)cols - mean * mean, 0.0f);
    const float rstd = rsqrtf(var + eps);
    *reinterpret_cast<float2 *>(out + r * 2) = make_float2(rstd, -mean);
}

int launch_fold_layernorm(const bf16 *w, const bf16 *gamma, const bf16 *beta, const bf16 *bias, int N, int K, bf16 *wf, float *csum, bf16 *bf,
                          hipStream_t s) {
    if (!w || !gamma || !beta || !wf || !csum || !bf || N <= 0 || K <= 0) return EILEV_E_BADARG;
    hipLaunchKernelGGL(fold_ln_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w, gamma, beta, bias, N, K, wf, csum, bf);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

int launch_ln_finalize(const float *part, int slots, int64_t rows, int cols, float eps, float *out, hipStream_t s) {
    if (!part || !out || slots <= 0 || rows <= 0 || cols <= 0) return EILEV_E_BADARG;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, part, slots, rows, cols, eps, out);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
