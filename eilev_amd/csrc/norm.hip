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

// ---- decode: split-K partial sums -> output row -> its LayerNorm ------------------------------------------------------------------------
// Replaces skinny_reduce_kernel + layernorm_kernel for the two residual GEMVs of an OPT block (hf modeling_opt.py:226-253: out_proj /
// fc2, + residual, then the next LayerNorm): partials summed s = 0 .. ks - 1 from 0, x wscale, + bias, + residual, one bf16 rounding, then
// the statistics over the rounded row — bit-identical to the two launches.  (Round 2 ran one WAVE per row; the workgroup-per-row form below
// replaced it in round 4: profiles/r04_reduce_ln_ab.log.)

// Round 4: the same reduction with ONE 16-byte chunk per thread — a workgroup of N / 8 threads per row (N = 2560: 5 waves) instead of one
// wave walking 5 chunks: a fifth of the loads per lane, gamma / beta requested with the partials (not after the statistics), the row
// statistics through LDS.  The one-wave form took 9.2-9.7 us per launch at 32 rows, twice per block = 12 % of the batch-32 decode step
// (profiles/r04_decode_b32_kernel_stats.md).  Summation order of the statistics differs from layernorm_kernel's (fp32, over 2560 values).
template <int KS>
__global__ __launch_bounds__(1024) void reduce_ln_wg_kernel(const float *__restrict__ part, int ks, int mr, int N, const float *__restrict__ wscale,
                                                           const bf16 *__restrict__ bias, const bf16 *__restrict__ resid, int64_t ldr,
                                                           bf16 *__restrict__ C, int64_t ldc, const bf16 *__restrict__ gamma,
                                                           const bf16 *__restrict__ beta, bf16 *__restrict__ y, float eps, int y_frag) {
    __shared__ float red[2][16];
    const int c = threadIdx.x, row = blockIdx.x, lane = c & 63, wid = c >> 6, nw = (blockDim.x + 63) >> 6;
    bf16 *yp = y + (y_frag ? frag32_index(row, c * 8) : (int64_t)row * N + c * 8);  // (y_frag: the row-block layout the decode GEMVs read, common.h)
    const bool on = c < (N >> 3);
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gv[8], bt[8];
    if (on) {
        auto add = [&](int s) {
            const float4 *pp = reinterpret_cast<const float4 *>(part + ((int64_t)s * mr + row) * N + c * 8);
            const float4 a = pp[0], b = pp[1];
            t[0] += a.x; t[1] += a.y; t[2] += a.z; t[3] += a.w; t[4] += b.x; t[5] += b.y; t[6] += b.z; t[7] += b.w;
        };
        if constexpr (KS > 0) {
#pragma unroll
            for (int s = 0; s < KS; ++s) add(s);
        } else {
            for (int s = 0; s < ks; ++s) add(s);
        }
        float bv[8], rv[8];
        if (bias) unpack8(*reinterpret_cast<const bf16x8 *>(bias + c * 8), bv);
        if (resid) unpack8(*reinterpret_cast<const bf16x8 *>(resid + (int64_t)row * ldr + c * 8), rv);
        unpack8(*reinterpret_cast<const bf16x8 *>(gamma + c * 8), gv);
        if (beta) unpack8(*reinterpret_cast<const bf16x8 *>(beta + c * 8), bt);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = t[e];
            if (wscale) x *= wscale[c * 8 + e];
            if (bias) x += bv[e];
            if (resid) x += rv[e];
            t[e] = x;
        }
        const bf16x8 o = pack8(t);
        *reinterpret_cast<bf16x8 *>(C + (int64_t)row * ldc + c * 8) = o;
        unpack8(o, t);  // statistics over the ROUNDED row, like the LayerNorm kernel that read it back
    }
    if (beta == nullptr) {  // RMS form (T5LayerNorm, hf modeling_t5.py:50-72): y = gamma * bf16(x * rsqrt(mean(x^2) + eps)), no mean, no beta —
        float ss = 0.0f;    // round 5: the three RMSNorms of a flan-t5 decoder block ride on the split-K reduces in front of them
        if (on) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf(t[e], t[e], ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) red[0][wid] = ss;
        __syncthreads();
        float tot2 = 0.0f;
        for (int w = 0; w < nw; ++w) tot2 += red[0][w];
        const float rs = rsqrtf(tot2 / (float)N + eps);
        if (on) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gv[e] * (float)(bf16)(t[e] * rs);
            *reinterpret_cast<bf16x8 *>(yp) = pack8(o);
        }
        return;
    }
    float s1 = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s1 += t[e];
    s1 = wave_sum(s1);
    if (lane == 0) red[0][wid] = s1;
    __syncthreads();
    float tot = 0.0f;
    for (int w = 0; w < nw; ++w) tot += red[0][w];
    const float mean = tot / (float)N;
    float s2 = 0.0f;
    if (on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s2 = fmaf(t[e] - mean, t[e] - mean, s2);
    }
    s2 = wave_sum(s2);
    if (lane == 0) red[1][wid] = s2;
    __syncthreads();
    tot = 0.0f;
    for (int w = 0; w < nw; ++w) tot += red[1][w];
    const float rstd = rsqrtf(tot / (float)N + eps);
    if (on) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (t[e] - mean) * rstd * gv[e] + bt[e];
        *reinterpret_cast<bf16x8 *>(yp) = pack8(o);
    }
}

int launch_reduce_ln(const float *part, int ks, int mr, int M, int N, const float *wscale, const bf16 *bias, const bf16 *resid, int64_t ldr, bf16 *C,
                     int64_t ldc, const bf16 *gamma, const bf16 *beta, bf16 *ln_out, float eps, hipStream_t s, int ln_frag) {
    if (!part || !C || !gamma || !ln_out || M <= 0 || (N & 7) || N > 8 * 512) return EILEV_E_UNSUPPORTED;  // beta == nullptr: the RMS form
    if (ln_frag && (M > 32 || (N & 31))) return EILEV_E_UNSUPPORTED;
    {
        const int threads = (((N >> 3) + 63) / 64) * 64;
#define EILEV_RLW(KS_) hipLaunchKernelGGL((reduce_ln_wg_kernel<KS_>), dim3(M), dim3(threads), 0, s, part, ks, mr, N, wscale, bias, resid, ldr, C, ldc, gamma, beta, ln_out, eps, ln_frag)
        if (ks == 2) EILEV_RLW(2); else if (ks == 4) EILEV_RLW(4); else EILEV_RLW(0);
#undef EILEV_RLW
        EILEV_LAUNCH_CHECK();
        return EILEV_OK;
    }
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
