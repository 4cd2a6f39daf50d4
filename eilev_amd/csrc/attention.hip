// attention.hip — softmax(q k^T * scale + mask) v on MFMA, flash style (no S x S matrix in HBM).
//
// Replaces the attention of Blip2Attention (hf modeling_blip_2.py:339-350; S=257, d=88), of the Q-Former
// self/cross attention (:592-604; Sq=32, Skv=32 / T*257, d=64) and of OPTAttention prefill
// (hf modeling_opt.py:163-176; causal, left/right padding mask, d=80), plus the single-query decode step.
//
// Prefill kernel: workgroup = 4 waves = 64 query rows of one (batch, head); K/V tiles of 64 keys are
// staged in LDS (K row-major, V transposed).  Scores are computed TRANSPOSED, S^T = K Q^T, so that after
// the 16x16x32 MFMA each lane owns one query row (col = lane & 15) and 16 of the tile's 64 keys: the
// online-softmax statistics are per-lane scalars (+2 cross-lane-group shuffles) and the probabilities
// are already in the B-operand layout of the second MFMA, O^T = V^T P^T (MFMA k-slots are an arbitrary
// but consistent permutation of the keys) — no LDS round trip for P.  Head dims 64/80/88/128 are
// zero-padded to DP in {64, 96, 128} on load.  fp32 softmax, bf16 P, fp32 accumulate.
#include "common.h"

namespace {

template <int DP>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const AttnArgs a) {
    constexpr int KD = DP / 32;        // MFMA k-steps over the head dim
    constexpr int DT = DP / 16;        // 16-row tiles of O^T
    constexpr int CH = DP / 8;         // 16-byte chunks per K/V row
    constexpr int KSTR = DP * 2 + 16;  // LDS row stride of the K tile (bytes): +16 keeps b128 reads conflict-free
    constexpr int VSTR = 64 * 2 + 16;  // LDS row stride of the transposed V tile (64 keys per row)
    __shared__ __attribute__((aligned(16))) char ks_[64 * KSTR];
    __shared__ __attribute__((aligned(16))) char vt_[DP * VSTR];
    __shared__ int msk_[64];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qrow = blockIdx.x * 64 + wid * 16 + l15;  // this lane's query row
    const int off = a.skv - a.sq;                        // causal offset: key j visible iff j <= i + off
    const float sl2 = a.scale * 1.44269504088896340736f;

    const bf16 *qp = a.q + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs;
    const bf16 *kp = a.k + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs;
    const bf16 *vp = a.v + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs;

    bf16x8 qf[KD];
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) {
        const int d0 = kd * 32 + lg * 8;
        qf[kd] = (qrow < a.sq && d0 < a.hd) ? *reinterpret_cast<const bf16x8 *>(qp + (int64_t)qrow * a.ldq + d0) : zero8();
    }

    float m_run = -1e30f, l_run = 0.0f;
    f32x4 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kv_end = a.skv;
    if (a.causal) {
        const int last_q = min(a.sq - 1, (int)blockIdx.x * 64 + 63);
        kv_end = min(a.skv, last_q + off + 1);
    }

    for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
        __syncthreads();  // everyone is done reading the previous tile
        // ---- stage K (row-major, coalesced 16-byte loads)
        for (int id = tid; id < 64 * CH; id += 256) {
            const int key = id / CH, c = id - key * CH;
            const int gk = kv0 + key;
            bf16x8 val = (gk < a.skv && c * 8 < a.hd) ? *reinterpret_cast<const bf16x8 *>(kp + (int64_t)gk * a.ldk + c * 8) : zero8();
            *reinterpret_cast<bf16x8 *>(ks_ + key * KSTR + c * 16) = val;
        }
        // ---- stage V transposed: vt[d][key]
        for (int id = tid; id < 64 * CH; id += 256) {
            const int key = id & 63, c = id >> 6;
            const int gk = kv0 + key;
            bf16x8 val = (gk < a.skv && c * 8 < a.hd) ? *reinterpret_cast<const bf16x8 *>(vp + (int64_t)gk * a.ldv + c * 8) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16 *>(vt_ + (c * 8 + e) * VSTR + key * 2) = val[e];
        }
        if (tid < 64) {
            const int gk = kv0 + tid;
            int ok = gk < a.skv;
            if (ok && a.key_mask) ok = a.key_mask[(int64_t)b * a.mask_ld + gk] != 0;
            msk_[tid] = ok;
        }
        __syncthreads();

        // ---- S^T = K Q^T : st[ct][r] = S[q = l15][key = kv0 + ct*16 + lg*4 + r]
        f32x4 st[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            st[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(ks_ + (ct * 16 + l15) * KSTR + (kd * 4 + lg) * 16);
                st[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kd], st[ct], 0, 0, 0);
            }
        }
        // ---- mask, online softmax (per-lane row statistics)
        float mx = -1e30f;
        bool okv[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = ct * 16 + lg * 4 + r;
                const bool ok = msk_[kl] != 0 && (!a.causal || (kv0 + kl) <= qrow + off);
                okv[ct][r] = ok;
                const float s = st[ct][r] * sl2;
                st[ct][r] = s;
                if (ok) mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        float rs = 0.0f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = okv[ct][r] ? exp2f(st[ct][r] - m_new) : 0.0f;
                st[ct][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[i] *= alpha;

        // ---- O^T += V^T P^T.  k-slot (lg, j) of step ks  <->  key kv0 + 32 ks + (j < 4 ? lg*4 + j : 16 + lg*4 + j - 4)
        bf16x8 pb[2];
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pb[ks2][j] = (bf16)st[2 * ks2][j];
                pb[ks2][4 + j] = (bf16)st[2 * ks2 + 1][j];
            }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const char *vrow = vt_ + (dt * 16 + l15) * VSTR + lg * 8;
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(vrow + ks2 * 64);
                const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(vrow + ks2 * 64 + 32);
                bf16x8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vf[j] = lo[j];
                    vf[4 + j] = hi[j];
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[ks2], o[dt], 0, 0, 0);
            }
        }
    }

    // ---- finalize: o[dt][r] = O[q = l15][d = dt*16 + lg*4 + r]
    if (qrow < a.sq) {
        const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
        bf16 *op = a.o + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs + (int64_t)qrow * a.ldo;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + lg * 4;
            if (d0 < a.hd) {
                bf16x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (bf16)(o[dt][r] * inv);
                *reinterpret_cast<bf16x4 *>(op + d0) = w;
            }
        }
    }
}

}  // namespace

int launch_attention(const AttnArgs &a, hipStream_t s) {
    if (a.batch <= 0 || a.sq <= 0) return EILEV_OK;
    if (!a.q || !a.k || !a.v || !a.o || a.skv <= 0) return EILEV_E_BADARG;
    if ((a.hd & 7) || a.hd > 128 || (a.ldq & 7) || (a.ldk & 7) || (a.ldv & 7) || (a.ldo & 3) || (a.q_hs & 7) ||
        (a.k_hs & 7) || (a.v_hs & 7) || (a.o_hs & 3) || (a.q_bs & 7) || (a.k_bs & 7) || (a.v_bs & 7) || (a.o_bs & 3))
        return EILEV_E_UNSUPPORTED;
    const dim3 grid((a.sq + 63) / 64, a.heads, a.batch), block(256);
    if (a.hd <= 64) hipLaunchKernelGGL(attn_prefill_kernel<64>, grid, block, 0, s, a);
    else if (a.hd <= 96) hipLaunchKernelGGL(attn_prefill_kernel<96>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(attn_prefill_kernel<128>, grid, block, 0, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
