// backward.hip — gradient kernels of the train_v2 path (ref:scripts/general/train_v2.py:124-130,207-217 —
// `loss = model(**batch).loss; accelerator.backward(loss)` with the ViT and the language model frozen).
//
// What autograd computes for the modules on the path, as gfx950 kernels:
//   * softmax attention backward (hf modeling_opt.py OPTAttention, modeling_blip_2.py Blip2QFormerMultiHeadAttention):
//     flash-style — P is recomputed from q.k and the row log-sum-exp, never stored.  Two kernels, no atomics:
//     `attn_bwd_dq_kernel` owns 64 query rows (pass 1: lse and delta = sum(o * d_o); pass 2: dQ), `attn_bwd_dkv_kernel`
//     owns 64 keys (dK, dV).  Every product is a 32x32x16 bf16 MFMA; the owned tile lives in registers as fragments, the
//     streamed tiles are staged row-major once and read either along their rows or through the transposing LDS read.
//   * LayerNorm backward (dx; dgamma / dbeta by a column-reduction kernel), erf-GELU / ReLU backward, bias gradients
//     (column sums), and the token cross-entropy with its logit gradient (hf loss_utils.ForCausalLMLoss).
// Linear layers need no new kernel: dX = dY . W and dW = dY^T . X are eilev_linear calls on transposed operands
// (eilev_amd/autograd.py).
#include "common.h"

namespace {

struct AttnBwdArgs {
    const bf16 *q, *k, *v, *o, *d_o;
    bf16 *dq, *dk, *dv;
    float *lse, *delta;  // (batch, heads, sq) each
    int batch, heads, sq, skv, hd;
    int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    float scale;
    int causal;
    const int32_t *key_mask;  // (batch, skv) or null
    // T5 relative position bias (frozen, additive before the softmax): rel_tab[h * rel_hs + clamp((key - query - (skv - sq)) + rel_off)]
    const float *rel_tab = nullptr;
    int64_t rel_hs = 0;
    int rel_off = 0, rel_n = 0;
    // dropout on the probabilities (same mask function as the forward kernel): o = (P * M / (1 - p)) V
    uint32_t drop_thr = 0, drop_seed = 0;
    float drop_scale = 1.0f;
};
// M(b, h, q, key) / (1 - p): 0 for a dropped probability
__device__ __forceinline__ float drop_factor(const AttnBwdArgs &a, int b, int h, int q, int key) {
    if (!a.drop_thr) return 1.0f;
    const uint64_t idx = (((uint64_t)b * a.heads + h) * a.sq + q) * (uint64_t)a.skv + key;
    return eilev_hash32(a.drop_seed, idx) >= a.drop_thr ? a.drop_scale : 0.0f;
}
__device__ __forceinline__ float rel_bias(const AttnBwdArgs &a, int h, int key, int qpos) {
    int ri = key - qpos + a.rel_off;
    ri = ri < 0 ? 0 : (ri >= a.rel_n ? a.rel_n - 1 : ri);
    return a.rel_tab[(int64_t)h * a.rel_hs + ri];
}

constexpr int LDT = 72;  // row stride (elements) of the 64-wide transposed / score tiles

// rows [row0, row0 + 64) x [0, hd) of a strided bf16 matrix into LDS: row-major [64][DP + 8] (zero beyond hd / nrows) and,
// when tr != null, transposed [DP][LDT] (kept for probes; the kernels use the transposing read instead)
template <int DP>
__device__ __forceinline__ void load_tile(const bf16 *src, int64_t ld, int row0, int nrows, int hd, bf16 *rm, bf16 *tr, int tid) {
    constexpr int LDR = DP + 8, CH = DP / 8;
    for (int i = tid; i < 64 * CH; i += 256) {
        const int r = i / CH, c = i - r * CH;
        bf16x8 v = zero8();
        const int gr = row0 + r;
        if (gr < nrows && c * 8 < hd) v = *reinterpret_cast<const bf16x8 *>(src + (int64_t)gr * ld + c * 8);
        *reinterpret_cast<bf16x8 *>(rm + r * LDR + c * 8) = v;
        if (tr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) tr[(c * 8 + e) * LDT + r] = v[e];
        }
    }
}

// the same tile in two steps, so that the global loads of tile t+1 are in flight while tile t is consumed: each thread
// holds DP/32 16-byte chunks
template <int DP>
struct TileRegs {
    bf16x8 v[DP / 32];
};
template <int DP>
__device__ __forceinline__ void fetch_tile(TileRegs<DP> &t, const bf16 *src, int64_t ld, int row0, int nrows, int hd, int tid) {
    constexpr int CH = DP / 8;
#pragma unroll
    for (int j = 0; j < DP / 32; ++j) {
        const int i = tid + j * 256, r = i / CH, c = i - r * CH, gr = row0 + r;
        t.v[j] = (gr < nrows && c * 8 < hd) ? *reinterpret_cast<const bf16x8 *>(src + (int64_t)gr * ld + c * 8) : zero8();
    }
}
template <int DP>
__device__ __forceinline__ void store_tile(const TileRegs<DP> &t, bf16 *rm, bf16 *tr, int tid) {
    constexpr int LDR = DP + 8, CH = DP / 8;
#pragma unroll
    for (int j = 0; j < DP / 32; ++j) {
        const int i = tid + j * 256, r = i / CH, c = i - r * CH;
        *reinterpret_cast<bf16x8 *>(rm + r * LDR + c * 8) = t.v[j];
        if (tr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) tr[(c * 8 + e) * LDT + r] = t.v[j][e];
        }
    }
}
__device__ __forceinline__ int fetch_key_mask(const AttnBwdArgs &a, int b, int kv0, int tid) {
    int ok = 0;
    if (tid < 64) {
        const int gk = kv0 + tid;
        ok = gk < a.skv;
        if (ok && a.key_mask) ok = a.key_mask[(int64_t)b * a.skv + gk] != 0;
    }
    return ok;
}

// MFMA layout used throughout (32x32x16 bf16): C[i][j] += sum_k A[i][k] * B[j][k]; lane l holds column j = l % 32 and rows
// i = (r & 3) + 8 * (r >> 2) + 4 * (l / 32) of the 32x32 block; an operand fragment of lane l is row l % 32, 8 k values of group l / 32.
// One operand's fragments are held in registers for the tile a workgroup owns for its whole life:
template <int KD>
__device__ __forceinline__ void load_frags(bf16x8 (&f)[KD], const bf16 *T, int ld, int lane) {
    const bf16 *p = T + (lane & 31) * ld + (lane >> 5) * 8;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) f[kd] = *reinterpret_cast<const bf16x8 *>(p + kd * 16);
}
template <int KD>
__device__ __forceinline__ void mma_ra(f32x16 &c, const bf16x8 (&af)[KD], const bf16 *B, int ldb, int lane) {
    const bf16 *bp = B + (lane & 31) * ldb + (lane >> 5) * 8;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kd], *reinterpret_cast<const bf16x8 *>(bp + kd * 16), c, 0, 0, 0);
}
template <int KD>
__device__ __forceinline__ void mma_rb(f32x16 &c, const bf16 *A, int lda, const bf16x8 (&bfr)[KD], int lane) {
    const bf16 *ap = A + (lane & 31) * lda + (lane >> 5) * 8;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(ap + kd * 16), bfr[kd], c, 0, 0, 0);
}
// C[i][j] += sum_k A[i][k] * T[k][j] over k = 0..63: A is [32][lda] with k contiguous, T is a ROW-MAJOR [64][ldt] tile (k = its
// rows, e.g. the q rows of dO when contracting over queries).  T's MFMA fragment comes from the transposing LDS read
// (`ds_read_b64_tr_b16`): no transposed copy of the streamed tiles is ever written.  The hardware hands lane (l % 32, l / 32) the
// k-slots (hi, jj) <-> k = 16 s + (jj < 4 ? 4 hi + jj : 8 + 4 hi + jj - 4) of step s; A is read in the same order (two 8-byte reads).
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
__device__ __forceinline__ void mma_at(f32x16 &c, const bf16 *A, int lda, const bf16 *T, int ldt, int lane) {
    const int l31 = lane & 31, hi = lane >> 5, g16 = lane >> 4, i16 = lane & 15;
    const bf16 *ap = A + l31 * lda + 4 * hi;
    const bf16 *tp = T + (4 * (g16 >> 1) + (i16 >> 2)) * ldt + 16 * (g16 & 1) + (i16 & 3) * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x4 a0 = *reinterpret_cast<const bf16x4 *>(ap + 16 * s), a1 = *reinterpret_cast<const bf16x4 *>(ap + 16 * s + 8);
        const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(tp + 16 * s * ldt));
        const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(tp + (16 * s + 8) * ldt));
        const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const bf16x8 tf = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, tf, c, 0, 0, 0);
    }
}
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

__device__ __forceinline__ void load_key_mask(const AttnBwdArgs &a, int b, int kv0, int *mk, int tid) {
    if (tid < 64) {
        const int gk = kv0 + tid;
        int ok = gk < a.skv;
        if (ok && a.key_mask) ok = a.key_mask[(int64_t)b * a.skv + gk] != 0;
        mk[tid] = ok;
    }
}

// ---- dQ (+ lse, delta) -------------------------------------------------------------------------------------------------------
// EXTRA: relative position bias and / or dropout on the probabilities (compiled out of the plain kernels: 126 vs 156 us per layer)
template <int DB, bool EXTRA>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnBwdArgs a) {
    constexpr int DP = DB * 32, LDR = DP + 8, DBH = (DB + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Q / dO are staged once, turned into MFMA fragments held in registers, and their LDS is reused for the K / V tiles
    bf16 *Qs = reinterpret_cast<bf16 *>(smem), *dOs = Qs + 64 * LDR, *Ks = Qs, *Vs = dOs;
    bf16 *dSs = dOs + 64 * LDR;
    float *lse_s = reinterpret_cast<float *>(dSs + 64 * LDT), *delta_s = lse_s + 64, *red = delta_s + 64;  // red[2][64][2]
    int *mk = reinterpret_cast<int *>(red + 256);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const int off = a.skv - a.sq;
    const bf16 *qp = a.q + (int64_t)b * a.sq * a.ldq + (int64_t)h * a.hd;
    const bf16 *kp = a.k + (int64_t)b * a.skv * a.ldk + (int64_t)h * a.hd;
    const bf16 *vp = a.v + (int64_t)b * a.skv * a.ldv + (int64_t)h * a.hd;
    const bf16 *op = a.o + (int64_t)b * a.sq * a.ldo + (int64_t)h * a.hd;
    const bf16 *dop = a.d_o + (int64_t)b * a.sq * a.ldo + (int64_t)h * a.hd;
    const int kv_end = a.causal ? min(a.skv, q0 + 63 + off + 1) : a.skv;

    load_tile<DP>(qp, a.ldq, q0, a.sq, a.hd, Qs, nullptr, tid);
    load_tile<DP>(dop, a.ldo, q0, a.sq, a.hd, dOs, nullptr, tid);
    if (tid < 64) delta_s[tid] = 0.0f;
    __syncthreads();
    {  // delta[q] = sum_d o * d_o (d_o from the tile just staged)
        constexpr int CH = DP / 8;
        for (int i = tid; i < 64 * CH; i += 256) {
            const int r = i / CH, c = i - r * CH;
            if (q0 + r < a.sq && c * 8 < a.hd) {
                float ov[8], gv[8], s = 0.0f;
                unpack8(*reinterpret_cast<const bf16x8 *>(op + (int64_t)(q0 + r) * a.ldo + c * 8), ov);
                unpack8(*reinterpret_cast<const bf16x8 *>(dOs + r * LDR + c * 8), gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += ov[e] * gv[e];
                atomicAdd(&delta_s[r], s);
            }
        }
    }

    constexpr int KD = DP / 16;
    bf16x8 qf[KD], gf[KD];  // rows (wid >> 1) * 32.. of Q and dO: B operand of pass 1, A operands of pass 2
    load_frags<KD>(qf, Qs + (wid >> 1) * 32 * LDR, LDR, lane);
    load_frags<KD>(gf, dOs + (wid >> 1) * 32 * LDR, LDR, lane);

    // pass 1: lse of every query row.  Wave (kb, qb) = (wid & 1, wid >> 1) scores keys kb*32.. against queries qb*32..
    {
        const int kb = wid & 1, qb = wid >> 1;
        const int qrow = q0 + qb * 32 + l31;
        float m_run = -1e30f, l_run = 0.0f;
        TileRegs<DP> kr;
        fetch_tile<DP>(kr, kp, a.ldk, 0, a.skv, a.hd, tid);
        int mr = fetch_key_mask(a, b, 0, tid);
        for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
            __syncthreads();
            store_tile<DP>(kr, Ks, nullptr, tid);
            if (tid < 64) mk[tid] = mr;
            if (kv0 + 64 < kv_end) {
                fetch_tile<DP>(kr, kp, a.ldk, kv0 + 64, a.skv, a.hd, tid);
                mr = fetch_key_mask(a, b, kv0 + 64, tid);
            }
            __syncthreads();
            f32x16 s = zero16();
            mma_rb<KD>(s, Ks + kb * 32 * LDR, LDR, qf, lane);
            float mx = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = kb * 32 + crow(r, hi);
                const bool ok = mk[kl] != 0 && (!a.causal || kv0 + kl <= qrow + off);
                s[r] = ok ? s[r] * a.scale + ((EXTRA && a.rel_tab) ? rel_bias(a, h, kv0 + kl, qrow + off) : 0.0f) : -1e30f;
                mx = fmaxf(mx, s[r]);
            }
            const float m_new = fmaxf(m_run, mx);
            float add = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) add += s[r] > -1e29f ? __expf(s[r] - m_new) : 0.0f;
            l_run = l_run * __expf(m_run - m_new) + add;
            m_run = m_new;
        }
        {  // merge the two lane halves, then the two key-block waves
            const float m2 = __shfl_xor(m_run, 32, 64), l2 = __shfl_xor(l_run, 32, 64);
            const float m = fmaxf(m_run, m2);
            l_run = l_run * __expf(m_run - m) + l2 * __expf(m2 - m);
            m_run = m;
        }
        if (hi == 0) {
            red[(kb * 64 + qb * 32 + l31) * 2] = m_run;
            red[(kb * 64 + qb * 32 + l31) * 2 + 1] = l_run;
        }
        __syncthreads();
        if (tid < 64) {
            const float m0 = red[tid * 2], l0 = red[tid * 2 + 1], m1 = red[(64 + tid) * 2], l1 = red[(64 + tid) * 2 + 1];
            const float m = fmaxf(m0, m1);
            const float l = l0 * __expf(m0 - m) + l1 * __expf(m1 - m);
            const float lse = l > 0.0f ? m + __logf(l) : 1e30f;  // no visible key: every P is 0
            lse_s[tid] = lse;
            if (q0 + tid < a.sq) {
                const int64_t idx = ((int64_t)b * a.heads + h) * a.sq + q0 + tid;
                a.lse[idx] = lse;
                a.delta[idx] = delta_s[tid];
            }
        }
    }

    // pass 2: dQ = scale * sum_keys dS K.  Scores by wave (qb2, kb2) = (wid >> 1, wid & 1), laid out [q][key];
    // dQ blocks by wave (qb, half) = (wid & 1, wid >> 1): d blocks [half * DBH, ...)
    const int qb = wid & 1, half = wid >> 1;
    f32x16 acc[DBH];
#pragma unroll
    for (int i = 0; i < DBH; ++i) acc[i] = zero16();
    TileRegs<DP> kr, vr;
    fetch_tile<DP>(kr, kp, a.ldk, 0, a.skv, a.hd, tid);
    fetch_tile<DP>(vr, vp, a.ldv, 0, a.skv, a.hd, tid);
    int mr = fetch_key_mask(a, b, 0, tid);
    for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
        __syncthreads();
        store_tile<DP>(kr, Ks, nullptr, tid);
        store_tile<DP>(vr, Vs, nullptr, tid);
        if (tid < 64) mk[tid] = mr;
        if (kv0 + 64 < kv_end) {
            fetch_tile<DP>(kr, kp, a.ldk, kv0 + 64, a.skv, a.hd, tid);
            fetch_tile<DP>(vr, vp, a.ldv, kv0 + 64, a.skv, a.hd, tid);
            mr = fetch_key_mask(a, b, kv0 + 64, tid);
        }
        __syncthreads();
        {
            const int qb2 = wid >> 1, kb2 = wid & 1;
            f32x16 s = zero16(), dp = zero16();
            mma_ra<KD>(s, qf, Ks + kb2 * 32 * LDR, LDR, lane);
            mma_ra<KD>(dp, gf, Vs + kb2 * 32 * LDR, LDR, lane);
            const int kl = kb2 * 32 + l31;
            const bool kok = mk[kl] != 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = qb2 * 32 + crow(r, hi);
                const bool ok = kok && q0 + ql < a.sq && (!a.causal || kv0 + kl <= q0 + ql + off);
                const float bias = (EXTRA && ok && a.rel_tab) ? rel_bias(a, h, kv0 + kl, q0 + ql + off) : 0.0f;
                const float p = ok ? __expf(s[r] * a.scale + bias - lse_s[ql]) : 0.0f;
                const float df = EXTRA ? drop_factor(a, b, h, q0 + ql, kv0 + kl) : 1.0f;
                dSs[ql * LDT + kl] = (bf16)(p * (dp[r] * df - delta_s[ql]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DBH; ++i) {
            const int db = half * DBH + i;
            if (db < DB) mma_at(acc[i], dSs + qb * 32 * LDT, LDT, Ks + db * 32, LDR, lane);
        }
    }
    bf16 *dqp = a.dq + (int64_t)b * a.sq * a.lddq + (int64_t)h * a.hd;
#pragma unroll
    for (int i = 0; i < DBH; ++i) {
        const int db = half * DBH + i, d = db * 32 + l31;
        if (db < DB && d < a.hd) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + qb * 32 + crow(r, hi);
                if (q < a.sq) dqp[(int64_t)q * a.lddq + d] = (bf16)(acc[i][r] * a.scale);
            }
        }
    }
}

// ---- dK, dV ------------------------------------------------------------------------------------------------------------------
template <int DB, bool EXTRA>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnBwdArgs a) {
    constexpr int DP = DB * 32, LDR = DP + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // K / V are staged once, turned into MFMA fragments held in registers, and their LDS is reused for the Q / dO tiles
    bf16 *Ks = reinterpret_cast<bf16 *>(smem), *Vs = Ks + 64 * LDR, *Qs = Ks, *dOs = Vs;
    bf16 *Pt = Vs + 64 * LDR, *dSt = Pt + 64 * LDT;
    float *lse_s = reinterpret_cast<float *>(dSt + 64 * LDT), *delta_s = lse_s + 64;
    int *mk = reinterpret_cast<int *>(delta_s + 64);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, kv0 = blockIdx.x * 64;
    const int off = a.skv - a.sq;
    const bf16 *qp = a.q + (int64_t)b * a.sq * a.ldq + (int64_t)h * a.hd;
    const bf16 *kp = a.k + (int64_t)b * a.skv * a.ldk + (int64_t)h * a.hd;
    const bf16 *vp = a.v + (int64_t)b * a.skv * a.ldv + (int64_t)h * a.hd;
    const bf16 *dop = a.d_o + (int64_t)b * a.sq * a.ldo + (int64_t)h * a.hd;
    const float *lsep = a.lse + ((int64_t)b * a.heads + h) * a.sq, *deltap = a.delta + ((int64_t)b * a.heads + h) * a.sq;

    load_tile<DP>(kp, a.ldk, kv0, a.skv, a.hd, Ks, nullptr, tid);
    load_tile<DP>(vp, a.ldv, kv0, a.skv, a.hd, Vs, nullptr, tid);
    load_key_mask(a, b, kv0, mk, tid);
    __syncthreads();
    constexpr int KD = DP / 16;
    bf16x8 kf[KD], vf[KD];  // keys (wid & 1) * 32.. : A operands of the score and dP products
    load_frags<KD>(kf, Ks + (wid & 1) * 32 * LDR, LDR, lane);
    load_frags<KD>(vf, Vs + (wid & 1) * 32 * LDR, LDR, lane);

    // wave (kb, half): key block kb; blocks t = half * DB + i: t < DB -> dV block t, else dK block t - DB
    const int kb = wid & 1, half = wid >> 1;
    f32x16 acc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) acc[i] = zero16();
    int q_begin = 0;
    if (a.causal) q_begin = max(0, kv0 - off) & ~63;  // queries below kv0 - off see none of these keys
    TileRegs<DP> qr, gr;
    float lr = 0.0f, dr = 0.0f;
    auto fetch_q = [&](int q0) {
        fetch_tile<DP>(qr, qp, a.ldq, q0, a.sq, a.hd, tid);
        fetch_tile<DP>(gr, dop, a.ldo, q0, a.sq, a.hd, tid);
        const bool ok = tid < 64 && q0 + tid < a.sq;
        lr = ok ? lsep[q0 + tid] : 0.0f;
        dr = ok ? deltap[q0 + tid] : 0.0f;
    };
    if (q_begin < a.sq) fetch_q(q_begin);
    for (int q0 = q_begin; q0 < a.sq; q0 += 64) {
        __syncthreads();
        store_tile<DP>(qr, Qs, nullptr, tid);
        store_tile<DP>(gr, dOs, nullptr, tid);
        if (tid < 64) {
            lse_s[tid] = lr;
            delta_s[tid] = dr;
        }
        if (q0 + 64 < a.sq) fetch_q(q0 + 64);
        __syncthreads();
        {
            const int kb2 = wid & 1, qb2 = wid >> 1;
            f32x16 s = zero16(), dp = zero16();
            mma_ra<KD>(s, kf, Qs + qb2 * 32 * LDR, LDR, lane);
            mma_ra<KD>(dp, vf, dOs + qb2 * 32 * LDR, LDR, lane);
            const int ql = qb2 * 32 + l31;
            const bool qok = q0 + ql < a.sq;
            const float lse = lse_s[ql], dl = delta_s[ql];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = kb2 * 32 + crow(r, hi);
                const bool ok = qok && mk[kl] != 0 && (!a.causal || kv0 + kl <= q0 + ql + off);
                const float bias = (EXTRA && ok && a.rel_tab) ? rel_bias(a, h, kv0 + kl, q0 + ql + off) : 0.0f;
                const float p = ok ? __expf(s[r] * a.scale + bias - lse) : 0.0f;
                const float df = EXTRA ? drop_factor(a, b, h, q0 + ql, kv0 + kl) : 1.0f;
                Pt[kl * LDT + ql] = (bf16)(p * df);  // dV sees the dropped, rescaled probabilities
                dSt[kl * LDT + ql] = (bf16)(p * (dp[r] * df - dl));
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DB; ++i) {
            const int t = half * DB + i;
            const bool is_k = t >= DB;
            const int db = is_k ? t - DB : t;
            mma_at(acc[i], (is_k ? dSt : Pt) + kb * 32 * LDT, LDT, (is_k ? Qs : dOs) + db * 32, LDR, lane);
        }
    }
#pragma unroll
    for (int i = 0; i < DB; ++i) {
        const int t = half * DB + i;
        const bool is_k = t >= DB;
        const int db = is_k ? t - DB : t, d = db * 32 + l31;
        if (d < a.hd) {
            bf16 *dst = is_k ? a.dk + (int64_t)b * a.skv * a.lddk + (int64_t)h * a.hd : a.dv + (int64_t)b * a.skv * a.lddv + (int64_t)h * a.hd;
            const int64_t ld = is_k ? a.lddk : a.lddv;
            const float f = is_k ? a.scale : 1.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + kb * 32 + crow(r, hi);
                if (key < a.skv) dst[(int64_t)key * ld + d] = (bf16)(acc[i][r] * f);
            }
        }
    }
}

template <int DB, bool EXTRA>
int launch_attn_bwd_e(const AttnBwdArgs &a, hipStream_t s) {
    constexpr int DP = DB * 32, LDR = DP + 8;
    const size_t smem_q = (size_t)(2 * 64 * LDR + 64 * LDT) * 2 + (64 + 64 + 256) * 4 + 64 * 4;
    const size_t smem_kv = (size_t)(2 * 64 * LDR + 2 * 64 * LDT) * 2 + (64 + 64) * 4 + 64 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dq_kernel<DB, EXTRA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_q));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_dkv_kernel<DB, EXTRA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv));
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DB, EXTRA>), dim3((a.sq + 63) / 64, a.heads, a.batch), dim3(256), smem_q, s, a);
    EILEV_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DB, EXTRA>), dim3((a.skv + 63) / 64, a.heads, a.batch), dim3(256), smem_kv, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
template <int DB>
int launch_attn_bwd(const AttnBwdArgs &a, hipStream_t s) {
    return (a.rel_tab || a.drop_thr) ? launch_attn_bwd_e<DB, true>(a, s) : launch_attn_bwd_e<DB, false>(a, s);
}

// ---- LayerNorm backward ------------------------------------------------------------------------------------------------------
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma.  One wave per row; stats (mean, rstd) are kept for the
// parameter-gradient kernel.
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16 *__restrict__ x, const bf16 *__restrict__ gamma, const bf16 *__restrict__ dy,
                                                            bf16 *__restrict__ dx, float *__restrict__ stats, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = cols >> 3;
    const bf16 *xr = x + row * cols, *gr = dy + row * cols;
    float v[MAXC][8], g[MAXC][8];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            unpack8(*reinterpret_cast<const bf16x8 *>(xr + c * 8), v[i]);
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
                v[i][e] -= mean;
                sq += v[i][e] * v[i][e];
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)cols + eps);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float gm[8];
            unpack8(*reinterpret_cast<const bf16x8 *>(gr + c * 8), g[i]);
            unpack8(*reinterpret_cast<const bf16x8 *>(gamma + c * 8), gm);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] *= rstd;  // xhat
                g[i][e] *= gm[e];
                s1 += g[i][e];
                s2 += g[i][e] * v[i][e];
            }
        }
    }
    s1 = wave_sum(s1) / (float)cols;
    s2 = wave_sum(s2) / (float)cols;
    bf16 *dr = dx + row * cols;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] - s1 - v[i][e] * s2);
            *reinterpret_cast<bf16x8 *>(dr + c * 8) = pack8(o);
        }
    }
    if (stats && lane == 0) {
        stats[row * 2] = mean;
        stats[row * 2 + 1] = rstd;
    }
}

// column reductions over rows: out0[c] += sum_r dy[r][c] * (LN ? xhat[r][c] : 1); LN also out1[c] += sum_r dy[r][c].
// Block = 32 columns x 8 row lanes; gridDim.y row slices; fp32 atomics into zero-initialised outputs.
template <bool LN>
__global__ __launch_bounds__(256) void col_reduce_kernel(const bf16 *__restrict__ x, const bf16 *__restrict__ dy, const float *__restrict__ stats,
                                                         float *__restrict__ out0, float *__restrict__ out1, int64_t rows, int cols) {
    __shared__ float red[2][8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float a0 = 0.0f, a1 = 0.0f;
    if (c < cols) {
        for (int64_t r = (int64_t)blockIdx.y * 8 + rl; r < rows; r += (int64_t)gridDim.y * 8) {
            const float g = (float)dy[r * cols + c];
            if (LN) {
                const float xh = ((float)x[r * cols + c] - stats[r * 2]) * stats[r * 2 + 1];
                a0 += g * xh;
                a1 += g;
            } else {
                a0 += g;
            }
        }
    }
    red[0][rl][cl] = a0;
    red[1][rl][cl] = a1;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            t0 += red[0][i][cl];
            t1 += red[1][i][cl];
        }
        atomicAdd(out0 + c, t0);
        if (LN) atomicAdd(out1 + c, t1);
    }
}

// ---- activation backward: dx = dy * act'(pre) ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16 *__restrict__ pre, const bf16 *__restrict__ dy, bf16 *__restrict__ dx, int64_t n8, int kind) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float p[8], g[8], o[8];
    unpack8(*reinterpret_cast<const bf16x8 *>(pre + i * 8), p);
    unpack8(*reinterpret_cast<const bf16x8 *>(dy + i * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float d;
        if (kind == 1) d = 0.5f * (1.0f + erff(p[e] * 0.70710678118654752440f)) + p[e] * 0.39894228040143267794f * __expf(-0.5f * p[e] * p[e]);
        else d = p[e] > 0.0f ? 1.0f : 0.0f;
        o[e] = g[e] * d;
    }
    *reinterpret_cast<bf16x8 *>(dx + i * 8) = pack8(o);
}
// forward of the activation alone (training keeps the pre-activation, so the GEMM runs with epilogue 0)
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16 *__restrict__ pre, bf16 *__restrict__ y, int64_t n8, int kind) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float p[8], o[8];
    unpack8(*reinterpret_cast<const bf16x8 *>(pre + i * 8), p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = kind == 1 ? gelu_erf(p[e]) : fmaxf(p[e], 0.0f);
    *reinterpret_cast<bf16x8 *>(y + i * 8) = pack8(o);
}

// ---- T5LayerNorm (RMS, no bias; hf modeling_t5.py:50-72) backward: dx = rstd * (g - xhat * mean(g * xhat)), g = dy * gamma, xhat = x * rstd ----
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16 *__restrict__ x, const bf16 *__restrict__ gamma, const bf16 *__restrict__ dy,
                                                          bf16 *__restrict__ dx, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = cols >> 3;
    const bf16 *xr = x + row * cols, *gr = dy + row * cols;
    float v[MAXC][8], g[MAXC][8];
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            unpack8(*reinterpret_cast<const bf16x8 *>(xr + c * 8), v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sq += v[i][e] * v[i][e];
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)cols + eps);
    float s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float gm[8];
            unpack8(*reinterpret_cast<const bf16x8 *>(gr + c * 8), g[i]);
            unpack8(*reinterpret_cast<const bf16x8 *>(gamma + c * 8), gm);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] *= rstd;
                g[i][e] *= gm[e];
                s2 += g[i][e] * v[i][e];
            }
        }
    }
    s2 = wave_sum(s2) / (float)cols;
    bf16 *dr = dx + row * cols;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] - v[i][e] * s2);
            *reinterpret_cast<bf16x8 *>(dr + c * 8) = pack8(o);
        }
    }
}

// ---- gated activation backward (T5DenseGatedActDense, hf modeling_t5.py:97-124): y = gelu_new(a) * b ----
__global__ __launch_bounds__(256) void gated_gelu_bwd_kernel(const bf16 *__restrict__ ab, const bf16 *__restrict__ dy, bf16 *__restrict__ dab,
                                                             int64_t rows, int F) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = F >> 3;
    if (idx >= rows * ch) return;
    const int64_t r = idx / ch;
    const int c = (int)(idx - r * ch);
    float a[8], b[8], g[8], da[8], db[8];
    unpack8(*reinterpret_cast<const bf16x8 *>(ab + r * 2 * F + c * 8), a);
    unpack8(*reinterpret_cast<const bf16x8 *>(ab + r * 2 * F + F + c * 8), b);
    unpack8(*reinterpret_cast<const bf16x8 *>(dy + r * F + c * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = a[e];
        const float u = 0.79788456080286535588f * (x + 0.044715f * x * x * x);
        const float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
        const float act = 0.5f * x * (1.0f + t);
        const float dact = 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.79788456080286535588f * (1.0f + 3.0f * 0.044715f * x * x);
        da[e] = g[e] * b[e] * dact;
        db[e] = g[e] * act;
    }
    *reinterpret_cast<bf16x8 *>(dab + r * 2 * F + c * 8) = pack8(da);
    *reinterpret_cast<bf16x8 *>(dab + r * 2 * F + F + c * 8) = pack8(db);
}

// ---- hidden-state dropout: y = x * M / (1 - p) (+ resid); the backward is the same call on dy without resid ----------------------
__global__ __launch_bounds__(256) void dropout_add_kernel(const bf16 *__restrict__ x, const bf16 *__restrict__ resid, bf16 *__restrict__ y, int64_t n8,
                                                          uint32_t thr, uint32_t seed, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float v[8], r[8], o[8];
    unpack8(*reinterpret_cast<const bf16x8 *>(x + i * 8), v);
    if (resid) unpack8(*reinterpret_cast<const bf16x8 *>(resid + i * 8), r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float kept = eilev_hash32(seed, (uint64_t)i * 8 + e) >= thr ? v[e] * scale : 0.0f;
        o[e] = resid ? (float)(bf16)kept + r[e] : kept;  // rounded like a separate dropout output before the residual add
    }
    *reinterpret_cast<bf16x8 *>(y + i * 8) = pack8(o);
}

// ---- token cross-entropy: row_loss = lse - logit[target]; dlogits = (softmax - onehot) * grad_scale (0 for ignored rows) ---------
__global__ __launch_bounds__(256) void ce_loss_kernel(const float *__restrict__ logits, const int64_t *__restrict__ targets, float grad_scale,
                                                      float *__restrict__ row_loss, bf16 *__restrict__ dlogits, int vocab) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float *lr = logits + row * vocab;
    bf16 *dr = dlogits ? dlogits + row * vocab : nullptr;  // null: loss only (eval / classify)
    const int64_t t = targets[row];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (t < 0 || t >= vocab) {
        for (int c = tid; dr && c < vocab; c += 256) dr[c] = (bf16)0.0f;
        if (tid == 0) row_loss[row] = 0.0f;
        return;
    }
    float mx = -3.0e38f;
    for (int c = tid; c < vocab; c += 256) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.0f;
    for (int c = tid; c < vocab; c += 256) sum += __expf(lr[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    const float lse = mx + __logf(sum), inv = grad_scale / sum;
    for (int c = tid; dr && c < vocab; c += 256) {
        const float p = __expf(lr[c] - mx) * inv;
        dr[c] = (bf16)(c == t ? p - grad_scale : p);
    }
    if (tid == 0) row_loss[row] = lse - lr[t];
}

}  // namespace

namespace {
int attn_bwd_impl(const void *q, const void *k, const void *v, const void *o, const void *d_o, void *dq, void *dk, void *dv, float *lse_delta,
                  int64_t batch, int64_t heads, int64_t sq, int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddq,
                  int64_t lddk, int64_t lddv, float scale, int causal, const int32_t *key_mask, const float *rel_tab, int64_t rel_stride,
                  int64_t rel_off, int64_t rel_n, float drop_p, uint32_t drop_seed, void *stream) {
    if (!q || !k || !v || !o || !d_o || !dq || !dk || !dv || !lse_delta || batch <= 0 || heads <= 0 || sq <= 0 || skv <= 0) return EILEV_E_BADARG;
    if (head_dim % 8 != 0 || head_dim > 128 || ((ldq | ldk | ldv | lddq | lddk | lddv) & 7)) return EILEV_E_UNSUPPORTED;
    AttnBwdArgs a;
    a.q = (const bf16 *)q; a.k = (const bf16 *)k; a.v = (const bf16 *)v; a.o = (const bf16 *)o; a.d_o = (const bf16 *)d_o;
    a.dq = (bf16 *)dq; a.dk = (bf16 *)dk; a.dv = (bf16 *)dv;
    a.lse = lse_delta; a.delta = lse_delta + batch * heads * sq;
    a.batch = (int)batch; a.heads = (int)heads; a.sq = (int)sq; a.skv = (int)skv; a.hd = (int)head_dim;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = heads * head_dim; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.scale = scale; a.causal = causal; a.key_mask = key_mask;
    if (rel_tab && (rel_n <= 0 || rel_stride < rel_n)) return EILEV_E_BADARG;
    a.rel_tab = rel_tab; a.rel_hs = rel_stride; a.rel_off = (int)rel_off; a.rel_n = (int)rel_n;
    if (drop_p > 0.0f) {
        a.drop_thr = eilev_drop_threshold(drop_p); a.drop_seed = drop_seed; a.drop_scale = 1.0f / (1.0f - drop_p);
    }
    hipStream_t s = (hipStream_t)stream;
    if (head_dim <= 64) return launch_attn_bwd<2>(a, s);
    if (head_dim <= 96) return launch_attn_bwd<3>(a, s);
    return launch_attn_bwd<4>(a, s);
}
}  // namespace

extern "C" int eilev_attention_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, void *dq, void *dk, void *dv,
                                   float *lse_delta, int64_t batch, int64_t heads, int64_t sq, int64_t skv, int64_t head_dim, int64_t ldq,
                                   int64_t ldk, int64_t ldv, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal,
                                   const int32_t *key_mask, void *stream) {
    return attn_bwd_impl(q, k, v, o, d_o, dq, dk, dv, lse_delta, batch, heads, sq, skv, head_dim, ldq, ldk, ldv, lddq, lddk, lddv, scale, causal,
                         key_mask, nullptr, 0, 0, 0, 0.0f, 0, stream);
}
extern "C" int eilev_attention_rel_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, void *dq, void *dk, void *dv,
                                       float *lse_delta, int64_t batch, int64_t heads, int64_t sq, int64_t skv, int64_t head_dim, int64_t ldq,
                                       int64_t ldk, int64_t ldv, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal,
                                       const int32_t *key_mask, const float *rel_tab, int64_t rel_stride, int64_t rel_off, int64_t rel_n,
                                       void *stream) {
    return attn_bwd_impl(q, k, v, o, d_o, dq, dk, dv, lse_delta, batch, heads, sq, skv, head_dim, ldq, ldk, ldv, lddq, lddk, lddv, scale, causal,
                         key_mask, rel_tab, rel_stride, rel_off, rel_n, 0.0f, 0, stream);
}
extern "C" int eilev_attention_dropout_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, void *dq, void *dk,
                                           void *dv, float *lse_delta, int64_t batch, int64_t heads, int64_t sq, int64_t skv, int64_t head_dim,
                                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddq, int64_t lddk, int64_t lddv, float scale,
                                           int causal, const int32_t *key_mask, const float *rel_tab, int64_t rel_stride, int64_t rel_off,
                                           int64_t rel_n, float dropout_p, uint32_t seed, void *stream) {
    if (!(dropout_p >= 0.0f && dropout_p < 1.0f)) return EILEV_E_BADARG;
    return attn_bwd_impl(q, k, v, o, d_o, dq, dk, dv, lse_delta, batch, heads, sq, skv, head_dim, ldq, ldk, ldv, lddq, lddk, lddv, scale, causal,
                         key_mask, rel_tab, rel_stride, rel_off, rel_n, dropout_p, seed, stream);
}

extern "C" int eilev_layernorm_bwd(const void *x, const void *gamma, const void *dy, void *dx, float *dgamma, float *dbeta, float *stats,
                                   int64_t rows, int64_t cols, float eps, void *stream) {
    if (!x || !gamma || !dy || !dx || rows <= 0 || cols <= 0) return EILEV_E_BADARG;
    if (cols % 8 != 0 || cols > 4096) return EILEV_E_UNSUPPORTED;
    if ((dgamma || dbeta) && !(dgamma && dbeta && stats)) return EILEV_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 3) / 4));
    const bf16 *xp = (const bf16 *)x, *gp = (const bf16 *)gamma, *dyp = (const bf16 *)dy;
    if (cols <= 1536) hipLaunchKernelGGL(layernorm_bwd_kernel<3>, grid, dim3(256), 0, s, xp, gp, dyp, (bf16 *)dx, stats, rows, (int)cols, eps);
    else if (cols <= 2560) hipLaunchKernelGGL(layernorm_bwd_kernel<5>, grid, dim3(256), 0, s, xp, gp, dyp, (bf16 *)dx, stats, rows, (int)cols, eps);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<8>, grid, dim3(256), 0, s, xp, gp, dyp, (bf16 *)dx, stats, rows, (int)cols, eps);
    EILEV_LAUNCH_CHECK();
    if (dgamma) {
        const unsigned slices = (unsigned)max((int64_t)1, min((int64_t)64, rows / 64));
        hipLaunchKernelGGL(col_reduce_kernel<true>, dim3((unsigned)((cols + 31) / 32), slices), dim3(256), 0, s, xp, dyp, stats, dgamma, dbeta, rows, (int)cols);
        EILEV_LAUNCH_CHECK();
    }
    return EILEV_OK;
}

extern "C" int eilev_colsum(const void *dy, float *out, int64_t rows, int64_t cols, void *stream) {
    if (!dy || !out || rows <= 0 || cols <= 0) return EILEV_E_BADARG;
    const unsigned slices = (unsigned)max((int64_t)1, min((int64_t)256, rows / 64));
    hipLaunchKernelGGL(col_reduce_kernel<false>, dim3((unsigned)((cols + 31) / 32), slices), dim3(256), 0, (hipStream_t)stream, nullptr,
                       (const bf16 *)dy, nullptr, out, nullptr, rows, (int)cols);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

extern "C" int eilev_act_fwd(const void *pre, void *y, int64_t n, int kind, void *stream) {
    if (!pre || !y || n <= 0 || (kind != 1 && kind != 2)) return EILEV_E_BADARG;
    if (n % 8 != 0) return EILEV_E_UNSUPPORTED;
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)pre, (bf16 *)y, n / 8, kind);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

extern "C" int eilev_act_bwd(const void *pre, const void *dy, void *dx, int64_t n, int kind, void *stream) {
    if (!pre || !dy || !dx || n <= 0 || (kind != 1 && kind != 2)) return EILEV_E_BADARG;
    if (n % 8 != 0) return EILEV_E_UNSUPPORTED;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)pre, (const bf16 *)dy,
                       (bf16 *)dx, n / 8, kind);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

extern "C" int eilev_ce_loss(const float *logits, const int64_t *targets, float grad_scale, float *row_loss, void *dlogits, int64_t rows,
                             int64_t vocab, void *stream) {
    if (!logits || !targets || !row_loss || rows <= 0 || vocab <= 0) return EILEV_E_BADARG;
    hipLaunchKernelGGL(ce_loss_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, targets, grad_scale, row_loss, (bf16 *)dlogits,
                       (int)vocab);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

// ---- encoder-decoder (T5) language model: forward building blocks the training graph composes, and their gradients ----------------
int launch_gated_gelu(const bf16 *ab, int64_t ld, bf16 *out, int64_t rows, int F, hipStream_t s);

extern "C" int eilev_attention_dropout(const void *q, const void *k, const void *v, void *o, int64_t batch, int64_t heads, int64_t sq,
                                       int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, float scale, int causal,
                                       const int32_t *key_mask, const float *rel_tab, int64_t rel_stride, int64_t rel_off, int64_t rel_n,
                                       float dropout_p, uint32_t seed, void *stream);
extern "C" int eilev_attention_rel(const void *q, const void *k, const void *v, void *o, int64_t batch, int64_t heads, int64_t sq, int64_t skv,
                                   int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, float scale, int causal, const int32_t *key_mask,
                                   const float *rel_tab, int64_t rel_stride, int64_t rel_off, int64_t rel_n, void *stream) {
    return eilev_attention_dropout(q, k, v, o, batch, heads, sq, skv, head_dim, ldq, ldk, ldv, scale, causal, key_mask, rel_tab, rel_stride, rel_off,
                                   rel_n, 0.0f, 0, stream);
}

extern "C" int eilev_attention_dropout(const void *q, const void *k, const void *v, void *o, int64_t batch, int64_t heads, int64_t sq,
                                       int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, float scale, int causal,
                                       const int32_t *key_mask, const float *rel_tab, int64_t rel_stride, int64_t rel_off, int64_t rel_n,
                                       float dropout_p, uint32_t seed, void *stream) {
    if (!q || !k || !v || !o || batch <= 0 || heads <= 0 || sq <= 0 || skv <= 0 || !(dropout_p >= 0.0f && dropout_p < 1.0f)) return EILEV_E_BADARG;
    if (rel_tab && (rel_n <= 0 || rel_stride < rel_n)) return EILEV_E_BADARG;
    AttnArgs a;
    a.q = (const bf16 *)q; a.k = (const bf16 *)k; a.v = (const bf16 *)v; a.o = (bf16 *)o;
    a.q_bs = sq * ldq; a.k_bs = skv * ldk; a.v_bs = skv * ldv; a.o_bs = sq * heads * head_dim;
    a.q_hs = a.k_hs = a.v_hs = a.o_hs = head_dim;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = heads * head_dim;
    a.batch = (int)batch; a.heads = (int)heads; a.sq = (int)sq; a.skv = (int)skv; a.hd = (int)head_dim; a.scale = scale;
    a.causal = causal; a.key_mask = key_mask; a.mask_ld = skv; a.dbg = 0;
    a.rel_tab = rel_tab; a.rel_hs = rel_stride; a.rel_off = (int)rel_off; a.rel_n = (int)rel_n;
    if (dropout_p > 0.0f) {
        a.drop_thr = eilev_drop_threshold(dropout_p); a.drop_seed = seed; a.drop_scale = 1.0f / (1.0f - dropout_p);
    }
    return launch_attention(a, (hipStream_t)stream);
}

extern "C" int eilev_dropout_add(const void *x, const void *resid, void *y, int64_t n, float dropout_p, uint32_t seed, void *stream) {
    if (!x || !y || n <= 0 || !(dropout_p >= 0.0f && dropout_p < 1.0f)) return EILEV_E_BADARG;
    if (n % 8 != 0) return EILEV_E_UNSUPPORTED;
    hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)x,
                       (const bf16 *)resid, (bf16 *)y, n / 8, eilev_drop_threshold(dropout_p), seed, 1.0f / (1.0f - dropout_p));
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

extern "C" int eilev_rmsnorm(const void *x, const void *gamma, void *y, int64_t rows, int64_t cols, float eps, void *stream) {
    if (!x || !gamma || !y || rows <= 0 || cols <= 0) return EILEV_E_BADARG;
    if (cols % 8 != 0 || cols > 4096) return EILEV_E_UNSUPPORTED;
    return launch_rmsnorm((const bf16 *)x, cols, (const bf16 *)gamma, (bf16 *)y, cols, rows, (int)cols, eps, (hipStream_t)stream);
}

extern "C" int eilev_rmsnorm_bwd(const void *x, const void *gamma, const void *dy, void *dx, int64_t rows, int64_t cols, float eps, void *stream) {
    if (!x || !gamma || !dy || !dx || rows <= 0 || cols <= 0) return EILEV_E_BADARG;
    if (cols % 8 != 0 || cols > 4096) return EILEV_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + 3) / 4));
    const bf16 *xp = (const bf16 *)x, *gp = (const bf16 *)gamma, *dyp = (const bf16 *)dy;
    if (cols <= 1536) hipLaunchKernelGGL(rmsnorm_bwd_kernel<3>, grid, dim3(256), 0, s, xp, gp, dyp, (bf16 *)dx, rows, (int)cols, eps);
    else if (cols <= 2560) hipLaunchKernelGGL(rmsnorm_bwd_kernel<5>, grid, dim3(256), 0, s, xp, gp, dyp, (bf16 *)dx, rows, (int)cols, eps);
    else hipLaunchKernelGGL(rmsnorm_bwd_kernel<8>, grid, dim3(256), 0, s, xp, gp, dyp, (bf16 *)dx, rows, (int)cols, eps);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

extern "C" int eilev_gated_gelu(const void *ab, void *out, int64_t rows, int64_t f, void *stream) {
    if (!ab || !out || rows <= 0 || f <= 0) return EILEV_E_BADARG;
    return launch_gated_gelu((const bf16 *)ab, 2 * f, (bf16 *)out, rows, (int)f, (hipStream_t)stream);
}

extern "C" int eilev_gated_gelu_bwd(const void *ab, const void *dy, void *dab, int64_t rows, int64_t f, void *stream) {
    if (!ab || !dy || !dab || rows <= 0 || f <= 0) return EILEV_E_BADARG;
    if (f % 8 != 0) return EILEV_E_UNSUPPORTED;
    const int64_t total = rows * (f >> 3);
    hipLaunchKernelGGL(gated_gelu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16 *)ab,
                       (const bf16 *)dy, (bf16 *)dab, rows, (int)f);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
