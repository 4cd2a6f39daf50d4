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
#include <type_traits>

#include "common.h"

int g_attn_force_v1 = 0, g_attn_dbg = 0;
extern "C" int eilev_debug_attn_v1(int on) { g_attn_force_v1 = on & 1; g_attn_dbg = on >> 1; return 0; }

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

// ---------------------------------------------------------------------------------------------------------
// v2 prefill kernel for the "awkward" head sizes of the real model (hd = 80 OPT-2.7B, 88 ViT-g; DP = 96).
//  * each wave owns 32 query rows (32x32x16 MFMA), a workgroup of NWQ waves covers 32*NWQ rows of one
//    (batch, head): NWQ = 9 covers a whole ViT frame (257 rows) so K/V are fetched once per (frame, head);
//  * K and V tiles of 64 keys go global -> LDS by LDS-DMA (global_load_lds_dwordx4), double buffered, in
//    their natural row-major layout (row stride hd*2 bytes: 160/176 B rows are bank-conflict-free for the
//    32-row ds_read_b128 fragment pattern) — no staging registers, no ds_write, no transposition pass;
//  * S^T = K Q^T as in v1 (lane owns one q row); the V^T fragments of O^T = V^T P^T come straight from the
//    row-major V tile through ds_read_b64_tr_b16 (hardware 4x16 transpose: output lane c, element j of a
//    16-lane group = input lane 4j + c/4, element c%4 — measured with tools/probes/tr_read.hip);
//  * columns d >= hd of a row alias the next row's first bytes (for the last K row: the first V row of the
//    same stage, landed by the same DMA batch): always finite bf16 data, multiplied by q columns that are
//    exactly 0 for K and producing only O^T rows that are never stored for V;
//  * O is staged per wave through LDS so that HBM sees whole hd*2-byte row segments, 16 bytes per lane.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

template <int NWQ>
__global__ __launch_bounds__(64 * NWQ) void attn_prefill_v2_kernel(const AttnArgs a) {
    constexpr int DB = 3;              // 32-wide d blocks (DP = 96)
    constexpr int KD = 6;              // k = 16 MFMA steps over DP
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int RS = a.hd * 2, CH = a.hd >> 3;  // LDS row stride (bytes), 16-byte chunks per row
    const int T = 64 * RS;                     // bytes per K (or V) tile
    int *msk = reinterpret_cast<int *>(smem + 4 * T + 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * NWQ + wid) * 32;
    const int qrow = q0 + l31;
    const int off = a.skv - a.sq;
    const float sl2 = a.scale * 1.44269504088896340736f;

    const bf16 *qp = a.q + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs;
    const bf16 *kp = a.k + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs;
    const bf16 *vp = a.v + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs;

    int kv_end = a.skv;
    if (a.causal) {
        const int last_q = min(a.sq - 1, (int)(blockIdx.x * NWQ + NWQ) * 32 - 1);
        kv_end = min(a.skv, last_q + off + 1);
    }
    const int ntiles = (kv_end + 63) / 64;

    // DMA: piece i covers linear 16-byte chunks p = i*64 + lane of a tile: key = p / CH, chunk = p % CH
    auto stage_in = [&](int buf, int t) {
        const int kv0 = t * 64;
        for (int i = wid; i < 2 * CH; i += NWQ) {  // pieces [0, CH) = K, [CH, 2CH) = V
            const bool isv = i >= CH;
            const int pi = isv ? i - CH : i;
            const int pch = pi * 64 + lane, key = pch / CH, c = pch - key * CH;
            int gk = kv0 + key;
            gk = gk < a.skv ? gk : a.skv - 1;
            const bf16 *src = isv ? vp + (int64_t)gk * a.ldv + c * 8 : kp + (int64_t)gk * a.ldk + c * 8;
            char *dst = smem + buf * 2 * T + (isv ? T : 0) + pi * 1024;
            __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)dst, 16, 0, 0);
        }
        if (tid < 64) {
            const int gk = kv0 + tid;
            int ok = gk < a.skv;
            if (ok && a.key_mask) ok = a.key_mask[(int64_t)b * a.mask_ld + gk] != 0;
            msk[buf * 64 + tid] = ok;
            const int all = __all(ok);
            if (tid == 0) msk[128 + buf] = all;
        }
    };

    bf16x8 qf[KD];
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) {
        const int d0 = kd * 16 + hi * 8;
        qf[kd] = (qrow < a.sq && d0 < a.hd) ? *reinterpret_cast<const bf16x8 *>(qp + (int64_t)qrow * a.ldq + d0) : zero8();
    }
    float m_run = -1e30f, l_run = 0.0f;
    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.0f;

    if (ntiles > 0) stage_in(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool active = q0 < a.sq;
    const int g16 = lane >> 4, i16 = lane & 15;
    // one KV tile = NB (1 or 2) blocks of 32 keys, processed together: independent MFMA chains, one softmax
    // update and one rescale of O per tile
    auto tile_body = [&](auto nb_tag, auto masked_tag, int cur, int kv0) {
        constexpr int NB = decltype(nb_tag)::value;
        constexpr bool MASKED = decltype(masked_tag)::value;
        const char *kt_ = smem + cur * 2 * T, *vt_ = kt_ + T;
        f32x16 st[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.0f;
        const int toff = kv0 & 63;  // 0, or 32 when the second half of a tile is processed on its own
        const char *krow = kt_ + (toff + l31) * RS + hi * 16;
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(krow + kb * 32 * RS + kd * 32);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kd], st[kb], 0, 0, 0);
            }
        }
        // st[kb][r] = S[q = l31][key = kv0 + 32 kb + (r&3) + 8*(r>>2) + 4*hi]   (raw q.k, scale folded below)
        if (MASKED) {
            const int *mk = msk + cur * 64;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = mk[toff + kl] != 0 && (!a.causal || (kv0 + kl) <= qrow + off);
                    st[kb][r] = ok ? st[kb][r] : -1e30f;
                }
        }
        float mx = -1e30f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = fmaxf(mx * sl2, -1e30f);  // scale > 0: max commutes with the scaling
        // lazy rescale (wave-uniform): the running max is only raised when some row outgrows it by 2^6
        if (!__all(mxs <= m_run + 6.0f)) {
            const float m_new = fmaxf(m_run, mxs);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        const float nm = -fmaxf(m_run, -1e20f);  // fully masked so far: keep exp2 arguments at -inf-like values
        float rs = 0.0f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[kb][r], sl2, nm));
                st[kb][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run += rs;
        // O^T += V^T P^T.  k-slot (hi, j) of step s of block kb <-> key kv0 + 32 kb + 16 s + (j < 4 ? 4 hi + j : 8 + 4 hi + j - 4)
        const char *vbase = vt_ + (toff + 4 * (g16 >> 1) + (i16 >> 2)) * RS + (16 * (g16 & 1) + (i16 & 3) * 4) * 2;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (bf16)st[kb][8 * s2 + j];
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const char *va = vbase + (kb * 32 + 16 * s2) * RS + db * 64;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)va);
                    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(va + 8 * RS));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[db], 0, 0, 0);
                }
            }
        }
    };

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles && !(a.dbg & 2)) stage_in(cur ^ 1, t + 1);
        if (active && !(a.dbg & 1)) {
            const int kv0 = t * 64;
            int vis_end = kv_end;  // first key no row of this wave may see
            if (a.causal) vis_end = min(vis_end, min(a.sq - 1, q0 + 31) + off + 1);
            // a tile needs no per-score masking when all 64 keys exist, pass the key mask and lie at or
            // below every row's causal limit
            const bool plain = msk[128 + cur] != 0 && (!a.causal || kv0 + 63 <= q0 + off);
            using two = std::integral_constant<int, 2>;
            using one = std::integral_constant<int, 1>;
            if (NWQ > 8) {
                // 9 waves per workgroup leave 168 VGPRs per wave: one 32-key block at a time
                if (kv0 < vis_end) {
                    if (plain) tile_body(one{}, std::false_type{}, cur, kv0);
                    else tile_body(one{}, std::true_type{}, cur, kv0);
                }
                if (kv0 + 32 < vis_end) {
                    if (plain) tile_body(one{}, std::false_type{}, cur, kv0 + 32);
                    else tile_body(one{}, std::true_type{}, cur, kv0 + 32);
                }
            } else if (kv0 + 32 < vis_end) {
                if (plain) tile_body(two{}, std::false_type{}, cur, kv0);
                else tile_body(two{}, std::true_type{}, cur, kv0);
            } else if (kv0 < vis_end) {
                tile_body(one{}, std::true_type{}, cur, kv0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // o[db][r] = O[q = l31][d = db*32 + (r&3) + 8*(r>>2) + 4*hi]; stage the wave's 32 x hd tile in LDS (the KV
    // stages are dead after the last barrier; each wave owns a private region) and store coalesced rows.
    {
        constexpr int OS = 200;  // staging row stride (bytes): 96 bf16 + pad
        char *reg = smem + wid * (32 * OS);
        const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (bf16)(o[db][g * 4 + e] * inv);
                *reinterpret_cast<bf16x4 *>(reg + l31 * OS + (db * 32 + g * 8 + hi * 4) * 2) = w;
            }
        bf16 *ob = a.o + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs;
        for (int pch = lane; pch < 32 * CH; pch += 64) {
            const int row = pch / CH, c = pch - row * CH;
            if (q0 + row < a.sq) {
                const bf16x4 *sp = reinterpret_cast<const bf16x4 *>(reg + row * OS + c * 16);
                const bf16x4 lo = sp[0], hi4 = sp[1];
                *reinterpret_cast<bf16x8 *>(ob + (int64_t)(q0 + row) * a.ldo + c * 8) =
                    (bf16x8){lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            }
        }
    }
}

template <int NWQ>
int launch_attn_v2(const AttnArgs &a, hipStream_t s) {
    size_t smem = (size_t)4 * 64 * a.hd * 2 + 256 + (2 * 64 + 2) * sizeof(int);
    if (smem < (size_t)NWQ * 32 * 200) smem = (size_t)NWQ * 32 * 200;
    const dim3 grid((a.sq + 32 * NWQ - 1) / (32 * NWQ), a.heads, a.batch), block(64 * NWQ);
    hipLaunchKernelGGL(attn_prefill_v2_kernel<NWQ>, grid, block, smem, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

}  // namespace

int launch_attention(const AttnArgs &a_in, hipStream_t s) {
    AttnArgs a = a_in;
    a.dbg = g_attn_dbg;
    if (a.batch <= 0 || a.sq <= 0) return EILEV_OK;
    if (!a.q || !a.k || !a.v || !a.o || a.skv <= 0) return EILEV_E_BADARG;
    if ((a.hd & 7) || a.hd > 128 || (a.ldq & 7) || (a.ldk & 7) || (a.ldv & 7) || (a.ldo & 7) || ((uintptr_t)a.o & 15) || (a.q_hs & 7) ||
        (a.k_hs & 7) || (a.v_hs & 7) || (a.o_hs & 7) || (a.q_bs & 7) || (a.k_bs & 7) || (a.v_bs & 7) || (a.o_bs & 7))
        return EILEV_E_UNSUPPORTED;
    if (!g_attn_force_v1 && (a.hd == 80 || a.hd == 88 || a.hd == 72) && a.skv >= 32) {
        const int qt = (a.sq + 31) / 32;
        if (qt == 9 || qt > 16) return (qt == 9) ? launch_attn_v2<9>(a, s) : launch_attn_v2<8>(a, s);
        if (qt >= 5) return launch_attn_v2<8>(a, s);
        if (qt >= 3) return launch_attn_v2<4>(a, s);
        return launch_attn_v2<2>(a, s);
    }
    const dim3 grid((a.sq + 63) / 64, a.heads, a.batch), block(256);
    if (a.hd <= 64) hipLaunchKernelGGL(attn_prefill_kernel<64>, grid, block, 0, s, a);
    else if (a.hd <= 96) hipLaunchKernelGGL(attn_prefill_kernel<96>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(attn_prefill_kernel<128>, grid, block, 0, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
