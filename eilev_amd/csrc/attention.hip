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
#include <type_traits>
#include <utility>

int g_attn_force_v1 = 0, g_attn_dbg = 0;
#ifdef EILEV_PROBES
extern "C" int eilev_debug_attn_v1(int on) { g_attn_force_v1 = on & 1; g_attn_dbg = on >> 1; return 0; }
#endif

namespace {

template <int DP, bool DROP = false>  // DROP: dropout on the probabilities (training graph), compiled out of the inference kernel
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
                float s = st[ct][r] * sl2;
                if (a.rel_tab) {  // T5 relative position bias, a function of (key - query position) per head
                    int ri = (kv0 + kl) - (qrow + off) + a.rel_off;
                    ri = ri < 0 ? 0 : (ri >= a.rel_n ? a.rel_n - 1 : ri);
                    s = fmaf(a.rel_tab[(int64_t)h * a.rel_hs + ri], 1.44269504088896340736f, s);
                }
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
                if (DROP) {  // training: the row sum keeps every probability, the product with V only the kept ones (scaled)
                    const uint64_t idx = (((uint64_t)b * a.heads + h) * a.sq + qrow) * (uint64_t)a.skv + (kv0 + ct * 16 + lg * 4 + r);
                    st[ct][r] = eilev_hash32(a.drop_seed, idx) >= a.drop_thr ? p * a.drop_scale : 0.0f;
                }
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

// Round 5: DB = 2 is the hd = 64 form (flan-t5: 32 heads x 64; no padded d block).  Its rows are 128 bytes — a power-of-two stride, so the
// 16-byte chunks of a row are XOR-swizzled (chunk c of key row r sits at chunk c ^ ((r >> 1) & 7), the GEMM kernels' swz()): the LDS-DMA
// applies it on the SOURCE side (lane p of a piece fetches the global chunk that belongs at LDS chunk p), the K fragment reads and the
// transposing V reads on their addresses.  REL: T5's relative position bias, bias(h, key - query) — the head's table (2 L - 1 floats for
// the encoder, <= 4096) is copied to LDS once per workgroup with 64 zeros of slack on both sides (rows / keys past the ends index there),
// added to the raw scores before masking.  (The v1 kernel this replaces for the T5 encoder ran at ~120 TFLOP/s: 41 % of the encoder's time.)
constexpr int ATTN_V2_REL_MAX = 4096, ATTN_V2_REL_SLACK = 64;
// bytes of the K / V stages + key-mask words, or of the O staging that overlays them at the end, whichever is larger (16-byte multiple):
// the relative-position table sits behind both
// Round 6: DB = 4 is the hd = 128 form (OPT-6.7B, BASELINE configs[4]; the round-1 kernel ran its prefill attention at 3.9 ms per block = 23 % of
// the fp8 prefill).  256-byte rows would put every row's chunk c on the same 16 banks, so the LDS image PADS a row to 17 chunks (272 bytes:
// the 16 lanes one ds_read_b128 services together then hit 16 different slots, like the 160 / 176-byte rows of DB = 3); the LDS-DMA fills the
// 17th chunk with a second copy of the 16th (lane -> source chunk is free on the source side), 6 % more bytes through the DMA, none read.
__host__ __device__ inline int attn_v2_row_bytes(int hd) { return hd == 128 ? 272 : hd * 2; }
__host__ __device__ inline int attn_v2_ostage_bytes(int hd) { return hd == 128 ? 272 : 200; }  // row stride of the per-wave O staging
__host__ __device__ inline int attn_v2_rel_offset(int nwq, int hd) {
    int b = 4 * 64 * attn_v2_row_bytes(hd) + 256 + (2 * 64 + 2) * (int)sizeof(int);
    if (b < nwq * 32 * attn_v2_ostage_bytes(hd)) b = nwq * 32 * attn_v2_ostage_bytes(hd);
    return (b + 15) & ~15;
}
template <int NWQ, int DB = 3, bool REL = false>
__global__ __launch_bounds__(64 * NWQ) void attn_prefill_v2_kernel(const AttnArgs a) {
    constexpr int KD = 2 * DB;         // k = 16 MFMA steps over DP = 32 DB (96: hd 72 / 80 / 88; 64: hd 64; 128: hd 128)
    constexpr bool SWZ = DB == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int RS = DB == 4 ? 272 : a.hd * 2, CH = RS >> 4;  // LDS row stride (bytes), 16-byte chunks per LDS row
    const int CHS = a.hd >> 3;                                // 16-byte chunks per row of q / k / v / o in memory
    const int T = 64 * RS;                     // bytes per K (or V) tile
    int *msk = reinterpret_cast<int *>(smem + 4 * T + 256);
    float *rel_lds = reinterpret_cast<float *>(smem + attn_v2_rel_offset(NWQ, a.hd));  // behind everything else (incl. the O staging)

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
            const int pch = pi * 64 + lane, key = pch / CH, cl = pch - key * CH;
            const int c = SWZ ? (cl ^ ((key >> 1) & 7)) : (DB == 4 && cl >= CHS ? CHS - 1 : cl);  // the global chunk that lives at LDS chunk cl of this row
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
    if constexpr (REL) {
        const float *rt = a.rel_tab + (int64_t)h * a.rel_hs;
        for (int i = tid; i < a.rel_n + 2 * ATTN_V2_REL_SLACK; i += 64 * NWQ) {
            const int j = i - ATTN_V2_REL_SLACK;
            rel_lds[i] = (j >= 0 && j < a.rel_n) ? rt[j] : 0.0f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool active = q0 < a.sq;
    const int g16 = lane >> 4, i16 = lane & 15;
    const float inv_scale = 1.0f / a.scale;
    // SWZ: per-lane chunk offsets of the K fragment reads (row & 7 pattern of the lane's key row is the same in every 32-key block)
    int kswz[SWZ ? KD : 1];
    if constexpr (SWZ) {
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) kswz[kd] = ((kd * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;
    }
    const int khi = hi * 16;
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
        const char *krow = kt_ + (toff + l31) * RS;
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(krow + kb * 32 * RS + (SWZ ? kswz[SWZ ? kd : 0] : kd * 32 + khi));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kd], st[kb], 0, 0, 0);
            }
        }
        // st[kb][r] = S[q = l31][key = kv0 + 32 kb + (r&3) + 8*(r>>2) + 4*hi]   (raw q.k, scale folded below)
        if constexpr (REL) {  // + bias(h, key - query) / scale (the scale is folded into the exponent below; T5: scale = 1)
            const float *rl = rel_lds + ATTN_V2_REL_SLACK + (kv0 - qrow - off + a.rel_off + 4 * hi);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = fmaf(rl[kb * 32 + (r & 3) + 8 * (r >> 2)], inv_scale, st[kb][r]);
        }
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
        // SWZ: the lane's 8 bytes are half of global chunk gb + 4 db of row R = toff + 4 (g16 >> 1) + (i16 >> 2) (+ 8 for the second read, + 16 s2
        // + 32 kb); (R >> 1) & 7 = x0 (+ 4 for the second read), x0 = 2 (g16 >> 1) + (i16 >> 3) < 4, gb < 4: chunk = (gb ^ x0) + 4 (db ^ second)
        const int v_gb = 2 * (g16 & 1) + ((i16 & 3) >> 1), v_x0 = 2 * (g16 >> 1) + (i16 >> 3);
        const char *vrow = vt_ + (toff + 4 * (g16 >> 1) + (i16 >> 2)) * RS;
        const char *vbase = SWZ ? vrow + ((v_gb ^ v_x0) << 4) + 8 * (i16 & 1) : vrow + (16 * (g16 & 1) + (i16 & 3) * 4) * 2;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (bf16)st[kb][8 * s2 + j];
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const char *va = vbase + (kb * 32 + 16 * s2) * RS + (SWZ ? 0 : db * 64);
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(va + (SWZ ? db * 64 : 0)));
                    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)(va + 8 * RS + (SWZ ? (db ^ 1) * 64 : 0)));
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
            if (NWQ > 8 || (DB == 4 && NWQ == 8)) {
                // 9 waves per workgroup leave 168 VGPRs per wave: one 32-key block at a time (round 6: also the hd = 128 form at 8 waves: its O
                // accumulators alone are 64 registers; with two blocks in flight hipcc spilled 31 VGPRs)
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
        constexpr int OS = DB == 4 ? 272 : 200;  // staging row stride (bytes): 96 (128) bf16 + pad
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
        for (int pch = lane; pch < 32 * CHS; pch += 64) {
            const int row = pch / CHS, c = pch - row * CHS;
            if (q0 + row < a.sq) {
                const bf16x4 *sp = reinterpret_cast<const bf16x4 *>(reg + row * OS + c * 16);
                const bf16x4 lo = sp[0], hi4 = sp[1];
                *reinterpret_cast<bf16x8 *>(ob + (int64_t)(q0 + row) * a.ldo + c * 8) =
                    (bf16x8){lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Frame attention: the ViT case (hf Blip2Attention, S = 257 tokens per frame, hd = 88, no mask), persistent.
//  * one workgroup (9 waves) per CU walks (frame, head) pairs; the whole K and V of a pair (257 x 176 B each)
//    live in LDS: ring [K(even pairs) | K(odd pairs) | V]; K of the NEXT pair is DMA'd while the current pair
//    computes, V of the current pair lands under the first S phase + softmax;
//  * a wave owns 16-row query tiles (16x16x32 MFMA; tiles w and w + 9 of the 17): S^T = K Q^T for ALL keys of
//    the row stays in registers (17 tiles x 4 = 68 fp32), so the softmax is exact single pass — no running
//    max, no rescale of O, no barrier inside a tile — and P is already in the B-operand layout of O^T = V^T P^T;
//  * k-slots of the S MFMA are a permutation of d chosen so that a K fragment is two ds_read_b64 that are
//    bank-conflict free on 176-byte rows (slot (g, j) <-> d = 32 ks + 16 (j >> 2) + 4 g + (j & 3));
//  * the 16 keys of a tile sit in the MFMA rows in the order 0 2 4 .. 14 1 3 .. 15: the V^T fragments come from
//    ds_read_b64_tr_b16 (32 lanes per LDS cycle), and with that order the 8 key rows a half-wave transposes are all even
//    (or all odd), which is what makes the 64 dwords it touches fall into 64 different banks on 176-byte rows;
//  * O^T leaves the MFMA with 4 consecutive d per lane and tile; lane pairs (g, g ^ 1) swap halves so that a store
//    is 16 bytes per lane and 64 contiguous bytes per query row, no LDS staging;
//  * rows >= S of the LDS images are copies of the last key row (DMA source clamped): finite, and masked (K) or
//    multiplied by P = 0 (V).
__device__ __forceinline__ void attn_dma16(__amdgpu_buffer_rsrc_t rsrc, lds_void_t *dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, 0, 0, 0);
}

__device__ unsigned long long g_attn_ts[9 * 8 * 16];  // probe: [wave][pair < 8][event < 16] s_memtime stamps of workgroup 0
#ifdef EILEV_PROBES
extern "C" int eilev_debug_attn_ts(void *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_ts), sizeof(g_attn_ts));
}
#endif
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) const bf16x4 lds_cbf16x4_t;

template <int HD, int NT>
__global__ __launch_bounds__(576) void attn_frame_kernel(const AttnArgs a) {
    constexpr int NW = 9;
    constexpr int CH = HD / 8, RS = HD * 2;
    constexpr int NPIECE = (NT * 16 * CH + (12 - CH) + 63) / 64;  // 1-KiB pieces per K (or V) image, incl. the d >= HD overhang
    constexpr int BUF = NPIECE * 1024;
    constexpr int PPW = (NPIECE + NW - 1) / NW;
    constexpr int KS = 3;               // k-steps of 32 over the padded head dim 96
    constexpr int NKS = (NT + 1) / 2;   // k-steps of 32 keys in O^T = V^T P^T
    constexpr int DT = 6;               // 16-row tiles of O^T
    static_assert(NT > NW && NT <= 2 * NW && HD <= 96 && HD % 8 == 0 && NT * 16 * RS < 65536 - 512, "tile split / ds offset field");
    static_assert(PPW == 6, "the counted s_waitcnt vmcnt(6) below assume 6 DMA pieces and 6 Q loads per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char_t *lds = (lds_char_t *)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int S = a.sq;
    const int npairs = a.batch * a.heads;
    const float sl2 = a.scale * 1.44269504088896340736f;

    // DMA: piece i covers 16-byte chunks 64 i + lane of an image, chunk p = (key p / CH, c = p % CH); wave w issues pieces
    // w, w + 9, ... (surplus slots repeat the last piece: same bytes to the same place).  The geometry is recomputed per
    // call (a handful of VALU ops) rather than kept in registers.
    auto stage = [&](const bf16 *src, int buf) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
        const unsigned ld2 = (unsigned)(a.ldk * 2);
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            int i = wid + NW * k;
            i = i < NPIECE ? i : NPIECE - 1;
            const int pch = i * 64 + lane;
            int key = pch / CH;
            const int c = pch - key * CH;
            key = key < S ? key : S - 1;
            attn_dma16(r, (lds_void_t *)(smem + buf * BUF + i * 1024), (unsigned)key * ld2 + c * 16);
        }
    };
    // Q fragments of a tile (3 x 16 bytes per lane), fetched a pair ahead: [0] first tile (w), [1] second tile (w + 9)
    bf16x8 qf[2][KS];
    auto load_q = [&](int pair, int ti, bf16x8 (&dst)[KS]) {
        const int b = pair / a.heads, h = pair - b * a.heads;
        const int row = (wid + NW * ti) * 16 + l15;
        const bf16 *rp = a.q + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs + (int64_t)(row < S ? row : S - 1) * a.ldq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 32 + 4 * g, d1 = d0 + 16;
            bf16x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
            if (row < S && d0 + 4 <= HD) lo = *reinterpret_cast<const bf16x4 *>(rp + d0);
            if (row < S && d1 + 4 <= HD) hi = *reinterpret_cast<const bf16x4 *>(rp + d1);
            dst[ks] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
#define FA_BARRIER()                       \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

    // A row i of a key tile <-> key 16 t + pi(i), pi = (0 2 4 .. 14 1 3 .. 15): lanes 0-31 of the V^T transpose reads then
    // touch even rows only, which makes them bank-conflict free on 176-byte rows
    const int prow = l15 < 8 ? 2 * l15 : 2 * l15 - 15;
    // K fragments are read with ds_read_b64 in inline asm: left to the compiler, pairs of them are fused into
    // ds_read2_b64, which runs at half the rate with a 32-bank mapping that conflicts on these rows
    const unsigned kbase = prow * RS + g * 8;
    const int vi = 4 * g + (l15 >> 2);
    const unsigned voff_t = 2 * BUF + (vi < 8 ? 2 * vi : 2 * vi - 15) * RS + (l15 & 3) * 8;
    // ---- building blocks.  sc[t][r] = q_row . k_(16 t + pi(4 g + r)); two key tiles per call = two independent MFMA chains
    // issue the 6 fragment reads of key tile t; kf[] is valid after the wait in s_mma
    auto s_issue = [&](unsigned kaddr, bf16x4 (&kf)[2 * KS], int t) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(kf[ks * 2 + hf]) : "v"(kaddr), "i"(t * 16 * RS + ks * 64 + hf * 32));
    };
    // NAFTER = number of younger ds_read_b64 already issued (the next tile's fragments): wait for everything older
    auto s_mma = [&](const bf16x8 (&q)[KS], bf16x4 (&kf)[2 * KS], auto &sc, int t, auto nafter) {
        constexpr int NAFTER = decltype(nafter)::value;
        // every fragment register is an in/out operand of the wait so that no MFMA can be scheduled above it
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]) : "n"(NAFTER));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x4 lo = kf[ks * 2], hi = kf[ks * 2 + 1];
            const bf16x8 k0 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, q[ks], acc, 0, 0, 0);
        }
        sc[t] = acc;
    };
    // row maximum over all keys (4 lanes share a row: xor 16, xor 32) -> the exponent offset -max * scale * log2 e
    auto row_offset = [&](auto &sc) -> float {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // keys of the last tile that do not exist (A row 4 g + r -> key)
            const int ar = 4 * g + r;
            if ((NT - 1) * 16 + (ar < 8 ? 2 * ar : 2 * ar - 15) >= S) sc[NT - 1][r] = -1e30f;
        }
        float m4[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
        for (int t = 0; t < NT; t += 2)
#pragma unroll
            for (int r = 0; r < 4; ++r) m4[r] = t + 1 < NT ? fmaxf(fmaxf(m4[r], sc[t][r]), sc[t + 1][r]) : fmaxf(m4[r], sc[t][r]);
        float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        return -mx * sl2;
    };
    // p = exp2(s * scale * log2 e + offset) for key tiles t, t + 1: packed fp32 arguments and row sums, bf16 P
    auto sm_tiles = [&](const auto &sc, bf16x4 (&pr)[NT], int t, float nm, f32x2 &la, f32x2 &lb) {
#pragma unroll
        for (int u = t; u < t + 2 && u < NT; ++u) {
            const f32x2 x0 = (f32x2){sc[u][0], sc[u][1]} * sl2 + nm, x1 = (f32x2){sc[u][2], sc[u][3]} * sl2 + nm;
            const f32x2 p0 = {__builtin_amdgcn_exp2f(x0.x), __builtin_amdgcn_exp2f(x0.y)};
            const f32x2 p1 = {__builtin_amdgcn_exp2f(x1.x), __builtin_amdgcn_exp2f(x1.y)};
            la += p0;
            lb += p1;
            pr[u] = (bf16x4){(bf16)p0.x, (bf16)p0.y, (bf16)p1.x, (bf16)p1.y};
        }
    };
    auto row_sum = [&](f32x2 la, f32x2 lb) -> float {
        float l = (la.x + la.y) + (lb.x + lb.y);
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        return l;
    };
    // O^T += V^T P^T for keys of tiles 2 k2, 2 k2 + 1: o[dt][r] = O[q_row][16 dt + 4 g + r].  The V^T fragments come from
    // ds_read_b64_tr_b16 in inline asm with explicit lgkmcnt waits: a compiler-visible LDS read after an LDS-DMA makes
    // hipcc drain vmcnt(0) first — i.e. wait for the K image of the NEXT pair that was issued a moment ago.
    const unsigned vaddr = (unsigned)(uintptr_t)lds + voff_t;
    auto pv_issue = [&](bf16x4 (&vf)[2 * DT], int k2) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vf[2 * dt]) : "v"(vaddr), "i"(2 * k2 * 16 * RS + dt * 32));
            if (2 * k2 + 1 < NT)
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vf[2 * dt + 1]) : "v"(vaddr), "i"((2 * k2 + 1) * 16 * RS + dt * 32));
        }
    };
    auto pv_mma = [&](f32x4 (&o)[DT], const bf16x4 (&pr)[NT], bf16x4 (&vf)[2 * DT], int k2, auto nafter) {
        constexpr int NAFTER = decltype(nafter)::value;
        constexpr bf16x4 z4 = {0, 0, 0, 0};
        asm volatile("s_waitcnt lgkmcnt(%12)"
                     : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(vf[4]), "+v"(vf[5]), "+v"(vf[6]), "+v"(vf[7]), "+v"(vf[8]),
                       "+v"(vf[9]), "+v"(vf[10]), "+v"(vf[11])
                     : "n"(NAFTER));
        const bool two = 2 * k2 + 1 < NT;
        const bf16x4 p0 = pr[2 * k2], p1 = two ? pr[2 * k2 + 1 < NT ? 2 * k2 + 1 : 0] : z4;
        const bf16x8 pb = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x4 lo = vf[2 * dt], hi = two ? vf[2 * dt + 1] : z4;
            const bf16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v8, pb, o[dt], 0, 0, 0);
        }
    };
    // O / l -> bf16.  Lane pairs (g, g ^ 1) swap halves so that every lane holds 8 consecutive d of one tile (even g:
    // tile 2 m, odd g: tile 2 m + 1) -> 16-byte stores, 64 contiguous bytes per query row
    auto store_o = [&](const f32x4 (&o)[DT], float l, bf16 *ob, int ti) {
        const int row = (wid + NW * ti) * 16 + l15;
        const float inv = __builtin_amdgcn_rcpf(l);  // (l >= 1; the same reciprocal as attn_frame3_kernel: patch rows stay bit-identical between the two)
        bf16 *op = ob + (int64_t)(row < S ? row : 0) * a.ldo + ((g & 1) * 16 + (g >> 1) * 8);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            union { bf16x4 v; int w[2]; } e, odd, give, got;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e.v[r] = (bf16)(o[2 * m][r] * inv);
                odd.v[r] = (bf16)(o[2 * m + 1][r] * inv);
            }
            give.v = (g & 1) ? e.v : odd.v;  // what the partner lane stores
            got.w[0] = __shfl_xor(give.w[0], 16, 64);
            got.w[1] = __shfl_xor(give.w[1], 16, 64);
            const bf16x4 mine = (g & 1) ? odd.v : e.v;
            const bf16x4 lo = (g & 1) ? got.v : mine, hi = (g & 1) ? mine : got.v;
            const int d0 = 32 * m + (g & 1) * 16 + (g >> 1) * 8;
            if (row < S && d0 + 8 <= HD)
                *reinterpret_cast<bf16x8 *>(op + 32 * m) = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };

    int pair = blockIdx.x;
    if (pair >= npairs) return;
    {
        const int b = pair / a.heads, h = pair - b * a.heads;
        stage(a.k + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs, 0);
    }
    const bool two_tiles = wid + NW < NT;
    load_q(pair, 0, qf[0]);
    if (two_tiles) load_q(pair, 1, qf[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool ts_on = (a.dbg & 256) && blockIdx.x == 0 && lane == 0;
#define FA_TS(ev)                                                                                  \
    do {                                                                                           \
        if (ts_on && it < 8) g_attn_ts[(wid * 8 + it) * 16 + (ev)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
    // one query tile: S^T (all keys, in registers) -> exact softmax -> O^T
    auto s_phase = [&](unsigned kaddr, const bf16x8 (&q)[KS], f32x4 (&sc)[NT]) {
        bf16x4 kf[2][2 * KS];  // double buffered: the reads of tile t + 1 are in flight under the MFMAs of tile t
        s_issue(kaddr, kf[0], 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) {
                s_issue(kaddr, kf[(t + 1) & 1], t + 1);
                s_mma(q, kf[t & 1], sc, t, std::integral_constant<int, 2 * KS>{});
            } else {
                s_mma(q, kf[t & 1], sc, t, std::integral_constant<int, 0>{});
            }
        }
    };
    auto softmax = [&](f32x4 (&sc)[NT], bf16x4 (&pr)[NT]) -> float {
        const float nm = row_offset(sc);
        f32x2 la = {0.f, 0.f}, lb = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; t += 2) sm_tiles(sc, pr, t, nm, la, lb);
        return row_sum(la, lb);
    };
    auto pv_phase = [&](const bf16x4 (&pr)[NT], float l, bf16 *ob, int ti) {
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x4 vf[2 * DT];
#pragma unroll
        for (int k2 = 0; k2 < NKS; ++k2) {
            pv_issue(vf, k2);
            pv_mma(o, pr, vf, k2, std::integral_constant<int, 0>{});
        }
        store_o(o, l, ob, ti);
    };
    auto run = [&](auto two_tag) {
        constexpr bool TWO = decltype(two_tag)::value;  // this wave owns query tiles (w, w + 9) or only w
        for (int it = 0; pair < npairs; pair += gridDim.x, ++it) {
            const int cur = it & 1;
            const int b = pair / a.heads, h = pair - b * a.heads;
            const int pn = pair + gridDim.x;
            const bool more = pn < npairs;
            FA_TS(0);
            // My pieces of K(pair) were issued after the previous pair's second barrier.  Loads younger than them: the Q
            // loads of the tiles I own (6 each), so "at most that many outstanding" implies the pieces have landed.
            if (it > 0) {
                if (TWO) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            // past this barrier: K(pair) is complete in LDS, every wave is done with V(previous pair) and K(previous pair)
            FA_BARRIER();
            FA_TS(1);
            stage(a.v + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs, 2);
            const unsigned kaddr = (unsigned)(uintptr_t)lds + cur * BUF + kbase;  // LDS byte address of this lane's K row
            bf16 *ob = a.o + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs;
            f32x4 sc[NT];
            bf16x4 pr[NT];
            s_phase(kaddr, qf[0], sc);
            FA_TS(2);
            float l = softmax(sc, pr);
            FA_TS(3);
            // V(pair) must be complete in LDS for every wave (nothing younger than its pieces is in flight)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FA_TS(4);
            FA_BARRIER();
            FA_TS(5);
            // K of the next pair streams in under the rest of this pair (its buffer is free since the first barrier)
            if (more) {
                const int bn = pn / a.heads, hn = pn - bn * a.heads;
                stage(a.k + (int64_t)bn * a.k_bs + (int64_t)hn * a.k_hs, cur ^ 1);
            }
            pv_phase(pr, l, ob, 0);
            if (more) load_q(pn, 0, qf[0]);  // consumed a pair from now
            FA_TS(6);
            if constexpr (TWO) {
                s_phase(kaddr, qf[1], sc);
                FA_TS(7);
                l = softmax(sc, pr);
                FA_TS(8);
                pv_phase(pr, l, ob, 1);
                if (more) load_q(pn, 1, qf[1]);
                FA_TS(9);
            }
        }
    };
    if (two_tiles) run(std::true_type{});
    else run(std::false_type{});
#undef FA_TS
#undef FA_BARRIER
}

template <int HD, int NT>
int launch_attn_frame(const AttnArgs &a, hipStream_t s) {
    constexpr int CH = HD / 8;
    constexpr int NPIECE = (NT * 16 * CH + (12 - CH) + 63) / 64;
    constexpr int smem = 3 * NPIECE * 1024;
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_frame_kernel<HD, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int dev = 0;
        EILEV_HIP_CHECK(hipGetDevice(&dev));
        EILEV_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_set = true;
    }
    const int npairs = a.batch * a.heads;
    const int ncu = eilev_grid_cus() < num_cu ? eilev_grid_cus() : num_cu;
    const int grid = npairs < ncu ? npairs : ncu;
    hipLaunchKernelGGL((attn_frame_kernel<HD, NT>), dim3(grid), dim3(576), smem, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

#include "attn_frame3.h"

template <int NWQ, int DB = 3, bool REL = false>
int launch_attn_v2(const AttnArgs &a, hipStream_t s) {
    size_t smem = attn_v2_rel_offset(NWQ, a.hd) + (REL ? (size_t)(a.rel_n + 2 * ATTN_V2_REL_SLACK) * sizeof(float) : 0);
    const dim3 grid((a.sq + 32 * NWQ - 1) / (32 * NWQ), a.heads, a.batch), block(64 * NWQ);
    if (smem > 64 * 1024) {
        static bool attr = false;
        if (!attr) {
            EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_prefill_v2_kernel<NWQ, DB, REL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
    }
    hipLaunchKernelGGL((attn_prefill_v2_kernel<NWQ, DB, REL>), grid, block, smem, s, a);
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
    if (a.hm) {  // the scattered q|k|v blocks (AttnArgs::hm): attn_frame3_kernel's HM form only
        if (a.hd != 88 || a.sq != 257 || a.skv != 257 || a.causal || a.key_mask || a.rel_tab || a.drop_thr || (a.q_hs & 7) || (a.q_bs & 7) ||
            a.q_hs != a.k_hs || a.q_hs != a.v_hs || (int64_t)a.sq * a.hd * 2 >= 0x7fff0000ll)
            return EILEV_E_UNSUPPORTED;
        return launch_attn_frame3<88, 17, true>(a, s);
    }
    // whole-frame ViT attention: S = 257 (17 tiles of 16), hd = 88, no mask, q / k / v rows of one fused buffer
    if (!g_attn_force_v1 && !(a.dbg & 4) && !a.rel_tab && !a.drop_thr && a.hd == 88 && a.sq == a.skv && a.sq > 256 && a.sq <= 272 && !a.causal && !a.key_mask &&
        a.ldk == a.ldv && !(a.ldq & 3) && (int64_t)a.sq * a.ldk * 2 < 0x7fff0000ll) {
        // >= 512 frames: two wave groups one phase apart (attn_frame3.h; 512: with phase stamps): 2-13 % faster at 544 / 1088 frames over
        // six boxes (profiles/r03_attn_frame3.log), slower below ~384 frames (its slots are longer: more exposed at the start and the end
        // of a workgroup's walk).  Probe flag 16 forces it, 32 forbids it.
#ifdef EILEV_PROBES
        if (a.dbg & 1024) {  // TIMING PROBE (wrong results): q / k / v read as if stored head-major ([frame][head][token][88]: an image is 45 KB contiguous)
            a.ldq = a.ldk = a.ldv = a.hd;
            a.q_hs = a.k_hs = a.v_hs = (int64_t)a.sq * a.hd;
            a.k = a.q + (int64_t)a.sq * a.heads * a.hd;  // three planes per frame inside the fused q|k|v rows: the same bytes are read, none twice
            a.v = a.q + 2 * (int64_t)a.sq * a.heads * a.hd;
        }
#endif
        if (a.sq == 257 && !(a.dbg & 32) && ((a.dbg & (16 | 512)) || a.batch >= 512)) return launch_attn_frame3<88, 17>(a, s);
        return launch_attn_frame<88, 17>(a, s);
    }
    // hd = 64 with >= 128 query rows (the flan-t5 encoder: L = 960, relative position bias; also its bias-free long forms): round 5
    // (its bias lookup does not clamp: the table must cover every distance of the launch, key - query - (skv - sq) in [-(skv - 1), sq - 1])
    if (!g_attn_force_v1 && !a.drop_thr && a.hd == 64 && a.sq >= 128 && a.skv >= 64 && a.scale > 0.0f &&
        (!a.rel_tab || (a.rel_n <= ATTN_V2_REL_MAX && a.rel_off >= a.skv - 1 && a.rel_off + a.sq <= a.rel_n))) {
        const int qt = (a.sq + 31) / 32;
        if (a.rel_tab) return qt >= 5 ? launch_attn_v2<8, 2, true>(a, s) : launch_attn_v2<4, 2, true>(a, s);
        return qt >= 5 ? launch_attn_v2<8, 2, false>(a, s) : launch_attn_v2<4, 2, false>(a, s);
    }
    // hd = 128 (OPT-6.7B prefill), >= 64 query rows: round 6
    if (!g_attn_force_v1 && !a.rel_tab && !a.drop_thr && a.hd == 128 && a.sq >= 64 && a.skv >= 64) {
        const int qt = (a.sq + 31) / 32;
#ifdef EILEV_PROBES
        if (a.dbg & 64) return launch_attn_v2<4, 4>(a, s);  // probe: 4 waves per workgroup (512 registers per wave, two blocks in flight)
#endif
        return qt >= 5 ? launch_attn_v2<8, 4>(a, s) : launch_attn_v2<4, 4>(a, s);
    }
    if (!g_attn_force_v1 && !a.rel_tab && !a.drop_thr && (a.hd == 80 || a.hd == 88 || a.hd == 72) && a.skv >= 32) {
        const int qt = (a.sq + 31) / 32;
        if (qt == 9 || qt > 16) return (qt == 9) ? launch_attn_v2<9>(a, s) : launch_attn_v2<8>(a, s);
        if (qt >= 5) return launch_attn_v2<8>(a, s);
        if (qt >= 3) return launch_attn_v2<4>(a, s);
        return launch_attn_v2<2>(a, s);
    }
    const dim3 grid((a.sq + 63) / 64, a.heads, a.batch), block(256);
    if (a.drop_thr) {
        if (a.hd <= 64) hipLaunchKernelGGL((attn_prefill_kernel<64, true>), grid, block, 0, s, a);
        else if (a.hd <= 96) hipLaunchKernelGGL((attn_prefill_kernel<96, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attn_prefill_kernel<128, true>), grid, block, 0, s, a);
    } else if (a.hd <= 64) hipLaunchKernelGGL(attn_prefill_kernel<64>, grid, block, 0, s, a);
    else if (a.hd <= 96) hipLaunchKernelGGL(attn_prefill_kernel<96>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(attn_prefill_kernel<128>, grid, block, 0, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
