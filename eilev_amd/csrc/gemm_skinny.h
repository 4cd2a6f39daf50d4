// gemm_skinny.h — weight-streaming kernels for M <= 32 rows (the decode step: HBM-bound), incl. the fp8-weight forms and the split-K reduce.
#pragma once
#include "gemm_common.h"

namespace {

// ---- skinny GEMM (M <= 16): weight-streaming, one 16-row block of W per workgroup ----------------
// grid = (ceil(N/16), KS).  Each of the 4 waves owns a contiguous slice of this workgroup's K range;
// per K-step of 32 a lane loads 16 B of W (row n0 + lane%16, k-group lane/16) and 16 B of A (batch row
// lane%16, zero beyond M) and issues one 16x16x32 MFMA; loads are issued 8 deep.  Wave partials are summed through LDS in a
// fixed order (deterministic).  KS == 1: epilogue applied here; KS > 1: fp32 partials to `part`
// ([KS][16][N]) for skinny_reduce_kernel.
struct SkinnyArgs {
    GemmArgs g;
    float *part;
    int ks;
    int mr;  // rows per split-K partial: 16 (M <= 16) or 32
};

__device__ __forceinline__ void skinny_epilogue(const GemmArgs &g, int row, int col, float v) {
    if (g.wscale) v *= g.wscale[col];
    if (g.bias) v += (float)g.bias[col];
    if (col < g.scale_cols) v *= g.scale;
    if (g.epi == 1) v = gelu_erf(v);
    else if (g.epi == 2) v = fmaxf(v, 0.0f);
    if (g.resid) v += (float)g.resid[(int64_t)row * g.ldr + col];
    if (g.out_f32) reinterpret_cast<float *>(g.C)[(int64_t)row * g.ldc + col] = v;
    else reinterpret_cast<bf16 *>(g.C)[(int64_t)row * g.ldc + col] = (bf16)v;
}

__global__ __launch_bounds__(256) void gemm_skinny_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ float red[4][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    int wrow = n0 + l15;
    wrow = wrow < g.N ? wrow : g.N - 1;
    // K range of this workgroup, then of this wave, in units of 32
    const int ksteps = (g.K + 31) / 32;
    const int per_wg = (ksteps + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ksteps, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 3) / 4;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);

    const bf16 *wp = g.W + (int64_t)wrow * g.ldw + lg * 8;
    const bf16 *ap = g.A + (int64_t)(l15 < g.M ? l15 : 0) * g.lda + lg * 8;
    const bool arow = l15 < g.M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = beg;
    // 8 independent 16-byte weight loads in flight per lane (the K tail of the matrix never lands here:
    // K % 256 == 0 for every decode shape; the remainder loop below handles the general case)
    for (; s + 8 <= end && (s + 8) * 32 <= g.K; s += 8) {
        bf16x8 wv[8], av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(wp + (s + u) * 32));
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = arow ? *reinterpret_cast<const bf16x8 *>(ap + (s + u) * 32) : zero8();
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u], wv[u], acc, 0, 0, 0);
    }
    for (; s < end; ++s) {
        const int k = s * 32;
        const bool kin = (k + lg * 8) < g.K;
        bf16x8 wv = kin ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(wp + k)) : zero8();
        bf16x8 av = (kin && arow) ? *reinterpret_cast<const bf16x8 *>(ap + k) : zero8();
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][lane][r] = acc[r];
    __syncthreads();
    if (wid == 0) {
        const int col = n0 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[w][lane][r];
            const int row = lg * 4 + r;
            if (row < g.M && col < g.N) {
                if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * 16 + row) * g.N + col] = v;
            }
        }
    }
}

// Skinny kernel, DMA-staged variant (K % 256 == 0): the 16-row weight block is streamed through per-wave LDS
// buffers with global_load_lds_dwordx4 so that every load instruction covers two whole 512-byte row segments
// (fully coalesced; the direct MFMA-layout loads above touch 64 separate 16-byte pieces per instruction).
// Each wave runs its own 2-deep pipeline on a private 2 x 8 KiB region: DMA(t+1) is issued before the
// counted s_waitcnt vmcnt(8) that retires DMA(t); no workgroup barrier in the K loop.  16-byte chunk c of
// row r is stored at chunk c ^ (r & 15) (swizzle applied on the source address) so the ds_read_b128 fragment
// reads of 16 rows x 512-byte stride are bank-conflict-free.
// MB = 1: M <= 16; MB = 2: M <= 32 (two 16-row activation tiles share every weight fragment: the weight stream, which
// bounds the kernel, is read once for twice the rows)
// PRE: a wave has at most 3 K-tiles (every decode shape): ALL its activation fragments are loaded up front (one exposed L2 round trip
// instead of one per tile: the per-tile loads were 40 % of the kernel) and the tile loop is three static iterations.  Same summation
// order as the rolled form.
template <int MB, bool PRE>
__global__ __launch_bounds__(256) void gemm_skinny_dma_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ __attribute__((aligned(16))) char wbuf[4][2][8192];
    __shared__ float red[4][MB][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    // K range of this workgroup / wave in units of 256 (= one 16 x 256 tile = 8 MFMA steps)
    const int ktiles = g.K / 256;
    const int per_wg = (ktiles + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ktiles, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 3) / 4;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);

    // DMA source: piece i (rows 2i, 2i+1): lane p -> row 2i + p/32, LDS slot p%32 <- global chunk slot ^ (row & 15)
    const int prow = lane >> 5, pslot = lane & 31;
    const bf16 *src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 2 * i + prow;
        int gr = n0 + row;
        gr = gr < g.N ? gr : g.N - 1;
        src[i] = g.W + (int64_t)gr * g.ldw + ((pslot ^ (row & 15)) << 3);
    }
    auto stage_in = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_void *)(src[i] + t * 256), (lds_void *)(&wbuf[wid][buf][i * 1024]), 16, 0, 0);
    };
    const bf16 *ap[MB];
    bool arow[MB];
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int r = mb * 16 + l15;
        arow[mb] = r < g.M;
        ap[mb] = g.A + (int64_t)(arow[mb] ? r : 0) * g.lda + lg * 8;
        acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (PRE) {
        bf16x8 av[MB][24];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
            if (beg + tt < end) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        av[mb][tt * 8 + u] = arow[mb] ? *reinterpret_cast<const bf16x8 *>(ap[mb] + (beg + tt) * 256 + u * 32) : zero8();
            }
        if (beg < end) stage_in(0, beg);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int t = beg + tt;
            if (t < end) {
                if (t + 1 < end) {
                    stage_in((tt & 1) ^ 1, t + 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile t and every activation fragment (older) have landed
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                const char *wb = &wbuf[wid][tt & 1][0] + l15 * 512;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wb + (((u * 4 + lg) ^ l15) << 4));
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][tt * 8 + u], wv, acc[mb], 0, 0, 0);
                }
            }
        }
    } else {
    if (beg < end) stage_in(0, beg);
        for (int t = beg; t < end; ++t) {
            const int cur = (t - beg) & 1;
            bf16x8 av[MB][8];
    #pragma unroll
            for (int mb = 0; mb < MB; ++mb)
    #pragma unroll
                for (int u = 0; u < 8; ++u) av[mb][u] = arow[mb] ? *reinterpret_cast<const bf16x8 *>(ap[mb] + t * 256 + u * 32) : zero8();
            if (t + 1 < end) {
                stage_in(cur ^ 1, t + 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the 8 pieces of tile t (older than the 8 just issued)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const char *wb = &wbuf[wid][cur][0] + l15 * 512;
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wb + (((u * 4 + lg) ^ l15) << 4));
    #pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][u], wv, acc[mb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid][mb][lane][r] = acc[mb][r];
    __syncthreads();
    if (wid < MB) {  // wave mb finishes row tile mb
        const int col = n0 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[w][wid][lane][r];
            const int row = wid * 16 + lg * 4 + r;
            if (row < g.M && col < g.N) {
                if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * (16 * MB) + row) * g.N + col] = v;
            }
        }
    }
}

// Skinny kernel for fp8 (OCP e4m3) weights: the same per-wave 2-deep LDS-DMA pipeline over 16-row x 256-K tiles, which are now
// 4 KiB (half the bytes of the bound stream).  Piece i = rows 4i .. 4i+3 x 256 B; 16-byte chunk c of row r is stored at chunk
// c ^ (r & 15).  A lane's 8 weights of an MFMA k-step are 8 bytes: v_cvt_pk_f32_fp8 + one v_perm_b32 per pair make the bf16
// fragment (every e4m3 value is exactly a bf16 value); the per-channel scale is applied to the fp32 sum in the epilogue.
// ---- weight streaming with the activations held in registers across several weight blocks (round 2) ---------------------------
// gemm_skinny_dma_kernel<MB, true> loads a wave's activation fragments (MB x 16 rows x its K slice, up to 192 VGPRs) and then streams
// ONE 16-row weight block (at most 3 tiles of 8 KB per wave): every workgroup pays 2 x its weight bytes in activation loads from L2
// and never reaches a steady stream (measured at batch 32: 2.1 TB/s).  Here a workgroup keeps the SAME activation fragments for NB
// consecutive weight blocks: the loads are paid once per NB blocks and each wave streams NB x its tiles through a 3-deep LDS-DMA
// ring (24 KB in flight per wave).  Partial sums of the 4 waves (K quarters) meet in LDS per block, in a fixed order.
template <int MB, int NB, int ST = 3>
__global__ __launch_bounds__(256) void gemm_skinny_nb_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    extern __shared__ __attribute__((aligned(16))) char smem_nb[];
    char(*wbuf)[ST][8192] = reinterpret_cast<char(*)[ST][8192]>(smem_nb);                                  // [4 waves][ST stages][8 KB]
    float(*red)[4][MB][64][4] = reinterpret_cast<float(*)[4][MB][64][4]>(smem_nb + 4 * ST * 8192);       // [2 (ping-pong)][4 waves][MB][64][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int nblocks = (g.N + 15) / 16;
    const int b0 = blockIdx.x * NB;
    const int ktiles = g.K / 256;
    const int per_wg = (ktiles + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ktiles, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 3) / 4;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);
    const int nt = max(end - beg, 0);  // tiles of this wave per weight block (<= 3)

    const int prow = lane >> 5, pslot = lane & 31;
    // piece i of block j: rows 2i, 2i+1 of the block; lane p -> row 2i + p/32, LDS slot p%32 <- global chunk slot ^ (row & 15)
    auto src = [&](int j, int i) {
        const int row = 2 * i + prow;
        int gr = (b0 + j) * 16 + row;
        gr = gr < g.N ? gr : g.N - 1;
        return g.W + (int64_t)gr * g.ldw + ((pslot ^ (row & 15)) << 3);
    };
    auto stage_in = [&](int stage, int j, int t) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_void *)(src(j, i) + t * 256), (lds_void *)(&wbuf[wid][stage][i * 1024]), 16, 0, 0);
    };
    // activation fragments of this wave's K slice, once
    bf16x8 av[MB][24];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
        if (tt < nt) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int r = mb * 16 + l15;
                const bf16 *ap = g.A + (int64_t)(r < g.M ? r : 0) * g.lda + lg * 8 + (beg + tt) * 256;
#pragma unroll
                for (int u = 0; u < 8; ++u) av[mb][tt * 8 + u] = r < g.M ? *reinterpret_cast<const bf16x8 *>(ap + u * 32) : zero8();
            }
        }
    const int nbl = min(NB, nblocks - b0);  // weight blocks of this workgroup
    const int total = nbl * nt;             // tiles this wave streams: flat index f = j * nt + tt
    // prologue: ST - 1 tiles in flight
    if (total > 0) stage_in(0, 0, beg);
    if (ST > 2 && total > 1) stage_in(1, 1 / nt, beg + 1 % nt);
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int f = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (j < nbl) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                if (tt < nt) {
                    // issue tile f + ST - 1, then wait until tile f has landed: the 8 pieces of each younger tile stay in flight
                    constexpr int AH = ST - 1;
                    if (f + AH < total) {
                        stage_in((f + AH) % ST, (f + AH) / nt, beg + (f + AH) % nt);
                        if (AH == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    } else if (AH == 2 && f + 1 < total) {
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const char *wb = &wbuf[wid][f % ST][0] + l15 * 512;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wb + (((u * 4 + lg) ^ l15) << 4));
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][tt * 8 + u], wv, acc[mb], 0, 0, 0);
                    }
                    ++f;
                }
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[j & 1][wid][mb][lane][r] = acc[mb][r];
            acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // one barrier per weight block; the partials ping-pong between two LDS regions, so the waves that finish block j (below) are
        // done before anybody writes region j & 1 again (after the barrier of block j + 1)
        __syncthreads();
        if (j < nbl) {  // (round 4) the MB x 4 (row tile, register) pairs of this block dealt over the four waves: the 4 K-quarter partials in a fixed order
            const int col = (b0 + j) * 16 + l15;
#pragma unroll
            for (int cb = 0; cb < MB; ++cb) {
                const int mb = (wid * MB + cb) >> 2, r = (wid * MB + cb) & 3;
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[j & 1][w][mb][lane][r];
                const int row = mb * 16 + lg * 4 + r;
                if (row < g.M && col < g.N) {
                    if (a.ks == 1) skinny_epilogue(g, row, col, v);
                    else a.part[((int64_t)blockIdx.y * (16 * MB) + row) * g.N + col] = v;
                }
            }
        }
    }
}

// (round 4, measured and removed: gemm_skinny5_kernel — one 16-row weight block x 5 waves x 2 K tiles per workgroup, every weight tile and
// activation fragment requested up front, no ring: 24.2 us per q|k|v / fc1 / fc2 launch against 19.8 us for the kernel above, 5.53 vs 5.40
// ms/token at batch 32 (profiles/r04_skinny5_rejected_*).  The ring depth was not the limit: each workgroup re-reads the 160 KB of
// activations from L2, and with one block per workgroup that is 2-3x the weight bytes entering every CU.)
// ---- round 4: 17..32 rows, ONE workgroup per CU, the activations loaded once per CU -----------------------------------------------------
// What bounds the kernels above at batch 32 is not the weight stream but the activations: every workgroup re-reads the 32 x K rows from
// L2 (160 KB at K = 2560) for 16-32 weight rows (80-160 KB) — 2-3x the weight bytes enter each CU (measured twice: 2 / 4 / 8 blocks per
// workgroup in round 2, and round 4's one-block-per-workgroup variant with every load up front, which was SLOWER: 24 vs 20 us).  Here the
// grid is the CUs.  A 512-thread workgroup splits K over its 8 waves (wave w: K / (8 ks) columns = KS k-steps of 32): its slice of the
// 32 rows is 80 VGPRs of MFMA A fragments, loaded ONCE; the CU then walks its share of the 16-row weight blocks, every wave streaming
// its K slice of a block straight into registers as B fragments (16 rows x 64 B per instruction, non-temporal) through a ring of RB
// blocks (30 KB per wave, 240 KB per CU in flight), 2 KS MFMAs per block, the 8 K-slice partials summed through LDS in a fixed order
// (ping-pong buffers, one barrier per block).  blockIdx.y = K split across CUs where the 8-wave slice would not fit the registers (fc2:
// K = 10240 -> 4) or the matrix has fewer blocks than CUs (out_proj): partials + reduce_ln_kernel as before.
template <int MB, int KS, int RB>
__global__ __launch_bounds__(512) void gemm_rows32_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ float red[2][8][MB][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int G = gridDim.x;
    // this CU's weight ROWS of K split blockIdx.y: [r0, r1) — N dealt row by row (round 5), not in 16-row blocks: fc1's 640 blocks on 256 CUs
    // were 3 blocks on one half of the chip and 2 on the other (the launch takes what the 3-block CUs take: +20 %); now every CU has 40 rows
    // = 2 blocks + 8 rows, the lanes of the missing rows requesting nothing.
    const int per = g.N / G, rem = g.N % G;
    const int r0 = blockIdx.x * per + min((int)blockIdx.x, rem), r1 = r0 + per + ((int)blockIdx.x < rem ? 1 : 0);
    const int nblk = (r1 - r0 + 15) >> 4;
    if (nblk <= 0) return;  // (uniform per workgroup)
    const int k0 = (blockIdx.y * 8 + wid) * (KS * 32);
    bf16x8 wv[RB][KS];
    unsigned short bv[RB];  // bias of the lane's output column (raw bf16 bits), requested WITH the block's weights: a load in the epilogue put one
                            // global round trip (~1.5 us) on the critical path of every block (measured without any operand loads: 7.5 us per
                            // q|k|v launch, 20.7 for the lm_head).  Round 5: requested AFTER the block's weights and kept unconverted until the
                            // epilogue — the bf16 -> float conversion at the request made hipcc wait (vmcnt(0)) for everything in flight before
                            // it issued the next block's loads: the "ring" held one block at a time.
    const bool plain_epi = a.ks == 1 && !g.wscale && !g.resid;
    const bool has_bias = plain_epi && g.bias;
    auto load_block = [&](int j, auto buf_c) {
        constexpr int B = decltype(buf_c)::value;
        const int gr = r0 + j * 16 + l15;
        if (gr < r1) {  // (a lane past the CU's rows keeps whatever its registers hold: that output column is never stored)
            const bf16 *wp = g.W + (int64_t)gr * g.ldw + k0 + lg * 8;
            int ustride = 32;
            if (g.Wp) {  // stream layout (round 5, launch_stream_pack below): the 16 x 32 fragment an instruction reads is one contiguous piece
                const int nv = min(16, r1 - (r0 + j * 16));
                wp = g.Wp + ((int64_t)r0 + (int64_t)j * 16) * g.K + (int64_t)((blockIdx.y * 8 + wid) * KS) * (nv * 32) + (l15 * 4 + lg) * 8;
                ustride = nv * 32;
            }
#pragma unroll
            for (int u = 0; u < KS; ++u) {
#ifdef ROWS32_NOW
                wv[B][u] = zero8();
#else
                wv[B][u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(wp + u * ustride));
#endif
            }
            bv[B] = reinterpret_cast<const unsigned short *>(has_bias ? g.bias : g.W)[has_bias ? gr : 0];
        }
    };
    bf16x8 av[MB][KS];
    auto load_x = [&]() {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int r = mb * 16 + l15;
            const bf16 *ap = g.A + (int64_t)(r < g.M ? r : 0) * g.lda + k0 + lg * 8;  // rows past M: row 0 again (their outputs are never stored)
            int ustride = 32;
            if (g.a_frag) {  // the rows in the row-block layout (common.h frag32_index): this load reads one contiguous kilobyte
                ap = g.A + (int64_t)(k0 / 32) * 1024 + mb * 512 + (l15 * 4 + lg) * 8;
                ustride = 1024;
            }
#pragma unroll
            for (int u = 0; u < KS; ++u) {
#ifdef ROWS32_NOX
                av[mb][u] = zero8();
#else
                av[mb][u] = *reinterpret_cast<const bf16x8 *>(ap + u * ustride);
#endif
            }
        }
    };
    auto consume = [&](int j, auto buf_c) {
        constexpr int B = decltype(buf_c)::value;
        f32x4 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][u], wv[B][u], acc[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[j & 1][wid][mb][lane][r] = acc[mb][r];
        __syncthreads();  // the partials of block j are complete; region (j + 1) & 1 is free again (its readers passed the previous barrier's successor)
        // every wave finishes ONE (row tile, accumulator register) pair of the block — 8 LDS reads and one store each — instead of waves
        // 0 .. MB - 1 finishing four: the next block's barrier waits for the finishers (same-box step 4.95 -> 4.79 ms/token).  (All of a CU's
        // blocks behind ONE barrier — MFMAs of every block first, then every reduction — was measured too: 5.31, the longer code spills.)
        if ((wid >> 2) < MB) {
            const int mb = wid >> 2, r = wid & 3, col = r0 + j * 16 + l15;
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[j & 1][w][mb][lane][r];
            const int row = mb * 16 + lg * 4 + r;
            if (row < g.M && col < r1) {
                if (plain_epi) {  // bias (prefetched) + activation + store: no load here
                    if (has_bias) v += __uint_as_float((unsigned)bv[B] << 16);
                    if (col < g.scale_cols) v *= g.scale;
                    if (g.epi == 1) v = gelu_erf(v);
                    else if (g.epi == 2) v = fmaxf(v, 0.0f);
                    if (g.out_f32) reinterpret_cast<float *>(g.C)[(int64_t)row * g.ldc + col] = v;
                    else reinterpret_cast<bf16 *>(g.C)[g.c_frag ? frag32_index(row, col) : (int64_t)row * g.ldc + col] = (bf16)v;
                } else if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * (16 * MB) + row) * g.N + col] = v;
            }
        }
    };
    // Request order (round 5): the 32 rows FIRST.  Loads return in order, so with the rows queued behind the ring (r4) the first block's
    // MFMAs waited for every weight byte of the workgroup: no block's reduction overlapped the stream (4.77 -> 4.73 ms / token at batch 32).
    if (nblk >= 2 * RB) {  // long: the lm_head (12 blocks per CU) — ring with a branch-free steady loop
        load_x();
        static_for<RB>([&](auto j_c) { load_block(decltype(j_c)::value, j_c); });
        int base = 0;
        for (; base + 2 * RB <= nblk; base += RB)
            static_for<RB>([&](auto j_c) {
                consume(base + decltype(j_c)::value, j_c);
                load_block(base + decltype(j_c)::value + RB, j_c);
            });
        static_for<RB>([&](auto j_c) {
            consume(base + decltype(j_c)::value, j_c);
            if (base + decltype(j_c)::value + RB < nblk) load_block(base + decltype(j_c)::value + RB, j_c);
        });
        base += RB;
        static_for<RB>([&](auto j_c) {
            if (base + decltype(j_c)::value < nblk) consume(base + decltype(j_c)::value, j_c);
        });
    } else {  // short: the block matrices (1-3 blocks per CU): everything requested up front
        load_x();
        static_for<RB>([&](auto j_c) {
            if (decltype(j_c)::value < nblk) load_block(decltype(j_c)::value, j_c);
        });
        static_for<RB>([&](auto j_c) {
            constexpr int J = decltype(j_c)::value;
            if (J < nblk) {
                consume(J, j_c);
                if (J + RB < nblk) load_block(J + RB, j_c);
            }
        });
        static_for<RB>([&](auto j_c) {
            if (decltype(j_c)::value + RB < nblk) consume(decltype(j_c)::value + RB, j_c);
        });
    }
}

static int skinny_n_cu() { return eilev_num_cu(); }

// Shape of a gemm_rows32_kernel launch, a function of (N, K, CUs) alone — the stream layout of a weight matrix is packed for it.
// K split: the 8-wave K slice must be 5 or 10 k-steps of 32 (8: K = 2048, flan-t5, round 5); more splits while the matrix has fewer 16-row
// blocks than CUs.  grid_x workgroups share the N weight rows row by row; with a K split the grid is still one workgroup per CU.
static bool rows32_shape(int N, int K, int n_cu, int &ks, int &ksteps, int &grid_x, bool first_fit = false) {
    if (K % 256 || N < 1) return false;
    const int nb = (N + 15) / 16, per_wave = K / 256;  // k-steps of 32 per wave without a split
    int k5 = 0;
    for (int c = 1; c <= 8; c *= 2)
        if (per_wave % c == 0 && (per_wave / c == 10 || per_wave / c == 5 || (per_wave / c == 8 && c == 1))) {
            k5 = c;
            if (nb * c >= n_cu || per_wave / c == 5 || first_fit) break;
        }
    if (!k5) return false;
    ks = k5;
    ksteps = per_wave / k5;
    const int cus = k5 > 1 ? (n_cu / k5 > 0 ? n_cu / k5 : 1) : n_cu;
    grid_x = nb < cus ? nb : cus;
    return true;
}

// the weight rows of workgroup x of grid_x: [r0, r0 + cnt)
__host__ __device__ static inline void rows32_rows(int N, int grid_x, int x, int &r0, int &cnt) {
    const int per = N / grid_x, rem = N % grid_x;
    r0 = x * per + (x < rem ? x : rem);
    cnt = per + (x < rem ? 1 : 0);
}

// Stream layout of W [N][K] for gemm_rows32_kernel at grid_x workgroups (same bytes, other order): the rows [r0, r0 + cnt) of workgroup x
// stay together at element r0 K; inside, 16-row block j (nv = min(16, cnt - 16 j) rows) at + 16 j K holds for every k-step of 32 the
// nv x 32 fragment as ONE piece of nv x 64 bytes in lane order — element (row l15, chunk lg) at ((l15 4 + lg) 8) — so that a wave's
// load instruction reads 1 KB contiguous instead of 16 segments of 64 bytes 5 KB apart (decode at batch 32: 4.35 -> 4.01 ms / token).
__global__ __launch_bounds__(256) void stream_pack_kernel(const bf16 *__restrict__ W, bf16 *__restrict__ out, int N, int K, int grid_x) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int kc = K >> 3;
    if (idx >= (int64_t)N * kc) return;
    const int n = (int)(idx / kc), k8 = (int)(idx - (int64_t)n * kc);
    const int per = N / grid_x, rem = N % grid_x, split = rem * (per + 1);
    const int x = n < split ? n / (per + 1) : rem + (n - split) / per;
    int r0, cnt;
    rows32_rows(N, grid_x, x, r0, cnt);
    const int j = (n - r0) >> 4, l15 = (n - r0) & 15, nv = min(16, cnt - 16 * j);
    const int kstep = k8 >> 2, lg = k8 & 3;
    const int64_t dst = ((int64_t)r0 + (int64_t)j * 16) * K + (int64_t)kstep * (nv * 32) + (l15 * 4 + lg) * 8;
    *reinterpret_cast<bf16x8 *>(out + dst) = *reinterpret_cast<const bf16x8 *>(W + (int64_t)n * K + (int64_t)k8 * 8);
}

static bool rows32_plan(const GemmArgs &g, int nb, int n_cu, SkinnyArgs &a, int &ks, int &ksteps, int &grid_x) {
    int k5 = 0;
    if (!rows32_shape(g.N, g.K, n_cu, k5, ksteps, grid_x, (g.dbg & 134217728) != 0)) return false;
    if (g.c_frag && (k5 > 1 || g.out_f32 || g.wscale || g.resid || (g.N & 31))) return false;  // c_frag: the plain bf16 epilogue only
    if (g.ln_frag && (k5 == 1 || !g.ln_out)) return false;                                      // ln_frag: the fused split-K reduce only
    if (g.a_frag && g.M > 32) return false;
    const int per_wave = g.K / 256;
    if (!k5) return false;
    // fewer blocks than CUs (out_proj: 160) and no further split: the unsplit form leaves a third of the chip idle and needs a separate LayerNorm
    // launch after it (the split-K reduce produces the LayerNorm for free) — those shapes keep the round-2 / round-3 kernels
    if (k5 == 1 && nb < n_cu && g.ln_out) return false;
#ifndef EILEV_ROWS32_SMALLN
    if (per_wave == 8 && nb < n_cu) return false;  // K = 2048 with fewer blocks than CUs (T5 o / cross q / cross o, N = 2048): the split-K kernels
#endif
    // The split-K forms (out_proj: 2 x 128 workgroups of 20 rows; fc2: 4 x 64 of 40 rows) lost to the round-3 kernels while the rows were dealt in
    // 16-row blocks (r4); with the rows dealt one by one every CU streams the same bytes and they win: 4.65 -> 4.50 ms / token at batch 32 (r5).
    // probe flag 1 << 27: the round-3 kernels for these shapes
    if (k5 > 1 && (g.dbg & 134217728)) return false;
    if (k5 > 1 && (!g.scratch || (size_t)k5 * a.mr * g.N * sizeof(float) > g.scratch_bytes)) return false;
    ks = k5;
    a.ks = k5;
    return true;
}

template <int MB, bool PRE>
__global__ __launch_bounds__(256) void gemm_skinny_w8_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    __shared__ __attribute__((aligned(16))) char wbuf[4][2][4096];
    __shared__ float red[4][MB][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int ktiles = g.K / 256;
    const int per_wg = (ktiles + a.ks - 1) / a.ks;
    const int wg_beg = blockIdx.y * per_wg, wg_end = min(ktiles, wg_beg + per_wg);
    const int per_w = (max(wg_end - wg_beg, 0) + 3) / 4;
    const int beg = wg_beg + wid * per_w, end = min(wg_end, beg + per_w);

    const int prow = lane >> 4, pslot = lane & 15;
    const uint8_t *src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + prow;
        int gr = n0 + row;
        gr = gr < g.N ? gr : g.N - 1;
        src[i] = g.W8 + (int64_t)gr * g.ldw + ((pslot ^ (row & 15)) << 4);
    }
    auto stage_in = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void *)(src[i] + t * 256), (lds_void *)(&wbuf[wid][buf][i * 1024]), 16, 0, 0);
    };
    const bf16 *ap[MB];
    bool arow[MB];
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int r = mb * 16 + l15;
        arow[mb] = r < g.M;
        ap[mb] = g.A + (int64_t)(arow[mb] ? r : 0) * g.lda + lg * 8;
        acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (PRE) {
        bf16x8 av[MB][24];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
            if (beg + tt < end) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        av[mb][tt * 8 + u] = arow[mb] ? *reinterpret_cast<const bf16x8 *>(ap[mb] + (beg + tt) * 256 + u * 32) : zero8();
            }
        if (beg < end) stage_in(0, beg);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int t = beg + tt;
            if (t < end) {
                if (t + 1 < end) {
                    stage_in((tt & 1) ^ 1, t + 1);
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                const char *wb = &wbuf[wid][tt & 1][0] + l15 * 256 + (lg & 1) * 8;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                    const u32x2_t q = *reinterpret_cast<const u32x2_t *>(wb + (((u * 2 + (lg >> 1)) ^ l15) << 4));
                    u32x4_t wbits;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], true);
                        const float l0 = lo.x, l1 = lo.y, h0 = hi2.x, h1 = hi2.y;
                        wbits[2 * h] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
                        wbits[2 * h + 1] = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(h0), 0x07060302u);
                    }
                    const bf16x8 wv = __builtin_bit_cast(bf16x8, wbits);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][tt * 8 + u], wv, acc[mb], 0, 0, 0);
                }
            }
        }
    } else {
    if (beg < end) stage_in(0, beg);
        for (int t = beg; t < end; ++t) {
            const int cur = (t - beg) & 1;
            bf16x8 av[MB][8];
    #pragma unroll
            for (int mb = 0; mb < MB; ++mb)
    #pragma unroll
                for (int u = 0; u < 8; ++u) av[mb][u] = arow[mb] ? *reinterpret_cast<const bf16x8 *>(ap[mb] + t * 256 + u * 32) : zero8();
            if (t + 1 < end) {
                stage_in(cur ^ 1, t + 1);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // the 4 pieces of tile t (older than the 4 just issued)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const char *wb = &wbuf[wid][cur][0] + l15 * 256 + (lg & 1) * 8;
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                const u32x2_t q = *reinterpret_cast<const u32x2_t *>(wb + (((u * 2 + (lg >> 1)) ^ l15) << 4));
                u32x4_t wbits;
    #pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // (element reads through float variables: __builtin_bit_cast of a vector subscript picks element 0 twice here)
                    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], true);
                    const float l0 = lo.x, l1 = lo.y, h0 = hi2.x, h1 = hi2.y;
                    // bf16 pair = high halves of the two floats (exact)
                    wbits[2 * h] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
                    wbits[2 * h + 1] = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(h0), 0x07060302u);
                }
                const bf16x8 wv = __builtin_bit_cast(bf16x8, wbits);
    #pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mb][u], wv, acc[mb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid][mb][lane][r] = acc[mb][r];
    __syncthreads();
    if (wid < MB) {
        const int col = n0 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[w][wid][lane][r];
            const int row = wid * 16 + lg * 4 + r;
            if (row < g.M && col < g.N) {
                if (a.ks == 1) skinny_epilogue(g, row, col, v);
                else a.part[((int64_t)blockIdx.y * (16 * MB) + row) * g.N + col] = v;
            }
        }
    }
}

// e4m3 bytes -> bf16 (exact), 16 bytes per thread
__global__ void w8_expand_kernel(const uint8_t *__restrict__ src, bf16 *__restrict__ dst, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const u32x4_t q = *reinterpret_cast<const u32x4_t *>(src + i * 16);
    u32x4_t o[2];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], true);
        const float l0 = lo.x, l1 = lo.y, h0 = hi2.x, h1 = hi2.y;
        o[h >> 1][(h & 1) * 2] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
        o[h >> 1][(h & 1) * 2 + 1] = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(h0), 0x07060302u);
    }
    *reinterpret_cast<u32x4_t *>(dst + i * 16) = o[0];
    *reinterpret_cast<u32x4_t *>(dst + i * 16 + 8) = o[1];
}

__global__ void skinny_reduce_kernel(const SkinnyArgs a) {
    const GemmArgs &g = a.g;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.M * g.N) return;
    const int row = idx / g.N, col = idx - row * g.N;
    float v = 0.0f;
    for (int s = 0; s < a.ks; ++s) v += a.part[((int64_t)s * a.mr + row) * g.N + col];
    skinny_epilogue(g, row, col, v);
}


}  // namespace
