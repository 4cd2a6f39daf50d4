// patch.hip — fused patch embedding + position + CLS + LayerNorm1 of ViT block 0 (BASELINE north_star "fused patch-embed+LayerNorm";
// SURVEY §9 K1).  Replaces hf Blip2VisionEmbeddings.forward (modeling_blip_2.py:243-255: Conv2d(3 -> D, k = s = P) + bias,
// [CLS ; patches] + position_embedding) AND the layer_norm1 of the first Blip2EncoderLayer (:390) in ONE kernel:
//
//   pixels (N, 3, T, H, W) --coalesced image-row reads--> LDS im2col tile (64 patches x 3*P*P) --MFMA 16x16x32--> fp32 tile
//   (64 patches x D, held by the 8 waves as 64 x D/8 slices) --+ bias + pos--> x (bf16 residual stream)  and, from the same
//   registers, LayerNorm(x) (bf16, the A operand of block 0's qkv GEMM).
//
// Round 1 ran this as im2col -> HBM (655 MB per 1088-frame launch) -> GEMM -> CLS kernel -> LayerNorm kernel.  Here the frame
// tensor is read ONCE with full image-row segments (P pixels per thread, adjacent threads adjacent patches: 448-byte rows of a
// 224-pixel image are read whole), nothing of the im2col matrix touches HBM, and x is never re-read for its statistics.
// A workgroup (512 threads) owns 64 consecutive patches of one frame and ALL D columns, so the row statistics are complete
// inside the workgroup (8 partial sums per row through LDS, two passes: mean, then centred second moment — like norm.hip).
// Weights come from the zero-padded [D][KP] copy the pipeline already builds (launch_pad_rows): every workgroup streams it from
// L2 (1.8 MB; HBM sees it once).  HBM-bound stage: per frame 3*H*W*2 B in, 2 * 257 * D * 2 B out.
#include "common.h"

unsigned long long *g_patch_trace = nullptr;
extern "C" int eilev_debug_patch_trace(void *buf) { g_patch_trace = (unsigned long long *)buf; return 0; }

namespace {

constexpr int TR = 64;      // patches (output rows) per workgroup
constexpr int JMAX = 11;    // 16-column blocks per wave: D <= 8 * 11 * 16 = 1408
constexpr int NWAVE = 8;

struct PatchArgs {
    const void *pix;        // (N, 3, T, IMG, IMG)
    const bf16 *wpad;       // [D][KP] zero padded beyond PK
    const bf16 *bias, *pos, *cls, *gamma, *beta;
    bf16 *x, *ln;           // (F * tok, D) each; ln may be null (no encoder block follows)
    int frames, T, IMG, P, G, G2, tok, D, PK, KP, KS, tiles_per_frame;
    float eps;
    unsigned long long *trace;  // probe-only: 8 s_memtime stamps per workgroup (tools/patch_trace.py)
};

template <typename PT>
__global__ __launch_bounds__(512) void patch_embed_ln_kernel(const PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int f = blockIdx.x / a.tiles_per_frame, tile = blockIdx.x % a.tiles_per_frame;
    const int p0 = tile * TR, nrows = min(TR, a.G2 - p0);
    const int LDA = a.KS * 32 + 8;                       // bf16 elements per LDS row (16-byte pad: rows land on different banks)
    bf16 *sA = reinterpret_cast<bf16 *>(smem);            // [TR][LDA]
    float *sRed = reinterpret_cast<float *>(smem + (size_t)TR * LDA * 2);  // [TR][NWAVE]
    bf16 *sOut = reinterpret_cast<bf16 *>(smem + (size_t)TR * LDA * 2 + TR * NWAVE * 4);  // [16][D + 8]
    const int D = a.D, NB = D >> 4;
    auto stamp = [&](int k) { if (a.trace && tid == 0) a.trace[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime(); };
    stamp(0);

    // ---- stage the im2col tile: item = (patch row r, channel c, dy) -> P contiguous pixels of one image row -------------------
    {
        const int n = f / a.T, t = f % a.T;
        const PT *px = reinterpret_cast<const PT *>(a.pix) + ((int64_t)n * 3 * a.T + t) * a.IMG * a.IMG;
        const int64_t cstride = (int64_t)a.T * a.IMG * a.IMG;
        const int items = TR * 3 * a.P;
        for (int it = tid; it < items; it += 512) {
            // r fastest within a patch row so that adjacent threads read adjacent P-pixel runs of the SAME image row
            const int r_in = it % a.G, rest = it / a.G;         // patch column within the patch row (if the tile spans whole rows)
            const int cd = rest % (3 * a.P), prow = rest / (3 * a.P);
            const int r = prow * a.G + r_in;                     // tile row; tiles start at multiples of 64 = whole patch rows iff G | 64
            if (r >= TR) continue;
            const int c = cd / a.P, dy = cd % a.P;
            bf16 *dst = sA + r * LDA + c * a.P * a.P + dy * a.P;
            if (r < nrows) {
                const int p = p0 + r, py = p / a.G, pxi = p % a.G;
                const PT *src = px + c * cstride + (int64_t)(py * a.P + dy) * a.IMG + pxi * a.P;
                for (int dx = 0; dx < a.P; ++dx) dst[dx] = (bf16)(float)src[dx];
            } else {
                for (int dx = 0; dx < a.P; ++dx) dst[dx] = (bf16)0.0f;
            }
        }
        // zero the K padding PK .. KS*32
        const int padk = a.KS * 32 - a.PK;
        for (int it = tid; it < TR * padk; it += 512) sA[(it / padk) * LDA + a.PK + it % padk] = (bf16)0.0f;
    }
    __syncthreads();
    stamp(1);

    // ---- K loop: wave w owns column blocks w, w + 8, ... (16 columns each); A from LDS, W straight from L2 ---------------------
    f32x4 acc[4][JMAX];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int j = 0; j < JMAX; ++j) acc[rb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nj = (NB - wid + NWAVE - 1) / NWAVE;  // blocks owned by this wave (wave-uniform)
    for (int ks = 0; ks < a.KS; ++ks) {
        bf16x8 af[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) af[rb] = *reinterpret_cast<const bf16x8 *>(sA + (rb * 16 + l15) * LDA + ks * 32 + lg * 8);
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            if (j < nj) {
                const int col = (wid + j * NWAVE) * 16 + l15;
                const bf16x8 wf = *reinterpret_cast<const bf16x8 *>(a.wpad + (int64_t)col * a.KP + ks * 32 + lg * 8);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rb], wf, acc[rb][j], 0, 0, 0);
            }
        }
    }

    stamp(2);
    // ---- epilogue: x = bf16(acc + bias + pos); LayerNorm over the bf16 values (what the unfused path normalises) ---------------
    // lane holds rows rb*16 + lg*4 + r (r = 0..3) of column (wid + j*8)*16 + l15
    float xv[4][JMAX][4];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        if (j < nj) {
            const int col = (wid + j * NWAVE) * 16 + l15;
            const float b = a.bias ? (float)a.bias[col] : 0.0f;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rb * 16 + lg * 4 + r;
                    const int tokn = 1 + p0 + (row < nrows ? row : 0);
                    // same association as the GEMM epilogue it replaces: (acc + bias) + pos, one rounding to bf16
                    xv[rb][j][r] = (float)(bf16)(acc[rb][j][r] + b + (float)a.pos[(int64_t)tokn * D + col]);
                }
        }
    }
    stamp(3);
    float mean[4][4], rstd[4][4];
    auto row_reduce = [&](float (&part)[4][4], float (&out)[4][4]) {
        // part[rb][r]: this lane's partial over its columns -> sum over the 16 lanes of a row group -> over the 8 waves via LDS
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = part[rb][r];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                if (l15 == 0) sRed[(rb * 16 + lg * 4 + r) * NWAVE + wid] = v;
            }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *q = sRed + (rb * 16 + lg * 4 + r) * NWAVE;
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) s += q[w];  // fixed order: deterministic
                out[rb][r] = s;
            }
        __syncthreads();
    };
    if (a.ln) {
        float part[4][4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < JMAX; ++j)
                    if (j < nj) s += xv[rb][j][r];
                part[rb][r] = s;
            }
        row_reduce(part, mean);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mean[rb][r] /= (float)D;
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < JMAX; ++j)
                    if (j < nj) {
                        const float d = xv[rb][j][r] - mean[rb][r];
                        s += d * d;
                    }
                part[rb][r] = s;
            }
        row_reduce(part, rstd);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) rstd[rb][r] = rsqrtf(rstd[rb][r] / (float)D + a.eps);
    }

    stamp(4);
    // ---- stores: one 16-row block at a time through LDS so that HBM sees 16-byte pieces of whole rows ---------------------------
    const int LDO = D + 8;
    const int chunks = D >> 3;  // 16-byte chunks per row
    auto flush = [&](int rb, bf16 *dst_base) {
        __syncthreads();
        for (int it = tid; it < 16 * chunks; it += 512) {
            const int rr = it / chunks, ch = it % chunks, row = rb * 16 + rr;
            if (row < nrows)
                *reinterpret_cast<bf16x8 *>(dst_base + ((int64_t)f * a.tok + 1 + p0 + row) * D + ch * 8) =
                    *reinterpret_cast<const bf16x8 *>(sOut + rr * LDO + ch * 8);
        }
        __syncthreads();
    };
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        if (rb * 16 >= nrows) break;
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
            if (j < nj) {
                const int col = (wid + j * NWAVE) * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) sOut[(lg * 4 + r) * LDO + col] = (bf16)xv[rb][j][r];
            }
        flush(rb, a.x);
        if (a.ln) {
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                if (j < nj) {
                    const int col = (wid + j * NWAVE) * 16 + l15;
                    const float gm = (float)a.gamma[col], bt = (float)a.beta[col];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sOut[(lg * 4 + r) * LDO + col] = (bf16)((xv[rb][j][r] - mean[rb][r]) * rstd[rb][r] * gm + bt);
                }
            flush(rb, a.ln);
        }
    }

    stamp(5);
    // ---- the frame's CLS row (token 0): x = cls + pos[0], LayerNorm of it — by the frame's first tile, wave 0 ----------------------
    if (tile == 0 && wid == 0) {
        float s = 0.0f;
        for (int c = lane; c < D; c += 64) s += (float)(bf16)((float)a.cls[c] + (float)a.pos[c]);
        const float mu = wave_sum(s) / (float)D;
        float q = 0.0f;
        for (int c = lane; c < D; c += 64) {
            const float d = (float)(bf16)((float)a.cls[c] + (float)a.pos[c]) - mu;
            q += d * d;
        }
        const float rs = rsqrtf(wave_sum(q) / (float)D + a.eps);
        for (int c = lane; c < D; c += 64) {
            const bf16 v = (bf16)((float)a.cls[c] + (float)a.pos[c]);
            a.x[(int64_t)f * a.tok * D + c] = v;
            if (a.ln) a.ln[(int64_t)f * a.tok * D + c] = (bf16)(((float)v - mu) * rs * (float)a.gamma[c] + (float)a.beta[c]);
        }
    }
}

}  // namespace

// Returns EILEV_E_UNSUPPORTED when the shape does not fit the fused kernel (the pipeline then takes the unfused kernels).
int launch_patch_embed_ln(const void *pix, int pix_dtype, const bf16 *wpad, const bf16 *bias, const bf16 *pos, const bf16 *cls,
                          const bf16 *gamma, const bf16 *beta, bf16 *x, bf16 *ln, int64_t frames_total, int frames_per_clip, int img,
                          int patch, int D, int KP, float eps, hipStream_t s) {
    const int G = img / patch, G2 = G * G, PK = 3 * patch * patch, KS = (PK + 31) / 32;
    if (D % 16 != 0 || D > NWAVE * JMAX * 16 || KS * 32 > KP || (KP & 7) || (TR % G != 0 && G2 > TR) || !wpad || !pos || !cls || !x) return EILEV_E_UNSUPPORTED;
    if (ln && (!gamma || !beta)) return EILEV_E_BADARG;
    const int LDA = KS * 32 + 8;
    const size_t smem = (size_t)TR * LDA * 2 + TR * NWAVE * 4 + (size_t)16 * (D + 8) * 2;
    if (smem > 160 * 1024) return EILEV_E_UNSUPPORTED;
    PatchArgs a;
    a.pix = pix; a.wpad = wpad; a.bias = bias; a.pos = pos; a.cls = cls; a.gamma = gamma; a.beta = beta; a.x = x; a.ln = ln;
    a.frames = (int)frames_total; a.T = frames_per_clip; a.IMG = img; a.P = patch; a.G = G; a.G2 = G2; a.tok = G2 + 1; a.D = D;
    a.PK = PK; a.KP = KP; a.KS = KS; a.tiles_per_frame = (G2 + TR - 1) / TR; a.eps = eps; a.trace = g_patch_trace;
    static bool attr = false;
    if (!attr) {
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(patch_embed_ln_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EILEV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(patch_embed_ln_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const dim3 grid((unsigned)(frames_total * a.tiles_per_frame));
    if (pix_dtype == EILEV_F32) hipLaunchKernelGGL(patch_embed_ln_kernel<float>, grid, dim3(512), smem, s, a);
    else hipLaunchKernelGGL(patch_embed_ln_kernel<bf16>, grid, dim3(512), smem, s, a);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}
