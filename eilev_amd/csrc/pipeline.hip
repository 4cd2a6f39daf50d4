// pipeline.hip — the C ABI of include/eilev.h on top of the gfx950 kernels: stage orchestration only
// (which kernel runs on which buffer); no arithmetic lives here.  Every launch goes to the caller's
// stream, nothing allocates or synchronises (except eilev_prof_collect), so a whole stage can be captured
// into a hipGraph by the caller.
#include <vector>

#include "common.h"

int launch_im2col(const void *pix, int dtype, bf16 *out, int64_t rows, int frames, int img, int patch, int kp, hipStream_t s);
int launch_pad_rows(const bf16 *w, bf16 *out, int rows, int k, int kp, hipStream_t s);
int launch_cls_rows(const bf16 *cls, const bf16 *pos, bf16 *x, int64_t frames_total, int tok, int d, hipStream_t s);
int launch_quant_rows_e4m3(const bf16 *x, int64_t ldx, uint8_t *q, float *scale, int64_t rows, int cols, hipStream_t s);
int launch_broadcast_rows(const bf16 *src, bf16 *dst, int64_t copies, int64_t n, hipStream_t s);
int launch_embed_scatter(const bf16 *embed, const int64_t *ids, const uint8_t *mask, const bf16 *feats, int64_t n_rows,
                         int64_t total, int vocab, bf16 *out, int d, hipStream_t s);
int launch_pos_embed(const bf16 *emb, const bf16 *pos, const int32_t *mask, int32_t *pid, bf16 *h, int batch, int L, int d, hipStream_t s,
                     int past = 0);
int launch_decode_embed(const bf16 *embed, const bf16 *pos, const int64_t *tokens, const int32_t *n_valid, const int32_t *state,
                        int vocab, int max_pid, bf16 *h, int batch, int d, hipStream_t s);
int launch_kv_write(const bf16 *qkv, bf16 *kc, bf16 *vc, int batch, int rows_per_b, int heads, int hd, int cap, int seq_len,
                    const int32_t *state, hipStream_t s, int slot0 = 0);
int launch_attn_decode(const bf16 *qkv, const bf16 *kc, const bf16 *vc, bf16 *out, const int32_t *attn_mask, const int32_t *state,
                       int batch, int seq_len, int cap, int heads, int hd, float *scratch, size_t scratch_bytes, hipStream_t s,
                       int64_t ldq = 0, const float *rel_tab = nullptr, int64_t rel_hs = 0, int rel_off = 0, int fuse_new = 0,
                       const bf16 *kg = nullptr, const bf16 *vg = nullptr, const int32_t *anc = nullptr, int beams = 1, int cap_g = 0, int out_frag = 0);
bool attn_decode_loop_ok(int batch, int heads, int hd, int cap_all, bool beam, const void *out, const void *state, int fuse_new, const void *rel_tab);
bool gemm_rows32_takes(const GemmArgs &g);  // gemm.hip
size_t attn_decode_scratch_bytes(int batch, int heads, int hd, int cap);
int launch_select(const float *logits, int batch, int vocab, int32_t *state, uint8_t *finished, int64_t eos_id, int64_t pad_id,
                  int64_t *tokens, int64_t *out_tokens, int64_t max_new, hipStream_t s);

int launch_t5_rel_table(const bf16 *rel_w, float *tab, int n, int off, int heads, int bidirectional, int num_buckets, int max_dist,
                        hipStream_t s, const int32_t *state = nullptr);
int launch_gated_gelu(const bf16 *ab, int64_t ld, bf16 *out, int64_t rows, int F, hipStream_t s);
int launch_rows_to_cache(const bf16 *src, int64_t ld, int col0, bf16 *plane, int batch, int rows_per_b, int heads, int hd, int cap,
                         int slot0, hipStream_t s, const int32_t *state = nullptr);

#define RC(expr)                 \
    do {                         \
        int _rc = (expr);        \
        if (_rc != 0) return _rc; \
    } while (0)

// ---- kernel profiler --------------------------------------------------------------------------------
namespace {
struct ProfRec {
    hipEvent_t a, b;
    int kind;
    double flops;
};
std::vector<ProfRec> g_recs;
size_t g_used = 0;
bool g_prof_on = false;
constexpr size_t kMaxRecs = 65536;
}  // namespace

void prof_begin(int kind, double flops, hipStream_t s) {
    if (!g_prof_on || g_used >= kMaxRecs) return;
    if (g_used == g_recs.size()) {
        ProfRec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        g_recs.push_back(r);
    }
    g_recs[g_used].kind = kind;
    g_recs[g_used].flops = flops;
    (void)hipEventRecord(g_recs[g_used].a, s);
}
void prof_end(hipStream_t s) {
    if (!g_prof_on || g_used >= kMaxRecs || g_used >= g_recs.size()) return;
    (void)hipEventRecord(g_recs[g_used].b, s);
    ++g_used;
}

extern "C" int eilev_prof_enable(int on) {
    g_prof_on = on != 0;
    g_used = 0;
    return 0;
}
extern "C" int eilev_prof_collect(int kind, int64_t *launches, double *total_ms, double *total_flops) {
    int64_t n = 0;
    double ms = 0.0, fl = 0.0;
    for (size_t i = 0; i < g_used; ++i) {
        if (kind != 0 && g_recs[i].kind != kind) continue;
        if (hipEventSynchronize(g_recs[i].b) != hipSuccess) continue;
        float t = 0.0f;
        if (hipEventElapsedTime(&t, g_recs[i].a, g_recs[i].b) != hipSuccess) continue;
        ms += t;
        fl += g_recs[i].flops;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return 0;
}

extern "C" int eilev_abi_version(void) { return EILEV_ABI_VERSION; }
// (r4) 24 576: tools/vit_small_launch.py — the fold wins from 96 frames per launch (53.3 vs 55.7 ms; 136 frames 69.3 vs 71.5), ties at 32-64
// frames and loses below (8 frames 11.2 vs 8.9 ms: too few 256 x 256 tiles for the persistent kernel)
constexpr int64_t kLnFoldMinRows = 24576;  // EilevVitWeights.fold_min_rows == 0
// The patch path: im2col_strip_kernel reads the frame tensor with coalesced 16-byte loads at 3.9 TB/s (168 us per 1088 frames) and im2col +
// GEMM + CLS rows take 0.93 ms per launch; the fused patch-embed + LayerNorm kernel of rounds 1-2 took 2.74 ms (0.69 TB/s on the pixels:
// 50 spilled VGPRs in its K loop, W re-streamed from L2 by every workgroup) and was retired to tools/probes/patch_fused.hip in round 5.
extern "C" const char *eilev_backend(void) { return "hip-gfx950"; }

namespace {

struct Carver {
    char *p;
    char *end;
    template <typename T>
    T *take(size_t n) {
        char *q = p;
        p += align_up(n * sizeof(T), 256);
        return reinterpret_cast<T *>(q);
    }
    bool ok() const { return p <= end; }
};

GemmArgs mk_gemm(const bf16 *A, int64_t lda, const void *W, int64_t ldw, const void *bias, const bf16 *resid, int64_t ldr,
                 void *C, int64_t ldc, int64_t M, int N, int K, int epi) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = (const bf16 *)W; g.ldw = ldw; g.bias = (const bf16 *)bias; g.resid = resid; g.ldr = ldr;
    g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K; g.epi = epi; g.out_f32 = 0; g.scale = 1.0f; g.scale_cols = 0;
    g.patch_group = 0; g.scratch = nullptr; g.scratch_bytes = 0; g.dbg = 0;
    return g;
}

inline int64_t vit_tok(const EilevDims *d) {
    const int64_t g = d->image_size / d->patch_size;
    return g * g + 1;
}
inline int patch_kp(const EilevDims *d) { return (3 * d->patch_size * d->patch_size + 63) / 64 * 64; }

bool dims_ok_vit(const EilevDims *d) {
    return d->v_hidden % 8 == 0 && d->v_inter % 8 == 0 && d->v_heads > 0 && d->v_hidden % d->v_heads == 0 &&
           (d->v_hidden / d->v_heads) % 8 == 0 && d->v_hidden / d->v_heads <= 128 && d->v_hidden <= 4096 &&
           d->image_size % d->patch_size == 0;
}
bool dims_ok_qf(const EilevDims *d) {
    return d->q_hidden % 8 == 0 && d->q_inter % 8 == 0 && d->q_heads > 0 && d->q_hidden % d->q_heads == 0 &&
           (d->q_hidden / d->q_heads) % 8 == 0 && d->q_hidden / d->q_heads <= 128 && d->q_hidden <= 4096 && d->q_cross_freq > 0;
}
bool dims_ok_opt(const EilevDims *d) {
    return d->t_hidden % 8 == 0 && d->t_ffn % 8 == 0 && d->t_heads > 0 && d->t_hidden % d->t_heads == 0 &&
           (d->t_hidden / d->t_heads) % 8 == 0 && d->t_hidden / d->t_heads <= 128 && d->t_hidden <= 4096;
}

int g_vit_head_major = 1;  // probe / test switch (eilev_debug_vit_head_major, probe build): 0 = row-major q|k|v in every ViT launch
constexpr size_t kSkinnyScratch = 16u << 20;  // halves: split-K partials of the decode GEMVs | flash-decoding partials (batch 32 x 40 heads x 9 key splits x (128 + 2) floats = 6 MB)

}  // namespace

// =====================================================================================================
// Stage 1: ViT
// =====================================================================================================
extern "C" size_t eilev_vit_workspace_bytes(const EilevDims *d, int64_t n_clips, int64_t frames) {
    const int64_t M = n_clips * frames * vit_tok(d);
    size_t b = 0;
    b += align_up((size_t)M * d->v_hidden * 2, 256) * 3;      // x, ln, att
    b += align_up((size_t)M * d->v_hidden * 3 * 2, 256);       // qkv
    b += align_up((size_t)M * (d->v_inter > patch_kp(d) ? d->v_inter : patch_kp(d)) * 2, 256);  // mlp / patch rows
    b += align_up((size_t)d->v_hidden * patch_kp(d) * 2, 256);  // padded patch weight
    b += align_up((size_t)M * ((d->v_hidden + 63) / 64) * 8, 256) + align_up((size_t)M * 8, 256);  // folded LayerNorm: row statistics
    return b + 256;
}

namespace {
// Debug output of the slow path: softmax(scale * q k^T) of one (frame, head) as a (tok, tok) bf16 matrix.  One wave per query
// row, a lane owns keys lane, lane + 64, ...; fp32 scores, max, sum — no tiling, no LDS: this is off the throughput path.
__global__ void __launch_bounds__(256) attn_probs_kernel(const bf16 *__restrict__ qkv, bf16 *__restrict__ probs, int tok, int heads,
                                                         int hd, int D, float scale) {
    const int fh = blockIdx.x, f = fh / heads, h = fh % heads;
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= tok) return;
    const bf16 *base = qkv + (int64_t)f * tok * 3 * D;
    const bf16 *q = base + (int64_t)i * 3 * D + h * hd;
    float sc[16];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int j = lane + 64 * t;
        float acc = -INFINITY;
        if (j < tok) {
            const bf16 *k = base + (int64_t)j * 3 * D + D + h * hd;
            acc = 0.0f;
            for (int e = 0; e < hd; e += 8) {
                const bf16x8 qa = *reinterpret_cast<const bf16x8 *>(q + e), ka = *reinterpret_cast<const bf16x8 *>(k + e);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fmaf((float)qa[u], (float)ka[u], acc);
            }
            acc *= scale;
        }
        sc[t] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        sc[t] = (lane + 64 * t < tok) ? __expf(sc[t] - mx) : 0.0f;
        sum += sc[t];
    }
    sum = wave_sum(sum);
    bf16 *out = probs + ((int64_t)fh * tok + i) * tok;
#pragma unroll
    for (int t = 0; t < 16; ++t)
        if (lane + 64 * t < tok) out[lane + 64 * t] = (bf16)(sc[t] / sum);
}

int vit_forward_impl(const EilevDims *d, const EilevVitWeights *w, const void *pixels, int pixels_dtype, int64_t n_clips, int64_t frames,
                     void *image_embeds, void *pooler, void *hidden_states, void *attentions, void *workspace, size_t workspace_bytes,
                     void *stream);
}  // namespace

extern "C" int eilev_vit_forward(const EilevDims *d, const EilevVitWeights *w, const void *pixels, int pixels_dtype,
                                 int64_t n_clips, int64_t frames, void *image_embeds, void *pooler, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    return vit_forward_impl(d, w, pixels, pixels_dtype, n_clips, frames, image_embeds, pooler, nullptr, nullptr, workspace,
                            workspace_bytes, stream);
}

// The reference's debug outputs (ref:eilev/model/v2.py:76-103, asserted by ref:tests/model/test_model_v2.py:57-83): the residual
// stream after the embeddings and after every block, and every block's attention probabilities.  Same kernels as
// eilev_vit_forward plus copies / the unfused probability kernel: a slow path, off the benchmark.
extern "C" int eilev_vit_forward_debug(const EilevDims *d, const EilevVitWeights *w, const void *pixels, int pixels_dtype,
                                       int64_t n_clips, int64_t frames, void *image_embeds, void *pooler, void *hidden_states,
                                       void *attentions, void *workspace, size_t workspace_bytes, void *stream) {
    return vit_forward_impl(d, w, pixels, pixels_dtype, n_clips, frames, image_embeds, pooler, hidden_states, attentions, workspace,
                            workspace_bytes, stream);
}

namespace {
int vit_forward_impl(const EilevDims *d, const EilevVitWeights *w, const void *pixels, int pixels_dtype, int64_t n_clips, int64_t frames,
                     void *image_embeds, void *pooler, void *hidden_states, void *attentions, void *workspace, size_t workspace_bytes,
                     void *stream) {
    if (!d || !w || !pixels || !image_embeds || !workspace || n_clips <= 0 || frames <= 0) return EILEV_E_BADARG;
    if (attentions && vit_tok(d) > 1024) return EILEV_E_UNSUPPORTED;
    if (!dims_ok_vit(d)) return EILEV_E_UNSUPPORTED;
    if (workspace_bytes < eilev_vit_workspace_bytes(d, n_clips, frames)) return EILEV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->v_hidden, Fi = d->v_inter, H = d->v_heads, hd = D / H;
    const int64_t tok = vit_tok(d), G2 = tok - 1, F = n_clips * frames, M = F * tok;
    if (M > 0x7fffffff / 2) return EILEV_E_UNSUPPORTED;
    const int KP = patch_kp(d), PK = 3 * d->patch_size * d->patch_size;
    Carver cv{(char *)workspace, (char *)workspace + workspace_bytes};
    bf16 *x = cv.take<bf16>((size_t)M * D), *ln = cv.take<bf16>((size_t)M * D), *att = cv.take<bf16>((size_t)M * D);
    bf16 *qkv = cv.take<bf16>((size_t)M * 3 * D);
    bf16 *mlp = cv.take<bf16>((size_t)M * (Fi > KP ? Fi : KP));
    bf16 *wpad = cv.take<bf16>((size_t)D * KP);
    const int slots = (D + 63) / 64;
    float *part = cv.take<float>((size_t)M * slots * 2), *lnrows = cv.take<float>((size_t)M * 2);
    if (!cv.ok()) return EILEV_E_WORKSPACE;
    // LayerNorm folded into qkv / fc1 (EilevVitWeights.layers_fold): the throughput path of large launches.  The debug outputs and
    // small launches (where other GEMM kernels than the persistent one win) keep the LayerNorm kernels.
    const bool fold = w->layers_fold && w->fold_min_rows >= 0 && M >= (w->fold_min_rows ? w->fold_min_rows : kLnFoldMinRows) && !hidden_states && !attentions && D % 64 == 0 && Fi % 64 == 0 &&
                      (int64_t)M * D * 2 < 0x7fff0000ll && (int64_t)Fi * D * 2 < 0x7fff0000ll;

    // patch embedding (+ bias + position, CLS rows): hf modeling_blip_2.py:243-255 as im2col (coalesced strip reads of the frame tensor) ->
    // GEMM with the position rows as its "residual" -> CLS rows; layer_norm1 of block 0 is the folded LayerNorm of its qkv GEMM (or the
    // LayerNorm kernel below).  (The ONE-kernel form — patch embedding + LayerNorm1 fused, rounds 1-2 — measured slower end to end in
    // round 3 and lives on as tools/probes/patch_fused.hip.)
    RC(launch_pad_rows((const bf16 *)w->patch_w, wpad, D, PK, KP, s));
    const bool ln0_done = false;
    {
        // prof kind 6: the pixel read of the frame tensor ("flops" carries the BYTES of pixels read: bench.py reports GB/s)
        prof_begin(6, (double)F * 3.0 * d->image_size * d->image_size * (pixels_dtype == 0 ? 4.0 : 2.0), s);
        RC(launch_im2col(pixels, pixels_dtype, mlp, F * G2, (int)frames, d->image_size, d->patch_size, KP, s));
        prof_end(s);
        GemmArgs g = mk_gemm(mlp, KP, wpad, KP, w->patch_b, (const bf16 *)w->pos, D, x, D, F * G2, D, KP, 0);
        g.patch_group = (int)G2;
        RC(launch_gemm(g, 5, s));
        RC(launch_cls_rows((const bf16 *)w->cls, (const bf16 *)w->pos, x, F, (int)tok, D, s));
    }
    const size_t hs_bytes = (size_t)M * D * sizeof(bf16);
    if (hidden_states) RC((int)hipMemcpyAsync(hidden_states, x, hs_bytes, hipMemcpyDeviceToDevice, s));

    const float scale = 1.0f / sqrtf((float)hd);
    for (int l = 0; l < d->v_layers; ++l) {
        const EilevVitLayer *L = &w->layers[l];
        const EilevVitLayerFold *LF = fold ? &w->layers_fold[l] : nullptr;
        bool hm = false;  // this block's q|k|v scattered into per-head blocks (GemmArgs::hm_tok): large folded launches whose attention is attn_frame3_kernel
        if (fold && l > 0) {  // fc2 of the previous block left the row statistics of x: qkv reads the raw stream
            GemmArgs g = mk_gemm(x, D, LF->qkv_w, D, LF->qkv_b, nullptr, 0, qkv, 3 * D, M, 3 * D, D, 0);
            g.ln_rows = lnrows;
            g.ln_csum = LF->qkv_csum;
            if (g_vit_head_major && w->layers_fold_hm && F >= 512 && tok == 257 && hd == 88) {
                const EilevVitLayerFoldHm *LH = &w->layers_fold_hm[l];  // the same folded matrix with its rows (output columns) in block order
                GemmArgs gh = mk_gemm(x, D, LH->qkv_w, D, LH->qkv_b, nullptr, 0, qkv, 3 * D, M, 3 * D, D, 0);
                gh.ln_rows = lnrows;
                gh.ln_csum = LH->qkv_csum;
                gh.hm_tok = (int)tok;
                gh.hm_heads = H;
                gh.hm_hd = hd;
                if (LH->qkv_w && LH->qkv_csum && hm_takes(gh)) {
                    g = gh;
                    hm = true;
                }
            }
            RC(launch_gemm(g, 3, s));
        } else {
            if (!(l == 0 && ln0_done)) RC(launch_layernorm(x, D, (const bf16 *)L->ln1_w, (const bf16 *)L->ln1_b, ln, D, M, D, d->v_eps, s));
            RC(launch_gemm(mk_gemm(ln, D, L->qkv_w, D, L->qkv_b, nullptr, 0, qkv, 3 * D, M, 3 * D, D, 0), 3, s));
        }
        if (attentions) {
            bf16 *pr = (bf16 *)attentions + (size_t)l * F * H * tok * tok;
            attn_probs_kernel<<<dim3((unsigned)(F * H), (unsigned)((tok + 3) / 4)), 256, 0, s>>>(qkv, pr, (int)tok, H, hd, D, scale);
            EILEV_LAUNCH_CHECK();
        }
        AttnArgs a;
        a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.o = att;
        a.q_bs = a.k_bs = a.v_bs = tok * 3 * (int64_t)D; a.o_bs = tok * (int64_t)D;
        a.q_hs = a.k_hs = a.v_hs = a.o_hs = hd;
        a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
        if (hm) {  // three planes per frame, one block of tok * hd elements per head ([tok][64] then [tok][24]: AttnArgs::hm)
            a.k = qkv + tok * (int64_t)D; a.v = qkv + 2 * tok * (int64_t)D;
            a.q_hs = a.k_hs = a.v_hs = tok * (int64_t)hd;
            a.hm = 1;
        }
        a.batch = (int)F; a.heads = H; a.sq = (int)tok; a.skv = (int)tok; a.hd = hd; a.scale = scale; a.causal = 0;
        a.key_mask = nullptr; a.mask_ld = 0;
        RC(launch_attention(a, s));
        if (fold) {
            GemmArgs gp = mk_gemm(att, D, L->proj_w, D, L->proj_b, x, D, x, D, M, D, D, 0);
            gp.stat_out = part;
            gp.stat_ld = M;
            RC(launch_gemm(gp, 4, s));
            RC(launch_ln_finalize(part, slots, M, D, d->v_eps, lnrows, s));
            GemmArgs g1 = mk_gemm(x, D, LF->fc1_w, D, LF->fc1_b, nullptr, 0, mlp, Fi, M, Fi, D, 1);
            g1.ln_rows = lnrows;
            g1.ln_csum = LF->fc1_csum;
            RC(launch_gemm(g1, 1, s));
            GemmArgs g2 = mk_gemm(mlp, Fi, L->fc2_w, Fi, L->fc2_b, x, D, x, D, M, D, Fi, 0);
            if (l + 1 < d->v_layers) {  // the next block's layer_norm1 is folded too; post_layernorm below stays a kernel
                g2.stat_out = part;
                g2.stat_ld = M;
            }
            RC(launch_gemm(g2, 2, s));
            if (l + 1 < d->v_layers) RC(launch_ln_finalize(part, slots, M, D, d->v_eps, lnrows, s));
        } else {
            RC(launch_gemm(mk_gemm(att, D, L->proj_w, D, L->proj_b, x, D, x, D, M, D, D, 0), 4, s));
            RC(launch_layernorm(x, D, (const bf16 *)L->ln2_w, (const bf16 *)L->ln2_b, ln, D, M, D, d->v_eps, s));
            RC(launch_gemm(mk_gemm(ln, D, L->fc1_w, D, L->fc1_b, nullptr, 0, mlp, Fi, M, Fi, D, 1), 1, s));
            RC(launch_gemm(mk_gemm(mlp, Fi, L->fc2_w, Fi, L->fc2_b, x, D, x, D, M, D, Fi, 0), 2, s));
        }
        if (hidden_states)
            RC((int)hipMemcpyAsync((char *)hidden_states + (size_t)(l + 1) * hs_bytes, x, hs_bytes, hipMemcpyDeviceToDevice, s));
    }
    RC(launch_layernorm(x, D, (const bf16 *)w->post_ln_w, (const bf16 *)w->post_ln_b, (bf16 *)image_embeds, D, M, D, d->v_eps, s));
    if (pooler)
        RC(launch_layernorm((const bf16 *)image_embeds, tok * D, (const bf16 *)w->post_ln_w, (const bf16 *)w->post_ln_b,
                            (bf16 *)pooler, D, F, D, d->v_eps, s));
    return EILEV_OK;
}
}  // namespace

// =====================================================================================================
// Stage 2: Q-Former
// =====================================================================================================
extern "C" size_t eilev_qformer_workspace_bytes(const EilevDims *d, int64_t n_clips, int64_t kv_len) {
    const int64_t R = n_clips * d->num_query;
    size_t b = 0;
    b += align_up((size_t)R * d->q_hidden * 2, 256) * 3;          // h, a, t
    b += align_up((size_t)R * d->q_hidden * 3 * 2, 256);           // qkv
    b += align_up((size_t)R * d->q_inter * 2, 256);                // f
    b += align_up((size_t)n_clips * kv_len * d->q_hidden * 2 * 2, 256);  // cross k|v
    b += align_up((size_t)d->num_query * d->q_hidden * 2, 256);    // normed query tokens
    return b + 256;
}

extern "C" int eilev_qformer_forward(const EilevDims *d, const EilevQfWeights *w, const void *image_embeds, int64_t n_clips,
                                     int64_t kv_len, void *query_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!d || !w || !image_embeds || !query_out || !workspace || n_clips <= 0 || kv_len <= 0) return EILEV_E_BADARG;
    if (!dims_ok_qf(d) || d->v_hidden % 8) return EILEV_E_UNSUPPORTED;
    if (workspace_bytes < eilev_qformer_workspace_bytes(d, n_clips, kv_len)) return EILEV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->q_hidden, Fi = d->q_inter, nq = d->num_query, H = d->q_heads, hd = D / H, Dv = d->v_hidden;
    const int64_t R = n_clips * nq, MK = n_clips * kv_len;
    Carver cv{(char *)workspace, (char *)workspace + workspace_bytes};
    bf16 *h = cv.take<bf16>((size_t)R * D), *a_ = cv.take<bf16>((size_t)R * D), *t = cv.take<bf16>((size_t)R * D);
    bf16 *qkv = cv.take<bf16>((size_t)R * 3 * D), *f = cv.take<bf16>((size_t)R * Fi);
    bf16 *ckv = cv.take<bf16>((size_t)MK * 2 * D), *q0 = cv.take<bf16>((size_t)nq * D);
    if (!cv.ok()) return EILEV_E_WORKSPACE;
    const bf16 *img = (const bf16 *)image_embeds;
    const float scale = 1.0f / sqrtf((float)hd);

    // embedding_output = layernorm(query_tokens), same for every clip (hf :913)
    RC(launch_layernorm((const bf16 *)w->query_tokens, D, (const bf16 *)w->ln_w, (const bf16 *)w->ln_b, q0, D, nq, D, d->q_eps, s));
    RC(launch_broadcast_rows(q0, h, n_clips, (int64_t)nq * D, s));

    for (int l = 0; l < d->q_layers; ++l) {
        const EilevQfLayer *L = &w->layers[l];
        // self-attention q|k|v: one GEMM when the three weights are packed contiguously
        const bf16 *sq = (const bf16 *)L->sq_w;
        const bool fused = (const bf16 *)L->sk_w == sq + (size_t)D * D && (const bf16 *)L->sv_w == sq + 2 * (size_t)D * D &&
                           (const bf16 *)L->sk_b == (const bf16 *)L->sq_b + D && (const bf16 *)L->sv_b == (const bf16 *)L->sq_b + 2 * D;
        if (fused) {
            RC(launch_gemm(mk_gemm(h, D, L->sq_w, D, L->sq_b, nullptr, 0, qkv, 3 * D, R, 3 * D, D, 0), 5, s));
        } else {
            RC(launch_gemm(mk_gemm(h, D, L->sq_w, D, L->sq_b, nullptr, 0, qkv, 3 * D, R, D, D, 0), 5, s));
            RC(launch_gemm(mk_gemm(h, D, L->sk_w, D, L->sk_b, nullptr, 0, qkv + D, 3 * D, R, D, D, 0), 5, s));
            RC(launch_gemm(mk_gemm(h, D, L->sv_w, D, L->sv_b, nullptr, 0, qkv + 2 * D, 3 * D, R, D, D, 0), 5, s));
        }
        AttnArgs a;
        a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.o = a_;
        a.q_bs = a.k_bs = a.v_bs = (int64_t)nq * 3 * D; a.o_bs = (int64_t)nq * D;
        a.q_hs = a.k_hs = a.v_hs = a.o_hs = hd;
        a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
        a.batch = (int)n_clips; a.heads = H; a.sq = nq; a.skv = nq; a.hd = hd; a.scale = scale; a.causal = 0;
        a.key_mask = nullptr; a.mask_ld = 0;
        RC(launch_attention(a, s));
        RC(launch_gemm(mk_gemm(a_, D, L->so_w, D, L->so_b, h, D, t, D, R, D, D, 0), 5, s));
        RC(launch_layernorm(t, D, (const bf16 *)L->sln_w, (const bf16 *)L->sln_b, h, D, R, D, d->q_eps, s));
        if (L->cq_w) {
            RC(launch_gemm(mk_gemm(h, D, L->cq_w, D, L->cq_b, nullptr, 0, qkv, 3 * D, R, D, D, 0), 5, s));
            const bf16 *ck = (const bf16 *)L->ck_w;
            const bool kvf = (const bf16 *)L->cv_w == ck + (size_t)D * Dv && (const bf16 *)L->cv_b == (const bf16 *)L->ck_b + D;
            if (kvf) {
                RC(launch_gemm(mk_gemm(img, Dv, L->ck_w, Dv, L->ck_b, nullptr, 0, ckv, 2 * D, MK, 2 * D, Dv, 0), 5, s));
            } else {
                RC(launch_gemm(mk_gemm(img, Dv, L->ck_w, Dv, L->ck_b, nullptr, 0, ckv, 2 * D, MK, D, Dv, 0), 5, s));
                RC(launch_gemm(mk_gemm(img, Dv, L->cv_w, Dv, L->cv_b, nullptr, 0, ckv + D, 2 * D, MK, D, Dv, 0), 5, s));
            }
            AttnArgs c;
            c.q = qkv; c.k = ckv; c.v = ckv + D; c.o = a_;
            c.q_bs = (int64_t)nq * 3 * D; c.k_bs = c.v_bs = kv_len * 2 * (int64_t)D; c.o_bs = (int64_t)nq * D;
            c.q_hs = c.k_hs = c.v_hs = c.o_hs = hd;
            c.ldq = 3 * D; c.ldk = c.ldv = 2 * D; c.ldo = D;
            c.batch = (int)n_clips; c.heads = H; c.sq = nq; c.skv = (int)kv_len; c.hd = hd; c.scale = scale; c.causal = 0;
            c.key_mask = nullptr; c.mask_ld = 0;
            RC(launch_attention(c, s));
            RC(launch_gemm(mk_gemm(a_, D, L->co_w, D, L->co_b, h, D, t, D, R, D, D, 0), 5, s));
            RC(launch_layernorm(t, D, (const bf16 *)L->cln_w, (const bf16 *)L->cln_b, h, D, R, D, d->q_eps, s));
        }
        RC(launch_gemm(mk_gemm(h, D, L->fi_w, D, L->fi_b, nullptr, 0, f, Fi, R, Fi, D, 1), 5, s));
        RC(launch_gemm(mk_gemm(f, Fi, L->fo_w, Fi, L->fo_b, h, D, t, D, R, D, Fi, 0), 5, s));
        RC(launch_layernorm(t, D, (const bf16 *)L->fln_w, (const bf16 *)L->fln_b,
                            l == d->q_layers - 1 ? (bf16 *)query_out : h, D, R, D, d->q_eps, s));
    }
    return EILEV_OK;
}

// =====================================================================================================
// Stage 3: projection, embedding, scatter
// =====================================================================================================
extern "C" int eilev_project_rows(const EilevDims *d, const void *proj_w, const void *proj_b, const void *query_out,
                                  int64_t n_rows, void *video_feats, void *stream) {
    if (!d || !proj_w || !query_out || !video_feats) return EILEV_E_BADARG;
    return launch_gemm(mk_gemm((const bf16 *)query_out, d->q_hidden, proj_w, d->q_hidden, proj_b, nullptr, 0, video_feats,
                               d->t_hidden, n_rows, d->t_hidden, d->q_hidden, 0), 5, (hipStream_t)stream);
}

extern "C" int eilev_embed_scatter(const EilevDims *d, const void *embed_tokens, const int64_t *input_ids,
                                   const uint8_t *video_mask, const void *video_feats, int64_t n_rows, int64_t batch,
                                   int64_t seq_len, void *inputs_embeds, void *stream) {
    if (!d || !embed_tokens || !input_ids || !inputs_embeds || batch <= 0 || seq_len <= 0) return EILEV_E_BADARG;
    if (d->t_hidden % 8) return EILEV_E_UNSUPPORTED;
    // (the count of set mask bits == n_rows contract is validated by the host wrapper; the kernel only
    //  guards against out-of-range ranks)
    return launch_embed_scatter((const bf16 *)embed_tokens, input_ids, video_mask, (const bf16 *)video_feats, n_rows,
                                batch * seq_len, d->vocab, (bf16 *)inputs_embeds, d->t_hidden, (hipStream_t)stream);
}

// =====================================================================================================
// Stage 4/5: OPT
// =====================================================================================================
// rows of the activation buffers: a decode step of 17..32 rows keeps its activations in the 32-row row-block layout (common.h frag32_index)
static inline int64_t opt_ws_rows(int64_t M) { return (M > 16 && M < 32) ? 32 : M; }
extern "C" size_t eilev_opt_workspace_bytes(const EilevDims *d, int64_t batch, int64_t seq_len) {
    const int64_t M = opt_ws_rows(batch * (seq_len > 1 ? seq_len : 1));
    size_t b = 0;
    b += align_up((size_t)M * d->t_hidden * 2, 256) * 3;       // h, x, att
    b += align_up((size_t)M * d->t_hidden * 3 * 2, 256);        // qkv
    b += align_up((size_t)M * d->t_ffn * 2, 256);               // ffn
    b += align_up((size_t)M * 4, 256);                          // position ids
    b += kSkinnyScratch;
    b += align_up((size_t)M * (d->t_ffn > d->t_hidden ? d->t_ffn : d->t_hidden), 256) + align_up((size_t)M * 4, 256);  // fp8 activations + row scales
    return b + 256;
}

extern "C" size_t eilev_opt_kv_cache_bytes(const EilevDims *d, int64_t batch, int64_t kv_capacity) {
    return (size_t)d->t_layers * 2 * batch * kv_capacity * d->t_hidden * 2;
}

namespace {

struct OptBufs {
    bf16 *h, *x, *att, *qkv, *ffn;
    int32_t *pid;
    float *scratch;
    uint8_t *a8;      // fp8 (e4m3) copy of the current linear's input rows (EilevOptWeights.w8_act_fp8)
    float *a8_scale;  // one scale per row
};

bool carve_opt(const EilevDims *d, int64_t M, void *ws, size_t bytes, OptBufs &b) {
    Carver cv{(char *)ws, (char *)ws + bytes};
    M = opt_ws_rows(M);
    b.h = cv.take<bf16>((size_t)M * d->t_hidden);
    b.x = cv.take<bf16>((size_t)M * d->t_hidden);
    b.att = cv.take<bf16>((size_t)M * d->t_hidden);
    b.qkv = cv.take<bf16>((size_t)M * 3 * d->t_hidden);
    b.ffn = cv.take<bf16>((size_t)M * d->t_ffn);
    b.pid = cv.take<int32_t>((size_t)M);
    b.scratch = cv.take<float>(kSkinnyScratch / sizeof(float));
    b.a8 = cv.take<uint8_t>((size_t)M * (d->t_ffn > d->t_hidden ? d->t_ffn : d->t_hidden));
    b.a8_scale = cv.take<float>((size_t)M);
    return cv.ok();
}

// fp8 form of a linear (EilevOptLayerW8): bytes + per-channel scales; large-M calls expand into w->w8_expand
int use_w8(GemmArgs &g, const EilevOptWeights *w, const uint8_t *w8, const float *sc, const OptBufs &b, hipStream_t s) {
    if (!w8 || !sc) return EILEV_E_BADARG;
    if (w->w8_act_fp8 && g.M > 32 && g.K % 128 == 0 && (int64_t)g.M * g.K < 0x7fff0000ll && (int64_t)g.N * g.K < 0x7fff0000ll) {
        // configs[4] "fp8 MFMA": quantise this linear's input rows per token and run the product on the fp8 MFMA
        RC(launch_quant_rows_e4m3(g.A, g.lda, b.a8, b.a8_scale, g.M, g.K, s));
        g.A8 = b.a8;
        g.ascale = b.a8_scale;
        g.lda = g.K;
        g.W8 = w8;
        g.wscale = sc;
        g.ldw = g.K;
        return EILEV_OK;
    }
    if (g.M > 32 || g.K % 256 != 0) {
        if (!w->w8_expand || w->w8_expand_bytes < (size_t)g.N * g.K * sizeof(bf16)) return EILEV_E_WORKSPACE;
        g.w8_scratch = (bf16 *)w->w8_expand;
    }
    g.W8 = w8;
    g.wscale = sc;
    g.ldw = g.K;
    return EILEV_OK;
}

// q|k|v projection of x into b.qkv (q pre-scaled by head_dim^-0.5, hf modeling_opt.py:151)
int opt_qkv(const EilevDims *d, const EilevOptWeights *w, int l, const OptBufs &b, int64_t M, hipStream_t s, int a_frag = 0) {
    const EilevOptLayer *L = &w->layers[l];
    const int D = d->t_hidden;
    const float scaling = 1.0f / sqrtf((float)(D / d->t_heads));
    const bf16 *qw = (const bf16 *)L->q_w;
    const bool bias_fused = (const bf16 *)L->k_b == (const bf16 *)L->q_b + D && (const bf16 *)L->v_b == (const bf16 *)L->q_b + 2 * D;
    if (w->layers_w8) {
        if (!bias_fused && (L->q_b || L->k_b || L->v_b)) return EILEV_E_UNSUPPORTED;  // fp8 q|k|v is one matrix: one bias vector
        GemmArgs g = mk_gemm(b.x, D, nullptr, D, L->q_b, nullptr, 0, b.qkv, 3 * D, M, 3 * D, D, 0);
        g.scale = scaling; g.scale_cols = D; g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
        RC(use_w8(g, w, w->layers_w8[l].qkv_w8, w->layers_w8[l].qkv_scale, b, s));
        return launch_gemm(g, 5, s);
    }
    const bool fused = (const bf16 *)L->k_w == qw + (size_t)D * D && (const bf16 *)L->v_w == qw + 2 * (size_t)D * D && bias_fused;
    if (fused) {
        GemmArgs g = mk_gemm(b.x, D, L->q_w, D, L->q_b, nullptr, 0, b.qkv, 3 * D, M, 3 * D, D, 0);
        g.scale = scaling; g.scale_cols = D; g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
        if (w->layers_stream && M > 16 && M <= 32) g.Wp = (const bf16 *)w->layers_stream[l].qkv_s;
        g.a_frag = a_frag;
        return launch_gemm(g, 5, s);
    }
    GemmArgs g = mk_gemm(b.x, D, L->q_w, D, L->q_b, nullptr, 0, b.qkv, 3 * D, M, D, D, 0);
    g.scale = scaling; g.scale_cols = D; g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2; g.a_frag = a_frag;
    RC(launch_gemm(g, 5, s));
    g = mk_gemm(b.x, D, L->k_w, D, L->k_b, nullptr, 0, b.qkv + D, 3 * D, M, D, D, 0);
    g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2; g.a_frag = a_frag;
    RC(launch_gemm(g, 5, s));
    g = mk_gemm(b.x, D, L->v_w, D, L->v_b, nullptr, 0, b.qkv + 2 * D, 3 * D, M, D, D, 0);
    g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2; g.a_frag = a_frag;
    return launch_gemm(g, 5, s);
}

// out_proj + residual, LN, fc1 + ReLU, fc2 + residual (hf modeling_opt.py:178-179, 226-247)
// next_ln_w / next_ln_b (decode): the LayerNorm that consumes this block's output (the next block's self_attn_layer_norm, or
// final_layer_norm) — then b.x leaves as that LayerNorm of b.h, and both LayerNorms of the block ride on the split-K reductions of
// out_proj / fc2 (GemmArgs::ln_out)
// frag (decode steps of 17..32 rows, eilev_opt_decode_step): b.att, b.x and b.ffn in the row-block layout (common.h frag32_index); b.h stays row-major
int opt_tail(const EilevDims *d, const EilevOptWeights *w, int l, const OptBufs &b, int64_t M, hipStream_t s, const void *next_ln_w = nullptr,
             const void *next_ln_b = nullptr, int frag = 0, bool dry = false) {
    const EilevOptLayer *L = &w->layers[l];
    const EilevOptLayerW8 *Q = w->layers_w8 ? &w->layers_w8[l] : nullptr;
    const int D = d->t_hidden, Ft = d->t_ffn;
    // (decode steps of 17..32 rows: the stream-layout copies of the three matrices, where the caller packed them)
    const EilevOptLayerStream *S = (w->layers_stream && !Q && M > 16 && M <= 32) ? &w->layers_stream[l] : nullptr;
    GemmArgs g = mk_gemm(b.att, D, L->o_w, D, L->o_b, b.h, D, b.h, D, M, D, D, 0);
    g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
    g.ln_gamma = (const bf16 *)L->ln2_w; g.ln_beta = (const bf16 *)L->ln2_b; g.ln_out = b.x; g.ln_eps = d->t_eps;
    if (S) g.Wp = (const bf16 *)S->o_s;
    g.a_frag = g.ln_frag = frag;
    if (dry && !gemm_rows32_takes(g)) return EILEV_E_UNSUPPORTED;
    if (Q) RC(use_w8(g, w, Q->o_w8, Q->o_scale, b, s));
    if (!dry) RC(launch_gemm(g, 5, s));
    g = mk_gemm(b.x, D, L->fc1_w, D, L->fc1_b, nullptr, 0, b.ffn, Ft, M, Ft, D, 2);
    g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
    if (S) g.Wp = (const bf16 *)S->fc1_s;
    g.a_frag = g.c_frag = frag;
    if (dry && !gemm_rows32_takes(g)) return EILEV_E_UNSUPPORTED;
    if (Q) RC(use_w8(g, w, Q->fc1_w8, Q->fc1_scale, b, s));
    if (!dry) RC(launch_gemm(g, 5, s));
    g = mk_gemm(b.ffn, Ft, L->fc2_w, Ft, L->fc2_b, b.h, D, b.h, D, M, D, Ft, 0);
    g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
    if (S) g.Wp = (const bf16 *)S->fc2_s;
    if (next_ln_w) {
        g.ln_gamma = (const bf16 *)next_ln_w; g.ln_beta = (const bf16 *)next_ln_b; g.ln_out = b.x; g.ln_eps = d->t_eps;
    }
    g.a_frag = frag;
    g.ln_frag = next_ln_w ? frag : 0;
    if (dry) return gemm_rows32_takes(g) ? EILEV_OK : EILEV_E_UNSUPPORTED;
    if (Q) RC(use_w8(g, w, Q->fc2_w8, Q->fc2_scale, b, s));
    return launch_gemm(g, 5, s);
}

// ---- small-batch decode (M <= 8 rows): the block as 5 launches of gemv.hip + the attention split ---------------------------------------
int g_decode_frag = 1;  // probe / test switch (eilev_debug_decode_frag, probe build): 0 = row-major activations in the 17..32-row decode step
int g_decode_rows = 1;  // probe / test switch (eilev_debug_decode_rows): 0 = the MFMA weight-streaming kernels at every batch size
constexpr int kDecodeKeys = 256;  // keys per flash-decoding split (misc.hip DEC_KEYS)

// Measured (tools/beam_probe.py, OPT-2.7B, L = 960, ms per token under hipGraph): rows 1: 2.27 against 2.64 for the MFMA weight-streaming
// kernels, 2: 2.67 (~2.7), 3: 2.89 (~2.8), 5: 4.07 against 2.87 — every extra row costs the dot-product kernel a pass of LDS reads and
// v_dot2c per weight chunk, the MFMA kernels nothing up to 16 rows.  So: M <= 2 (latency mode; one sample per GPU of a strong-scaled step).
bool opt_rows_usable(const EilevDims *d, const EilevOptWeights *w, int64_t M) {
    if (!g_decode_rows || w->layers_w8 || M > 2) return false;
    const int D = d->t_hidden;
    if (!gemv_rows_ok((int)M, D, D) || !gemv_rows_ok((int)M, D, d->t_ffn) || (D / d->t_heads) % 8) return false;
    for (int l = 0; l < d->t_layers; ++l) {  // q | k | v must be ONE [3 D, D] matrix with one bias vector (the engine packs them so)
        const EilevOptLayer *L = &w->layers[l];
        const bf16 *qw = (const bf16 *)L->q_w, *qb = (const bf16 *)L->q_b;
        if ((const bf16 *)L->k_w != qw + (size_t)D * D || (const bf16 *)L->v_w != qw + 2 * (size_t)D * D || !qb || (const bf16 *)L->k_b != qb + D ||
            (const bf16 *)L->v_b != qb + 2 * D)
            return false;
    }
    return true;
}
// batch 1 (latency mode; one sample per GPU of a strong-scaled step): gemv1_kernel + attn_decode1_kernel.  eilev_debug_decode_rows(3) = off
bool opt_rows1_usable(const EilevDims *d, const EilevOptWeights *w, int64_t M, int64_t cap) {
    if (M != 1 || g_decode_rows == 3 || !opt_rows_usable(d, w, M)) return false;
    const int D = d->t_hidden;
    return gemv1_ok(3 * D, D, 1) && gemv1_ok(D, D, 0) && gemv1_ok(d->t_ffn, D, 1) && gemv1_ok(D, d->t_ffn, 0) && gemv1_ok(d->vocab, D, 1) &&
           attn_decode1_ok(1, (int)cap, D / d->t_heads);
}
// 2..4 rows (beam search, a few samples per GPU): gemvm_kernel for every K = t_hidden linear, the one-pass attention where it applies
bool opt_rowsm_usable(const EilevDims *d, const EilevOptWeights *w, int64_t M) {
    // measured (OPT-2.7B, L = 960, ms per token): 2 rows 2.06 (row-dot kernels of round 3: 2.67), 4 rows 2.51, 5 rows 3.20 against 2.87 for the MFMA
    // weight-streaming kernels, whose cost is flat up to 16 rows: every extra row costs this kernel a pass of LDS reads + dot products
    if (M < 2 || M > 4 || g_decode_rows == 3 || !g_decode_rows || w->layers_w8) return false;
    const int D = d->t_hidden;
    if (!gemvm_ok((int)M, 3 * D, D, 1) || !gemvm_ok((int)M, D, D, 0) || (D / d->t_heads) % 8 || (d->vocab & 1)) return false;
    for (int l = 0; l < d->t_layers; ++l) {  // q | k | v must be ONE [3 D, D] matrix with one bias vector (the engine packs them so)
        const EilevOptLayer *L = &w->layers[l];
        const bf16 *qw = (const bf16 *)L->q_w, *qb = (const bf16 *)L->q_b;
        if ((const bf16 *)L->k_w != qw + (size_t)D * D || (const bf16 *)L->v_w != qw + 2 * (size_t)D * D || !qb || (const bf16 *)L->k_b != qb + D ||
            (const bf16 *)L->v_b != qb + 2 * D)
            return false;
    }
    return true;
}
// block l without its attention: LayerNorm + q|k|v (before), out_proj + residual, LayerNorm + fc1 + ReLU, fc2 + residual (after)
int opt_rowsm_qkv(const EilevDims *d, const EilevOptWeights *w, int l, const OptBufs &b, int64_t M, hipStream_t s) {
    const EilevOptLayer *L = &w->layers[l];
    const int D = d->t_hidden;
    return launch_gemvm(1, b.h, D, (const bf16 *)L->ln1_w, (const bf16 *)L->ln1_b, d->t_eps, (const bf16 *)L->q_w, (const bf16 *)L->q_b, nullptr, 0, b.qkv, 3 * D, 0, (int)M,
                        3 * D, D, 0, 1.0f / sqrtf((float)(D / d->t_heads)), D, s);
}
int opt_rowsm_tail(const EilevDims *d, const EilevOptWeights *w, int l, const OptBufs &b, int64_t M, hipStream_t s) {
    const EilevOptLayer *L = &w->layers[l];
    const int D = d->t_hidden, Ft = d->t_ffn;
    RC(launch_gemvm(0, b.att, D, nullptr, nullptr, 0.f, (const bf16 *)L->o_w, (const bf16 *)L->o_b, b.h, D, b.h, D, 0, (int)M, D, D, 0, 1.0f, 0, s));
    RC(launch_gemvm(1, b.h, D, (const bf16 *)L->ln2_w, (const bf16 *)L->ln2_b, d->t_eps, (const bf16 *)L->fc1_w, (const bf16 *)L->fc1_b, nullptr, 0, b.ffn, Ft, 0, (int)M, Ft, D,
                    2, 1.0f, 0, s));
    if (gemv_rows_ok((int)M, D, Ft))  // K = t_ffn: the LDS-staged row-dot kernel (gemvm_kernel spills at K = 10240)
        return launch_gemv_rows(0, b.ffn, Ft, nullptr, nullptr, 0.f, nullptr, 0, 0, 0, (const bf16 *)L->fc2_w, (const bf16 *)L->fc2_b, b.h, D, b.h, D, 0, (int)M, D, Ft, 0, 1.0f,
                                0, s);
    GemmArgs g = mk_gemm(b.ffn, Ft, L->fc2_w, Ft, L->fc2_b, b.h, D, b.h, D, M, D, Ft, 0);
    g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
    return launch_gemm(g, 5, s);
}
int opt_rowsm_head(const EilevDims *d, const EilevOptWeights *w, const OptBufs &b, int64_t M, float *logits, hipStream_t s) {
    const int D = d->t_hidden;
    return launch_gemvm(1, b.h, D, (const bf16 *)w->final_ln_w, (const bf16 *)w->final_ln_b, d->t_eps, (const bf16 *)w->embed_tokens, nullptr, nullptr, 0, logits, d->vocab, 1,
                        (int)M, d->vocab, D, 0, 1.0f, 0, s);
}
// self_attn_layer_norm + q|k|v of block l from b.h into b.qkv (q pre-scaled)
int opt_rows_qkv(const EilevDims *d, const EilevOptWeights *w, int l, const OptBufs &b, int64_t M, hipStream_t s) {
    const EilevOptLayer *L = &w->layers[l];
    const int D = d->t_hidden;
    return launch_gemv_rows(1, b.h, D, (const bf16 *)L->ln1_w, (const bf16 *)L->ln1_b, d->t_eps, nullptr, 0, 0, 0, (const bf16 *)L->q_w, (const bf16 *)L->q_b,
                            nullptr, 0, b.qkv, 3 * D, 0, (int)M, 3 * D, D, 0, 1.0f / sqrtf((float)(D / d->t_heads)), D, s);
}
// (merge +) out_proj + residual, final_layer_norm + fc1 + ReLU, fc2 + residual: b.h -> b.h.  part != nullptr: the attention partials are
// merged in the prologue of out_proj (M <= 2: every workgroup repeats the merge); otherwise b.att holds the merged rows.
int opt_rows_tail(const EilevDims *d, const EilevOptWeights *w, int l, const OptBufs &b, int64_t M, const float *part, int nsplit, hipStream_t s) {
    const EilevOptLayer *L = &w->layers[l];
    const int D = d->t_hidden, Ft = d->t_ffn, H = d->t_heads;
    if (part) RC(launch_gemv_rows(2, nullptr, 0, nullptr, nullptr, 0.f, part, H, D / H, nsplit, (const bf16 *)L->o_w, (const bf16 *)L->o_b, b.h, D, b.h, D, 0,
                                  (int)M, D, D, 0, 1.0f, 0, s));
    else RC(launch_gemv_rows(0, b.att, D, nullptr, nullptr, 0.f, nullptr, 0, 0, 0, (const bf16 *)L->o_w, (const bf16 *)L->o_b, b.h, D, b.h, D, 0, (int)M, D, D, 0,
                             1.0f, 0, s));
    RC(launch_gemv_rows(1, b.h, D, (const bf16 *)L->ln2_w, (const bf16 *)L->ln2_b, d->t_eps, nullptr, 0, 0, 0, (const bf16 *)L->fc1_w, (const bf16 *)L->fc1_b, nullptr,
                        0, b.ffn, Ft, 0, (int)M, Ft, D, 2, 1.0f, 0, s));
    return launch_gemv_rows(0, b.ffn, Ft, nullptr, nullptr, 0.f, nullptr, 0, 0, 0, (const bf16 *)L->fc2_w, (const bf16 *)L->fc2_b, b.h, D, b.h, D, 0, (int)M, D, Ft,
                            0, 1.0f, 0, s);
}
// final_layer_norm + lm_head -> fp32 logits
int opt_rows_head(const EilevDims *d, const EilevOptWeights *w, const OptBufs &b, int64_t M, float *logits, hipStream_t s) {
    const int D = d->t_hidden;
    return launch_gemv_rows(1, b.h, D, (const bf16 *)w->final_ln_w, (const bf16 *)w->final_ln_b, d->t_eps, nullptr, 0, 0, 0, (const bf16 *)w->embed_tokens, nullptr,
                            nullptr, 0, logits, d->vocab, 1, (int)M, d->vocab, D, 0, 1.0f, 0, s);
}

}  // namespace

#ifdef EILEV_PROBES
extern "C" int eilev_debug_decode_rows(int on) { g_decode_rows = on; return 0; }
extern "C" int eilev_debug_decode_frag(int on) { g_decode_frag = on; return 0; }
extern "C" int eilev_debug_vit_head_major(int on) { g_vit_head_major = on; return 0; }
#endif
namespace {
int opt_prefill_impl(const EilevDims *d, const EilevOptWeights *w, const void *inputs_embeds, const int32_t *attn_mask, int64_t batch,
                     int64_t seq_len, void *kv_cache, int64_t kv_capacity, float *logits_last, float *logits_all, bf16 *hidden,
                     void *workspace, size_t workspace_bytes, void *stream) {
    if (!d || !w || !inputs_embeds || !attn_mask || !kv_cache || !workspace || batch <= 0 || seq_len <= 0) return EILEV_E_BADARG;
    if (seq_len > kv_capacity || seq_len > d->max_pos) return EILEV_E_BADARG;
    if (!dims_ok_opt(d)) return EILEV_E_UNSUPPORTED;
    if (workspace_bytes < eilev_opt_workspace_bytes(d, batch, seq_len)) return EILEV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->t_hidden, H = d->t_heads, hd = D / H;
    const int64_t M = batch * seq_len;
    OptBufs b;
    if (!carve_opt(d, M, workspace, workspace_bytes, b)) return EILEV_E_WORKSPACE;
    RC(launch_pos_embed((const bf16 *)inputs_embeds, (const bf16 *)w->embed_positions, attn_mask, b.pid, b.h, (int)batch,
                        (int)seq_len, D, s));
    const size_t per_layer = (size_t)2 * batch * H * kv_capacity * hd;
    const size_t hs_elems = (size_t)M * D;  // one entry of the hidden_states tuple
    for (int l = 0; l < d->t_layers; ++l) {
        const EilevOptLayer *L = &w->layers[l];
        bf16 *kc = (bf16 *)kv_cache + l * per_layer, *vc = kc + per_layer / 2;
        if (hidden) EILEV_HIP_CHECK(hipMemcpyAsync(hidden + l * hs_elems, b.h, hs_elems * sizeof(bf16), hipMemcpyDeviceToDevice, s));  // the block's input
        RC(launch_layernorm(b.h, D, (const bf16 *)L->ln1_w, (const bf16 *)L->ln1_b, b.x, D, M, D, d->t_eps, s));
        RC(opt_qkv(d, w, l, b, M, s));
        RC(launch_kv_write(b.qkv, kc, vc, (int)batch, (int)seq_len, H, hd, (int)kv_capacity, (int)seq_len, nullptr, s));
        AttnArgs a;
        a.q = b.qkv; a.k = b.qkv + D; a.v = b.qkv + 2 * D; a.o = b.att;
        a.q_bs = a.k_bs = a.v_bs = seq_len * 3 * (int64_t)D; a.o_bs = seq_len * (int64_t)D;
        a.q_hs = a.k_hs = a.v_hs = a.o_hs = hd;
        a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
        a.batch = (int)batch; a.heads = H; a.sq = (int)seq_len; a.skv = (int)seq_len; a.hd = hd; a.scale = 1.0f; a.causal = 1;
        a.key_mask = attn_mask; a.mask_ld = seq_len;
        RC(launch_attention(a, s));
        RC(opt_tail(d, w, l, b, M, s));
    }
    RC(launch_layernorm(b.h, D, (const bf16 *)w->final_ln_w, (const bf16 *)w->final_ln_b, b.x, D, M, D, d->t_eps, s));
    if (hidden) EILEV_HIP_CHECK(hipMemcpyAsync(hidden + (size_t)d->t_layers * hs_elems, b.x, hs_elems * sizeof(bf16), hipMemcpyDeviceToDevice, s));
    if (logits_all) {
        GemmArgs g = mk_gemm(b.x, D, w->embed_tokens, D, nullptr, nullptr, 0, logits_all, d->vocab, M, d->vocab, D, 0);
        g.out_f32 = 1;
        RC(launch_gemm(g, 5, s));
    }
    if (logits_last) {
        // last position of every row: a strided [batch, D] view of x
        GemmArgs g = mk_gemm(b.x + (seq_len - 1) * (int64_t)D, seq_len * (int64_t)D, w->embed_tokens, D, nullptr, nullptr, 0,
                             logits_last, d->vocab, batch, d->vocab, D, 0);
        g.out_f32 = 1; g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
        RC(launch_gemm(g, 5, s));
    }
    return EILEV_OK;
}
}  // namespace

extern "C" int eilev_opt_prefill(const EilevDims *d, const EilevOptWeights *w, const void *inputs_embeds,
                                 const int32_t *attn_mask, int64_t batch, int64_t seq_len, void *kv_cache, int64_t kv_capacity,
                                 float *logits_last, float *logits_all, void *workspace, size_t workspace_bytes, void *stream) {
    return opt_prefill_impl(d, w, inputs_embeds, attn_mask, batch, seq_len, kv_cache, kv_capacity, logits_last, logits_all, nullptr, workspace,
                            workspace_bytes, stream);
}
extern "C" int eilev_opt_prefill_debug(const EilevDims *d, const EilevOptWeights *w, const void *inputs_embeds,
                                       const int32_t *attn_mask, int64_t batch, int64_t seq_len, void *kv_cache, int64_t kv_capacity,
                                       float *logits_last, float *logits_all, void *hidden_states, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    if (!hidden_states) return EILEV_E_BADARG;
    return opt_prefill_impl(d, w, inputs_embeds, attn_mask, batch, seq_len, kv_cache, kv_capacity, logits_last, logits_all, (bf16 *)hidden_states,
                            workspace, workspace_bytes, stream);
}

extern "C" int eilev_opt_extend(const EilevDims *d, const EilevOptWeights *w, const void *inputs_embeds, const int32_t *attn_mask,
                                int64_t batch, int64_t new_len, int64_t past_len, void *kv_cache, int64_t kv_capacity,
                                float *logits_all, void *workspace, size_t workspace_bytes, void *stream) {
    if (!d || !w || !inputs_embeds || !attn_mask || !kv_cache || !logits_all || !workspace || batch <= 0 || new_len <= 0 || past_len < 0)
        return EILEV_E_BADARG;
    const int64_t total = past_len + new_len;
    if (total > kv_capacity || total > d->max_pos) return EILEV_E_BADARG;
    if (!dims_ok_opt(d)) return EILEV_E_UNSUPPORTED;
    if (workspace_bytes < eilev_opt_workspace_bytes(d, batch, total)) return EILEV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->t_hidden, H = d->t_heads, hd = D / H;
    const int64_t M = batch * new_len;
    OptBufs b;
    if (!carve_opt(d, batch * total, workspace, workspace_bytes, b)) return EILEV_E_WORKSPACE;
    RC(launch_pos_embed((const bf16 *)inputs_embeds, (const bf16 *)w->embed_positions, attn_mask, b.pid, b.h, (int)batch, (int)total, D,
                        s, (int)past_len));
    const size_t per_layer = (size_t)2 * batch * H * kv_capacity * hd;
    for (int l = 0; l < d->t_layers; ++l) {
        const EilevOptLayer *L = &w->layers[l];
        bf16 *kc = (bf16 *)kv_cache + l * per_layer, *vc = kc + per_layer / 2;
        RC(launch_layernorm(b.h, D, (const bf16 *)L->ln1_w, (const bf16 *)L->ln1_b, b.x, D, M, D, d->t_eps, s));
        RC(opt_qkv(d, w, l, b, M, s));
        RC(launch_kv_write(b.qkv, kc, vc, (int)batch, (int)new_len, H, hd, (int)kv_capacity, (int)total, nullptr, s, (int)past_len));
        // queries: the new rows (in the q|k|v buffer); keys / values: the cache, slots [0, total)
        AttnArgs a;
        a.q = b.qkv; a.k = kc; a.v = vc; a.o = b.att;
        a.q_bs = new_len * 3 * (int64_t)D; a.o_bs = new_len * (int64_t)D;
        a.k_bs = a.v_bs = (int64_t)H * kv_capacity * hd;
        a.q_hs = a.o_hs = hd; a.k_hs = a.v_hs = kv_capacity * (int64_t)hd;
        a.ldq = 3 * D; a.ldk = a.ldv = hd; a.ldo = D;
        a.batch = (int)batch; a.heads = H; a.sq = (int)new_len; a.skv = (int)total; a.hd = hd; a.scale = 1.0f; a.causal = 1;
        a.key_mask = attn_mask; a.mask_ld = total; a.dbg = 0;
        RC(launch_attention(a, s));
        RC(opt_tail(d, w, l, b, M, s));
    }
    RC(launch_layernorm(b.h, D, (const bf16 *)w->final_ln_w, (const bf16 *)w->final_ln_b, b.x, D, M, D, d->t_eps, s));
    GemmArgs g = mk_gemm(b.x, D, w->embed_tokens, D, nullptr, nullptr, 0, logits_all, d->vocab, M, d->vocab, D, 0);
    g.out_f32 = 1; g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
    return launch_gemm(g, 5, s);
}

extern "C" int eilev_greedy_select(const float *logits, int64_t batch, int64_t vocab, int32_t *state, uint8_t *finished,
                                   int64_t eos_id, int64_t pad_id, int64_t *tokens, int64_t *out_tokens, int64_t max_new,
                                   void *stream) {
    if (!logits || !state || !finished || !tokens || !out_tokens || batch <= 0) return EILEV_E_BADARG;
    return launch_select(logits, (int)batch, (int)vocab, state, finished, eos_id, pad_id, tokens, out_tokens, max_new,
                         (hipStream_t)stream);
}

extern "C" int eilev_topk_logprob(const float *logits, const float *row_score, int64_t rows, int64_t vocab, int64_t keep, float *out_val,
                                  int32_t *out_idx, void *stream) {
    if (!logits || !out_val || !out_idx || rows < 0 || vocab <= 0 || keep <= 0 || keep > vocab) return EILEV_E_BADARG;
    if (rows == 0) return EILEV_OK;
    return launch_topk_logprob(logits, row_score, (int)rows, (int)vocab, (int)keep, out_val, out_idx, (hipStream_t)stream);
}

extern "C" size_t eilev_beam_scratch_bytes(int64_t batch, int64_t beams, int64_t keep, int64_t max_new) {
    return sizeof(int64_t) * (size_t)batch * (size_t)(keep + 2 * beams) * (size_t)max_new;
}
extern "C" int eilev_beam_advance(const float *row_lp, const int32_t *row_tok, int64_t batch, int64_t beams, int64_t keep, int64_t max_new,
                                  const int32_t *state, const int64_t *eos_ids, int64_t n_eos, const float *len_pow, int len_pow_reciprocal,
                                  int early_stopping, int64_t *run_seq, float *run_score, int64_t *fin_seq, float *fin_score, int64_t *fin_len,
                                  uint8_t *finished, uint8_t *can_improve, int64_t *tokens, int32_t *anc, int64_t gen_cap, void *scratch,
                                  size_t scratch_bytes, void *stream) {
    if (!row_lp || !row_tok || !state || !len_pow || !run_seq || !run_score || !fin_seq || !fin_score || !fin_len || !finished || !can_improve ||
        !tokens || batch <= 0 || beams <= 0 || keep < beams || max_new <= 0 || n_eos < 0 || (n_eos > 0 && !eos_ids))
        return EILEV_E_BADARG;
    if (!scratch || scratch_bytes < eilev_beam_scratch_bytes(batch, beams, keep, max_new)) return EILEV_E_WORKSPACE;
    return launch_beam_advance(row_lp, row_tok, (int)batch, (int)beams, (int)keep, (int)max_new, state, eos_ids, (int)n_eos, len_pow,
                               len_pow_reciprocal, early_stopping, run_seq, run_score, fin_seq, fin_score, fin_len, finished, can_improve, tokens,
                               anc, (int)gen_cap, (int64_t *)scratch, (hipStream_t)stream);
}

extern "C" int eilev_opt_decode_step(const EilevDims *d, const EilevOptWeights *w, int64_t *tokens, int32_t *state,
                                     const int32_t *attn_mask, const int32_t *n_valid, int64_t batch, int64_t seq_len,
                                     void *kv_cache, int64_t kv_capacity, float *logits, uint8_t *finished, int64_t eos_id,
                                     int64_t pad_id, int64_t *out_tokens, int64_t max_new, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    if (!d || !w || !tokens || !state || !attn_mask || !n_valid || !kv_cache || !logits || !finished || !out_tokens || !workspace)
        return EILEV_E_BADARG;
    if (batch <= 0 || seq_len + max_new > kv_capacity + 1) return EILEV_E_BADARG;
    if (!dims_ok_opt(d)) return EILEV_E_UNSUPPORTED;
    if (workspace_bytes < eilev_opt_workspace_bytes(d, batch, 1)) return EILEV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->t_hidden, H = d->t_heads, hd = D / H;
    OptBufs b;
    if (!carve_opt(d, batch, workspace, workspace_bytes, b)) return EILEV_E_WORKSPACE;
    RC(launch_decode_embed((const bf16 *)w->embed_tokens, (const bf16 *)w->embed_positions, tokens, n_valid, state, d->vocab,
                           d->max_pos + 1, b.h, (int)batch, D, s));
    const size_t per_layer = (size_t)2 * batch * H * kv_capacity * hd;
    if (opt_rows1_usable(d, w, batch, kv_capacity)) {  // ONE row (round 4): register-resident activations, one-pass attention: 5 launches per block
        for (int l = 0; l < d->t_layers; ++l) {
            const EilevOptLayer *L = &w->layers[l];
            bf16 *kc = (bf16 *)kv_cache + l * per_layer, *vc = kc + per_layer / 2;
            const int Ft = d->t_ffn;
            RC(launch_gemv1(1, b.h, (const bf16 *)L->ln1_w, (const bf16 *)L->ln1_b, d->t_eps, (const bf16 *)L->q_w, (const bf16 *)L->q_b, nullptr, b.qkv, 0, 3 * D, D, 0,
                            1.0f / sqrtf((float)hd), D, s));
            if (hd == 80 && g_decode_rows != 5) {  // 128-key splits over all CUs, merged in out_proj's prologue (eilev_debug_decode_rows(5): one workgroup per head)
                float *part = b.scratch + kSkinnyScratch / 2 / sizeof(float);
                RC(launch_attn_decode_part(b.qkv, kc, vc, part, kSkinnyScratch / 2, attn_mask, state, 1, (int)seq_len, (int)kv_capacity, H, hd, s));
                RC(launch_gemv1(2, nullptr, nullptr, nullptr, 0.f, (const bf16 *)L->o_w, (const bf16 *)L->o_b, b.h, b.h, 0, D, D, 0, 1.0f, 0, s, part, H, hd,
                                attn_decode_part_splits((int)kv_capacity)));
            } else {
                RC(launch_attn_decode1(b.qkv, kc, vc, b.att, attn_mask, state, 1, (int)seq_len, (int)kv_capacity, H, hd, s));
                RC(launch_gemv1(0, b.att, nullptr, nullptr, 0.f, (const bf16 *)L->o_w, (const bf16 *)L->o_b, b.h, b.h, 0, D, D, 0, 1.0f, 0, s));
            }
            RC(launch_gemv1(1, b.h, (const bf16 *)L->ln2_w, (const bf16 *)L->ln2_b, d->t_eps, (const bf16 *)L->fc1_w, (const bf16 *)L->fc1_b, nullptr, b.ffn, 0, Ft, D, 2,
                            1.0f, 0, s));
            RC(launch_gemv1(0, b.ffn, nullptr, nullptr, 0.f, (const bf16 *)L->fc2_w, (const bf16 *)L->fc2_b, b.h, b.h, 0, D, Ft, 0, 1.0f, 0, s));
        }
        RC(launch_gemv1(1, b.h, (const bf16 *)w->final_ln_w, (const bf16 *)w->final_ln_b, d->t_eps, (const bf16 *)w->embed_tokens, nullptr, nullptr, logits, 1, d->vocab,
                        D, 0, 1.0f, 0, s));
        return launch_select(logits, 1, d->vocab, state, finished, eos_id, pad_id, tokens, out_tokens, max_new, s);
    }
    if (opt_rowsm_usable(d, w, batch)) {  // 2..4 rows (round 4)
        const bool one_pass = attn_decode1_ok((int)batch, (int)kv_capacity, hd);
        for (int l = 0; l < d->t_layers; ++l) {
            bf16 *kc = (bf16 *)kv_cache + l * per_layer, *vc = kc + per_layer / 2;
            RC(opt_rowsm_qkv(d, w, l, b, batch, s));
            if (one_pass) RC(launch_attn_decode1(b.qkv, kc, vc, b.att, attn_mask, state, (int)batch, (int)seq_len, (int)kv_capacity, H, hd, s));
            else RC(launch_attn_decode(b.qkv, kc, vc, b.att, attn_mask, state, (int)batch, (int)seq_len, (int)kv_capacity, H, hd,
                                       b.scratch + kSkinnyScratch / 2 / sizeof(float), kSkinnyScratch / 2, s, 0, nullptr, 0, 0, 1));
            RC(opt_rowsm_tail(d, w, l, b, batch, s));
        }
        RC(opt_rowsm_head(d, w, b, batch, logits, s));
        return launch_select(logits, (int)batch, d->vocab, state, finished, eos_id, pad_id, tokens, out_tokens, max_new, s);
    }
    if (opt_rows_usable(d, w, batch)) {  // M <= 8: row-dot kernels with LayerNorm / merge in their prologues (gemv.hip): 5 launches + attention per block
        float *part = b.scratch + kSkinnyScratch / 2 / sizeof(float);
        const int nsplit = (int)((kv_capacity + kDecodeKeys - 1) / kDecodeKeys);
        const bool fuse_merge = batch <= 2;
        for (int l = 0; l < d->t_layers; ++l) {
            bf16 *kc = (bf16 *)kv_cache + l * per_layer, *vc = kc + per_layer / 2;
            RC(opt_rows_qkv(d, w, l, b, batch, s));
            RC(launch_attn_decode(b.qkv, kc, vc, fuse_merge ? nullptr : b.att, attn_mask, state, (int)batch, (int)seq_len, (int)kv_capacity, H, hd, part,
                                  kSkinnyScratch / 2, s, 0, nullptr, 0, 0, 1));
            RC(opt_rows_tail(d, w, l, b, batch, fuse_merge ? part : nullptr, nsplit, s));
        }
        RC(opt_rows_head(d, w, b, batch, logits, s));
        return launch_select(logits, (int)batch, d->vocab, state, finished, eos_id, pad_id, tokens, out_tokens, max_new, s);
    }
    // 17..32 rows (round 5): the activations between the kernels of a block (attention rows, LayerNorm rows, fc1 rows) in the row-block layout
    // (common.h frag32_index) when every linear of the block runs on gemm_rows32_kernel and the attention on attn_decode_loop_kernel — a
    // dry run of the block's launches decides; the first q|k|v projection reads the row-major LayerNorm of the embedding rows
    GemmArgs gh = mk_gemm(b.x, D, w->embed_tokens, D, nullptr, nullptr, 0, logits, d->vocab, batch, d->vocab, D, 0);
    gh.out_f32 = 1; gh.scratch = b.scratch; gh.scratch_bytes = kSkinnyScratch / 2;
    if (w->lm_head_stream && batch > 16 && batch <= 32) gh.Wp = (const bf16 *)w->lm_head_stream;
    int frag = 0;
    if (g_decode_frag && batch > 16 && batch <= 32 && !w->layers_w8 && D % 32 == 0 && d->t_ffn % 32 == 0 &&
        attn_decode_loop_ok((int)batch, H, hd, (int)kv_capacity, false, b.att, state, 1, nullptr)) {
        gh.a_frag = 1;
        GemmArgs gq = mk_gemm(b.x, D, w->layers[0].q_w, D, w->layers[0].q_b, nullptr, 0, b.qkv, 3 * D, batch, D, D, 0);  // q, k, v one by one
        gq.scratch = b.scratch; gq.scratch_bytes = kSkinnyScratch / 2; gq.a_frag = 1;
        GemmArgs gq3 = gq;  // ... or as one [3 D, D] matrix (opt_qkv decides per block)
        gq3.N = 3 * D;
        frag = gemm_rows32_takes(gh) && gemm_rows32_takes(gq) && gemm_rows32_takes(gq3) &&
               opt_tail(d, w, 0, b, batch, s, w->final_ln_w, w->final_ln_b, 1, true) == EILEV_OK;
        gh.a_frag = frag;
    }
    for (int l = 0; l < d->t_layers; ++l) {
        const EilevOptLayer *L = &w->layers[l];
        bf16 *kc = (bf16 *)kv_cache + l * per_layer, *vc = kc + per_layer / 2;
        if (l == 0) RC(launch_layernorm(b.h, D, (const bf16 *)L->ln1_w, (const bf16 *)L->ln1_b, b.x, D, batch, D, d->t_eps, s));
        RC(opt_qkv(d, w, l, b, batch, s, l > 0 ? frag : 0));
        // the new token's K / V go into the cache inside the attention kernel (fuse_new): one launch less per layer
        RC(launch_attn_decode(b.qkv, kc, vc, b.att, attn_mask, state, (int)batch, (int)seq_len, (int)kv_capacity, H, hd,
                              b.scratch + kSkinnyScratch / 2 / sizeof(float), kSkinnyScratch / 2, s, 0, nullptr, 0, 0, 1, nullptr, nullptr, nullptr, 1, 0, frag));
        // the block's output goes straight into the LayerNorm that reads it next (the next block's, or final_layer_norm): b.x
        const bool last = l + 1 == d->t_layers;
        RC(opt_tail(d, w, l, b, batch, s, last ? w->final_ln_w : w->layers[l + 1].ln1_w, last ? w->final_ln_b : w->layers[l + 1].ln1_b, frag));
    }
    RC(launch_gemm(gh, 5, s));
    return launch_select(logits, (int)batch, d->vocab, state, finished, eos_id, pad_id, tokens, out_tokens, max_new, s);
}

namespace {
__global__ void bump_step_kernel(int32_t *state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[0] += 1;
}
}  // namespace

// One decode step of beam search WITHOUT moving the KV cache (include/eilev.h): the prompt's keys / values stay in the prefill cache, one
// row per SAMPLE; the generated tokens' in a generation cache, one row per beam SLOT; `ancestors[g][r]` names the slot that holds the g-th
// generated token of the hypothesis now living in row r.  (Before: torch index_select of the whole cache per step — 1.6 GB at 5 beams.)
extern "C" int eilev_opt_decode_step_beam(const EilevDims *d, const EilevOptWeights *w, const int64_t *tokens, int32_t *state,
                                          const int32_t *attn_mask, const int32_t *n_valid, int64_t rows, int64_t beams, int64_t seq_len,
                                          const void *kv_prompt, void *kv_gen, int64_t gen_capacity, const int32_t *ancestors, float *logits,
                                          void *workspace, size_t workspace_bytes, void *stream) {
    if (!d || !w || !tokens || !state || !attn_mask || !n_valid || !kv_prompt || !kv_gen || !ancestors || !logits || !workspace) return EILEV_E_BADARG;
    if (rows <= 0 || beams <= 0 || rows % beams || seq_len <= 0 || gen_capacity <= 0 || rows > 32) return EILEV_E_BADARG;
    if (!dims_ok_opt(d)) return EILEV_E_UNSUPPORTED;
    if (workspace_bytes < eilev_opt_workspace_bytes(d, rows, 1)) return EILEV_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->t_hidden, H = d->t_heads, hd = D / H;
    const int64_t samples = rows / beams;
    OptBufs b;
    if (!carve_opt(d, rows, workspace, workspace_bytes, b)) return EILEV_E_WORKSPACE;
    if (attn_decode_scratch_bytes((int)rows, H, hd, (int)(seq_len + gen_capacity)) > kSkinnyScratch / 2) return EILEV_E_WORKSPACE;
    RC(launch_decode_embed((const bf16 *)w->embed_tokens, (const bf16 *)w->embed_positions, tokens, n_valid, state, d->vocab, d->max_pos + 1, b.h,
                           (int)rows, D, s));
    const size_t per_p = (size_t)2 * samples * H * seq_len * hd, per_g = (size_t)2 * rows * H * gen_capacity * hd;
    if (opt_rowsm_usable(d, w, rows)) {  // 2..4 rows (e.g. 4 beams of one sample; 5 beams take the MFMA row kernels), round 4: gemvm_kernel around the beam attention
        for (int l = 0; l < d->t_layers; ++l) {
            const bf16 *kc = (const bf16 *)kv_prompt + l * per_p, *vc = kc + per_p / 2;
            bf16 *kg = (bf16 *)kv_gen + l * per_g, *vg = kg + per_g / 2;
            RC(opt_rowsm_qkv(d, w, l, b, rows, s));
            RC(launch_attn_decode(b.qkv, kc, vc, b.att, attn_mask, state, (int)rows, (int)seq_len, (int)seq_len, H, hd,
                                  b.scratch + kSkinnyScratch / 2 / sizeof(float), kSkinnyScratch / 2, s, 0, nullptr, 0, 0, 1, kg, vg, ancestors, (int)beams,
                                  (int)gen_capacity));
            RC(opt_rowsm_tail(d, w, l, b, rows, s));
        }
        RC(opt_rowsm_head(d, w, b, rows, logits, s));
        bump_step_kernel<<<1, 64, 0, s>>>(state);
        EILEV_LAUNCH_CHECK();
        return EILEV_OK;
    }
    if (opt_rows_usable(d, w, rows)) {  // <= 8 rows (the sample script: 5 beams of one sample): the small-batch block of gemv.hip
        float *part = b.scratch + kSkinnyScratch / 2 / sizeof(float);
        const int nsplit = (int)((seq_len + gen_capacity + kDecodeKeys - 1) / kDecodeKeys);
        const bool fuse_merge = rows <= 2;
        for (int l = 0; l < d->t_layers; ++l) {
            const bf16 *kc = (const bf16 *)kv_prompt + l * per_p, *vc = kc + per_p / 2;
            bf16 *kg = (bf16 *)kv_gen + l * per_g, *vg = kg + per_g / 2;
            RC(opt_rows_qkv(d, w, l, b, rows, s));
            RC(launch_attn_decode(b.qkv, kc, vc, fuse_merge ? nullptr : b.att, attn_mask, state, (int)rows, (int)seq_len, (int)seq_len, H, hd, part,
                                  kSkinnyScratch / 2, s, 0, nullptr, 0, 0, 1, kg, vg, ancestors, (int)beams, (int)gen_capacity));
            RC(opt_rows_tail(d, w, l, b, rows, fuse_merge ? part : nullptr, nsplit, s));
        }
        RC(opt_rows_head(d, w, b, rows, logits, s));
        bump_step_kernel<<<1, 64, 0, s>>>(state);
        EILEV_LAUNCH_CHECK();
        return EILEV_OK;
    }
    for (int l = 0; l < d->t_layers; ++l) {
        const EilevOptLayer *L = &w->layers[l];
        const bf16 *kc = (const bf16 *)kv_prompt + l * per_p, *vc = kc + per_p / 2;
        bf16 *kg = (bf16 *)kv_gen + l * per_g, *vg = kg + per_g / 2;
        if (l == 0) RC(launch_layernorm(b.h, D, (const bf16 *)L->ln1_w, (const bf16 *)L->ln1_b, b.x, D, rows, D, d->t_eps, s));
        RC(opt_qkv(d, w, l, b, rows, s));
        RC(launch_attn_decode(b.qkv, kc, vc, b.att, attn_mask, state, (int)rows, (int)seq_len, (int)seq_len, H, hd,
                              b.scratch + kSkinnyScratch / 2 / sizeof(float), kSkinnyScratch / 2, s, 0, nullptr, 0, 0, 1, kg, vg, ancestors, (int)beams,
                              (int)gen_capacity));
        const bool last = l + 1 == d->t_layers;
        RC(opt_tail(d, w, l, b, rows, s, last ? w->final_ln_w : w->layers[l + 1].ln1_w, last ? w->final_ln_b : w->layers[l + 1].ln1_b));
    }
    GemmArgs g = mk_gemm(b.x, D, w->embed_tokens, D, nullptr, nullptr, 0, logits, d->vocab, rows, d->vocab, D, 0);
    g.out_f32 = 1; g.scratch = b.scratch; g.scratch_bytes = kSkinnyScratch / 2;
    if (w->lm_head_stream && rows > 16 && rows <= 32) g.Wp = (const bf16 *)w->lm_head_stream;
    RC(launch_gemm(g, 5, s));
    bump_step_kernel<<<1, 64, 0, s>>>(state);  // the step counter lives on the device: a captured step replays for every step
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

// =====================================================================================================
// Building blocks (unit parity tests, roofline probe)
// =====================================================================================================
namespace {
// eilev_attention_probs: one wave per (batch, head, query row); a lane owns keys lane, lane + 64, ... (up to 64 per lane = 4096 keys)
__global__ void __launch_bounds__(256) attn_probs_masked_kernel(const bf16 *__restrict__ q, const bf16 *__restrict__ k, bf16 *__restrict__ probs, int heads,
                                                                int sq, int skv, int hd, int64_t ldq, int64_t ldk, float scale, int causal,
                                                                const int32_t *__restrict__ key_mask, const float *__restrict__ rel_tab, int64_t rel_stride,
                                                                int rel_off, int rel_n) {
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= sq) return;
    const bf16 *qr = q + ((int64_t)b * sq + i) * ldq + h * hd;
    bf16 *out = probs + (((int64_t)b * heads + h) * sq + i) * skv;
    const int nt = (skv + 63) / 64;
    float mx = -INFINITY;
    // two passes over the keys (scores recomputed): no per-lane array of 64 scores
    for (int pass = 0; pass < 2; ++pass) {
        float sum = 0.0f;
        float keep_mx = mx;
        for (int t = 0; t < nt; ++t) {
            const int j = lane + 64 * t;
            float s = -INFINITY;
            if (j < skv && (!causal || j <= i + (skv - sq)) && (!key_mask || key_mask[(int64_t)b * skv + j] != 0)) {
                const bf16 *kr = k + ((int64_t)b * skv + j) * ldk + h * hd;
                s = 0.0f;
                for (int e = 0; e < hd; e += 8) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8 *>(qr + e), ka = *reinterpret_cast<const bf16x8 *>(kr + e);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = fmaf((float)qa[u], (float)ka[u], s);
                }
                s *= scale;
                if (rel_tab) s += rel_tab[(int64_t)h * rel_stride + min(max(j - i - (skv - sq) + rel_off, 0), rel_n - 1)];
            }
            if (pass == 0) mx = fmaxf(mx, s);
            else sum += s == -INFINITY ? 0.0f : __expf(s - keep_mx);
        }
        if (pass == 0) {
            mx = wave_max(mx);
            continue;
        }
        sum = wave_sum(sum);
        for (int t = 0; t < nt; ++t) {  // third walk: write (scores recomputed once more: this is the debug path)
            const int j = lane + 64 * t;
            if (j >= skv) continue;
            float p = 0.0f;
            if ((!causal || j <= i + (skv - sq)) && (!key_mask || key_mask[(int64_t)b * skv + j] != 0)) {
                const bf16 *kr = k + ((int64_t)b * skv + j) * ldk + h * hd;
                float s = 0.0f;
                for (int e = 0; e < hd; e += 8) {
                    const bf16x8 qa = *reinterpret_cast<const bf16x8 *>(qr + e), ka = *reinterpret_cast<const bf16x8 *>(kr + e);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = fmaf((float)qa[u], (float)ka[u], s);
                }
                s *= scale;
                if (rel_tab) s += rel_tab[(int64_t)h * rel_stride + min(max(j - i - (skv - sq) + rel_off, 0), rel_n - 1)];
                p = sum > 0.0f ? __expf(s - keep_mx) / sum : 0.0f;
            }
            out[j] = (bf16)p;
        }
    }
}
}  // namespace

extern "C" int eilev_attention_probs(const void *q, const void *k, void *probs, int64_t batch, int64_t heads, int64_t sq, int64_t skv, int64_t head_dim,
                                     int64_t ldq, int64_t ldk, float scale, int causal, const int32_t *key_mask, const float *rel_tab,
                                     int64_t rel_stride, int64_t rel_off, int64_t rel_n, void *stream) {
    if (!q || !k || !probs || batch < 0 || heads <= 0 || sq < 0 || skv <= 0 || skv > 4096) return EILEV_E_BADARG;
    if (rel_tab && (rel_n <= 0 || rel_stride < rel_n)) return EILEV_E_BADARG;
    if ((head_dim & 7) || (ldq & 7) || (ldk & 7) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15)) return EILEV_E_UNSUPPORTED;
    if (batch == 0 || sq == 0) return EILEV_OK;
    attn_probs_masked_kernel<<<dim3((unsigned)(batch * heads), (unsigned)((sq + 3) / 4)), 256, 0, (hipStream_t)stream>>>(
        (const bf16 *)q, (const bf16 *)k, (bf16 *)probs, (int)heads, (int)sq, (int)skv, (int)head_dim, ldq, ldk, scale, causal, key_mask, rel_tab, rel_stride,
        (int)rel_off, (int)rel_n);
    EILEV_LAUNCH_CHECK();
    return EILEV_OK;
}

extern "C" int eilev_linear(const void *a, const void *w, const void *bias, const void *residual, void *c, int64_t m,
                            int64_t n, int64_t k, int epilogue, int out_f32, void *stream) {
    if (!a || !w || !c || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffff || n > 0x7fffffff) return EILEV_E_BADARG;
    GemmArgs g = mk_gemm((const bf16 *)a, k, w, k, bias, (const bf16 *)residual, n, c, n, m, (int)n, (int)k, epilogue);
    g.out_f32 = out_f32;
    return launch_gemm(g, 5, (hipStream_t)stream);
}

extern "C" size_t eilev_linear_w8_scratch_bytes(int64_t m, int64_t n, int64_t k) {
    const size_t expand = m > 32 || k % 256 != 0 ? (size_t)n * k * sizeof(bf16) : 0;       // large M: weights expanded to bf16
    const size_t partials = m <= 32 ? (size_t)64 * 32 * n * sizeof(float) : 0;             // small M: split-K partial sums
    return expand > partials ? expand : partials;
}

extern "C" int eilev_linear_w8(const void *a, const uint8_t *w8, const float *w_scale, const void *bias, const void *residual, void *c,
                               int64_t m, int64_t n, int64_t k, int epilogue, int out_f32, void *scratch, size_t scratch_bytes, void *stream) {
    if (!a || !w8 || !w_scale || !c || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffff || n > 0x7fffffff) return EILEV_E_BADARG;
    GemmArgs g = mk_gemm((const bf16 *)a, k, nullptr, k, bias, (const bf16 *)residual, n, c, n, m, (int)n, (int)k, epilogue);
    g.out_f32 = out_f32;
    g.W8 = w8;
    g.wscale = w_scale;
    if (m > 32 || k % 256 != 0) {
        if (!scratch || scratch_bytes < (size_t)n * k * sizeof(bf16)) return EILEV_E_WORKSPACE;
        g.w8_scratch = (bf16 *)scratch;
    } else if (scratch) {
        g.scratch = (float *)scratch;
        g.scratch_bytes = scratch_bytes;
    }
    return launch_gemm(g, 5, (hipStream_t)stream);
}

// LayerNorm folded into the consuming linear: the stages the folded ViT blocks are made of (include/eilev.h, ABI version 9)
extern "C" int eilev_fold_layernorm(const void *w, const void *gamma, const void *beta, const void *bias, int64_t n, int64_t k, void *w_out,
                                    float *csum, void *bias_out, void *stream) {
    if (n > 0x7fffffff || k > 0x7fffffff) return EILEV_E_BADARG;
    return launch_fold_layernorm((const bf16 *)w, (const bf16 *)gamma, (const bf16 *)beta, (const bf16 *)bias, (int)n, (int)k, (bf16 *)w_out,
                                 csum, (bf16 *)bias_out, (hipStream_t)stream);
}

extern "C" int eilev_linear_stats(const void *a, const void *w, const void *bias, const void *residual, void *c, int64_t m, int64_t n,
                                  int64_t k, float *stats, void *stream) {
    if (!a || !w || !c || !residual || !stats || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffff || n > 0x7fffffff) return EILEV_E_BADARG;
    GemmArgs g = mk_gemm((const bf16 *)a, k, w, k, bias, (const bf16 *)residual, n, c, n, m, (int)n, (int)k, 0);
    g.stat_out = stats;
    g.stat_ld = m;
    return launch_gemm(g, 5, (hipStream_t)stream);
}

extern "C" int eilev_ln_finalize(const float *stats, int64_t m, int64_t n, float eps, float *ln_rows, void *stream) {
    if (n > 0x7fffffff) return EILEV_E_BADARG;
    return launch_ln_finalize(stats, (int)((n + 63) / 64), m, (int)n, eps, ln_rows, (hipStream_t)stream);
}

extern "C" int eilev_linear_lnfold(const void *a, const void *w_f, const void *bias_f, const float *csum, const float *ln_rows, void *c,
                                   int64_t m, int64_t n, int64_t k, int epilogue, void *stream) {
    if (!a || !w_f || !csum || !ln_rows || !c || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffff || n > 0x7fffffff) return EILEV_E_BADARG;
    if (epilogue != 0 && epilogue != 1) return EILEV_E_UNSUPPORTED;
    GemmArgs g = mk_gemm((const bf16 *)a, k, w_f, k, bias_f, nullptr, 0, c, n, m, (int)n, (int)k, epilogue);
    g.ln_rows = ln_rows;
    g.ln_csum = csum;
    return launch_gemm(g, 5, (hipStream_t)stream);
}

extern "C" int eilev_quant_rows_e4m3(const void *x, uint8_t *q, float *scale, int64_t rows, int64_t cols, void *stream) {
    if (!x || !q || !scale || rows < 0 || cols <= 0 || cols > 0x7fffffff) return EILEV_E_BADARG;
    return launch_quant_rows_e4m3((const bf16 *)x, cols, q, scale, rows, (int)cols, (hipStream_t)stream);
}

extern "C" int eilev_linear_a8w8(const uint8_t *a8, const float *a_scale, const uint8_t *w8, const float *w_scale, const void *bias,
                                 const void *residual, void *c, int64_t m, int64_t n, int64_t k, int epilogue, int out_f32, void *stream) {
    if (!a8 || !a_scale || !w8 || !w_scale || !c || m < 0 || n <= 0 || k <= 0 || m > 0x7fffffff || n > 0x7fffffff) return EILEV_E_BADARG;
    if (epilogue != 0 && epilogue != 2) return EILEV_E_UNSUPPORTED;
    GemmArgs g = mk_gemm(nullptr, k, nullptr, k, bias, (const bf16 *)residual, n, c, n, m, (int)n, (int)k, epilogue);
    g.out_f32 = out_f32;
    g.A8 = a8; g.ascale = a_scale; g.W8 = w8; g.wscale = w_scale;
    return launch_gemm(g, 5, (hipStream_t)stream);
}

extern "C" int eilev_linear_rows(const void *x, const void *ln_gamma, const void *ln_beta, float eps, const void *w, const void *bias,
                                 const void *residual, void *c, int64_t m, int64_t n, int64_t k, int epilogue, int out_f32, void *stream) {
    if (!x || !w || !c || m <= 0 || n <= 0 || k <= 0 || n > 0x7fffffff || k > 0x7fffffff || (ln_gamma != nullptr) != (ln_beta != nullptr)) return EILEV_E_BADARG;
    if (m > 8 || !gemv_rows_ok((int)m, (int)n, (int)k) || (epilogue != 0 && epilogue != 2)) return EILEV_E_UNSUPPORTED;
    if (m == 1 && g_decode_rows != 3 && gemv1_ok((int)n, (int)k, ln_gamma ? 1 : 0) && !(((uintptr_t)bias | (uintptr_t)residual) & 3) && !(n & 1))  // round 4: one row (bias / residual are fetched as 32-bit pairs: even n only)
        return launch_gemv1(ln_gamma ? 1 : 0, (const bf16 *)x, (const bf16 *)ln_gamma, (const bf16 *)ln_beta, eps, (const bf16 *)w, (const bf16 *)bias,
                            (const bf16 *)residual, c, out_f32, (int)n, (int)k, epilogue, 1.0f, 0, (hipStream_t)stream);
    if (m >= 2 && g_decode_rows != 3 && gemvm_ok((int)m, (int)n, (int)k, ln_gamma ? 1 : 0) && !(((uintptr_t)bias | (uintptr_t)residual) & 3) && !(n & 1))
        return launch_gemvm(ln_gamma ? 1 : 0, (const bf16 *)x, k, (const bf16 *)ln_gamma, (const bf16 *)ln_beta, eps, (const bf16 *)w, (const bf16 *)bias,
                            (const bf16 *)residual, n, c, n, out_f32, (int)m, (int)n, (int)k, epilogue, 1.0f, 0, (hipStream_t)stream);
    return launch_gemv_rows(ln_gamma ? 1 : 0, (const bf16 *)x, k, (const bf16 *)ln_gamma, (const bf16 *)ln_beta, eps, nullptr, 0, 0, 0, (const bf16 *)w,
                            (const bf16 *)bias, (const bf16 *)residual, n, c, n, out_f32, (int)m, (int)n, (int)k, epilogue, 1.0f, 0, (hipStream_t)stream);
}

extern "C" int eilev_layernorm(const void *x, const void *gamma, const void *beta, void *y, int64_t rows, int64_t cols,
                               float eps, void *stream) {
    return launch_layernorm((const bf16 *)x, cols, (const bf16 *)gamma, (const bf16 *)beta, (bf16 *)y, cols, rows, (int)cols,
                            eps, (hipStream_t)stream);
}

extern "C" int eilev_attention(const void *q, const void *k, const void *v, void *o, int64_t batch, int64_t heads, int64_t sq,
                               int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, float scale, int causal,
                               const int32_t *key_mask, void *stream) {
    AttnArgs a;
    a.q = (const bf16 *)q; a.k = (const bf16 *)k; a.v = (const bf16 *)v; a.o = (bf16 *)o;
    a.q_bs = sq * ldq; a.k_bs = skv * ldk; a.v_bs = skv * ldv; a.o_bs = sq * heads * head_dim;
    a.q_hs = a.k_hs = a.v_hs = a.o_hs = head_dim;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = heads * head_dim;
    a.batch = (int)batch; a.heads = (int)heads; a.sq = (int)sq; a.skv = (int)skv; a.hd = (int)head_dim; a.scale = scale;
    a.causal = causal; a.key_mask = key_mask; a.mask_ld = skv;
    return launch_attention(a, (hipStream_t)stream);
}

// =====================================================================================================
// Encoder-decoder language model: flan-t5 (hf models/t5/modeling_t5.py; include/eilev.h "encoder-decoder")
// =====================================================================================================
namespace {
bool dims_ok_t5(const EilevT5Dims *d) {
    return d->d_model % 8 == 0 && d->d_kv % 8 == 0 && d->d_kv <= 128 && d->heads > 0 && d->d_ff % 8 == 0 && d->d_model <= 4096 &&
           d->rel_buckets >= 4 && d->rel_max_dist > d->rel_buckets / 4;
}
struct T5Bufs {
    bf16 *h, *x, *qkv, *att, *ff, *gate;
    float *rel, *scratch;
    int64_t rel_n;
};
bool carve_t5(const EilevT5Dims *d, int64_t M, int64_t rel_n, void *ws, size_t bytes, T5Bufs &b) {
    const size_t I = (size_t)d->heads * d->d_kv;
    Carver cv{(char *)ws, (char *)ws + bytes};
    b.h = cv.take<bf16>((size_t)M * d->d_model);
    b.x = cv.take<bf16>((size_t)M * d->d_model);
    b.qkv = cv.take<bf16>((size_t)M * 3 * I);
    b.att = cv.take<bf16>((size_t)M * I);
    b.ff = cv.take<bf16>((size_t)M * 2 * d->d_ff);
    b.gate = cv.take<bf16>((size_t)M * d->d_ff);
    b.rel = cv.take<float>((size_t)d->heads * rel_n);
    b.scratch = cv.take<float>(kSkinnyScratch / sizeof(float));
    b.rel_n = rel_n;
    return cv.ok();
}
GemmArgs t5_gemm(const T5Bufs &b, const bf16 *A, int64_t lda, const void *W, int64_t ldw, const bf16 *resid, int64_t ldr, void *Cp,
                 int64_t ldc, int64_t M, int N, int K) {
    GemmArgs g = mk_gemm(A, lda, W, ldw, nullptr, resid, ldr, Cp, ldc, M, N, K, 0);
    g.scratch = b.scratch;
    g.scratch_bytes = kSkinnyScratch / 2;  // the other half holds the decode-attention partials
    return g;
}
// up to three projections of x with a shared input: one GEMM when the weights sit back to back in memory (the engine packs them)
int t5_proj(const T5Bufs &b, const bf16 *x, int D, const void *w0, const void *w1, const void *w2, int n_each, bf16 *out, int64_t ldo,
            int64_t M, hipStream_t s) {
    const bf16 *p0 = (const bf16 *)w0, *p1 = (const bf16 *)w1, *p2 = (const bf16 *)w2;
    const int cnt = 1 + (p1 != nullptr) + (p2 != nullptr);
    const bool fused = (cnt < 2 || p1 == p0 + (size_t)n_each * D) && (cnt < 3 || p2 == p0 + 2 * (size_t)n_each * D);
    if (fused) return launch_gemm(t5_gemm(b, x, D, w0, D, nullptr, 0, out, ldo, M, cnt * n_each, D), 5, s);
    const bf16 *ws[3] = {p0, p1, p2};
    for (int i = 0; i < cnt; ++i) RC(launch_gemm(t5_gemm(b, x, D, ws[i], D, nullptr, 0, out + (size_t)i * n_each, ldo, M, n_each, D), 5, s));
    return EILEV_OK;
}
// h += wo(gelu_new(wi_0 x) * wi_1 x) with x = rmsnorm(h)   [T5LayerFF :126-141, T5DenseGatedActDense :97-124]
// normed_in: b.x already holds RMSNorm_ff(h) (written by the reduce of the GEMV before); next_ln: the RMSNorm weight whose output of the new h
// the wo GEMV's reduce should leave in b.x (decode steps: see t5_decode_impl)
int t5_ff(const EilevT5Dims *d, const EilevT5Layer *L, const T5Bufs &b, int64_t M, hipStream_t s, bool normed_in = false, const void *next_ln = nullptr) {
    const int D = d->d_model, F = d->d_ff;
    if (!normed_in) RC(launch_rmsnorm(b.h, D, (const bf16 *)L->ln_ff, b.x, D, M, D, d->eps, s));
    RC(t5_proj(b, b.x, D, L->wi0_w, L->wi1_w, nullptr, F, b.ff, 2 * F, M, s));
    RC(launch_gated_gelu(b.ff, 2 * F, b.gate, M, F, s));
    GemmArgs g = t5_gemm(b, b.gate, F, L->wo_w, F, b.h, D, b.h, D, M, D, F);
    if (next_ln) { g.ln_gamma = (const bf16 *)next_ln; g.ln_beta = nullptr; g.ln_out = b.x; g.ln_eps = d->eps; }
    return launch_gemm(g, 5, s);
}
}  // namespace

extern "C" size_t eilev_t5_workspace_bytes(const EilevT5Dims *d, int64_t batch, int64_t rows, int64_t kv_len) {
    const size_t M = (size_t)batch * rows, I = (size_t)d->heads * d->d_kv;
    const size_t rel_n = (size_t)(rows + kv_len + 1);
    return (M * (2 * (size_t)d->d_model + 4 * I + 3 * (size_t)d->d_ff)) * sizeof(bf16) + d->heads * rel_n * sizeof(float) + kSkinnyScratch +
           16 * 256;
}

// hidden_out (nullable): (enc_layers + 1, batch, enc_len, D) = hf T5Stack's hidden_states tuple: every block's input, then the output of
// final_layer_norm (modeling_t5.py T5Stack.forward: all_hidden_states)
static int t5_encode_impl(const EilevT5Dims *d, const EilevT5Weights *w, const void *inputs_embeds, const int32_t *attn_mask,
                          int64_t batch, int64_t enc_len, void *enc_out, void *hidden_out, void *workspace, size_t workspace_bytes,
                          void *stream) {
    if (!d || !w || !inputs_embeds || !attn_mask || !enc_out || !workspace || batch <= 0 || enc_len <= 0) return EILEV_E_BADARG;
    if (!dims_ok_t5(d)) return EILEV_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->d_model, H = d->heads, hd = d->d_kv, I = H * hd;
    const int64_t M = batch * enc_len;
    T5Bufs b;
    if (!carve_t5(d, M, 2 * enc_len + 1, workspace, workspace_bytes, b)) return EILEV_E_WORKSPACE;
    EILEV_HIP_CHECK(hipMemcpyAsync(b.h, inputs_embeds, (size_t)M * D * sizeof(bf16), hipMemcpyDeviceToDevice, s));
    // bias(i, j) depends on j - i in [-(L-1), L-1]: table index (j - i) + L - 1
    RC(launch_t5_rel_table((const bf16 *)w->enc_rel_bias, b.rel, (int)(2 * enc_len - 1), (int)enc_len - 1, H, 1, d->rel_buckets,
                           d->rel_max_dist, s));
    const size_t hid_bytes = (size_t)M * D * sizeof(bf16);
    for (int l = 0; l < d->enc_layers; ++l) {
        const EilevT5Layer *L = &w->enc_layers[l];
        if (hidden_out) EILEV_HIP_CHECK(hipMemcpyAsync((char *)hidden_out + l * hid_bytes, b.h, hid_bytes, hipMemcpyDeviceToDevice, s));
        RC(launch_rmsnorm(b.h, D, (const bf16 *)L->ln_sa, b.x, D, M, D, d->eps, s));
        RC(t5_proj(b, b.x, D, L->q_w, L->k_w, L->v_w, I, b.qkv, 3 * I, M, s));
        AttnArgs a;
        a.q = b.qkv; a.k = b.qkv + I; a.v = b.qkv + 2 * I; a.o = b.att;
        a.q_bs = a.k_bs = a.v_bs = enc_len * 3 * (int64_t)I; a.o_bs = enc_len * (int64_t)I;
        a.q_hs = a.k_hs = a.v_hs = a.o_hs = hd;
        a.ldq = a.ldk = a.ldv = 3 * I; a.ldo = I;
        a.batch = (int)batch; a.heads = H; a.sq = a.skv = (int)enc_len; a.hd = hd; a.scale = 1.0f; a.causal = 0;
        a.key_mask = attn_mask; a.mask_ld = enc_len; a.dbg = 0;
        a.rel_tab = b.rel; a.rel_hs = 2 * enc_len - 1; a.rel_off = (int)enc_len - 1; a.rel_n = (int)(2 * enc_len - 1);
        RC(launch_attention(a, s));
        RC(launch_gemm(t5_gemm(b, b.att, I, L->o_w, I, b.h, D, b.h, D, M, D, I), 5, s));
        RC(t5_ff(d, L, b, M, s));
    }
    RC(launch_rmsnorm(b.h, D, (const bf16 *)w->enc_final_ln, (bf16 *)enc_out, D, M, D, d->eps, s));
    if (hidden_out)
        EILEV_HIP_CHECK(hipMemcpyAsync((char *)hidden_out + d->enc_layers * hid_bytes, enc_out, hid_bytes, hipMemcpyDeviceToDevice, s));
    return EILEV_OK;
}

extern "C" int eilev_t5_encode(const EilevT5Dims *d, const EilevT5Weights *w, const void *inputs_embeds, const int32_t *attn_mask,
                               int64_t batch, int64_t enc_len, void *enc_out, void *workspace, size_t workspace_bytes, void *stream) {
    return t5_encode_impl(d, w, inputs_embeds, attn_mask, batch, enc_len, enc_out, nullptr, workspace, workspace_bytes, stream);
}
extern "C" int eilev_t5_encode_debug(const EilevT5Dims *d, const EilevT5Weights *w, const void *inputs_embeds, const int32_t *attn_mask,
                                     int64_t batch, int64_t enc_len, void *enc_out, void *hidden_out, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    return t5_encode_impl(d, w, inputs_embeds, attn_mask, batch, enc_len, enc_out, hidden_out, workspace, workspace_bytes, stream);
}

extern "C" size_t eilev_t5_cross_kv_bytes(const EilevT5Dims *d, int64_t batch, int64_t enc_len) {
    return sizeof(bf16) * (size_t)2 * d->dec_layers * batch * d->heads * enc_len * d->d_kv;
}
extern "C" size_t eilev_t5_self_kv_bytes(const EilevT5Dims *d, int64_t batch, int64_t kv_capacity) {
    return sizeof(bf16) * (size_t)2 * d->dec_layers * batch * d->heads * kv_capacity * d->d_kv;
}

// k|v of every decoder block from the encoder output: one GEMM per layer (two when the weights are not packed back to back)
// into the workspace, then re-tiled per head into the cache planes.
extern "C" int eilev_t5_cross_kv(const EilevT5Dims *d, const EilevT5Weights *w, const void *enc_out, int64_t batch, int64_t enc_len,
                                 void *cross_kv, void *workspace, size_t workspace_bytes, void *stream) {
    if (!d || !w || !enc_out || !cross_kv || !workspace || batch <= 0 || enc_len <= 0) return EILEV_E_BADARG;
    if (!dims_ok_t5(d)) return EILEV_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->d_model, H = d->heads, hd = d->d_kv, I = H * hd;
    const int64_t M = batch * enc_len;
    T5Bufs b;
    if (!carve_t5(d, M, 1, workspace, workspace_bytes, b)) return EILEV_E_WORKSPACE;
    const size_t plane = (size_t)batch * H * enc_len * hd;
    for (int l = 0; l < d->dec_layers; ++l) {
        const EilevT5Layer *L = &w->dec_layers[l];
        bf16 *kc = (bf16 *)cross_kv + 2 * (size_t)l * plane, *vc = kc + plane;
        RC(t5_proj(b, (const bf16 *)enc_out, D, L->ck_w, L->cv_w, nullptr, I, b.qkv, 2 * I, M, s));
        RC(launch_rows_to_cache(b.qkv, 2 * I, 0, kc, (int)batch, (int)enc_len, H, hd, (int)enc_len, 0, s));
        RC(launch_rows_to_cache(b.qkv, 2 * I, I, vc, (int)batch, (int)enc_len, H, hd, (int)enc_len, 0, s));
    }
    return EILEV_OK;
}

// `state` != null: single-token step whose position is state[0] on the device (host past_len = 0, the buffers are sized
// for the whole capacity); null: positions past_len .. past_len + new_len - 1 given by the host
static int t5_decode_impl(const EilevT5Dims *d, const EilevT5Weights *w, const int64_t *dec_ids, const int32_t *enc_mask,
                          int64_t batch, int64_t new_len, int64_t past_len, const int32_t *state, void *self_kv, int64_t kv_capacity,
                          const void *cross_kv, int64_t enc_len, float *logits, void *workspace, size_t workspace_bytes,
                          void *stream, const int32_t *dec_mask = nullptr, void *hidden_out = nullptr) {
    if (!d || !w || !dec_ids || !enc_mask || !self_kv || !cross_kv || !logits || !workspace) return EILEV_E_BADARG;
    if (state && (dec_mask || hidden_out)) return EILEV_E_BADARG;
    if (batch <= 0 || new_len <= 0 || past_len < 0 || past_len + new_len > kv_capacity || enc_len <= 0) return EILEV_E_BADARG;
    if (state && (new_len != 1 || past_len != 0)) return EILEV_E_BADARG;
    if (!dims_ok_t5(d)) return EILEV_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int D = d->d_model, H = d->heads, hd = d->d_kv, I = H * hd;
    const int64_t M = batch * new_len, total = state ? kv_capacity : past_len + new_len;  // with `state`: an upper bound
    T5Bufs b;
    if (!carve_t5(d, M, total + 1, workspace, workspace_bytes, b)) return EILEV_E_WORKSPACE;
    RC(launch_embed_scatter((const bf16 *)w->shared, dec_ids, nullptr, nullptr, 0, M, d->vocab, b.h, D, s));
    // causal self-attention: key j of query at absolute position p: rel = j - p in [-(total-1), 0]: index rel + total - 1
    // (with `state` the table holds the single query row at position state[0]: entry j = bias of key j)
    RC(launch_t5_rel_table((const bf16 *)w->dec_rel_bias, b.rel, (int)total, (int)total - 1, H, 0, d->rel_buckets, d->rel_max_dist, s,
                           state));
    const size_t splane = (size_t)batch * H * kv_capacity * hd, cplane = (size_t)batch * H * enc_len * hd;
    // single-query steps use the split decode-attention kernel; its partials live in the second half of the skinny scratch
    const size_t skinny_f = kSkinnyScratch / 2 / sizeof(float);
    const int64_t kmax = kv_capacity > enc_len ? kv_capacity : enc_len;
    const bool single = new_len == 1 && attn_decode_scratch_bytes((int)batch, H, hd, (int)kmax) <= kSkinnyScratch / 2;
    if (state && !single) return EILEV_E_UNSUPPORTED;
    const size_t hid_bytes = (size_t)M * D * sizeof(bf16);
    const bool fuse_norm = single && M <= 32 && !hidden_out;  // the weight-streaming GEMVs of a decode step (their reduce can carry a norm)
    for (int l = 0; l < d->dec_layers; ++l) {
        const EilevT5Layer *L = &w->dec_layers[l];
        bf16 *kc = (bf16 *)self_kv + 2 * (size_t)l * splane, *vc = kc + splane;
        const bf16 *ck = (const bf16 *)cross_kv + 2 * (size_t)l * cplane, *cv = ck + cplane;
        if (hidden_out) EILEV_HIP_CHECK(hipMemcpyAsync((char *)hidden_out + l * hid_bytes, b.h, hid_bytes, hipMemcpyDeviceToDevice, s));
        // ---- self-attention against the cache (T5LayerSelfAttention :372-401)
        // (decode steps, round 5: the RMSNorm in front of every projection is produced by the residual GEMV before it — its split-K reduce
        //  writes h and RMSNorm(h) in one launch (GemmArgs::ln_out with ln_beta == nullptr) — so only block 0 normalises here)
        if (!fuse_norm || l == 0) RC(launch_rmsnorm(b.h, D, (const bf16 *)L->ln_sa, b.x, D, M, D, d->eps, s));
        RC(t5_proj(b, b.x, D, L->q_w, L->k_w, L->v_w, I, b.qkv, 3 * I, M, s));
        // (graph-replayed decode steps, round 5: the new token's K / V go into the cache inside the attention kernel — fuse_new, as in the OPT step)
        if (!state) {
            RC(launch_rows_to_cache(b.qkv, 3 * I, I, kc, (int)batch, (int)new_len, H, hd, (int)kv_capacity, (int)past_len, s, state));
            RC(launch_rows_to_cache(b.qkv, 3 * I, 2 * I, vc, (int)batch, (int)new_len, H, hd, (int)kv_capacity, (int)past_len, s, state));
        }
        if (single) {
            // one query row per sequence: flash-decoding split kernel (keys 0 .. total - 1 of the cache, bias of a single row)
            if (state)  // kv_total = 1 + state[0] on the device; the table is this query's row (entry j = key j)
                RC(launch_attn_decode(b.qkv, kc, vc, b.att, nullptr, state, (int)batch, 1, (int)kv_capacity, H, hd, b.scratch + skinny_f,
                                      kSkinnyScratch / 2, s, 3 * (int64_t)I, b.rel, total, -1, 1));
            else
                RC(launch_attn_decode(b.qkv, kc, vc, b.att, dec_mask, nullptr, (int)batch, (int)total, (int)kv_capacity, H, hd,
                                      b.scratch + skinny_f, kSkinnyScratch / 2, s, 3 * (int64_t)I, b.rel, total, (int)total - 1));
        } else {
        AttnArgs a;
            a.q = b.qkv; a.k = kc; a.v = vc; a.o = b.att;
            a.q_bs = new_len * 3 * (int64_t)I; a.o_bs = new_len * (int64_t)I;
            a.k_bs = a.v_bs = (int64_t)H * kv_capacity * hd;
            a.q_hs = a.o_hs = hd; a.k_hs = a.v_hs = kv_capacity * (int64_t)hd;
            a.ldq = 3 * I; a.ldk = a.ldv = hd; a.ldo = I;
            a.batch = (int)batch; a.heads = H; a.sq = (int)new_len; a.skv = (int)total; a.hd = hd; a.scale = 1.0f; a.causal = 1;
            a.key_mask = dec_mask; a.mask_ld = dec_mask ? total : 0; a.dbg = 0;  // decoder_attention_mask: keys of padded target positions
            a.rel_tab = b.rel; a.rel_hs = total; a.rel_off = (int)total - 1; a.rel_n = (int)total;
            RC(launch_attention(a, s));
        }
        {
            GemmArgs go = t5_gemm(b, b.att, I, L->o_w, I, b.h, D, b.h, D, M, D, I);
            if (fuse_norm) { go.ln_gamma = (const bf16 *)L->ln_ca; go.ln_beta = nullptr; go.ln_out = b.x; go.ln_eps = d->eps; }
            RC(launch_gemm(go, 5, s));
        }
        // ---- cross-attention over the encoder output (T5LayerCrossAttention :404-432): no position bias, padding mask
        if (!fuse_norm) RC(launch_rmsnorm(b.h, D, (const bf16 *)L->ln_ca, b.x, D, M, D, d->eps, s));
        RC(launch_gemm(t5_gemm(b, b.x, D, L->cq_w, D, nullptr, 0, b.qkv, I, M, I, D), 5, s));
        if (single) {
            RC(launch_attn_decode(b.qkv, ck, cv, b.att, enc_mask, nullptr, (int)batch, (int)enc_len, (int)enc_len, H, hd, b.scratch + skinny_f,
                                  kSkinnyScratch / 2, s, (int64_t)I));
        } else {
        AttnArgs c;
            c.q = b.qkv; c.k = ck; c.v = cv; c.o = b.att;
            c.q_bs = new_len * (int64_t)I; c.o_bs = new_len * (int64_t)I;
            c.k_bs = c.v_bs = (int64_t)H * enc_len * hd;
            c.q_hs = c.o_hs = hd; c.k_hs = c.v_hs = enc_len * (int64_t)hd;
            c.ldq = I; c.ldk = c.ldv = hd; c.ldo = I;
            c.batch = (int)batch; c.heads = H; c.sq = (int)new_len; c.skv = (int)enc_len; c.hd = hd; c.scale = 1.0f; c.causal = 0;
            c.key_mask = enc_mask; c.mask_ld = enc_len; c.dbg = 0;
            RC(launch_attention(c, s));
        }
        {
            GemmArgs gc = t5_gemm(b, b.att, I, L->co_w, I, b.h, D, b.h, D, M, D, I);
            if (fuse_norm) { gc.ln_gamma = (const bf16 *)L->ln_ff; gc.ln_beta = nullptr; gc.ln_out = b.x; gc.ln_eps = d->eps; }
            RC(launch_gemm(gc, 5, s));
        }
        RC(t5_ff(d, L, b, M, s, fuse_norm, fuse_norm ? (l + 1 < d->dec_layers ? w->dec_layers[l + 1].ln_sa : w->dec_final_ln) : nullptr));
    }
    if (!fuse_norm) RC(launch_rmsnorm(b.h, D, (const bf16 *)w->dec_final_ln, b.x, D, M, D, d->eps, s));
    if (hidden_out) EILEV_HIP_CHECK(hipMemcpyAsync((char *)hidden_out + d->dec_layers * hid_bytes, b.x, hid_bytes, hipMemcpyDeviceToDevice, s));
    GemmArgs g = t5_gemm(b, b.x, D, w->lm_head, D, nullptr, 0, logits, d->vocab, M, d->vocab, D);
    g.out_f32 = 1;
    if (d->scale_decoder_outputs) { g.scale = 1.0f / sqrtf((float)D); g.scale_cols = d->vocab; }
    return launch_gemm(g, 5, s);
}

extern "C" int eilev_t5_decode(const EilevT5Dims *d, const EilevT5Weights *w, const int64_t *dec_ids, const int32_t *enc_mask,
                               int64_t batch, int64_t new_len, int64_t past_len, void *self_kv, int64_t kv_capacity,
                               const void *cross_kv, int64_t enc_len, float *logits, void *workspace, size_t workspace_bytes,
                               void *stream) {
    return t5_decode_impl(d, w, dec_ids, enc_mask, batch, new_len, past_len, nullptr, self_kv, kv_capacity, cross_kv, enc_len, logits,
                          workspace, workspace_bytes, stream);
}

// eilev_t5_decode + decoder_attention_mask (dec_mask (batch, past_len + new_len) int32, nullable: keys of the target the self-attention must
// not see, on top of the causal rule; hf T5Stack: create_causal_mask(attention_mask = decoder_attention_mask)) + the per-block tensors
// (hidden_out (dec_layers + 1, batch, new_len, D), nullable)
extern "C" int eilev_t5_decode_debug(const EilevT5Dims *d, const EilevT5Weights *w, const int64_t *dec_ids, const int32_t *enc_mask,
                                     const int32_t *dec_mask, int64_t batch, int64_t new_len, int64_t past_len, void *self_kv,
                                     int64_t kv_capacity, const void *cross_kv, int64_t enc_len, float *logits, void *hidden_out,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    return t5_decode_impl(d, w, dec_ids, enc_mask, batch, new_len, past_len, nullptr, self_kv, kv_capacity, cross_kv, enc_len, logits,
                          workspace, workspace_bytes, stream, dec_mask, hidden_out);
}

extern "C" int eilev_t5_decode_step(const EilevT5Dims *d, const EilevT5Weights *w, const int64_t *tokens, const int32_t *state,
                                    const int32_t *enc_mask, int64_t batch, void *self_kv, int64_t kv_capacity, const void *cross_kv,
                                    int64_t enc_len, float *logits, void *workspace, size_t workspace_bytes, void *stream) {
    if (!state) return EILEV_E_BADARG;
    return t5_decode_impl(d, w, tokens, enc_mask, batch, 1, 0, state, self_kv, kv_capacity, cross_kv, enc_len, logits, workspace,
                          workspace_bytes, stream);
}
