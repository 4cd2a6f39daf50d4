/*
 * eilev.h — C ABI of the MI355X-native VideoBLIP / EILeV forward path.
 *
 * The reference (yukw777/EILEV) has no FFI: its boundary for this path is the Python class
 * eilev.model.v2.VideoBlipForConditionalGeneration (ref:eilev/model/v2.py:106-324), whose
 * arithmetic is executed by third-party `transformers` modules.  This header is the C-ABI a
 * maintainer binds (ctypes stub in INTEGRATION.md) so that the stages of that class's
 * forward()/generate() run on hand-written gfx950 kernels.  Each entry point names the
 * reference code it replaces.
 *
 * Two shared libraries export exactly these symbols:
 *   eilev_amd/csrc/libeilev_hip.so  — the product: HIP kernels for gfx950.  All data pointers
 *                                     are DEVICE pointers; parameters/activations are bf16
 *                                     (uint16 storage) unless stated; `stream` is a hipStream_t.
 *   oracle/libeilev_ref.so          — TEST INFRASTRUCTURE ONLY: plain-C CPU restatement.  All
 *                                     pointers are HOST pointers; parameters/activations fp32;
 *                                     `stream` is ignored.
 *
 * Conventions: the caller owns every buffer (weights, activations, workspace, KV cache); the
 * library never allocates device memory and never synchronises the stream.  Returns 0 on
 * success, a negative EILEV_E_* for bad arguments / unsupported dimensions, a positive value
 * = passthrough hipError_t.  Global mutable state: the product library (which exports exactly the symbols declared here:
 * csrc/exports.map) has ONE process-global, the timing recorder behind eilev_prof_enable / eilev_prof_collect (off by default; a
 * caller that turns it on owns it for the process), and no probe switch; every tuning threshold is an argument or a struct field.  The
 * `eilev_debug_*` switches of the tools (tile-configuration overrides, phase stamps, alternative kernels for A/B runs) exist only in the
 * PROBE build of the same sources (`python eilev_amd/csrc/build.py --variant probes -DEILEV_PROBES` -> libeilev_hip_probes.so, loaded by
 * tools/ and by the few tests that compare an alternative kernel): listed at the end of this header.
 * Linear weights are in the checkpoint's layout: [out_features, in_features] row-major.
 */
#ifndef EILEV_H
#define EILEV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EILEV_ABI_VERSION 16  /* 16: EilevVitWeights.fold_min_rows instead of a process-global knob (round 6); 15: EilevVitWeights.layers_fold_hm (round 5); 14: EilevOptWeights.layers_stream / lm_head_stream, eilev_stream_layout_pack (round 5); 10: eilev_opt_decode_step_beam, eilev_linear_rows; 11: eilev_opt_prefill_debug (round 3); 12: eilev_attention_probs; 13: eilev_t5_encode_debug, eilev_t5_decode_debug, rel_tab of eilev_attention_probs, eilev_topk_logprob, eilev_beam_advance (round 4) */

#define EILEV_OK 0
#define EILEV_E_BADARG (-1)
#define EILEV_E_UNSUPPORTED (-2)
#define EILEV_E_WORKSPACE (-3)
/* RCCL failures of the exchange entries: EILEV_E_RCCL_BASE + ncclResult_t (hipError_t passthrough stays below 1000) */
#define EILEV_E_RCCL_BASE 10000

/* element types of `pixels` */
#define EILEV_F32 0
#define EILEV_BF16 1

/* Model dimensions: the fields of Blip2Config the path reads (hf:models/blip_2/configuration_blip_2.py). */
typedef struct EilevDims {
    /* vision (ViT) */
    int32_t image_size, patch_size, v_hidden, v_inter, v_layers, v_heads;
    float v_eps;
    /* Q-Former */
    int32_t q_hidden, q_inter, q_layers, q_heads, q_cross_freq, num_query;
    float q_eps;
    /* text model (OPT, pre-LN, ReLU, learned positions with offset 2) */
    int32_t t_hidden, t_ffn, t_layers, t_heads, vocab, max_pos;
    float t_eps;
    /* oracle only: 1 = round activations to bf16 where the HIP path stores bf16 */
    int32_t emulate_bf16;
} EilevDims;

/* One ViT block: hf Blip2EncoderLayer (modeling_blip_2.py:383-402). */
typedef struct EilevVitLayer {
    const void *ln1_w, *ln1_b;     /* [Dv] */
    const void *qkv_w, *qkv_b;     /* [3Dv, Dv], [3Dv] (k-bias slots stored, structurally 0) */
    const void *proj_w, *proj_b;   /* [Dv, Dv], [Dv] */
    const void *ln2_w, *ln2_b;     /* [Dv] */
    const void *fc1_w, *fc1_b;     /* [Fv, Dv], [Fv] */
    const void *fc2_w, *fc2_b;     /* [Dv, Fv], [Dv] */
} EilevVitLayer;

/* ABI version 9.  layer_norm1 / layer_norm2 of a block folded into the linear that consumes them (eilev_fold_layernorm applied to
 * qkv and fc1): w [N, Dv] bf16 = gamma (.) W, b [N] bf16 = b + W . beta, csum [N] f32 = row sums of w. */
typedef struct EilevVitLayerFold {
    const void *qkv_w, *qkv_b;
    const float *qkv_csum;
    const void *fc1_w, *fc1_b;
    const float *fc1_csum;
} EilevVitLayerFold;

typedef struct EilevVitWeights {
    const void *patch_w, *patch_b; /* [Dv, 3, P, P], [Dv] */
    const void *cls, *pos;         /* [Dv], [1 + (image/patch)^2, Dv] */
    const void *post_ln_w, *post_ln_b;
    const EilevVitLayer *layers;   /* host array, v_layers entries */
    /* ABI version 9.  NULL, or a host array of v_layers entries.  With it, launches of at least 24 576 token rows (round 4; 65 536 before) run the blocks
     * WITHOUT LayerNorm kernels: proj / fc2 (+ residual) also emit per-row (sum, sum of squares) of the stream they write, and
     * qkv / fc1 read the raw stream and compute rstd * (x . w^T - mean * csum) + b (hf modeling_blip_2.py:383-402:
     * the same function; the bf16 rounding the reference puts on the LayerNorm output sits on gamma (.) W instead).  `layers`
     * stays required (smaller launches, the debug outputs and block 0's layer_norm1 use it). */
    const EilevVitLayerFold *layers_fold;
    /* ABI version 15.  NULL, or (with layers_fold) a host array of v_layers entries: the folded q|k|v matrix of a block once more with its
     * ROWS (= output columns) reordered, per q | k | v third, as [head 0 dims 0..63, head 1 dims 0..63, ..., head 0 dims 64.., head 1 dims
     * 64.., ...].  Launches of >= 512 frames with 257 tokens and head size 88 then write q, k, v of a (frame, head) as one block [token][64]
     * followed by [token][head_dim - 64] (frame region: tokens_per_frame * 3 Dv elements, thirds in the order q, k, v, heads in order) and
     * the frame attention stages a head's image from two contiguous runs; everything else runs from `layers_fold` / `layers`.  Same values,
     * same arithmetic, bit-identical image_embeds. */
    const struct EilevVitLayerFoldHm *layers_fold_hm;
    /* ABI version 16.  Minimum token rows of a launch for the folded path: 0 = the library's default (24 576: tools/vit_small_launch.py — the
     * fold wins from 96 frames per launch, ties at 32-64 and loses below: too few 256 x 256 tiles for the persistent kernel); 1 = every launch
     * (with layers_fold set); negative = never.  (Rounds 4-5: a process-global knob, eilev_debug_ln_fold_min_rows.) */
    int64_t fold_min_rows;
} EilevVitWeights;

typedef struct EilevVitLayerFoldHm {
    const void *qkv_w, *qkv_b;
    const float *qkv_csum;
} EilevVitLayerFoldHm;

/* One Q-Former block: hf Blip2QFormerLayer (modeling_blip_2.py:701-752). cross_* are NULL on
 * layers without cross-attention (layer_idx % q_cross_freq != 0). */
typedef struct EilevQfLayer {
    const void *sq_w, *sq_b, *sk_w, *sk_b, *sv_w, *sv_b; /* self-attn q/k/v [Dq,Dq] */
    const void *so_w, *so_b, *sln_w, *sln_b;             /* self output dense + LayerNorm */
    const void *cq_w, *cq_b;                             /* cross query [Dq,Dq] */
    const void *ck_w, *ck_b, *cv_w, *cv_b;               /* cross key/value [Dq,Dv] */
    const void *co_w, *co_b, *cln_w, *cln_b;             /* cross output dense + LayerNorm */
    const void *fi_w, *fi_b;                             /* intermediate_query [Fq,Dq] */
    const void *fo_w, *fo_b, *fln_w, *fln_b;             /* output_query [Dq,Fq] + LayerNorm */
} EilevQfLayer;

typedef struct EilevQfWeights {
    const void *query_tokens;      /* [num_query, Dq] */
    const void *ln_w, *ln_b;       /* qformer.layernorm */
    const EilevQfLayer *layers;    /* host array, q_layers entries */
} EilevQfWeights;

/* One OPT block: hf OPTDecoderLayer (modeling_opt.py:202-253). */
typedef struct EilevOptLayer {
    const void *ln1_w, *ln1_b;                           /* self_attn_layer_norm */
    const void *q_w, *q_b, *k_w, *k_b, *v_w, *v_b;       /* [Dt,Dt] */
    const void *o_w, *o_b;                               /* out_proj */
    const void *ln2_w, *ln2_b;                           /* final_layer_norm (of the block) */
    const void *fc1_w, *fc1_b, *fc2_w, *fc2_b;           /* [Ft,Dt], [Dt,Ft] */
} EilevOptLayer;

/* Optional fp8 (OCP e4m3) form of the four linears of a block (eilev_linear_w8: one fp32 scale per output channel).  q|k|v are
 * one [3 Dt, Dt] matrix (rows q, k, v).  The biases stay the bf16 ones of EilevOptLayer. */
typedef struct EilevOptLayerW8 {
    const uint8_t *qkv_w8; const float *qkv_scale;   /* [3 Dt, Dt], [3 Dt] */
    const uint8_t *o_w8;   const float *o_scale;     /* [Dt, Dt] */
    const uint8_t *fc1_w8; const float *fc1_scale;   /* [Ft, Dt] */
    const uint8_t *fc2_w8; const float *fc2_scale;   /* [Dt, Ft] */
} EilevOptLayerW8;

typedef struct EilevOptWeights {
    const void *embed_tokens;      /* [vocab, Dt]; also the tied lm_head */
    const void *embed_positions;   /* [max_pos + 2, Dt] */
    const void *final_ln_w, *final_ln_b;
    const EilevOptLayer *layers;   /* host array, t_layers entries */
    /* NULL: bf16 weights.  Else a host array of t_layers entries: the q/k/v/out_proj/fc1/fc2 WEIGHT pointers of `layers` are
     * ignored (they may be NULL) and the fp8 matrices are used (decode streams the bytes; prefill expands one matrix at a time
     * into w8_expand, at least max(Ft, 3 Dt) * Dt * 2 bytes of device memory).  ABI version 2. */
    const EilevOptLayerW8 *layers_w8;
    void *w8_expand;
    size_t w8_expand_bytes;
    /* ABI version 8.  Non-zero with layers_w8: prefill / extend (more than 32 rows) quantise the INPUT of each of the four linears
     * per token to e4m3 (eilev_quant_rows_e4m3) and run the products on the fp8 MFMA (eilev_linear_a8w8) instead of expanding the
     * weights to bf16 — BASELINE configs[4] "fp8 MFMA".  Decode steps keep bf16 activations (weight-streaming, HBM-bound). */
    int32_t w8_act_fp8;
    /* ABI version 14.  Optional second copy of the bf16 decode matrices in the STREAM LAYOUT of the 17..32-row decode kernel
     * (eilev_stream_layout_pack): NULL, or a host array of t_layers entries (an entry's NULL members fall back to `layers`); lm_head_stream:
     * NULL or the packed embed_tokens.  Only decode steps of 17..32 rows read them; results are bit-identical to the plain layout. */
    const struct EilevOptLayerStream *layers_stream;
    const void *lm_head_stream;
} EilevOptWeights;

/* q|k|v as ONE [3 Dt, Dt] matrix (rows q, k, v — what `layers[l].q_w` points at when the three are contiguous), out_proj, fc1, fc2 */
typedef struct EilevOptLayerStream {
    const void *qkv_s, *o_s, *fc1_s, *fc2_s;
} EilevOptLayerStream;

int eilev_abi_version(void);
/* "hip-gfx950" or "cpu-oracle" */
const char *eilev_backend(void);

/* ---- stage 1: vision encoder --------------------------------------------------------------
 * Replaces VideoBlipVisionModel.forward (ref:eilev/model/v2.py:24-103) =
 * Blip2VisionEmbeddings + 39 x Blip2EncoderLayer + post_layernorm (hf modeling_blip_2.py:243-255,
 * 383-402, 505-533).  pixels: (N, 3, T, H, W) contiguous, f32 or bf16 (the (n,t) flatten of
 * v2.py:57 is done by addressing, nothing is materialised).  image_embeds: (N, T*tokens, Dv).
 * pooler: nullable (N, T, Dv) = post_layernorm applied twice to each frame's CLS row. */
size_t eilev_vit_workspace_bytes(const EilevDims *d, int64_t n_clips, int64_t frames);
int eilev_vit_forward(const EilevDims *d, const EilevVitWeights *w, const void *pixels, int pixels_dtype,
                      int64_t n_clips, int64_t frames, void *image_embeds, void *pooler,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Debug outputs of the vision wrapper (ref:eilev/model/v2.py:76-103; shapes asserted by ref:tests/model/test_model_v2.py:57-83):
 * eilev_vit_forward plus, when non-NULL,
 *   hidden_states: (v_layers + 1, N, T*tokens, Dv) — embeddings output, then the residual stream after every block (what
 *                  hf Blip2Encoder :466-478 collects with output_hidden_states=True);
 *   attentions:    (v_layers, N, T, heads, tokens, tokens) — softmax(scale q k^T) of every block (hf eager attention weights).
 * A slow path (copies + an unfused probability kernel); tokens <= 1024 for `attentions`. */
int eilev_vit_forward_debug(const EilevDims *d, const EilevVitWeights *w, const void *pixels, int pixels_dtype,
                            int64_t n_clips, int64_t frames, void *image_embeds, void *pooler, void *hidden_states,
                            void *attentions, void *workspace, size_t workspace_bytes, void *stream);

/* ---- stage 2: Q-Former --------------------------------------------------------------------
 * Replaces Blip2QFormerModel.forward as called from ref:eilev/model/v2.py:291-300 with
 * all-ones masks (hf modeling_blip_2.py:889-949).  image_embeds: (N, kv_len, Dv).
 * query_out: (N, num_query, Dq). */
size_t eilev_qformer_workspace_bytes(const EilevDims *d, int64_t n_clips, int64_t kv_len);
int eilev_qformer_forward(const EilevDims *d, const EilevQfWeights *w, const void *image_embeds,
                          int64_t n_clips, int64_t kv_len, void *query_out,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- stage 3: language projection + token embedding + scatter ---------------------------------
 * Replaces ref:eilev/model/v2.py:308-316 (language_projection, get_input_embeddings()(input_ids),
 * inputs_embeds[video_input_mask] = video_features).  query_out: (n_rows, Dq) with
 * n_rows = N*num_query; the k-th nonzero of video_mask (row-major over (B, L)) receives projected
 * row k.  n_rows must equal the number of nonzeros (else EILEV_E_BADARG, like torch's index_put
 * shape error).  video_feats: nullable (n_rows, Dt) copy of the projected rows (what the RCCL
 * all-gather exchanges).  If query_out is NULL and video_feats_in is given, rows are taken from
 * video_feats_in (already projected, e.g. gathered from other ranks). */
int eilev_project_rows(const EilevDims *d, const void *proj_w, const void *proj_b, const void *query_out,
                       int64_t n_rows, void *video_feats, void *stream);
int eilev_embed_scatter(const EilevDims *d, const void *embed_tokens, const int64_t *input_ids,
                        const uint8_t *video_mask, const void *video_feats, int64_t n_rows,
                        int64_t batch, int64_t seq_len, void *inputs_embeds, void *stream);

/* ---- stage 3b: clip-token exchange between the ranks of a node (RCCL over xGMI; HIP library) -----------------------
 * New capability named by BASELINE.json north_star ("RCCL gather over xGMI of the per-clip query tokens before the LM");
 * closest reference ancestor: accelerate's pad + all-gather, ref:scripts/general/generate_narration_texts.py:124-127.
 * One process per GPU; clips are dealt to ranks (eilev_amd/sharding.py), each rank runs stages 1-3 on its clips, then the
 * projected rows (num_query x Dt bf16 per clip = 164 KB at OPT-2.7B) travel to the rank whose language-model pass needs them.
 *   `comm` is an ncclComm_t (opaque pointer; the header does not need rccl.h).  RCCL is resolved at run time from the copy
 *   already mapped into the process: eilev_comm_bind(path) (NULL = "librccl.so.1"); eilev_comm_unique_id fills
 *   EILEV_COMM_ID_BYTES bytes on one rank, the caller ships them to the others (any channel), every rank calls
 *   eilev_comm_init on ITS device.  Rows are opaque byte rows of `row_bytes`.  All calls are asynchronous launches on `stream`
 *   (a side stream: the exchange of one encode chunk overlaps the ViT of the next) and need no comm when world == 1.
 * eilev_gather_clip_tokens: every rank receives every rank's rows, rank-major in `all` (rows[r] rows from rank r;
 *   ncclAllGather when the counts are equal, grouped send/recv otherwise).
 * eilev_exchange_clip_tokens: all-to-all-v — send_rows[r] rows from row send_off[r] of `send` go to rank r; recv_rows[q] rows
 *   from rank q land at row recv_off[q] of `recv` (each rank receives only the clips of its own samples).
 * Errors: EILEV_E_RCCL_BASE + ncclResult_t.  The oracle library implements world == 1 (a copy) and returns
 * EILEV_E_UNSUPPORTED otherwise. */
#define EILEV_COMM_ID_BYTES 128
int eilev_comm_bind(const char *librccl_path);
int eilev_comm_unique_id(void *id);
int eilev_comm_init(void **comm, int world, int rank, const void *id);
int eilev_comm_destroy(void *comm);
int eilev_gather_clip_tokens(void *comm, const void *local, void *all, const int64_t *rows, int world, int rank,
                             int64_t row_bytes, void *stream);
int eilev_exchange_clip_tokens(void *comm, const void *send, const int64_t *send_rows, const int64_t *send_off, void *recv,
                               const int64_t *recv_rows, const int64_t *recv_off, int world, int rank, int64_t row_bytes,
                               void *stream);

/* ---- fp8-weight linear (BASELINE configs[4]: "fp8 MFMA weights") ------------------------------------
 * nn.Linear with the weight matrix stored as OCP e4m3 bytes (torch.float8_e4m3fn) and one fp32 scale per output channel:
 *     C[m, n] = epilogue((sum_k A[m, k] * dq(Wq[n, k])) * w_scale[n] + bias[n]) (+ residual)
 * i.e. what F.linear(x, (Wq.float() * w_scale[:, None])) computes with the scale factored out of the sum (every e4m3 value
 * is exactly a bf16 value, products accumulate in fp32).  The reference has no fp8 path: this is the weight format of the
 * configs[4] deployment; parity is against the oracle run on the dequantised weights.  M <= 32 (decode) streams the bytes;
 * larger M expands them to bf16 in `scratch` (eilev_linear_w8_scratch_bytes) and runs the bf16 kernels.  a / bias /
 * residual / c as in eilev_linear; epilogue 0 none, 1 erf-GELU, 2 ReLU. */
size_t eilev_linear_w8_scratch_bytes(int64_t m, int64_t n, int64_t k);
int eilev_linear_w8(const void *a, const uint8_t *w8, const float *w_scale, const void *bias, const void *residual, void *c, int64_t m,
                    int64_t n, int64_t k, int epilogue, int out_f32, void *scratch, size_t scratch_bytes, void *stream);

/* LayerNorm folded into the linear that consumes it (ABI version 9; the ViT blocks with EilevVitWeights.layers_fold use these
 * stages; exported so that each can be checked against layernorm + linear of the reference, hf modeling_blip_2.py:383-402).
 *   eilev_fold_layernorm   w [n, k], gamma / beta [k], bias [n] or NULL -> w_out [n, k] = bf16(gamma (.) w), csum [n] = sum_k w_out
 *                          (f32), bias_out [n] = bf16(bias + w . beta)
 *   eilev_linear_stats     c = a . w^T + bias + residual (as eilev_linear, bf16 out) and, per 64-column slot s = column / 64 and row,
 *                          stats[(s * m + row) * 2 + {0, 1}] = (sum, sum of squares) of the values written (before their bf16 rounding)
 *   eilev_ln_finalize      ln_rows[row * 2 + {0, 1}] = (rstd, -mean) from the slots of `stats` summed in slot order
 *   eilev_linear_lnfold    c = act(ln_rows[m, 0] * (a . w_f^T + ln_rows[m, 1] * csum[n]) + bias_f[n]);  epilogue 0 none, 1 erf-GELU
 *                          (the accumulators start from ln_rows[m, 1] * csum[n]; the epilogue multiplies by ln_rows[m, 0])
 * k % 64 == 0; operands below 2 GiB; csum 16-byte aligned.  (The oracle restates the same algebra in f32 / double.) */
int eilev_fold_layernorm(const void *w, const void *gamma, const void *beta, const void *bias, int64_t n, int64_t k, void *w_out,
                         float *csum, void *bias_out, void *stream);
int eilev_linear_stats(const void *a, const void *w, const void *bias, const void *residual, void *c, int64_t m, int64_t n, int64_t k,
                       float *stats, void *stream);
int eilev_ln_finalize(const float *stats, int64_t m, int64_t n, float eps, float *ln_rows, void *stream);
int eilev_linear_lnfold(const void *a, const void *w_f, const void *bias_f, const float *csum, const float *ln_rows, void *c, int64_t m,
                        int64_t n, int64_t k, int epilogue, void *stream);
/* PROBE build only (-DEILEV_PROBES; not exported by libeilev_hip.so): process-global switches, all default 0 = the product path
 *   eilev_debug_gemm_flags : int (int)        force a GEMM tile configuration / remove phases (tools/gemm_probe.py documents the bits)
 *   eilev_debug_gemm_trace : int (void*, int) per-tile time stamps of the persistent GEMM (tools/gemm_trace.py, tools/gemm_itrace.py)
 *   eilev_debug_attn_v1 : int (int)           route attention to the round-1 kernels / select frame-attention variants (A/B tests)
 *   eilev_debug_attn_ts : int (void*)         phase stamps of the frame attention kernel (tools/attn_ts.py)
 *   eilev_debug_decode_rows : int (int)       0: MFMA weight-streaming kernels at every batch size; 1 (default): row-dot kernels at <= 4 rows; 3: the round-3 row-dot kernels
 *   eilev_debug_grid_cus : int (int)          persistent kernels size their grids for n CUs (a CU-masked stream: tools/overlap_probe.py); 0 = all
 *   eilev_debug_beam_part : int (int)         0: beam-search attention at <= 8 rows through the 256-key split kernel (round 3); 1 (default): 128-key ranges
 *   eilev_debug_attn_part32 : int (int)       plain decode attention of head size 80: 0 the 256-key split kernel; 1 (default) by batch size; 2 / 3 force the 256-key ranges / the per-head loop
 *   eilev_debug_vit_head_major : int (int)    0: row-major q|k|v rows in every ViT launch; 1 (default): per-head blocks in launches of >= 512 frames (layers_fold_hm)
 *   eilev_debug_decode_frag : int (int)       0: row-major activations inside the 17..32-row decode step; 1 (default): the row-block layout where every kernel of the block supports it */

/* fp8 ACTIVATIONS x fp8 weights on the fp8 MFMA (BASELINE configs[4] "fp8 MFMA weights"; v_mfma_f32_32x32x64_f8f6f4, twice the
 * bf16 MFMA rate).  The reference has no fp8 path; parity is against the oracle on the same quantised operands.
 * eilev_quant_rows_e4m3: dynamic per-row (per-token) quantisation of x (rows, cols) [bf16 / oracle f32]:
 *     amax = max_c |x[r, c]|;  scale[r] = amax / 448 (1 if amax = 0);  q[r, c] = e4m3_rne(clamp(x[r, c] * (448 / amax), +-448))
 *   (fp32 arithmetic: one division for 448 / amax, one multiplication per element, round-to-nearest-even to OCP e4m3).
 * eilev_linear_a8w8: C[m, n] = epilogue((sum_k dq(a8[m, k]) * dq(w8[n, k])) * a_scale[m] * w_scale[n] + bias[n]) (+ residual);
 *   every product of two e4m3 values is exact in fp32, sums accumulate in fp32.  epilogue 0 none, 2 ReLU.  k % 128 == 0. */
int eilev_quant_rows_e4m3(const void *x, uint8_t *q, float *scale, int64_t rows, int64_t cols, void *stream);
int eilev_linear_a8w8(const uint8_t *a8, const float *a_scale, const uint8_t *w8, const float *w_scale, const void *bias,
                      const void *residual, void *c, int64_t m, int64_t n, int64_t k, int epilogue, int out_f32, void *stream);

/* ---- stage 0: frames -> pixel_values (device-side process()) -------------------------------------------
 * Replaces the image half of process() (ref:eilev/model/utils.py:5-26: frames flattened into the HF image batch,
 * Blip2Processor -> BlipImageProcessor: PIL BICUBIC resize to size x size on uint8, x * (1/255), (x - mean) / std
 * [hf:models/blip/image_processing_blip.py], folded back to (B, C, T, H, W)).  Bit-exact restatement of Pillow's
 * two-pass 8-bit resample (horizontal then vertical, 22-bit fixed-point coefficients, rounding, clip to 0..255;
 * Pillow src/libImaging/Resample.c): the coefficient / bounds tables depend only on the sizes and are built on the
 * host (eilev_amd/preprocess.py, oracle/eilev_ref.c::eilev_resample_coeffs) exactly as precompute_coeffs +
 * normalize_coeffs_8bpc do; rescale + normalize of a byte is a 3 x 256 table of the fp32 values the HF code produces.
 *   video:   uint8 (B, 3, T, Hin, Win)      out: (B, 3, T, Hout, Wout), out_dtype 0 = fp32 (what process() returns), 1 = bf16
 *   coef_h:  int32 (Wout, ksize_h), bounds_h: int32 (Wout, 2) = (first input column, taps); NULL when Win == Wout
 *   coef_v:  int32 (Hout, ksize_v), bounds_v: int32 (Hout, 2);                               NULL when Hin == Hout
 *   lut:     fp32 (3, 256)
 * workspace: the horizontally resized planes, uint8 (B * 3 * T, Hin, Wout). */
size_t eilev_process_workspace_bytes(int64_t batch, int64_t frames, int64_t h_in, int64_t w_out);
int eilev_process_frames(const uint8_t *video, int64_t batch, int64_t frames, int64_t h_in, int64_t w_in, int64_t h_out, int64_t w_out,
                         const int32_t *coef_h, const int32_t *bounds_h, int32_t ksize_h, const int32_t *coef_v,
                         const int32_t *bounds_v, int32_t ksize_v, const float *lut, void *out, int32_t out_dtype,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- stage 4: OPT prefill -----------------------------------------------------------------
 * Replaces OPTForCausalLM.forward on inputs_embeds (hf modeling_opt.py:321-396, 464-524) as called
 * from ref:eilev/model/v2.py:220-227 (forward) and through GenerationMixin._prefill from
 * v2.py:318-322 (generate).  attn_mask: (B, L) int32, 1 = attend (left or right padding).
 * kv_cache: [t_layers][2][B][t_heads][kv_capacity][head_dim]; logits_last: (B, vocab) f32 of the
 * last position; logits_all: nullable (B, L, vocab) f32. */
size_t eilev_opt_workspace_bytes(const EilevDims *d, int64_t batch, int64_t seq_len);
size_t eilev_opt_kv_cache_bytes(const EilevDims *d, int64_t batch, int64_t kv_capacity);
int eilev_opt_prefill(const EilevDims *d, const EilevOptWeights *w, const void *inputs_embeds,
                      const int32_t *attn_mask, int64_t batch, int64_t seq_len, void *kv_cache,
                      int64_t kv_capacity, float *logits_last, float *logits_all,
                      void *workspace, size_t workspace_bytes, void *stream);
/* eilev_opt_prefill that also exports what hf returns under `output_hidden_states=True` (ref:eilev/model/v2.py:220-227 hands the flag
 * to the language model; hf modeling_opt.py OPTDecoder.forward collects the input of every block and, last, the output of
 * final_layer_norm): hidden_states [t_layers + 1][B][L][t_hidden] in the activation dtype — [0] inputs_embeds + positions,
 * [l] the residual stream after block l - 1, [t_layers] AFTER final_layer_norm (the rows the lm_head sees).  Same arithmetic and
 * the same logits as eilev_opt_prefill; the extra cost is one device copy per block. */
int eilev_opt_prefill_debug(const EilevDims *d, const EilevOptWeights *w, const void *inputs_embeds,
                            const int32_t *attn_mask, int64_t batch, int64_t seq_len, void *kv_cache,
                            int64_t kv_capacity, float *logits_last, float *logits_all, void *hidden_states,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---- stage 4b: continue a prefilled sequence (classify) ---------------------------------------------
 * Runs `new_len` further positions per row against a KV cache that already holds `past_len` entries.
 * Replaces the second language-model call of classify() (ref:eilev/model/v2.py:461-466:
 * input_ids = class tokens, past_key_values = the prompt's cache repeated per class).
 * inputs_embeds: (batch, new_len, Dt); attn_mask: (batch, past_len + new_len) int32;
 * logits_all: (batch, new_len, vocab) f32.  Workspace: eilev_opt_workspace_bytes(d, batch, past_len + new_len). */
int eilev_opt_extend(const EilevDims *d, const EilevOptWeights *w, const void *inputs_embeds, const int32_t *attn_mask,
                     int64_t batch, int64_t new_len, int64_t past_len, void *kv_cache, int64_t kv_capacity,
                     float *logits_all, void *workspace, size_t workspace_bytes, void *stream);

/* ---- encoder-decoder language model: flan-t5 (BASELINE configs[3]) -----------------------
 * Replaces T5ForConditionalGeneration as driven by ref:eilev/model/v2.py:228-238 (forward) and :318-322 (generate):
 * hf models/t5/modeling_t5.py T5Stack :640-752, T5Block :435-510, T5Attention :176-370 (no 1/sqrt(d) scaling; relative
 * position bias of block 0 shared by every block of the stack; :217-262 bucket function), T5LayerNorm :50-72 (RMS, no
 * bias), T5DenseGatedActDense :97-124 (gelu_new(wi_0 x) * wi_1 x), lm_head :1030-1036.  No linear layer has a bias. */
typedef struct EilevT5Dims {
    int32_t d_model, d_kv, heads, d_ff, enc_layers, dec_layers, vocab;
    int32_t rel_buckets, rel_max_dist;   /* relative_attention_num_buckets / _max_distance */
    float eps;                           /* layer_norm_epsilon */
    int32_t scale_decoder_outputs;       /* 1: sequence_output *= d_model^-0.5 before lm_head (T5 v1.0 configs) */
    int32_t emulate_bf16;                /* oracle only */
} EilevT5Dims;

typedef struct EilevT5Layer {
    const void *ln_sa;                          /* layer[0].layer_norm.weight [D] */
    const void *q_w, *k_w, *v_w, *o_w;          /* SelfAttention q/k/v [H*dkv, D], o [D, H*dkv] */
    const void *ln_ca;                          /* decoder: layer[1].layer_norm (NULL in the encoder) */
    const void *cq_w, *ck_w, *cv_w, *co_w;      /* decoder: EncDecAttention (NULL in the encoder) */
    const void *ln_ff;                          /* layer[-1].layer_norm */
    const void *wi0_w, *wi1_w, *wo_w;           /* DenseReluDense wi_0/wi_1 [F, D], wo [D, F] */
} EilevT5Layer;

typedef struct EilevT5Weights {
    const void *shared;                         /* [vocab, D] input embedding of encoder and decoder */
    const void *lm_head;                        /* [vocab, D] (== shared when tied) */
    const void *enc_rel_bias, *dec_rel_bias;    /* block[0].layer[0].SelfAttention.relative_attention_bias [buckets, H] */
    const void *enc_final_ln, *dec_final_ln;    /* [D] */
    const EilevT5Layer *enc_layers, *dec_layers;/* host arrays */
} EilevT5Weights;

/* rows = max tokens per sequence processed by one call (encoder length or decoder new_len), kv_len = max keys */
size_t eilev_t5_workspace_bytes(const EilevT5Dims *d, int64_t batch, int64_t rows, int64_t kv_len);
/* encoder stack on inputs_embeds (batch, enc_len, D) (= shared[input_ids] with the video rows scattered in,
 * eilev_embed_scatter); attn_mask (batch, enc_len) int32; enc_out (batch, enc_len, D) = encoder last_hidden_state. */
int eilev_t5_encode(const EilevT5Dims *d, const EilevT5Weights *w, const void *inputs_embeds, const int32_t *attn_mask,
                    int64_t batch, int64_t enc_len, void *enc_out, void *workspace, size_t workspace_bytes, void *stream);
/* cross-attention keys/values of every decoder block, computed once per encoder output:
 * [dec_layer][k|v][batch][head][enc_len][d_kv] */
size_t eilev_t5_cross_kv_bytes(const EilevT5Dims *d, int64_t batch, int64_t enc_len);
int eilev_t5_cross_kv(const EilevT5Dims *d, const EilevT5Weights *w, const void *enc_out, int64_t batch, int64_t enc_len,
                      void *cross_kv, void *workspace, size_t workspace_bytes, void *stream);
/* decoder self-attention cache: [dec_layer][k|v][batch][head][capacity][d_kv] */
size_t eilev_t5_self_kv_bytes(const EilevT5Dims *d, int64_t batch, int64_t kv_capacity);
/* decoder over new_len positions past_len .. past_len + new_len - 1 of every sequence (teacher forcing: past_len = 0,
 * new_len = target length; generation: new_len = 1).  dec_ids (batch, new_len) int64; enc_mask (batch, enc_len) int32;
 * logits (batch, new_len, vocab) f32. */
int eilev_t5_decode(const EilevT5Dims *d, const EilevT5Weights *w, const int64_t *dec_ids, const int32_t *enc_mask,
                    int64_t batch, int64_t new_len, int64_t past_len, void *self_kv, int64_t kv_capacity,
                    const void *cross_kv, int64_t enc_len, float *logits, void *workspace, size_t workspace_bytes,
                    void *stream);

/* The same two stacks with what the reference's forward can also ask of them (ref:eilev/model/v2.py:228-238 hands decoder_attention_mask,
 * output_hidden_states to hf T5ForConditionalGeneration):
 *  - hidden_out (nullable): hf T5Stack's `hidden_states` tuple as one tensor (layers + 1, batch, rows, D): every block's input, then the
 *    output of final_layer_norm (modeling_t5.py T5Stack.forward, all_hidden_states);
 *  - dec_mask (nullable): decoder_attention_mask (batch, past_len + new_len) int32: target keys the decoder self-attention must not
 *    see, on top of the causal rule (key 0 of every row must be visible: no query row may lose all of its keys). */
int eilev_t5_encode_debug(const EilevT5Dims *d, const EilevT5Weights *w, const void *inputs_embeds, const int32_t *attn_mask,
                          int64_t batch, int64_t enc_len, void *enc_out, void *hidden_out, void *workspace, size_t workspace_bytes,
                          void *stream);
int eilev_t5_decode_debug(const EilevT5Dims *d, const EilevT5Weights *w, const int64_t *dec_ids, const int32_t *enc_mask,
                          const int32_t *dec_mask, int64_t batch, int64_t new_len, int64_t past_len, void *self_kv,
                          int64_t kv_capacity, const void *cross_kv, int64_t enc_len, float *logits, void *hidden_out,
                          void *workspace, size_t workspace_bytes, void *stream);

/* One generation step of the decoder with the step counter on the DEVICE (so that a captured hipGraph replays for every
 * step): position = state[0] = number of tokens fed so far; tokens (batch) int64 = the ids to feed (start token at step 0,
 * then what eilev_greedy_select wrote); logits (batch, vocab) f32.  Same arithmetic as eilev_t5_decode(new_len = 1). */
int eilev_t5_decode_step(const EilevT5Dims *d, const EilevT5Weights *w, const int64_t *tokens, const int32_t *state,
                         const int32_t *enc_mask, int64_t batch, void *self_kv, int64_t kv_capacity, const void *cross_kv,
                         int64_t enc_len, float *logits, void *workspace, size_t workspace_bytes, void *stream);

/* ---- stage 5: greedy decode ---------------------------------------------------------------
 * Replaces one iteration of GenerationMixin._sample with do_sample=False
 * (hf generation/utils.py:2876-2937): argmax of the fp32 last-row logits, pad-after-EOS,
 * unfinished bookkeeping.  eilev_greedy_select consumes logits (B, vocab) and writes
 * tokens[b] (the token fed to the next step) and out_tokens[b*max_new + state[0]'s step].
 *
 * state (int32, device for the HIP library): [0] = number of tokens generated so far (step),
 * [1] = number of unfinished rows.  finished: (B) uint8.  eos_id < 0 disables EOS.
 * eilev_opt_decode_step runs one full step: embed tokens[b] at position n_valid[b] + step - 1,
 * 32 blocks against the KV cache (slot seq_len + step - 1), final LN, lm_head, then
 * eilev_greedy_select.  Every launch parameter that changes between steps is read from `state`
 * on the device, so the call can be captured once into a hipGraph and replayed. */
/* ABI version 13 (round 4).  Beam search, the vocabulary-sized part of one step (hf generation/utils.py `_beam_search` ->
 * `_get_top_k_continuations`, the sample script's default: ref:samples/eilev_generate_action_narration.py:60-73): for every row r
 *   out_val[r, 0..keep) = the `keep` largest of log_softmax(logits[r, :]) + row_score[r], descending (ties: lower token id first),
 *   out_idx[r, 0..keep) = their token ids.
 * The top 2K over a sample's K rows x vocabulary are among its rows' top 2K: the merge is K x 2K numbers (eilev_amd/beam.py).
 * log_softmax as torch evaluates it in fp32: (x - max) - log(sum exp(x - max)).  row_score nullable (0).  HIP: vocab <= 65536, % 4 == 0. */
int eilev_topk_logprob(const float *logits, const float *row_score, int64_t rows, int64_t vocab, int64_t keep, float *out_val,
                       int32_t *out_idx, void *stream);
/* ABI version 13 (round 4).  The bookkeeping of that step for every sample, on eilev_topk_logprob's output (rows = batch * beams, row b * beams + j =
 * beam j of sample b): merge to the sample's top `keep` continuations, next running beams (the best `beams` that did not stop), finished set
 * (only the top `beams` candidates may finish; score / len ** length_penalty; frozen once nothing can improve), the early_stopping=False
 * heuristic (hf `_get_running_beams_for_next_iteration`, `_update_finished_beams`, `_check_early_stop_heuristic`), then what the decode step
 * reads: tokens (rows) to feed and, when anc != NULL, the ancestor table of eilev_opt_decode_step_beam updated in place (column r continues the
 * hypothesis that lived in row src[r]; row `cur` = identity).  cur = state[0] - 1 (state = the decode step's counter, on the device for the HIP
 * library: preset to 1 before the first step).  len_pow[n - 1] = float(n) ** length_penalty, or its fp32 reciprocal when len_pow_reciprocal
 * (x / scalar is x * (1 / scalar) in torch's GPU kernel, a true division on the CPU: eilev_amd/beam.py keeps either bit for bit).
 * early_stopping: 0 False / "never" with length_penalty <= 0, 1 True, 2 "never" with length_penalty > 0.  eos_ids: host array (n_eos <= 8).
 * Sequences (batch, beams, max_new) int64, scores f32, fin_len int64, finished / can_improve uint8, all updated in place.  Equal scores: the
 * lower candidate index wins.  HIP: beams <= 32, keep <= 64, beams * keep <= 2048, gen_cap * beams <= 2048.
 * Precondition 1 <= state[0] <= max_new (a device value the entry cannot validate): outside it the call changes nothing. */
size_t eilev_beam_scratch_bytes(int64_t batch, int64_t beams, int64_t keep, int64_t max_new);
int eilev_beam_advance(const float *row_lp, const int32_t *row_tok, int64_t batch, int64_t beams, int64_t keep, int64_t max_new,
                       const int32_t *state, const int64_t *eos_ids, int64_t n_eos, const float *len_pow, int len_pow_reciprocal,
                       int early_stopping, int64_t *run_seq, float *run_score, int64_t *fin_seq, float *fin_score, int64_t *fin_len,
                       uint8_t *finished, uint8_t *can_improve, int64_t *tokens, int32_t *anc, int64_t gen_cap, void *scratch,
                       size_t scratch_bytes, void *stream);
int eilev_greedy_select(const float *logits, int64_t batch, int64_t vocab, int32_t *state,
                        uint8_t *finished, int64_t eos_id, int64_t pad_id, int64_t *tokens,
                        int64_t *out_tokens, int64_t max_new, void *stream);
int eilev_opt_decode_step(const EilevDims *d, const EilevOptWeights *w, int64_t *tokens, int32_t *state,
                          const int32_t *attn_mask, const int32_t *n_valid, int64_t batch,
                          int64_t seq_len, void *kv_cache, int64_t kv_capacity, float *logits,
                          uint8_t *finished, int64_t eos_id, int64_t pad_id, int64_t *out_tokens,
                          int64_t max_new, void *workspace, size_t workspace_bytes, void *stream);

/* Beam search (the sample script's default call, ref:samples/eilev_generate_action_narration.py:60-73 -> hf
 * GenerationMixin._beam_search generation/utils.py:3208-3560, which reorders the KV cache by the surviving beams' parents every
 * step: `_temporary_reorder_cache`).  Here the cache is NOT moved: row r = beam r % beams of sample r / beams;
 *   kv_prompt   the cache eilev_opt_prefill filled for the `rows / beams` SAMPLES with kv_capacity == seq_len (read only);
 *   kv_gen      eilev_opt_kv_cache_bytes(d, rows, gen_capacity) bytes: slot g of row r = the g-th generated token that row r produced;
 *   ancestors   (gen_capacity, rows) int32, device: ancestors[g][r] = the row whose slot g belongs to the hypothesis now in row r
 *               (the caller gathers it by the step's parent indices: a few hundred integers instead of the cache);
 *   attn_mask   (rows / beams, seq_len): the prompts' masks; n_valid (rows): number of visible prompt tokens per row;
 *   state[0]    number of tokens generated so far (>= 1: tokens[r] is the state[0]-th, its K / V go to slot state[0] - 1 of row r;
 *               the caller sets ancestors[state[0] - 1][r] = r); incremented by the call, so a captured step replays.
 * logits (rows, vocab) f32.  Selection (top-2K, length penalty) stays with the caller. */
int eilev_opt_decode_step_beam(const EilevDims *d, const EilevOptWeights *w, const int64_t *tokens, int32_t *state,
                               const int32_t *attn_mask, const int32_t *n_valid, int64_t rows, int64_t beams,
                               int64_t seq_len, const void *kv_prompt, void *kv_gen, int64_t gen_capacity,
                               const int32_t *ancestors, float *logits, void *workspace, size_t workspace_bytes,
                               void *stream);

/* ---- building blocks exported for unit parity tests and the roofline probe --------------------
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N]); epilogue: 0 none, 1 GELU(erf),
 * 2 ReLU.  out_f32 != 0 writes f32 instead of bf16 (HIP library). */
int eilev_linear(const void *a, const void *w, const void *bias, const void *residual, void *c,
                 int64_t m, int64_t n, int64_t k, int epilogue, int out_f32, void *stream);
int eilev_layernorm(const void *x, const void *gamma, const void *beta, void *y, int64_t rows,
                    int64_t cols, float eps, void *stream);
/* The small-batch form of the decode step's linears (m <= 8 rows, k % 512 == 0): C = epilogue(LN(x) . W^T + bias) (+ residual)
 * with the LayerNorm that feeds the linear (hf OPTDecoderLayer self_attn_layer_norm -> q|k|v, final_layer_norm -> fc1, decoder
 * final_layer_norm -> lm_head: modeling_opt.py:226-247, 386) evaluated in the same launch; ln_gamma == NULL: plain linear.  The
 * normalised rows are rounded to bf16 before the product (HIP library), as the separate LayerNorm kernel stores them.
 * epilogue: 0 none, 2 ReLU.  Other shapes: EILEV_E_UNSUPPORTED (HIP library; eilev_linear / eilev_layernorm serve them). */
/* Weight-streaming layout of a bf16 nn.Linear weight [n, k] (row-major, no row padding) for decode steps of 17..32 rows: the same n * k
 * values reordered so that every load instruction of the decode kernel reads one contiguous kilobyte (the checkpoint layout gives it 16
 * segments of 64 bytes, a row apart).  `out`: n * k bf16, device memory, must not alias w.  The order depends on (n, k) and on the CU count
 * of the current device.  EILEV_E_UNSUPPORTED: the decode kernel does not take this shape (k % 256, or no 5 / 8 / 10-step K slice) — keep
 * the plain layout.  hf counterpart: none (the arithmetic of nn.Linear in OPTDecoderLayer, modeling_opt.py:226-247, is unchanged). */
int eilev_stream_layout_pack(const void *w, int64_t n, int64_t k, void *out, void *stream);
int eilev_linear_rows(const void *x, const void *ln_gamma, const void *ln_beta, float eps, const void *w, const void *bias,
                      const void *residual, void *c, int64_t m, int64_t n, int64_t k, int epilogue, int out_f32, void *stream);
/* Multi-head attention over packed projections. q: rows of length ldq with head h at column
 * h*head_dim (same for k, v with ldk, ldv); o: (batch, sq, heads*head_dim).  causal != 0 applies
 * key <= query + (skv - sq); key_mask: nullable (batch, skv) int32. scale multiplies q.k. */
int eilev_attention(const void *q, const void *k, const void *v, void *o, int64_t batch, int64_t heads,
                    int64_t sq, int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv,
                    float scale, int causal, const int32_t *key_mask, void *stream);
/* ABI version 12 (round 4).  The attention WEIGHTS of the same call: probs (batch, heads, sq, skv) = softmax over the visible keys of
 * scale * q . k (same layout, causal and key_mask arguments as eilev_attention; masked keys get 0; a row without a visible key is all 0) —
 * what hf's eager attention returns as `attn_weights` and the reference passes through under `output_attentions=True`
 * (ref:eilev/model/v2.py:187-193 Q-Former [hf modeling_blip_2.py Blip2QFormerMultiHeadAttention], :220-227 language model [hf modeling_opt.py
 * eager_attention_forward]).  Slow path: one wave per query row, nothing tiled; probs bf16 (HIP) / f32 (oracle).  skv <= 4096.
 * ABI version 13: + rel_tab (nullable; the additive relative position bias of eilev_attention_rel, same indexing: the T5 stacks' weights,
 * hf modeling_t5.py T5Attention `attn_weights` = softmax(q . k + position_bias + mask), scale 1). */
int eilev_attention_probs(const void *q, const void *k, void *probs, int64_t batch, int64_t heads, int64_t sq, int64_t skv, int64_t head_dim,
                          int64_t ldq, int64_t ldk, float scale, int causal, const int32_t *key_mask, const float *rel_tab, int64_t rel_stride,
                          int64_t rel_off, int64_t rel_n, void *stream);

/* ---- gradient building blocks of the train_v2 path -------------------------------------------------
 * ref:scripts/general/train_v2.py:124-130,207-217: `loss = model(**batch).loss; accelerator.backward(loss)` with the
 * ViT and the language model frozen (gradients flow *through* the LM to the Q-Former, language projection and query
 * tokens).  These are what torch.autograd runs for the modules on the path; a linear layer's dX = dY . W and
 * dW = dY^T . X are eilev_linear calls on transposed operands (eilev_amd/autograd.py).  bf16 tensors (HIP) / f32 (oracle).
 *
 * eilev_attention_bwd: gradients of eilev_attention (same layout arguments; o and d_o are (batch, sq, heads*head_dim);
 *   dq/dk/dv rows have strides lddq/lddk/lddv with head h at column h*head_dim).  P is recomputed from q, k and the row
 *   log-sum-exp; lse_delta is a (2, batch, heads, sq) f32 workspace (out: lse, then delta = sum_d o * d_o).
 * eilev_layernorm_bwd: dx of nn.LayerNorm (hf modeling_opt.py:215,226,387; modeling_blip_2.py:619,675,913); when dgamma
 *   and dbeta (f32, cols) are given their gradients are ACCUMULATED into them, with stats a (rows, 2) f32 workspace.
 * eilev_colsum: out[c] += sum_r dy[r, c] (bias gradient; f32, accumulated).
 * eilev_act_fwd / eilev_act_bwd: y = act(pre) / dx = dy * act'(pre); kind 1 erf-GELU, 2 ReLU.
 * eilev_ce_loss: hf loss_utils.ForCausalLMLoss per row: row_loss[r] = logsumexp(logits[r]) - logits[r, target[r]],
 *   dlogits[r] = (softmax(logits[r]) - onehot(target[r])) * grad_scale; rows with target < 0 (ignore_index -100) get 0.
 *   dlogits may be NULL (loss only: the eval-mode loss of forward(labels=...) and classify's log-likelihoods). */
int eilev_attention_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, void *dq,
                        void *dk, void *dv, float *lse_delta, int64_t batch, int64_t heads, int64_t sq,
                        int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddq,
                        int64_t lddk, int64_t lddv, float scale, int causal, const int32_t *key_mask,
                        void *stream);
int eilev_layernorm_bwd(const void *x, const void *gamma, const void *dy, void *dx, float *dgamma, float *dbeta,
                        float *stats, int64_t rows, int64_t cols, float eps, void *stream);
int eilev_colsum(const void *dy, float *out, int64_t rows, int64_t cols, void *stream);
int eilev_act_fwd(const void *pre, void *y, int64_t n, int kind, void *stream);
int eilev_act_bwd(const void *pre, const void *dy, void *dx, int64_t n, int kind, void *stream);
int eilev_ce_loss(const float *logits, const int64_t *targets, float grad_scale, float *row_loss, void *dlogits,
                  int64_t rows, int64_t vocab, void *stream);

/* ---- encoder-decoder (T5) building blocks for the training graph (hf models/t5/modeling_t5.py) ---------------------------
 * eilev_attention_rel / eilev_attention_rel_bwd: eilev_attention / eilev_attention_bwd with T5's additive relative position bias
 *   (T5Attention :176-369, frozen on the train_v2 path: no gradient for it):
 *   score(h, i, j) = scale * q_i . k_j + rel_tab[h * rel_stride + clamp((j - i - (skv - sq)) + rel_off, 0, rel_n - 1)];
 *   rel_tab (f32, heads x rel_stride) holds the bias per relative distance; null = no bias.
 * eilev_rmsnorm / eilev_rmsnorm_bwd: T5LayerNorm :50-72 (y = gamma * x * rsqrt(mean(x^2) + eps)) and its dx.
 * eilev_gated_gelu / eilev_gated_gelu_bwd: T5DenseGatedActDense :97-124 on ab = [a | b] rows of 2 f columns:
 *   out = gelu_new(a) * b (tanh form, hf activations NewGELUActivation); dab = [dy * b * gelu_new'(a) | dy * gelu_new(a)]. */
int eilev_attention_rel(const void *q, const void *k, const void *v, void *o, int64_t batch, int64_t heads, int64_t sq,
                        int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, float scale, int causal,
                        const int32_t *key_mask, const float *rel_tab, int64_t rel_stride, int64_t rel_off,
                        int64_t rel_n, void *stream);
int eilev_attention_rel_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, void *dq,
                            void *dk, void *dv, float *lse_delta, int64_t batch, int64_t heads, int64_t sq,
                            int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddq,
                            int64_t lddk, int64_t lddv, float scale, int causal, const int32_t *key_mask,
                            const float *rel_tab, int64_t rel_stride, int64_t rel_off, int64_t rel_n, void *stream);
int eilev_rmsnorm(const void *x, const void *gamma, void *y, int64_t rows, int64_t cols, float eps, void *stream);
int eilev_rmsnorm_bwd(const void *x, const void *gamma, const void *dy, void *dx, int64_t rows, int64_t cols, float eps,
                      void *stream);
int eilev_gated_gelu(const void *ab, void *out, int64_t rows, int64_t f, void *stream);
int eilev_gated_gelu_bwd(const void *ab, const void *dy, void *dab, int64_t rows, int64_t f, void *stream);

/* ---- dropout of the training graph (`model.train()` under Trainer: hidden_dropout_prob / attention_probs_dropout_prob of the
 * Q-Former, `dropout` of OPT, `dropout_rate` of T5) ------------------------------------------------------------------------
 * Masks are a pure function of (seed, element index): keep iff hash(seed, index) >= p * 2^32 with (`eilev_hash32` in csrc/common.h)
 *   z = index + 0x9E3779B97F4A7C15 * (seed + 1); z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9; z = (z ^ z >> 27) * 0x94D049BB133111EB;
 *   hash = (z ^ z >> 31) >> 32                                                     (splitmix64 finaliser, 64-bit wrap-around)
 * so the backward recomputes the mask of the forward and nothing is stored; torch's Philox stream is NOT reproduced (no two
 * dropout implementations share masks; parity is against the oracle, which restates the same function).
 * eilev_dropout_add: y = x * M / (1 - p) (+ resid when non-null; the dropped value is rounded to the storage type first); its
 *   gradient w.r.t. x is the same call on dy with resid = null.  index = element offset.
 * eilev_attention_dropout(_bwd): eilev_attention_rel(_bwd) with dropout on the probabilities AFTER the softmax normalisation
 *   (hf Blip2QFormerMultiHeadAttention / T5Attention): o = (softmax(s) * M / (1 - p)) v; index = ((b * heads + h) * sq + i) * skv + j. */
int eilev_dropout_add(const void *x, const void *resid, void *y, int64_t n, float dropout_p, uint32_t seed, void *stream);
int eilev_attention_dropout(const void *q, const void *k, const void *v, void *o, int64_t batch, int64_t heads, int64_t sq,
                            int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, float scale, int causal,
                            const int32_t *key_mask, const float *rel_tab, int64_t rel_stride, int64_t rel_off,
                            int64_t rel_n, float dropout_p, uint32_t seed, void *stream);
int eilev_attention_dropout_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, void *dq,
                                void *dk, void *dv, float *lse_delta, int64_t batch, int64_t heads, int64_t sq,
                                int64_t skv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t lddq,
                                int64_t lddk, int64_t lddv, float scale, int causal, const int32_t *key_mask,
                                const float *rel_tab, int64_t rel_stride, int64_t rel_off, int64_t rel_n,
                                float dropout_p, uint32_t seed, void *stream);

/* ---- kernel profiler (HIP library; no-ops returning 0 in the oracle) ----------------------------
 * When enabled, the dominant GEMM launches are bracketed with hipEvents on the launch stream.
 * eilev_prof_collect synchronises those events and returns per-kind launch count, total ms and
 * total algorithmic FLOPs since the last reset.  kind: 0 = all tiled GEMMs, 1 = ViT fc1,
 * 2 = ViT fc2, 3 = ViT qkv, 4 = ViT proj. */
int eilev_prof_enable(int on);
int eilev_prof_collect(int kind, int64_t *launches, double *total_ms, double *total_flops);

#ifdef __cplusplus
}
#endif
#endif /* EILEV_H */
