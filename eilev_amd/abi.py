"""ctypes binding of include/eilev.h.

Both shared libraries export the same symbols; this module only describes the ABI and
builds the weight structs from a ``name -> address`` callback.  It contains no arithmetic
and no fallback: ``load_hip()`` raises if ``libeilev_hip.so`` is missing.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "csrc", "libeilev_hip.so")

vp = C.c_void_p


class Dims(C.Structure):
    _fields_ = [
        ("image_size", C.c_int32), ("patch_size", C.c_int32), ("v_hidden", C.c_int32),
        ("v_inter", C.c_int32), ("v_layers", C.c_int32), ("v_heads", C.c_int32), ("v_eps", C.c_float),
        ("q_hidden", C.c_int32), ("q_inter", C.c_int32), ("q_layers", C.c_int32), ("q_heads", C.c_int32),
        ("q_cross_freq", C.c_int32), ("num_query", C.c_int32), ("q_eps", C.c_float),
        ("t_hidden", C.c_int32), ("t_ffn", C.c_int32), ("t_layers", C.c_int32), ("t_heads", C.c_int32),
        ("vocab", C.c_int32), ("max_pos", C.c_int32), ("t_eps", C.c_float),
        ("emulate_bf16", C.c_int32),
    ]


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(f, vp) for f in fields]})


VIT_LAYER_FIELDS = ["ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_w", "ln2_b",
                    "fc1_w", "fc1_b", "fc2_w", "fc2_b"]
QF_LAYER_FIELDS = ["sq_w", "sq_b", "sk_w", "sk_b", "sv_w", "sv_b", "so_w", "so_b", "sln_w", "sln_b",
                   "cq_w", "cq_b", "ck_w", "ck_b", "cv_w", "cv_b", "co_w", "co_b", "cln_w", "cln_b",
                   "fi_w", "fi_b", "fo_w", "fo_b", "fln_w", "fln_b"]
OPT_LAYER_FIELDS = ["ln1_w", "ln1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b",
                    "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b"]

T5_LAYER_FIELDS = ["ln_sa", "q_w", "k_w", "v_w", "o_w", "ln_ca", "cq_w", "ck_w", "cv_w", "co_w", "ln_ff", "wi0_w", "wi1_w", "wo_w"]

VitLayer = _ptr_struct("VitLayer", VIT_LAYER_FIELDS)
T5Layer = _ptr_struct("T5Layer", T5_LAYER_FIELDS)
QfLayer = _ptr_struct("QfLayer", QF_LAYER_FIELDS)
OptLayer = _ptr_struct("OptLayer", OPT_LAYER_FIELDS)


VitLayerFold = _ptr_struct("VitLayerFold", ["qkv_w", "qkv_b", "qkv_csum", "fc1_w", "fc1_b", "fc1_csum"])


VitLayerFoldHm = _ptr_struct("VitLayerFoldHm", ["qkv_w", "qkv_b", "qkv_csum"])


class VitWeights(C.Structure):
    _fields_ = [("patch_w", vp), ("patch_b", vp), ("cls", vp), ("pos", vp), ("post_ln_w", vp),
                ("post_ln_b", vp), ("layers", C.POINTER(VitLayer)), ("layers_fold", C.POINTER(VitLayerFold)),
                ("layers_fold_hm", C.POINTER(VitLayerFoldHm)), ("fold_min_rows", C.c_int64)]


class QfWeights(C.Structure):
    _fields_ = [("query_tokens", vp), ("ln_w", vp), ("ln_b", vp), ("layers", C.POINTER(QfLayer))]


class OptLayerW8(C.Structure):
    _fields_ = [(n, vp) for n in ("qkv_w8", "qkv_scale", "o_w8", "o_scale", "fc1_w8", "fc1_scale", "fc2_w8", "fc2_scale")]


class OptLayerStream(C.Structure):
    _fields_ = [(n, vp) for n in ("qkv_s", "o_s", "fc1_s", "fc2_s")]


class OptWeights(C.Structure):
    _fields_ = [("embed_tokens", vp), ("embed_positions", vp), ("final_ln_w", vp), ("final_ln_b", vp),
                ("layers", C.POINTER(OptLayer)), ("layers_w8", C.POINTER(OptLayerW8)), ("w8_expand", vp), ("w8_expand_bytes", C.c_size_t),
                ("w8_act_fp8", C.c_int32), ("layers_stream", C.POINTER(OptLayerStream)), ("lm_head_stream", vp)]


class T5Dims(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("d_kv", C.c_int32), ("heads", C.c_int32), ("d_ff", C.c_int32),
                ("enc_layers", C.c_int32), ("dec_layers", C.c_int32), ("vocab", C.c_int32),
                ("rel_buckets", C.c_int32), ("rel_max_dist", C.c_int32), ("eps", C.c_float),
                ("scale_decoder_outputs", C.c_int32), ("emulate_bf16", C.c_int32)]


class T5Weights(C.Structure):
    _fields_ = [("shared", vp), ("lm_head", vp), ("enc_rel_bias", vp), ("dec_rel_bias", vp), ("enc_final_ln", vp),
                ("dec_final_ln", vp), ("enc_layers", C.POINTER(T5Layer)), ("dec_layers", C.POINTER(T5Layer))]


# HF state-dict key suffixes for each struct field.
_VIT_KEYS = {
    "ln1_w": "layer_norm1.weight", "ln1_b": "layer_norm1.bias",
    "qkv_w": "self_attn.qkv.weight", "qkv_b": "self_attn.qkv.bias",
    "proj_w": "self_attn.projection.weight", "proj_b": "self_attn.projection.bias",
    "ln2_w": "layer_norm2.weight", "ln2_b": "layer_norm2.bias",
    "fc1_w": "mlp.fc1.weight", "fc1_b": "mlp.fc1.bias", "fc2_w": "mlp.fc2.weight", "fc2_b": "mlp.fc2.bias",
}
_QF_KEYS = {
    "sq_w": "attention.attention.query.weight", "sq_b": "attention.attention.query.bias",
    "sk_w": "attention.attention.key.weight", "sk_b": "attention.attention.key.bias",
    "sv_w": "attention.attention.value.weight", "sv_b": "attention.attention.value.bias",
    "so_w": "attention.output.dense.weight", "so_b": "attention.output.dense.bias",
    "sln_w": "attention.output.LayerNorm.weight", "sln_b": "attention.output.LayerNorm.bias",
    "cq_w": "crossattention.attention.query.weight", "cq_b": "crossattention.attention.query.bias",
    "ck_w": "crossattention.attention.key.weight", "ck_b": "crossattention.attention.key.bias",
    "cv_w": "crossattention.attention.value.weight", "cv_b": "crossattention.attention.value.bias",
    "co_w": "crossattention.output.dense.weight", "co_b": "crossattention.output.dense.bias",
    "cln_w": "crossattention.output.LayerNorm.weight", "cln_b": "crossattention.output.LayerNorm.bias",
    "fi_w": "intermediate_query.dense.weight", "fi_b": "intermediate_query.dense.bias",
    "fo_w": "output_query.dense.weight", "fo_b": "output_query.dense.bias",
    "fln_w": "output_query.LayerNorm.weight", "fln_b": "output_query.LayerNorm.bias",
}
_OPT_KEYS = {
    "ln1_w": "self_attn_layer_norm.weight", "ln1_b": "self_attn_layer_norm.bias",
    "q_w": "self_attn.q_proj.weight", "q_b": "self_attn.q_proj.bias",
    "k_w": "self_attn.k_proj.weight", "k_b": "self_attn.k_proj.bias",
    "v_w": "self_attn.v_proj.weight", "v_b": "self_attn.v_proj.bias",
    "o_w": "self_attn.out_proj.weight", "o_b": "self_attn.out_proj.bias",
    "ln2_w": "final_layer_norm.weight", "ln2_b": "final_layer_norm.bias",
    "fc1_w": "fc1.weight", "fc1_b": "fc1.bias", "fc2_w": "fc2.weight", "fc2_b": "fc2.bias",
}

VIT_PREFIX = "vision_model.encoder.layers.{}."
QF_PREFIX = "qformer.encoder.layer.{}."
OPT_PREFIX = "language_model.model.decoder.layers.{}."


def vit_layer_keys(i):
    return {f: VIT_PREFIX.format(i) + s for f, s in _VIT_KEYS.items()}


def qf_layer_keys(i, has_cross):
    return {f: QF_PREFIX.format(i) + s for f, s in _QF_KEYS.items() if has_cross or not f.startswith("c")}


def opt_layer_keys(i):
    return {f: OPT_PREFIX.format(i) + s for f, s in _OPT_KEYS.items()}


def t5_layer_keys(stack: str, i: int):
    """struct field -> state-dict key of T5 block i of `stack` ("encoder" / "decoder")."""
    p = f"language_model.{stack}.block.{i}.layer."
    ff = 2 if stack == "decoder" else 1
    k = {"ln_sa": p + "0.layer_norm.weight"}
    for n in "qkvo":
        k[f"{n}_w"] = p + f"0.SelfAttention.{n}.weight"
    if stack == "decoder":
        k["ln_ca"] = p + "1.layer_norm.weight"
        for n in "qkvo":
            k[f"c{n}_w"] = p + f"1.EncDecAttention.{n}.weight"
    k["ln_ff"] = p + f"{ff}.layer_norm.weight"
    for n, f in (("wi_0", "wi0_w"), ("wi_1", "wi1_w"), ("wo", "wo_w")):
        k[f] = p + f"{ff}.DenseReluDense.{n}.weight"
    return k


def t5_dims_from_config(config, emulate_bf16: bool = False) -> T5Dims:
    t = config.text_config
    if t.feed_forward_proj != "gated-gelu":
        raise NotImplementedError("only the gated-gelu (T5 v1.1 / flan-t5) feed-forward is built")
    d = T5Dims()
    d.d_model, d.d_kv, d.heads, d.d_ff = t.d_model, t.d_kv, t.num_heads, t.d_ff
    d.enc_layers, d.dec_layers, d.vocab = t.num_layers, t.num_decoder_layers, t.vocab_size
    d.rel_buckets, d.rel_max_dist, d.eps = t.relative_attention_num_buckets, t.relative_attention_max_distance, t.layer_norm_epsilon
    d.scale_decoder_outputs = int(bool(getattr(t, "scale_decoder_outputs", getattr(t, "tie_word_embeddings", True))))
    d.emulate_bf16 = int(emulate_bf16)
    return d


def dims_from_config(config, emulate_bf16: bool = False) -> Dims:
    """Fill EilevDims from a transformers Blip2Config (eps/sizes are read, never hard-coded)."""
    v, q, t = config.vision_config, config.qformer_config, config.text_config
    mt = getattr(t, "model_type", "opt")
    if mt not in ("opt", "t5"):
        raise NotImplementedError(f"language model type {mt!r}: only OPT (decoder-only) and T5 (encoder-decoder) are built")
    if mt == "opt" and (not getattr(t, "do_layer_norm_before", True) or getattr(t, "word_embed_proj_dim", t.hidden_size) != t.hidden_size):
        raise NotImplementedError("OPT variants with post-LN or projected embeddings (opt-350m) are not supported")
    d = Dims()
    d.image_size, d.patch_size = v.image_size, v.patch_size
    d.v_hidden, d.v_inter, d.v_layers, d.v_heads = v.hidden_size, v.intermediate_size, v.num_hidden_layers, v.num_attention_heads
    d.v_eps = v.layer_norm_eps
    d.q_hidden, d.q_inter, d.q_layers, d.q_heads = q.hidden_size, q.intermediate_size, q.num_hidden_layers, q.num_attention_heads
    d.q_cross_freq, d.num_query, d.q_eps = q.cross_attention_frequency, config.num_query_tokens, q.layer_norm_eps
    if mt == "t5":  # the OPT fields stay 0; projection / embedding only read t_hidden and vocab
        d.t_hidden, d.vocab = t.d_model, t.vocab_size
    else:
        d.t_hidden, d.t_ffn, d.t_layers, d.t_heads = t.hidden_size, t.ffn_dim, t.num_hidden_layers, t.num_attention_heads
        d.vocab, d.max_pos, d.t_eps = t.vocab_size, t.max_position_embeddings, 1e-5  # nn.LayerNorm default (hf modeling_opt.py:215)
    d.emulate_bf16 = int(emulate_bf16)
    return d


class WeightPack:
    """The three weight structs plus the ctypes arrays they point to (kept alive here)."""

    def __init__(self, dims: Dims, addr, t5dims: "T5Dims | None" = None):
        """``addr(key) -> int`` returns the address of the state-dict tensor ``key``."""
        self.dims = dims
        self.t5dims = t5dims
        self._vit_layers = (VitLayer * dims.v_layers)()
        for i in range(dims.v_layers):
            for f, k in vit_layer_keys(i).items():
                setattr(self._vit_layers[i], f, addr(k))
        self.vit = VitWeights(
            addr("vision_model.embeddings.patch_embedding.weight"), addr("vision_model.embeddings.patch_embedding.bias"),
            addr("vision_model.embeddings.class_embedding"), addr("vision_model.embeddings.position_embedding"),
            addr("vision_model.post_layernorm.weight"), addr("vision_model.post_layernorm.bias"),
            C.cast(self._vit_layers, C.POINTER(VitLayer)))
        self._qf_layers = (QfLayer * dims.q_layers)()
        for i in range(dims.q_layers):
            for f, k in qf_layer_keys(i, i % dims.q_cross_freq == 0).items():
                setattr(self._qf_layers[i], f, addr(k))
        self.qf = QfWeights(addr("query_tokens"), addr("qformer.layernorm.weight"), addr("qformer.layernorm.bias"),
                            C.cast(self._qf_layers, C.POINTER(QfLayer)))
        self._opt_layers = (OptLayer * dims.t_layers)()
        for i in range(dims.t_layers):
            for f, k in opt_layer_keys(i).items():
                setattr(self._opt_layers[i], f, addr(k))
        if t5dims is None:
            self.opt = OptWeights(
                addr("language_model.model.decoder.embed_tokens.weight"),
                addr("language_model.model.decoder.embed_positions.weight"),
                addr("language_model.model.decoder.final_layer_norm.weight"),
                addr("language_model.model.decoder.final_layer_norm.bias"),
                C.cast(self._opt_layers, C.POINTER(OptLayer)), None, None, 0)
            self.embed_tokens = self.opt.embed_tokens
        else:
            self._t5_layers = {}
            for stack, n in (("encoder", t5dims.enc_layers), ("decoder", t5dims.dec_layers)):
                arr = (T5Layer * n)()
                for i in range(n):
                    for f, k in t5_layer_keys(stack, i).items():
                        setattr(arr[i], f, addr(k))
                self._t5_layers[stack] = arr
            rb = "language_model.{}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"
            shared = addr("language_model.shared.weight")
            try:  # untied head (original flan-t5 checkpoints); the installed transformers always ties it to `shared`
                head = addr("language_model.lm_head.weight") or shared
            except KeyError:
                head = shared
            self.t5 = T5Weights(shared, head, addr(rb.format("encoder")),
                                addr(rb.format("decoder")), addr("language_model.encoder.final_layer_norm.weight"),
                                addr("language_model.decoder.final_layer_norm.weight"),
                                C.cast(self._t5_layers["encoder"], C.POINTER(T5Layer)),
                                C.cast(self._t5_layers["decoder"], C.POINTER(T5Layer)))
            self.embed_tokens = shared
        self.proj_w = addr("language_projection.weight")
        self.proj_b = addr("language_projection.bias")


EXPORTS = [
    "eilev_abi_version", "eilev_backend", "eilev_vit_workspace_bytes", "eilev_vit_forward", "eilev_vit_forward_debug",
    "eilev_qformer_workspace_bytes", "eilev_qformer_forward", "eilev_project_rows", "eilev_embed_scatter",
    "eilev_opt_workspace_bytes", "eilev_opt_kv_cache_bytes", "eilev_opt_prefill", "eilev_opt_prefill_debug", "eilev_opt_extend", "eilev_greedy_select", "eilev_topk_logprob", "eilev_beam_scratch_bytes", "eilev_beam_advance",
    "eilev_opt_decode_step", "eilev_opt_decode_step_beam", "eilev_linear", "eilev_linear_rows", "eilev_layernorm", "eilev_attention", "eilev_attention_probs", "eilev_prof_enable",
    "eilev_prof_collect", "eilev_t5_workspace_bytes", "eilev_t5_encode", "eilev_t5_cross_kv_bytes", "eilev_t5_cross_kv",
    "eilev_t5_self_kv_bytes", "eilev_t5_decode", "eilev_t5_decode_step", "eilev_t5_encode_debug", "eilev_t5_decode_debug", "eilev_process_workspace_bytes", "eilev_process_frames",
    "eilev_linear_w8_scratch_bytes", "eilev_linear_w8", "eilev_quant_rows_e4m3", "eilev_linear_a8w8", "eilev_attention_bwd", "eilev_layernorm_bwd", "eilev_colsum",
    "eilev_act_fwd", "eilev_act_bwd", "eilev_ce_loss", "eilev_attention_rel", "eilev_attention_rel_bwd", "eilev_rmsnorm",
    "eilev_rmsnorm_bwd", "eilev_gated_gelu", "eilev_gated_gelu_bwd", "eilev_dropout_add", "eilev_attention_dropout",
    "eilev_attention_dropout_bwd", "eilev_comm_bind", "eilev_comm_unique_id", "eilev_comm_init", "eilev_comm_destroy",
    "eilev_gather_clip_tokens", "eilev_exchange_clip_tokens", "eilev_fold_layernorm", "eilev_linear_stats", "eilev_ln_finalize",
    "eilev_linear_lnfold", "eilev_stream_layout_pack",
]


def attach_vit_fold(pack, per_layer):
    """Point ``pack.vit`` at LayerNorm-folded qkv / fc1 right-hand sides: per_layer = [{"qkv": (w_ptr, b_ptr, csum_ptr), "fc1": ...}, ...]."""
    arr = (VitLayerFold * len(per_layer))()
    for i, d in enumerate(per_layer):
        for name in ("qkv", "fc1"):
            for f, v in zip(("w", "b", "csum"), d[name]):
                setattr(arr[i], f"{name}_{f}", v)
    pack._vit_layers_fold = arr
    pack.vit.layers_fold = C.cast(arr, C.POINTER(VitLayerFold))


def attach_vit_fold_hm(pack, per_layer):
    """Point ``pack.vit`` at the block-ordered copies of the folded q|k|v matrices (EilevVitWeights.layers_fold_hm): per_layer =
    [(w_ptr, b_ptr, csum_ptr), ...]; ``per_layer=None`` detaches them."""
    if per_layer is None:
        pack._vit_layers_fold_hm = None
        pack.vit.layers_fold_hm = None
        return
    arr = (VitLayerFoldHm * len(per_layer))()
    for i, (w, b, cs) in enumerate(per_layer):
        arr[i].qkv_w, arr[i].qkv_b, arr[i].qkv_csum = w, b, cs
    pack._vit_layers_fold_hm = arr
    pack.vit.layers_fold_hm = C.cast(arr, C.POINTER(VitLayerFoldHm))


def attach_opt_stream(pack, per_layer, lm_head_ptr):
    """Point ``pack.opt`` at stream-layout copies of the decode matrices (eilev_stream_layout_pack): per_layer = [{"qkv": ptr or None,
    "o": ..., "fc1": ..., "fc2": ...}, ...]; ``per_layer=None`` detaches them."""
    if per_layer is None:
        pack._opt_layers_stream = None
        pack.opt.layers_stream = None
        pack.opt.lm_head_stream = None
        return
    arr = (OptLayerStream * len(per_layer))()
    for i, d in enumerate(per_layer):
        for name in ("qkv", "o", "fc1", "fc2"):
            setattr(arr[i], f"{name}_s", d.get(name))
    pack._opt_layers_stream = arr
    pack.opt.layers_stream = C.cast(arr, C.POINTER(OptLayerStream))
    pack.opt.lm_head_stream = lm_head_ptr


def attach_opt_w8(pack, per_layer, expand_ptr: int, expand_bytes: int, act_fp8: bool = False):
    """Point ``pack.opt`` at fp8 (e4m3) linears: per_layer = [{"qkv": (bytes_ptr, scale_ptr), "o": ..., "fc1": ..., "fc2": ...}, ...]."""
    arr = (OptLayerW8 * len(per_layer))()
    for i, d in enumerate(per_layer):
        for name in ("qkv", "o", "fc1", "fc2"):
            setattr(arr[i], f"{name}_w8", d[name][0])
            setattr(arr[i], f"{name}_scale", d[name][1])
    pack._opt_layers_w8 = arr
    pack.opt.layers_w8 = C.cast(arr, C.POINTER(OptLayerW8))
    pack.opt.w8_expand = expand_ptr
    pack.opt.w8_expand_bytes = expand_bytes
    pack.opt.w8_act_fp8 = 1 if act_fp8 else 0


def bind(lib: C.CDLL) -> C.CDLL:
    i64, i32, f32, sz = C.c_int64, C.c_int, C.c_float, C.c_size_t
    DP = C.POINTER(Dims)
    lib.eilev_abi_version.restype = i32
    lib.eilev_backend.restype = C.c_char_p
    lib.eilev_vit_workspace_bytes.restype = sz
    lib.eilev_vit_workspace_bytes.argtypes = [DP, i64, i64]
    lib.eilev_vit_forward.restype = i32
    lib.eilev_vit_forward.argtypes = [DP, C.POINTER(VitWeights), vp, i32, i64, i64, vp, vp, vp, sz, vp]
    lib.eilev_vit_forward_debug.restype = i32
    lib.eilev_vit_forward_debug.argtypes = [DP, C.POINTER(VitWeights), vp, i32, i64, i64, vp, vp, vp, vp, vp, sz, vp]
    lib.eilev_qformer_workspace_bytes.restype = sz
    lib.eilev_qformer_workspace_bytes.argtypes = [DP, i64, i64]
    lib.eilev_qformer_forward.restype = i32
    lib.eilev_qformer_forward.argtypes = [DP, C.POINTER(QfWeights), vp, i64, i64, vp, vp, sz, vp]
    lib.eilev_project_rows.restype = i32
    lib.eilev_project_rows.argtypes = [DP, vp, vp, vp, i64, vp, vp]
    lib.eilev_linear_w8_scratch_bytes.restype = sz
    lib.eilev_linear_w8_scratch_bytes.argtypes = [i64, i64, i64]
    lib.eilev_linear_w8.restype = i32
    lib.eilev_linear_w8.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, vp, sz, vp]
    lib.eilev_quant_rows_e4m3.restype = i32
    lib.eilev_quant_rows_e4m3.argtypes = [vp, vp, vp, i64, i64, vp]
    lib.eilev_fold_layernorm.restype = i32
    lib.eilev_fold_layernorm.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, vp, vp]
    lib.eilev_linear_stats.restype = i32
    lib.eilev_linear_stats.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, vp, vp]
    lib.eilev_ln_finalize.restype = i32
    lib.eilev_ln_finalize.argtypes = [vp, i64, i64, f32, vp, vp]
    lib.eilev_linear_lnfold.restype = i32
    lib.eilev_linear_lnfold.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]
    lib.eilev_linear_a8w8.restype = i32
    lib.eilev_linear_a8w8.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, vp]
    lib.eilev_process_workspace_bytes.restype = sz
    lib.eilev_process_workspace_bytes.argtypes = [i64, i64, i64, i64]
    lib.eilev_process_frames.restype = i32
    lib.eilev_process_frames.argtypes = [vp, i64, i64, i64, i64, i64, i64, vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, sz, vp]
    lib.eilev_embed_scatter.restype = i32
    lib.eilev_embed_scatter.argtypes = [DP, vp, vp, vp, vp, i64, i64, i64, vp, vp]
    lib.eilev_opt_workspace_bytes.restype = sz
    lib.eilev_opt_workspace_bytes.argtypes = [DP, i64, i64]
    lib.eilev_opt_kv_cache_bytes.restype = sz
    lib.eilev_opt_kv_cache_bytes.argtypes = [DP, i64, i64]
    lib.eilev_opt_prefill.restype = i32
    lib.eilev_opt_prefill.argtypes = [DP, C.POINTER(OptWeights), vp, vp, i64, i64, vp, i64, vp, vp, vp, sz, vp]
    lib.eilev_opt_prefill_debug.restype = i32
    lib.eilev_opt_prefill_debug.argtypes = [DP, C.POINTER(OptWeights), vp, vp, i64, i64, vp, i64, vp, vp, vp, vp, sz, vp]
    lib.eilev_opt_extend.restype = i32
    lib.eilev_opt_extend.argtypes = [DP, C.POINTER(OptWeights), vp, vp, i64, i64, i64, vp, i64, vp, vp, sz, vp]
    lib.eilev_greedy_select.restype = i32
    lib.eilev_greedy_select.argtypes = [vp, i64, i64, vp, vp, i64, i64, vp, vp, i64, vp]
    lib.eilev_opt_decode_step.restype = i32
    lib.eilev_opt_decode_step.argtypes = [DP, C.POINTER(OptWeights), vp, vp, vp, vp, i64, i64, vp, i64, vp, vp,
                                          i64, i64, vp, i64, vp, sz, vp]
    lib.eilev_opt_decode_step_beam.restype = i32
    lib.eilev_opt_decode_step_beam.argtypes = [DP, C.POINTER(OptWeights), vp, vp, vp, vp, i64, i64, i64, vp, vp, i64, vp, vp, vp, sz, vp]
    lib.eilev_linear.restype = i32
    lib.eilev_linear.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, vp]
    lib.eilev_stream_layout_pack.restype = i32
    lib.eilev_stream_layout_pack.argtypes = [vp, i64, i64, vp, vp]
    lib.eilev_linear_rows.restype = i32
    lib.eilev_linear_rows.argtypes = [vp, vp, vp, f32, vp, vp, vp, vp, i64, i64, i64, i32, i32, vp]
    lib.eilev_layernorm.restype = i32
    lib.eilev_layernorm.argtypes = [vp, vp, vp, vp, i64, i64, f32, vp]
    lib.eilev_attention.restype = i32
    lib.eilev_attention.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, f32, i32, vp, vp]
    lib.eilev_attention_probs.restype = i32
    lib.eilev_attention_probs.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, f32, i32, vp, vp, i64, i64, i64, vp]
    lib.eilev_attention_bwd.restype = i32
    lib.eilev_attention_bwd.argtypes = [vp] * 9 + [i64] * 11 + [f32, i32, vp, vp]
    lib.eilev_layernorm_bwd.restype = i32
    lib.eilev_layernorm_bwd.argtypes = [vp] * 7 + [i64, i64, f32, vp]
    lib.eilev_colsum.restype = i32
    lib.eilev_colsum.argtypes = [vp, vp, i64, i64, vp]
    lib.eilev_act_fwd.restype = i32
    lib.eilev_act_fwd.argtypes = [vp, vp, i64, i32, vp]
    lib.eilev_act_bwd.restype = i32
    lib.eilev_act_bwd.argtypes = [vp, vp, vp, i64, i32, vp]
    lib.eilev_ce_loss.restype = i32
    lib.eilev_ce_loss.argtypes = [vp, vp, f32, vp, vp, i64, i64, vp]
    lib.eilev_attention_rel.restype = i32
    lib.eilev_attention_rel.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, f32, i32, vp, vp, i64, i64, i64, vp]
    lib.eilev_attention_rel_bwd.restype = i32
    lib.eilev_attention_rel_bwd.argtypes = [vp] * 9 + [i64] * 11 + [f32, i32, vp, vp, i64, i64, i64, vp]
    lib.eilev_rmsnorm.restype = i32
    lib.eilev_rmsnorm.argtypes = [vp, vp, vp, i64, i64, f32, vp]
    lib.eilev_rmsnorm_bwd.restype = i32
    lib.eilev_rmsnorm_bwd.argtypes = [vp, vp, vp, vp, i64, i64, f32, vp]
    lib.eilev_gated_gelu.restype = i32
    lib.eilev_gated_gelu.argtypes = [vp, vp, i64, i64, vp]
    lib.eilev_gated_gelu_bwd.restype = i32
    lib.eilev_gated_gelu_bwd.argtypes = [vp, vp, vp, i64, i64, vp]
    u32 = C.c_uint32
    lib.eilev_dropout_add.restype = i32
    lib.eilev_dropout_add.argtypes = [vp, vp, vp, i64, f32, u32, vp]
    lib.eilev_attention_dropout.restype = i32
    lib.eilev_attention_dropout.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, f32, i32, vp, vp, i64, i64, i64, f32, u32, vp]
    lib.eilev_attention_dropout_bwd.restype = i32
    lib.eilev_attention_dropout_bwd.argtypes = [vp] * 9 + [i64] * 11 + [f32, i32, vp, vp, i64, i64, i64, f32, u32, vp]
    TP = C.POINTER(T5Dims)
    lib.eilev_t5_workspace_bytes.restype = sz
    lib.eilev_t5_workspace_bytes.argtypes = [TP, i64, i64, i64]
    lib.eilev_t5_encode.restype = i32
    lib.eilev_t5_encode.argtypes = [TP, C.POINTER(T5Weights), vp, vp, i64, i64, vp, vp, sz, vp]
    lib.eilev_t5_cross_kv_bytes.restype = sz
    lib.eilev_t5_cross_kv_bytes.argtypes = [TP, i64, i64]
    lib.eilev_t5_cross_kv.restype = i32
    lib.eilev_t5_cross_kv.argtypes = [TP, C.POINTER(T5Weights), vp, i64, i64, vp, vp, sz, vp]
    lib.eilev_t5_self_kv_bytes.restype = sz
    lib.eilev_t5_self_kv_bytes.argtypes = [TP, i64, i64]
    lib.eilev_t5_decode.restype = i32
    lib.eilev_t5_decode.argtypes = [TP, C.POINTER(T5Weights), vp, vp, i64, i64, i64, vp, i64, vp, i64, vp, vp, sz, vp]
    lib.eilev_beam_scratch_bytes.restype = sz
    lib.eilev_beam_scratch_bytes.argtypes = [i64, i64, i64, i64]
    lib.eilev_beam_advance.restype = i32
    lib.eilev_beam_advance.argtypes = [vp, vp, i64, i64, i64, i64, vp, vp, i64, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, sz, vp]
    lib.eilev_topk_logprob.restype = i32
    lib.eilev_topk_logprob.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp]
    lib.eilev_t5_encode_debug.restype = i32
    lib.eilev_t5_encode_debug.argtypes = [TP, C.POINTER(T5Weights), vp, vp, i64, i64, vp, vp, vp, sz, vp]
    lib.eilev_t5_decode_debug.restype = i32
    lib.eilev_t5_decode_debug.argtypes = [TP, C.POINTER(T5Weights), vp, vp, vp, i64, i64, i64, vp, i64, vp, i64, vp, vp, vp, sz, vp]
    lib.eilev_t5_decode_step.restype = i32
    lib.eilev_t5_decode_step.argtypes = [TP, C.POINTER(T5Weights), vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, sz, vp]
    lib.eilev_comm_bind.restype = i32
    lib.eilev_comm_bind.argtypes = [C.c_char_p]
    lib.eilev_comm_unique_id.restype = i32
    lib.eilev_comm_unique_id.argtypes = [vp]
    lib.eilev_comm_init.restype = i32
    lib.eilev_comm_init.argtypes = [C.POINTER(vp), i32, i32, vp]
    lib.eilev_comm_destroy.restype = i32
    lib.eilev_comm_destroy.argtypes = [vp]
    lib.eilev_gather_clip_tokens.restype = i32
    lib.eilev_gather_clip_tokens.argtypes = [vp, vp, vp, vp, i32, i32, i64, vp]
    lib.eilev_exchange_clip_tokens.restype = i32
    lib.eilev_exchange_clip_tokens.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, vp]
    lib.eilev_prof_enable.restype = i32
    lib.eilev_prof_enable.argtypes = [i32]
    lib.eilev_prof_collect.restype = i32
    lib.eilev_prof_collect.argtypes = [i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return lib


def load_library(path: str) -> C.CDLL:
    lib = bind(C.CDLL(path))
    if lib.eilev_abi_version() != 16:
        raise RuntimeError(f"{path}: ABI version mismatch")
    return lib


_hip = None


def load_hip() -> C.CDLL:
    """Load the HIP product library.  No fallback: a missing build is an error."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(
                f"{HIP_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback on the product path.")
        _hip = load_library(HIP_LIB_PATH)
        if _hip.eilev_backend() != b"hip-gfx950":
            raise RuntimeError("libeilev_hip.so reports an unexpected backend")
    return _hip


PROBES_LIB_PATH = os.path.join(_HERE, "csrc", "libeilev_hip_probes.so")
_probes = None


def load_probes():
    """The PROBE build of the same sources (-DEILEV_PROBES: the product entry points + the process-global `eilev_debug_*` switches of
    include/eilev.h's last comment) — tools/ and the few tests that compare an alternative kernel.  None when it has not been built
    (`python eilev_amd/csrc/build.py --variant probes -DEILEV_PROBES`; __graft_entry__.build() builds it next to the product library).
    Never loaded by the product path."""
    global _probes
    if _probes is None and os.path.exists(PROBES_LIB_PATH):
        _probes = load_library(PROBES_LIB_PATH)
    return _probes


def use_probes() -> None:
    """tools/ only: make this process run on the probe build (load_hip() and HIP_LIB_PATH then refer to libeilev_hip_probes.so)."""
    global _hip, HIP_LIB_PATH
    lib = load_probes()
    if lib is None:
        raise RuntimeError(f"{PROBES_LIB_PATH} not found: python eilev_amd/csrc/build.py --variant probes -DEILEV_PROBES")
    _hip, HIP_LIB_PATH = lib, PROBES_LIB_PATH


def check(rc: int, what: str) -> None:
    if rc != 0:
        names = {-1: "EILEV_E_BADARG", -2: "EILEV_E_UNSUPPORTED", -3: "EILEV_E_WORKSPACE"}
        what_rc = names.get(rc, f"ncclResult {rc - 10000}" if rc >= 10000 else f"hipError {rc}")
        raise RuntimeError(f"{what} failed: {what_rc}")
