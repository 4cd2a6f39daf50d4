"""Names and shapes of the VideoBLIP state dict (the checkpoint contract)."""
from __future__ import annotations

from . import abi


def state_dict_shapes(config):
    """(key -> shape) of the VideoBLIP state dict for a Blip2Config (names: SURVEY §8a-W)."""
    v, q, t = config.vision_config, config.qformer_config, config.text_config
    is_t5 = getattr(t, "model_type", "opt") == "t5"
    Dv, Fv, Dq, Fq, Dt = v.hidden_size, v.intermediate_size, q.hidden_size, q.intermediate_size, t.hidden_size
    Ft = t.d_ff if is_t5 else t.ffn_dim
    tok = (v.image_size // v.patch_size) ** 2 + 1
    s = {"query_tokens": (1, config.num_query_tokens, Dq),
         "vision_model.embeddings.class_embedding": (1, 1, Dv),
         "vision_model.embeddings.position_embedding": (1, tok, Dv),
         "vision_model.embeddings.patch_embedding.weight": (Dv, 3, v.patch_size, v.patch_size),
         "vision_model.embeddings.patch_embedding.bias": (Dv,),
         "vision_model.post_layernorm.weight": (Dv,), "vision_model.post_layernorm.bias": (Dv,),
         "qformer.layernorm.weight": (Dq,), "qformer.layernorm.bias": (Dq,),
         "language_projection.weight": (Dt, Dq), "language_projection.bias": (Dt,)}
    if is_t5:
        s.update(t5_shapes(t))
    else:
        s.update({"language_model.model.decoder.embed_tokens.weight": (t.vocab_size, Dt),
                  "language_model.model.decoder.embed_positions.weight": (t.max_position_embeddings + 2, Dt),
                  "language_model.model.decoder.final_layer_norm.weight": (Dt,),
                  "language_model.model.decoder.final_layer_norm.bias": (Dt,)})
    vs = {"ln1_w": (Dv,), "ln1_b": (Dv,), "qkv_w": (3 * Dv, Dv), "qkv_b": (3 * Dv,), "proj_w": (Dv, Dv), "proj_b": (Dv,),
          "ln2_w": (Dv,), "ln2_b": (Dv,), "fc1_w": (Fv, Dv), "fc1_b": (Fv,), "fc2_w": (Dv, Fv), "fc2_b": (Dv,)}
    for i in range(v.num_hidden_layers):
        for f, k in abi.vit_layer_keys(i).items():
            s[k] = vs[f]
    for i in range(q.num_hidden_layers):
        cross = i % q.cross_attention_frequency == 0
        for f, k in abi.qf_layer_keys(i, cross).items():
            if f in ("ck_w", "cv_w"):
                shp = (Dq, q.encoder_hidden_size)
            elif f == "fi_w":
                shp = (Fq, Dq)
            elif f == "fi_b":
                shp = (Fq,)
            elif f == "fo_w":
                shp = (Dq, Fq)
            elif f.endswith("_w") and "ln" not in f:
                shp = (Dq, Dq)
            else:
                shp = (Dq,)
            s[k] = shp
    os_ = {"fc1_w": (Ft, Dt), "fc1_b": (Ft,), "fc2_w": (Dt, Ft)}
    for i in range(0 if is_t5 else t.num_hidden_layers):
        for f, k in abi.opt_layer_keys(i).items():
            s[k] = os_.get(f, (Dt, Dt) if f in ("q_w", "k_w", "v_w", "o_w") else (Dt,))
    return s


def t5_shapes(t):
    """(key -> shape) of the T5ForConditionalGeneration part (hf models/t5/modeling_t5.py; `lm_head` and the two
    `embed_tokens` are the same tensor as `shared` under the installed transformers and are not listed)."""
    D, I, F = t.d_model, t.num_heads * t.d_kv, t.d_ff
    s = {"language_model.shared.weight": (t.vocab_size, D),
         "language_model.encoder.final_layer_norm.weight": (D,),
         "language_model.decoder.final_layer_norm.weight": (D,)}
    for stack, n in (("encoder", t.num_layers), ("decoder", t.num_decoder_layers)):
        for i in range(n):
            for f, k in abi.t5_layer_keys(stack, i).items():
                if f.startswith("ln"):
                    s[k] = (D,)
                elif f in ("o_w", "co_w"):
                    s[k] = (D, I)
                elif f in ("wi0_w", "wi1_w"):
                    s[k] = (F, D)
                elif f == "wo_w":
                    s[k] = (D, F)
                else:
                    s[k] = (I, D)
        s[f"language_model.{stack}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = (t.relative_attention_num_buckets, t.num_heads)
    return s
