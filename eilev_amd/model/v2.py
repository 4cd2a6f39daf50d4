"""VideoBLIP / EILeV model API on the MI355X-native forward path.

Keeps the class surface the reference's callers import (ref:eilev/model/v2.py:20-324): ``VideoBlipVisionModel`` and
``VideoBlipForConditionalGeneration`` with ``forward`` / ``generate`` / ``from_pretrained`` / ``save_pretrained`` /
``.config`` / ``.device`` / ``.dtype`` / ``.vision_model`` / ``.qformer`` / ``.language_model`` /
``.language_projection`` / ``.query_tokens``.  The modules below are PARAMETER CONTAINERS with the checkpoint's
state-dict names; none of them has arithmetic in its ``forward``.  All FLOPs run in
``libeilev_hip.so`` through :class:`eilev_amd.engine.HipEngine`.  There is no CPU / eager fallback: calling
``forward`` or ``generate`` on a model that is not on an AMD GPU raises.

Not built (raise ``NotImplementedError``): contrastive / group-beam decoding (greedy, multinomial sampling, beam search and beam-search sampling
are).  ``output_hidden_states`` / ``output_attentions`` inside the full model's ``forward`` are served from slow paths for the vision wrapper,
the Q-Former (self- and cross-attention weights), the OPT language model and the T5 stacks (hidden states; self- and cross-attention weights).  ``decoder_attention_mask`` with padding is honoured on the evaluation route (the first target position of a row must stay visible).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from transformers import Blip2Config, PreTrainedModel
from transformers.modeling_outputs import (
    BaseModelOutputWithPooling,
    BaseModelOutputWithPoolingAndCrossAttentions,
    CausalLMOutputWithPast,
)
from transformers.models.blip_2.modeling_blip_2 import Blip2ForConditionalGenerationModelOutput


# ---- parameter containers (names = checkpoint keys) -----------------------------------------------------
def _bag(**children) -> nn.Module:
    m = nn.Module()
    for k, v in children.items():
        if isinstance(v, nn.Parameter):
            m.register_parameter(k, v)
        else:
            m.add_module(k, v)
    return m


def _vit_params(c) -> nn.Module:
    d, f = c.hidden_size, c.intermediate_size
    tok = (c.image_size // c.patch_size) ** 2 + 1
    emb = _bag(class_embedding=nn.Parameter(torch.zeros(1, 1, d)), position_embedding=nn.Parameter(torch.zeros(1, tok, d)),
               patch_embedding=nn.Conv2d(3, d, c.patch_size, c.patch_size))
    layers = nn.ModuleList(
        _bag(self_attn=_bag(qkv=nn.Linear(d, 3 * d), projection=nn.Linear(d, d)), layer_norm1=nn.LayerNorm(d, eps=c.layer_norm_eps),
             mlp=_bag(fc1=nn.Linear(d, f), fc2=nn.Linear(f, d)), layer_norm2=nn.LayerNorm(d, eps=c.layer_norm_eps))
        for _ in range(c.num_hidden_layers))
    return emb, _bag(layers=layers), nn.LayerNorm(d, eps=c.layer_norm_eps)


def _qformer_params(c) -> nn.Module:
    d, f = c.hidden_size, c.intermediate_size

    def attn(kv):
        return _bag(attention=_bag(query=nn.Linear(d, d), key=nn.Linear(kv, d), value=nn.Linear(kv, d)),
                    output=_bag(dense=nn.Linear(d, d), LayerNorm=nn.LayerNorm(d, eps=c.layer_norm_eps)))

    blocks = []
    for i in range(c.num_hidden_layers):
        parts = dict(attention=attn(d))
        if i % c.cross_attention_frequency == 0:
            parts["crossattention"] = attn(c.encoder_hidden_size)
        parts["intermediate_query"] = _bag(dense=nn.Linear(d, f))
        parts["output_query"] = _bag(dense=nn.Linear(f, d), LayerNorm=nn.LayerNorm(d, eps=c.layer_norm_eps))
        blocks.append(_bag(**parts))
    return _bag(layernorm=nn.LayerNorm(d, eps=c.layer_norm_eps), encoder=_bag(layer=nn.ModuleList(blocks)))


class _OptParams(nn.Module):
    """OPTForCausalLM's parameter tree (hf modeling_opt.py:443-524) without its arithmetic."""

    def __init__(self, c):
        super().__init__()
        d, f = c.hidden_size, c.ffn_dim
        layers = nn.ModuleList(
            _bag(self_attn=_bag(k_proj=nn.Linear(d, d), v_proj=nn.Linear(d, d), q_proj=nn.Linear(d, d), out_proj=nn.Linear(d, d)),
                 self_attn_layer_norm=nn.LayerNorm(d), fc1=nn.Linear(d, f), fc2=nn.Linear(f, d), final_layer_norm=nn.LayerNorm(d))
            for _ in range(c.num_hidden_layers))
        dec = _bag(embed_tokens=nn.Embedding(c.vocab_size, d, c.pad_token_id),
                   embed_positions=nn.Embedding(c.max_position_embeddings + 2, d), final_layer_norm=nn.LayerNorm(d), layers=layers)
        self.model = _bag(decoder=dec)
        self.lm_head = nn.Linear(d, c.vocab_size, bias=False)
        self.lm_head.weight = dec.embed_tokens.weight  # tied (hf modeling_opt.py:444)
        self.config = c

    def get_input_embeddings(self):
        return self.model.decoder.embed_tokens

    def forward(self, *a, **k):
        raise RuntimeError("language_model is a parameter container; call the parent VideoBlipForConditionalGeneration")


class _RmsW(nn.Module):
    """T5LayerNorm's parameter (hf models/t5/modeling_t5.py:50-57): a weight, no bias."""

    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))


class _T5Params(nn.Module):
    """T5ForConditionalGeneration's parameter tree (hf modeling_t5.py:898-937) without its arithmetic; `lm_head` and both
    `embed_tokens` are tied to `shared` like the installed transformers does."""

    def __init__(self, c):
        super().__init__()
        if c.feed_forward_proj != "gated-gelu":
            raise NotImplementedError("only the gated-gelu (T5 v1.1 / flan-t5) feed-forward is built on the HIP path")
        d, inner, f = c.d_model, c.num_heads * c.d_kv, c.d_ff
        lin = lambda i, o: nn.Linear(i, o, bias=False)

        def attn(rel):
            parts = dict(q=lin(d, inner), k=lin(d, inner), v=lin(d, inner), o=lin(inner, d))
            if rel:
                parts["relative_attention_bias"] = nn.Embedding(c.relative_attention_num_buckets, c.num_heads)
            return _bag(**parts)

        def stack(n, decoder):
            blocks = []
            for i in range(n):
                layers = [_bag(SelfAttention=attn(i == 0), layer_norm=_RmsW(d))]
                if decoder:
                    layers.append(_bag(EncDecAttention=attn(False), layer_norm=_RmsW(d)))
                layers.append(_bag(DenseReluDense=_bag(wi_0=lin(d, f), wi_1=lin(d, f), wo=lin(f, d)), layer_norm=_RmsW(d)))
                blocks.append(_bag(layer=nn.ModuleList(layers)))
            return _bag(embed_tokens=nn.Embedding(c.vocab_size, d), block=nn.ModuleList(blocks), final_layer_norm=_RmsW(d))

        self.shared = nn.Embedding(c.vocab_size, d)
        self.encoder = stack(c.num_layers, False)
        self.decoder = stack(c.num_decoder_layers, True)
        self.lm_head = lin(d, c.vocab_size)
        self.encoder.embed_tokens.weight = self.shared.weight
        self.decoder.embed_tokens.weight = self.shared.weight
        self.lm_head.weight = self.shared.weight
        self.config = c

    def get_input_embeddings(self):
        return self.shared

    def forward(self, *a, **k):
        raise RuntimeError("language_model is a parameter container; call the parent VideoBlipForConditionalGeneration")


def _require_gpu(t: torch.Tensor, what: str):
    if t.device.type != "cuda":
        raise RuntimeError(f"{what}: the MI355X-native path has no CPU fallback — move the model to an AMD GPU (.to('cuda'))")


class VideoBlipVisionModel(nn.Module):
    """(num_videos, C, T, H, W) -> ViT-g over every frame -> (num_videos, T * tokens, D)  [ref:eilev/model/v2.py:20-103]."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings, self.encoder, self.post_layernorm = _vit_params(config)
        self._engine_ref = None  # set by the parent model; a stand-alone vision model builds a ViT-only engine
        self._own_engine = None

    def _engine(self):
        if self._engine_ref is not None:
            return self._engine_ref()
        key = _params_key(self)
        if self._own_engine is None or self._own_engine[0] != key:
            from ..engine import HipEngine
            from transformers import Blip2Config as _C

            named = {"vision_model." + k: v for k, v in self.state_dict().items()}
            full = _C(vision_config=self.config.to_dict())
            self._own_engine = (key, HipEngine(full, named, device=next(self.parameters()).device, parts=("vit",)))
        return self._own_engine[1]

    @torch.no_grad()
    def forward(self, pixel_values=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        _require_gpu(next(self.parameters()), "VideoBlipVisionModel.forward")
        dtype = next(self.parameters()).dtype
        hidden = attn = None
        if output_attentions or output_hidden_states:
            # debug outputs: the slow path of the library (eilev_vit_forward_debug), shapes as ref:eilev/model/v2.py:76-103:
            # hidden_states = layers + 1 tensors (N, T*tokens, D), attentions = layers tensors (N, T, heads, tokens, tokens)
            last, pooled, hid, att = self._engine().vit_debug(pixel_values, bool(output_hidden_states), bool(output_attentions))
            if hid is not None:
                hidden = tuple(h.to(dtype) for h in hid)
            if att is not None:
                attn = tuple(a.to(dtype) for a in att)
        else:
            last, pooled = self._engine().vit(pixel_values, want_pooler=True)
        last, pooled = last.to(dtype), pooled.to(dtype)
        if return_dict is False:
            return (last, pooled, hidden, attn)  # fixed 4-tuple with None placeholders, as ref:eilev/model/v2.py:103
        return BaseModelOutputWithPooling(last_hidden_state=last, pooler_output=pooled, hidden_states=hidden, attentions=attn)


def _params_key(module: nn.Module):
    return tuple((p.data_ptr(), p._version, p.dtype) for p in module.parameters())


class VideoBlipForConditionalGeneration(PreTrainedModel):
    config_class = Blip2Config
    config: Blip2Config
    base_model_prefix = "blip"
    main_input_name = "pixel_values"
    _tied_weights_keys = {"language_model.lm_head.weight": "language_model.model.decoder.embed_tokens.weight"}
    _supports_sdpa = False

    def __init__(self, config: Blip2Config) -> None:
        super().__init__(config)
        mt = config.text_config.model_type
        if (config.use_decoder_only_language_model and mt != "opt") or (not config.use_decoder_only_language_model and mt != "t5"):
            raise NotImplementedError(f"language model {mt!r}: OPT (decoder-only) and T5 (encoder-decoder) are built on the HIP path")
        self._is_t5 = mt == "t5"
        if self._is_t5:
            self._tied_weights_keys = {f"language_model.{k}": "language_model.shared.weight"
                                       for k in ("lm_head.weight", "encoder.embed_tokens.weight", "decoder.embed_tokens.weight")}
        self.vision_model = VideoBlipVisionModel(config.vision_config)
        self.query_tokens = nn.Parameter(torch.zeros(1, config.num_query_tokens, config.qformer_config.hidden_size))
        self.qformer = _qformer_params(config.qformer_config)
        self.language_projection = nn.Linear(config.qformer_config.hidden_size, config.text_config.hidden_size)
        self.language_model = _T5Params(config.text_config) if self._is_t5 else _OptParams(config.text_config)
        self._hip = None
        import weakref

        me = weakref.ref(self)
        self.vision_model._engine_ref = lambda: me().engine()
        self.post_init()

    # ---- HF plumbing -------------------------------------------------------------------------------------
    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, (nn.Linear, nn.Conv2d, nn.Embedding)):
            nn.init.normal_(module.weight, mean=0.0, std=std)
            if getattr(module, "bias", None) is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            nn.init.ones_(module.weight)
            nn.init.zeros_(module.bias)

    def get_input_embeddings(self):
        return self.language_model.get_input_embeddings()

    def set_input_embeddings(self, value):
        if self._is_t5:
            self.language_model.shared = value
        else:
            self.language_model.model.decoder.embed_tokens = value

    def get_output_embeddings(self):
        return self.language_model.lm_head

    def tie_weights(self, *a, **k):
        lm = self.language_model
        if self._is_t5:
            lm.encoder.embed_tokens.weight = lm.decoder.embed_tokens.weight = lm.lm_head.weight = lm.shared.weight
        else:
            lm.lm_head.weight = lm.model.decoder.embed_tokens.weight

    def _preprocess_accelerate(self):
        """ref:eilev/model/v2.py:276-278 calls this when `accelerate` attached an `hf_device_map`.  The HIP engine keeps ONE bf16 copy of
        all weights on the model's device, so there is nothing to re-home: a no-op (a map that spreads the model over several devices
        is refused when the engine is built: every parameter must live on one AMD GPU)."""
        return None

    # ---- engine ------------------------------------------------------------------------------------------
    def engine(self):
        """bf16 device copy of the weights + the HIP library; rebuilt when a parameter changed (optimizer step,
        load_state_dict, .to())."""
        lm_weights = getattr(self, "hip_lm_weights", "bf16")  # "fp8": e4m3 weight-only OPT linears (set the attribute before use)
        ln_fold = bool(getattr(self, "hip_vit_ln_fold", True))  # False: large ViT launches keep their LayerNorm kernels
        key = (_params_key(self), lm_weights, ln_fold)
        if self._hip is None or self._hip[0] != key:
            from ..engine import HipEngine

            _require_gpu(self.query_tokens, type(self).__name__)
            self._hip = (key, HipEngine(self.config, dict(self.state_dict()), device=self.query_tokens.device, lm_weights=lm_weights,
                                        vit_ln_fold=ln_fold))
        return self._hip[1]

    def _qformer_attn_tuple(self, qa, dtype):
        """`qformer_outputs.attentions` as the installed transformers fills it: the weights of EVERY attention module of the Q-Former in
        execution order — self_0, cross_0, self_1, ... (cross-attention every `cross_attention_frequency` blocks); `cross_attentions` holds
        the cross ones again (pinned by tests/golden/mid_attndebug.npz)."""
        selfs, crosses = qa
        out, ci = [], 0
        for i in range(selfs.shape[0]):
            out.append(selfs[i].to(dtype))
            if i % self.config.qformer_config.cross_attention_frequency == 0:
                out.append(crosses[ci].to(dtype))
                ci += 1
        return tuple(out)

    def _encode(self, pixel_values, input_ids, video_input_mask, vision_debug=(False, False)):
        eng = self.engine()
        feats = None
        vision = qf = None
        if pixel_values is not None:
            assert video_input_mask is not None
            if any(vision_debug):  # slow path: per-block hidden states / attention maps of the ViT
                img, pooled, hid, att = eng.vit_debug(pixel_values, want_hidden=vision_debug[0], want_attn=vision_debug[1])
                self._vision_debug = (None if hid is None else tuple(hid), None if att is None else tuple(att))
            else:
                img, pooled = eng.vit(pixel_values, want_pooler=True)
            q = eng.qformer(img)
            qh = eng.qformer_hidden_states(img) if any(vision_debug) else None
            self._qformer_debug = qh if vision_debug[0] else None
            self._qformer_attn = eng.qformer_attentions(img, qh) if vision_debug[1] else None
            feats = eng.project(q)
            vision, qf = (img, pooled), q
        emb = eng.embed_scatter(input_ids, video_input_mask if feats is not None else None, feats)
        return emb, vision, qf

    # ---- API ---------------------------------------------------------------------------------------------
    def forward(self, input_ids, attention_mask=None, pixel_values=None, video_input_mask=None, decoder_input_ids=None,
                decoder_attention_mask=None, output_attentions=None, output_hidden_states=None, labels=None, return_dict=None):
        """pixel_values: (num_videos, C, T, H, W); video_input_mask: (batch, seq_len)  [ref:eilev/model/v2.py:132-252].

        With ``labels``, autograd enabled and at least one trainable parameter (what `Trainer` does for train_v2:
        ref:scripts/general/train_v2.py:124-130, 207-217) the call returns a loss with a gradient, computed by the training
        graph of eilev_amd/train.py (logits come from the same forward, detached); dropout is applied in ``train()`` mode only.
        Otherwise (no labels, ``torch.no_grad()``, or nothing trainable) it runs the inference kernels without autograd."""
        # the differentiable route is chosen by what the CALL needs (labels, autograd on, a train_v2-style trainable set), see
        # _wants_graph: `model.eval()` + `loss.backward()` (fine-tuning with dropout off) gets a loss with a graph too; the module
        # mode decides whether dropout is applied, as in the reference
        if labels is not None and torch.is_grad_enabled() and self._wants_graph():
            if decoder_input_ids is not None or (decoder_attention_mask is not None and not bool((decoder_attention_mask != 0).all())):
                # (the reference's collators emit neither: ref:eilev/data/utils.py:148-200; hf derives decoder_input_ids from labels)
                raise NotImplementedError("the training graph shifts `labels` into decoder_input_ids itself: explicit decoder_input_ids / a padded "
                                          "decoder_attention_mask are served by the evaluation route only (torch.no_grad())")
            return self._forward_train(input_ids, attention_mask, pixel_values, video_input_mask, labels, return_dict)
        return self._forward_eval(input_ids, attention_mask, pixel_values, video_input_mask, decoder_input_ids, decoder_attention_mask,
                                  output_attentions, output_hidden_states, labels, return_dict)

    def _wants_graph(self) -> bool:
        """train() mode with anything trainable -> the training graph (which refuses, loudly, a trainable ViT / language model).
        eval() mode -> the graph only when the model is frozen the way train_v2 freezes it (ViT + language model frozen, something
        else trainable): `model.eval(); loss.backward()` then works, while a freshly built model (every parameter trainable) that
        is merely evaluated without `torch.no_grad()` keeps getting logits from the inference route."""
        trainable = [n for n, p in self.named_parameters() if p.requires_grad]
        if not trainable:
            return False
        if self.training:
            return True
        return not any(n.startswith(("vision_model.", "language_model.")) for n in trainable)

    def _forward_train(self, input_ids, attention_mask, pixel_values, video_input_mask, labels, return_dict):
        from ..engine import HipEngine
        from ..train import TrainGraph

        named = dict(self.named_parameters())
        params = {k: p for k, p in named.items() if p.requires_grad}
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in named.values() if not p.requires_grad)
        cached = getattr(self, "_hip_train", None)
        if cached is None or cached[0] != key:  # frozen weights -> bf16 device copies, once (trainable ones are read live)
            _require_gpu(self.query_tokens, type(self).__name__)
            cached = self._hip_train = (key, HipEngine(self.config, dict(self.state_dict()), device=self.query_tokens.device))
        # train() mode = dropout on, as under the reference's Trainer; a fresh mask seed every call (torch's seed + a call counter)
        self._hip_train_calls = getattr(self, "_hip_train_calls", 0) + 1
        seed = (torch.initial_seed() * 1000003 + self._hip_train_calls) & 0xFFFFFFFF
        graph = TrainGraph(cached[1], params, dropout=self.training and getattr(self, "hip_train_dropout", True), seed=seed)
        loss = graph.loss(input_ids, attention_mask, pixel_values, video_input_mask, labels)
        # the reference's output always carries the logits (ref:eilev/model/v2.py:239-252): lm_head over the hidden states this very
        # forward produced, outside the graph (one GEMM, ~0.3 ms and 96 MB per 960-token sample at OPT-2.7B);
        # `model.hip_train_logits = False` skips them (Trainer only consumes the loss)
        logits = graph.logits().to(self.dtype) if getattr(self, "hip_train_logits", True) else None
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        if self._is_t5:
            from transformers.modeling_outputs import Seq2SeqLMOutput

            lm_out = Seq2SeqLMOutput(loss=loss, logits=logits)
        else:
            lm_out = CausalLMOutputWithPast(loss=loss, logits=logits)
        if not return_dict:
            return (loss, logits, None, None, lm_out)
        return Blip2ForConditionalGenerationModelOutput(loss=loss, logits=logits, vision_outputs=None, qformer_outputs=None,
                                                        language_model_outputs=lm_out)

    @torch.no_grad()
    def _forward_eval(self, input_ids, attention_mask=None, pixel_values=None, video_input_mask=None, decoder_input_ids=None,
                      decoder_attention_mask=None, output_attentions=None, output_hidden_states=None, labels=None, return_dict=None):
        if pixel_values is not None:
            assert video_input_mask is not None
        # output_hidden_states / output_attentions: the VISION outputs carry both (slow path of the library, what
        # ref:tests/model/test_model_v2.py:57-83 asserts on the vision wrapper) and the Q-Former output carries hidden_states (r3: the stack
        # re-run with its first i blocks) and the OPT language model's output carries hidden_states (r3: eilev_opt_prefill_debug); r4: the
        # attention weights of the Q-Former (self and cross) and of the OPT language model (eilev_attention_probs on q / k recomputed from the
        # per-block inputs); the T5 stacks serve both (eilev_t5_*_debug, eilev_attention_probs with the relative position bias)
        self._vision_debug = (None, None)
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        dtype = self.dtype
        emb, vision, qf = self._encode(pixel_values, input_ids, video_input_mask,
                                       vision_debug=(bool(output_hidden_states), bool(output_attentions)))
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if self._is_t5:
            return self._forward_t5(emb, vision, qf, attention_mask, decoder_input_ids, decoder_attention_mask, labels, return_dict,
                                    bool(output_hidden_states), bool(output_attentions))
        lm_hidden = lm_attn = None
        if output_hidden_states or output_attentions:  # hf OPTDecoder's tuple: every block's input, then the output of final_layer_norm
            _, logits32, _, hs = self.engine().prefill(emb, attention_mask, all_logits=True, last_logits=False, hidden_states=True)
            if output_hidden_states:
                lm_hidden = tuple(h.to(dtype) for h in hs.unbind(0))
            if output_attentions:  # (round 4) hf eager `attn_weights` of every block: (B, heads, L, L)
                lm_attn = tuple(a.to(dtype) for a in self.engine().lm_attentions(hs, attention_mask).unbind(0))
        else:
            _, logits32, _ = self.engine().prefill(emb, attention_mask, all_logits=True, last_logits=False)
        loss = None
        if labels is not None:
            # HF causal-LM loss: shifted CE, ignore_index -100 (hf loss_utils.ForCausalLMLoss)
            shift = logits32[:, :-1].reshape(-1, logits32.size(-1))
            tgt = labels.to(logits32.device)[:, 1:].reshape(-1)
            loss = self.engine().ce_mean(shift, tgt).to(dtype)
        logits = logits32.to(dtype)
        vis_out = qf_out = None
        if vision is not None:
            vh, va = getattr(self, "_vision_debug", (None, None))
            vis_out = BaseModelOutputWithPooling(last_hidden_state=vision[0].to(dtype), pooler_output=vision[1].to(dtype),
                                                 hidden_states=None if vh is None else tuple(h.to(dtype) for h in vh),
                                                 attentions=None if va is None else tuple(a.to(dtype) for a in va))
            qh = getattr(self, "_qformer_debug", None)
            qa = getattr(self, "_qformer_attn", None)
            qf_out = BaseModelOutputWithPoolingAndCrossAttentions(last_hidden_state=qf.to(dtype), pooler_output=qf[:, 0].to(dtype),
                                                                  hidden_states=None if qh is None else tuple(h.to(dtype) for h in qh),
                                                                  attentions=None if qa is None else self._qformer_attn_tuple(qa, dtype),
                                                                  cross_attentions=None if qa is None else tuple(a.to(dtype) for a in qa[1]))
        lm_out = CausalLMOutputWithPast(loss=loss, logits=logits, hidden_states=lm_hidden, attentions=lm_attn)
        if not return_dict:
            out = (logits, vis_out, qf_out, lm_out)
            return ((loss,) + out) if loss is not None else out
        return Blip2ForConditionalGenerationModelOutput(loss=loss, logits=logits, vision_outputs=vis_out, qformer_outputs=qf_out,
                                                        language_model_outputs=lm_out)

    def _forward_t5(self, emb, vision, qf, attention_mask, decoder_input_ids, decoder_attention_mask, labels, return_dict,
                    output_hidden_states=False, output_attentions=False):
        """Encoder-decoder branch [ref:eilev/model/v2.py:228-238 -> hf T5ForConditionalGeneration.forward :939-1055].  decoder_attention_mask
        with padding and output_hidden_states (both stacks' tuples) go through the *_debug entries (round 4); output_attentions: the three
        hf tuples (encoder / decoder self-attention, cross-attention) from eilev_attention_probs on q / k recomputed from the block inputs."""
        from transformers.modeling_outputs import Seq2SeqLMOutput

        t = self.config.text_config
        if decoder_input_ids is None:
            if labels is None:
                raise ValueError("You have to specify either decoder_input_ids or labels")
            # T5._shift_right: start token, labels shifted, -100 -> pad
            start = t.decoder_start_token_id if t.decoder_start_token_id is not None else t.pad_token_id
            decoder_input_ids = torch.cat((torch.full_like(labels[:, :1], start), labels[:, :-1]), dim=1)
            decoder_input_ids = decoder_input_ids.masked_fill(decoder_input_ids == -100, t.pad_token_id)
        dtype = self.dtype
        enc_hs = dec_hs = None
        padded = decoder_attention_mask is not None and not bool((decoder_attention_mask != 0).all())
        enc_at = dec_at = cross_at = None
        if padded or output_hidden_states or output_attentions:
            logits32, enc, enc_hs, dec_hs = self.engine().t5_forward_debug(emb, attention_mask, decoder_input_ids,
                                                                         decoder_attention_mask if padded else None,
                                                                         output_hidden_states or output_attentions)
            if output_attentions:
                enc_at, dec_at, cross_at = (tuple(a.to(self.dtype) for a in t_.unbind(0)) for t_ in
                                            self.engine().t5_attentions(enc_hs, dec_hs, attention_mask, decoder_attention_mask if padded else None))
            if not output_hidden_states:
                enc_hs = dec_hs = None
        else:
            logits32, enc = self.engine().t5_forward(emb, attention_mask, decoder_input_ids)
        loss = None
        if labels is not None:
            loss = self.engine().ce_mean(logits32.reshape(-1, logits32.size(-1)), labels.to(logits32.device).reshape(-1)).to(dtype)
        logits = logits32.to(dtype)
        vis_out = qf_out = None
        if vision is not None:
            vh, va = getattr(self, "_vision_debug", (None, None))
            vis_out = BaseModelOutputWithPooling(last_hidden_state=vision[0].to(dtype), pooler_output=vision[1].to(dtype),
                                                 hidden_states=None if vh is None else tuple(h.to(dtype) for h in vh),
                                                 attentions=None if va is None else tuple(a.to(dtype) for a in va))
            qh = getattr(self, "_qformer_debug", None)
            qa = getattr(self, "_qformer_attn", None)
            qf_out = BaseModelOutputWithPoolingAndCrossAttentions(last_hidden_state=qf.to(dtype), pooler_output=qf[:, 0].to(dtype),
                                                                  hidden_states=None if qh is None else tuple(h.to(dtype) for h in qh),
                                                                  attentions=None if qa is None else self._qformer_attn_tuple(qa, dtype),
                                                                  cross_attentions=None if qa is None else tuple(a.to(dtype) for a in qa[1]))
        lm_out = Seq2SeqLMOutput(loss=loss, logits=logits, encoder_last_hidden_state=enc.to(dtype),
                                 encoder_hidden_states=None if enc_hs is None else tuple(h.to(dtype) for h in enc_hs.unbind(0)),
                                 decoder_hidden_states=None if dec_hs is None else tuple(h.to(dtype) for h in dec_hs.unbind(0)),
                                 encoder_attentions=enc_at, decoder_attentions=dec_at, cross_attentions=cross_at)
        if not return_dict:
            out = (logits, vis_out, qf_out, lm_out)
            return ((loss,) + out) if loss is not None else out
        return Blip2ForConditionalGenerationModelOutput(loss=loss, logits=logits, vision_outputs=vis_out, qformer_outputs=qf_out,
                                                        language_model_outputs=lm_out)

    @torch.no_grad()
    def generate(self, input_ids, pixel_values=None, video_input_mask=None, attention_mask=None, **generate_kwargs):
        """Greedy decoding on the HIP path; returns only the NEW tokens like the reference does for OPT
        (ref:eilev/model/v2.py:254-324 drives the LM with inputs_embeds)."""
        assert not (input_ids is None and pixel_values is None)
        if pixel_values is not None:
            assert video_input_mask is not None
        if hasattr(self, "hf_device_map"):
            self._preprocess_accelerate()
        kw = dict(generate_kwargs)
        num_beams = kw.pop("num_beams", 1)
        do_sample = kw.pop("do_sample", False)
        length_penalty = kw.pop("length_penalty", 1.0)  # only affects beam search
        early_stopping = kw.pop("early_stopping", False)
        num_return = kw.pop("num_return_sequences", 1)
        num_beams = 1 if num_beams is None else int(num_beams)
        if kw.get("penalty_alpha") or kw.get("num_beam_groups", 1) not in (None, 1):
            raise NotImplementedError("contrastive / group-beam decoding is not built on the HIP path (greedy, multinomial sampling, beam "
                                      "search and beam-search sampling are)")
        sampler = None
        if do_sample:  # hf GenerationConfig defaults: temperature 1.0, top_k 50, top_p 1.0
            gen_cfg0 = getattr(self, "generation_config", None)
            dflt = lambda name, v: getattr(gen_cfg0, name, v) if gen_cfg0 is not None and getattr(gen_cfg0, name, None) is not None else v
            sampler = dict(temperature=float(kw.pop("temperature", dflt("temperature", 1.0))), top_k=int(kw.pop("top_k", dflt("top_k", 50)) or 0),
                           top_p=float(kw.pop("top_p", dflt("top_p", 1.0))), generator=kw.pop("generator", None))
        else:
            for k in ("temperature", "top_k", "top_p"):
                kw.pop(k, None)  # ignored by greedy / beam search, as in HF (which only warns)
        max_new = kw.pop("max_new_tokens", None)
        if max_new is None:
            max_len = kw.pop("max_length", None)
            if max_len is None:
                max_len = 20  # HF GenerationConfig default max_length
                max_new = 19 if self._is_t5 else 20
            elif self._is_t5:
                max_new = int(max_len) - 1  # encoder-decoder: max_length counts the decoder tokens incl. the start token
            else:
                max_new = int(max_len) - input_ids.shape[1]
        min_new = int(kw.pop("min_new_tokens", 0) or 0)
        gen_cfg = getattr(self, "generation_config", None)
        eos = kw.pop("eos_token_id", getattr(gen_cfg, "eos_token_id", None) if gen_cfg is not None else None)
        if eos is None:
            eos = self.config.text_config.eos_token_id
        from ..sampling import eos_list

        eos_ids = eos_list(eos)  # hf accepts an int or a list of ids; any of them finishes a row
        pad = kw.pop("pad_token_id", None)
        if pad is None:
            pad = self.config.text_config.pad_token_id if self.config.text_config.pad_token_id is not None else (eos_ids[0] if eos_ids else 0)
        had_eos = bool(eos_ids)
        if min_new >= max_new:
            eos_ids = []  # EOS can never fire before the budget is exhausted
            min_new = 0
        kw.pop("use_cache", None)
        # hf forwards these to GenerationMixin (ref:eilev/model/v2.py:318-322): the result becomes a ModelOutput with `sequences` (+ `scores` /
        # `logits`: one (rows, vocab) fp32 tensor per generated token).  Scores are served for greedy search on the decoder-only LM — the
        # per-step logits of the eager decode loop (`return_step_logits`), which for greedy search without processors ARE hf's processed scores;
        # like hf, output_scores / output_logits without return_dict_in_generate change nothing.
        want_dict = bool(kw.pop("return_dict_in_generate", False))
        want_scores, want_logits = bool(kw.pop("output_scores", False)) and want_dict, bool(kw.pop("output_logits", False)) and want_dict
        # hf hands every other kwarg to GenerationMixin (ref:eilev/model/v2.py:318-322).  Logits processors and stopping criteria run in the
        # host loops over the same HIP decode step, with transformers' own processor classes (exactly hf's arithmetic and order).
        from transformers import (LogitsProcessorList, MaxTimeCriteria, NoRepeatNGramLogitsProcessor, RepetitionPenaltyLogitsProcessor,
                                  StoppingCriteriaList)

        procs = LogitsProcessorList()
        rp = kw.pop("repetition_penalty", None)
        if rp is not None and float(rp) != 1.0:
            procs.append(RepetitionPenaltyLogitsProcessor(penalty=float(rp)))
        ngram = kw.pop("no_repeat_ngram_size", None)
        if ngram:
            procs.append(NoRepeatNGramLogitsProcessor(int(ngram)))
        user_procs = kw.pop("logits_processor", None)
        if user_procs:
            procs.extend(user_procs)
        crit = StoppingCriteriaList(kw.pop("stopping_criteria", None) or [])
        max_time = kw.pop("max_time", None)
        if max_time is not None:
            crit.append(MaxTimeCriteria(max_time=float(max_time)))
        rules = None
        if len(procs) or len(crit):
            rules = dict(processors=procs if len(procs) else None, stopping=crit if len(crit) else None)
            if had_eos and not eos_ids:  # the EOS list was emptied above, the model still has an EOS id: hf fills shortened hypotheses with the pad id
                rules["fill_id"] = int(pad)
        if kw:
            raise NotImplementedError(f"unsupported generate() arguments on the HIP path: {sorted(kw)}")
        if want_dict and not (want_scores or want_logits):  # sequences only: any decoding mode, wrapped like hf wraps it
            from transformers.generation.utils import GenerateDecoderOnlyOutput, GenerateEncoderDecoderOutput

            plain = {k: v for k, v in generate_kwargs.items() if k not in ("return_dict_in_generate", "output_scores", "output_logits")}
            seq = self.generate(input_ids, pixel_values=pixel_values, video_input_mask=video_input_mask, attention_mask=attention_mask, **plain)
            return (GenerateEncoderDecoderOutput if self._is_t5 else GenerateDecoderOnlyOutput)(sequences=seq)
        if num_beams == 1 and not do_sample and int(num_return) != 1:
            raise ValueError("Greedy methods without beam search do not support `num_return_sequences` different than 1")
        if num_beams > 1 and int(num_return) > num_beams:
            raise ValueError("`num_return_sequences` has to be smaller or equal to `num_beams`")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        emb, _, _ = self._encode(pixel_values, input_ids, video_input_mask)
        if sampler is not None and num_beams == 1 and int(num_return) > 1:
            # hf `_expand_inputs_for_generation`: every prompt is repeated num_return_sequences times (rows of one prompt adjacent)
            emb = emb.repeat_interleave(int(num_return), dim=0)
            attention_mask = attention_mask.repeat_interleave(int(num_return), dim=0)
        # the captured device step knows ONE eos id and no minimum length; several ids or 0 < min_new_tokens < max_new_tokens run the
        # same HIP decode step with the stopping rule applied by the host loop (eilev_amd/sampling.py, beam.py)
        eos1 = eos_ids[0] if eos_ids else -1
        host_rules = len(eos_ids) > 1 or min_new > 0 or rules is not None
        eng = self.engine()
        if want_dict:
            plain_greedy = num_beams == 1 and sampler is None and not host_rules
            if (want_scores or want_logits) and not (plain_greedy and not self._is_t5):
                raise NotImplementedError("generate(output_scores / output_logits) is served for greedy search on the decoder-only language model only")
            from transformers.generation.utils import GenerateDecoderOnlyOutput

            if plain_greedy and not self._is_t5 and (want_scores or want_logits):
                ids, steps = eng.greedy_decode(emb, attention_mask, int(max_new), eos_id=int(eos1), pad_id=int(pad), use_graph=False, return_step_logits=True)
                steps = tuple(st_[:, : self.config.text_config.vocab_size].float() for st_ in steps[: ids.shape[1]])
                return GenerateDecoderOnlyOutput(sequences=ids, scores=steps if want_scores else None, logits=steps if want_logits else None)
            raise AssertionError("unreachable: the sequences-only form returned before the encode")
        if self._is_t5:
            t = self.config.text_config
            start = t.decoder_start_token_id if t.decoder_start_token_id is not None else t.pad_token_id
            if rules is not None:  # hf's processors see the decoder ids, which begin with the start token
                rules["prefix"] = torch.full((emb.shape[0], 1), int(start), dtype=torch.int64, device=emb.device)
            if num_beams > 1:
                return eng.t5_beam(emb, attention_mask, int(max_new), num_beams, float(length_penalty), eos_id=eos_ids if host_rules else eos1,
                                   pad_id=int(pad), start_id=int(start), early_stopping=early_stopping,
                                   num_return_sequences=int(num_return), sampler=sampler, min_new_tokens=min_new, rules=rules)
            if sampler is not None or host_rules:
                rule = dict(sampler) if sampler is not None else dict(greedy=True)
                return eng.t5_beam(emb, attention_mask, int(max_new), 1, eos_id=eos_ids if host_rules else eos1, pad_id=int(pad),
                                   start_id=int(start), sampler=dict(rule, min_new_tokens=min_new), rules=rules)
            return eng.t5_greedy(emb, attention_mask, int(max_new), eos_id=int(eos1), pad_id=int(pad), start_id=int(start))
        if num_beams > 1:
            return eng.beam_decode(emb, attention_mask, int(max_new), num_beams, float(length_penalty), eos_id=eos_ids if host_rules else eos1,
                                   pad_id=int(pad), early_stopping=early_stopping, num_return_sequences=int(num_return), sampler=sampler,
                                   min_new_tokens=min_new, rules=rules)
        if sampler is not None or host_rules:
            rule = dict(sampler) if sampler is not None else dict(greedy=True)
            return eng.beam_decode(emb, attention_mask, int(max_new), 1, eos_id=eos_ids if host_rules else eos1, pad_id=int(pad),
                                   sampler=dict(rule, min_new_tokens=min_new), rules=rules)
        return eng.greedy_decode(emb, attention_mask, int(max_new), eos_id=int(eos1), pad_id=int(pad))

    @torch.no_grad()
    def classify(self, prompt_input_ids, class_input_ids, prompt_attention_mask=None, pixel_values=None,
                 prompt_video_input_mask=None, class_attention_mask=None, class_batch_size=None):
        """Mean log-likelihood of each class text after each (left-padded) prompt: (batch, num_classes)
        [ref:eilev/model/v2.py:326-501].  Prompt prefilled once on the HIP path, class tokens continue its KV cache."""
        assert self.config.use_decoder_only_language_model
        if pixel_values is not None:
            assert prompt_video_input_mask is not None
        emb, _, _ = self._encode(pixel_values, prompt_input_ids, prompt_video_input_mask)
        if prompt_attention_mask is None:
            prompt_attention_mask = torch.ones_like(prompt_input_ids)
        ll = self.engine().classify_loglik(emb, prompt_attention_mask, class_input_ids, class_attention_mask, class_batch_size)
        return ll.to(self.dtype)
