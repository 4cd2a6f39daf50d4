"""Training-mode forward of the hot path: `model(**batch).loss` with a gradient.

ref:scripts/general/train_v2.py:124-130 — the ViT and the language model are frozen, the Q-Former (+ query tokens +
language projection) trains; ref:eilev/model/v2.py:132-252 is the forward whose loss this reproduces:

    frames -> ViT (frozen: the inference kernels, no graph) -> Q-Former -> language_projection -> scatter into the token
    embeddings -> OPT decoder (frozen weights, activation gradients) -> shifted token cross-entropy
                | flan-t5 encoder -> decoder on the shifted labels (frozen weights, activation gradients) -> token cross-entropy

Everything after the ViT is composed from the autograd-wrapped HIP kernels of eilev_amd/autograd.py, so
`loss.backward()` runs the gradient kernels of eilev_amd/csrc/backward.hip and leaves `.grad` on the trainable
parameters exactly as the reference's `accelerator.backward(loss)` does.  Dropout (0.1 in the Q-Former / OPT / T5 configs
while `model.train()`) is applied at the reference's sites when `TrainGraph(dropout=True)` (the model class does so in train()
mode), with counter-based masks recomputed in the backward (profiles/HISTORY.md §5h); dropout=False is the deterministic function.
"""
from __future__ import annotations

import math

import torch

from . import abi
from . import autograd as ag


class _EmbedScatter(torch.autograd.Function):
    """inputs_embeds = embed_tokens[input_ids] with the video positions replaced by the projected query tokens
    [ref:eilev/model/v2.py:316]; the gradient of the video rows is the gather of those positions."""

    @staticmethod
    def forward(ctx, feats, engine, input_ids, video_mask):
        vm = (video_mask.to(feats.device) != 0)
        ctx.save_for_backward(vm)
        return engine.embed_scatter(input_ids, vm, feats)

    @staticmethod
    def backward(ctx, d_emb):
        (vm,) = ctx.saved_tensors
        return d_emb[vm].contiguous(), None, None, None


class TrainGraph:
    """Loss of one batch on the HIP kernels.  ``params``: trainable tensors by state-dict key (fp32 masters or bf16);
    every other weight is read from the engine's frozen bf16 copies."""

    def __init__(self, engine, params: dict, dropout: bool = False, seed: int = 0):
        """dropout=True applies the configuration's dropout probabilities at every site the reference applies them in `train()` mode
        (Q-Former hidden / attention-probability dropout, OPT residual-branch dropout, T5 dropout_rate sites) with masks derived from
        ``seed`` (pass a different seed every step); dropout=False is the deterministic function the goldens pin."""
        self.dropout = bool(dropout)
        self.seed = int(seed)
        self._site = 0
        self.eng = engine
        self.params = dict(params)
        ok = ("qformer.", "query_tokens", "language_projection.")
        bad = [k for k in self.params if not k.startswith(ok)]
        if bad:
            raise NotImplementedError(f"only the Q-Former, query tokens and language projection train on this path (train_v2 freezes the rest): {bad[:3]}")

    def W(self, key):
        p = self.params.get(key)
        return p if p is not None else self.eng._keep[key]

    # ---- dropout sites: every call draws the next mask seed of the step (same order forward after forward) ----
    def _seed(self):
        self._site += 1
        return (self.seed * 2654435761 + self._site * 40503) & 0xFFFFFFFF

    def _lin_res(self, x, w, b, resid, p):
        """dense -> dropout -> + residual (hf *SelfOutput / *Output / OPTDecoderLayer / T5Layer*); fused into the GEMM without dropout"""
        if self.dropout and p > 0.0:
            return ag.dropout_add(ag.linear(x, w, b), resid, p, self._seed())
        return ag.linear(x, w, b, residual=resid)

    def _drop(self, x, p):
        return ag.dropout_add(x, None, p, self._seed()) if self.dropout and p > 0.0 else x

    def _adrop(self, p):
        return (p, self._seed()) if self.dropout and p > 0.0 else None

    # ---- Q-Former (hf modeling_blip_2.py Blip2QFormerModel: post-LN BERT layers, cross-attention every q_cross_freq) ----
    def qformer(self, image_embeds: torch.Tensor) -> torch.Tensor:
        d = self.eng.dims
        N, kv, Dv = image_embeds.shape
        D, H, nq = d.q_hidden, d.q_heads, d.num_query
        scale = (D // H) ** -0.5
        img = image_embeds.reshape(N * kv, Dv)
        qt = self.W("query_tokens").reshape(nq, D)
        q0 = ag.layer_norm(qt.to(torch.bfloat16), self.W("qformer.layernorm.weight"), self.W("qformer.layernorm.bias"), d.q_eps)
        qc = self.eng.config.qformer_config
        ph, pa = float(qc.hidden_dropout_prob), float(qc.attention_probs_dropout_prob)
        h = self._drop(q0.unsqueeze(0).expand(N, nq, D).reshape(N * nq, D), ph)  # hf :913 dropout(layernorm(query_embeds))
        for l in range(d.q_layers):
            cross = l % d.q_cross_freq == 0
            k = abi.qf_layer_keys(l, cross)
            w = lambda f: self.W(k[f])
            q = ag.linear(h, w("sq_w"), w("sq_b")).view(N, nq, D)
            kk = ag.linear(h, w("sk_w"), w("sk_b")).view(N, nq, D)
            v = ag.linear(h, w("sv_w"), w("sv_b")).view(N, nq, D)
            ctx = ag.attention(q, kk, v, H, scale, drop=self._adrop(pa)).view(N * nq, D)
            h = ag.layer_norm(self._lin_res(ctx, w("so_w"), w("so_b"), h, ph), w("sln_w"), w("sln_b"), d.q_eps)
            if cross:
                q = ag.linear(h, w("cq_w"), w("cq_b")).view(N, nq, D)
                kk = ag.linear(img, w("ck_w"), w("ck_b")).view(N, kv, D)
                v = ag.linear(img, w("cv_w"), w("cv_b")).view(N, kv, D)
                ctx = ag.attention(q, kk, v, H, scale, drop=self._adrop(pa)).view(N * nq, D)
                h = ag.layer_norm(self._lin_res(ctx, w("co_w"), w("co_b"), h, ph), w("cln_w"), w("cln_b"), d.q_eps)
            f = ag.gelu(ag.linear(h, w("fi_w"), w("fi_b")))
            h = ag.layer_norm(self._lin_res(f, w("fo_w"), w("fo_b"), h, ph), w("fln_w"), w("fln_b"), d.q_eps)
        return h  # (N * nq, D)

    def _opt_qkv(self, l, k):
        """q|k|v projection of frozen OPT layer l as one [3 D, D] matrix (built once per engine)."""
        cache = self.eng.__dict__.setdefault("_train_fused_qkv", {})
        hit = cache.get(l)
        if hit is None:
            ws = [self.W(k[f]) for f in ("q_w", "k_w", "v_w")]
            bs = [self.W(k[f]) for f in ("q_b", "k_b", "v_b")]
            if any(t.requires_grad for t in ws + bs):
                raise NotImplementedError("the language model is frozen on the train_v2 path")
            hit = cache[l] = (torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous())
        return hit

    # ---- OPT decoder with frozen weights (hf modeling_opt.py OPTDecoder: pre-LN, ReLU FFN, learned positions + 2) ----
    def opt_hidden(self, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        d = self.eng.dims
        B, L, D = inputs_embeds.shape
        H = d.t_heads
        scale = (D // H) ** -0.5
        am = attention_mask.to(inputs_embeds.device, torch.int64)
        pos = (torch.cumsum(am, 1) * am - 1 + 2).clamp_(min=0)  # hf OPTLearnedPositionalEmbedding (offset 2)
        pe = self.W("language_model.model.decoder.embed_positions.weight")
        h = (inputs_embeds + pe[pos]).reshape(B * L, D)
        km = am.to(torch.int32)
        tc = self.eng.config.text_config
        po = float(getattr(tc, "dropout", 0.0))
        if self.dropout and float(getattr(tc, "attention_dropout", 0.0)) > 0.0:
            raise NotImplementedError("OPT attention_dropout > 0 (the released configurations use 0.0)")
        for l in range(d.t_layers):
            k = abi.opt_layer_keys(l)
            w = lambda f: self.W(k[f])
            x = ag.layer_norm(h, w("ln1_w"), w("ln1_b"), d.t_eps)
            wqkv, bqkv = self._opt_qkv(l, k)
            qkv = ag.linear(x, wqkv, bqkv).view(B, L, 3 * D)  # one GEMM of N = 3 D instead of three of N = D (160 tiles each at L = 960)
            ctx = ag.attention_packed(qkv, H, scale, causal=True, key_mask=km).view(B * L, D)
            h = self._lin_res(ctx, w("o_w"), w("o_b"), h, po)
            x = ag.layer_norm(h, w("ln2_w"), w("ln2_b"), d.t_eps)
            f = ag.linear_relu(x, w("fc1_w"), w("fc1_b"))  # frozen layer: ReLU in the GEMM epilogue, the output is the saved activation
            h = self._lin_res(f, w("fc2_w"), w("fc2_b"), h, po)
        return ag.layer_norm(h, self.W("language_model.model.decoder.final_layer_norm.weight"),
                             self.W("language_model.model.decoder.final_layer_norm.bias"), d.t_eps).view(B, L, D)

    # ---- flan-t5 with frozen weights (hf modeling_t5.py: T5Stack encoder + decoder, RMS norms, relative position bias, ----
    # ---- gated-GELU feed-forward, no 1/sqrt(d) scaling; ref:eilev/model/v2.py:228-238) ------------------------------------
    def _frozen(self, tag, build):
        cache = self.eng.__dict__.setdefault("_train_fused_t5", {})
        hit = cache.get(tag)
        if hit is None:
            hit = cache[tag] = build()
        return hit

    def _t5_cat(self, tag, keys):
        def build():
            ws = [self.W(k) for k in keys]
            if any(t.requires_grad for t in ws):
                raise NotImplementedError("the language model is frozen on the train_v2 path")
            return torch.cat(ws, 0).contiguous()
        return self._frozen(tag, build)

    def _t5_rel(self, stack: str, L: int):
        """f32 (heads, 2 L - 1) bias over key - query in [-(L-1), L-1] (T5Attention._relative_position_bucket :217-262, float32
        arithmetic in the same order as the torch ops) and the offset L - 1."""
        d = self.eng.t5dims
        bidirectional = stack == "encoder"

        def build():
            w = self.W(f"language_model.{stack}.block.0.layer.0.SelfAttention.relative_attention_bias.weight")  # (buckets, heads)
            rp = torch.arange(-(L - 1), L, device=w.device)  # memory position - context position
            nb = d.rel_buckets
            ret = torch.zeros_like(rp)
            if bidirectional:
                nb //= 2
                ret = ret + (rp > 0).to(torch.long) * nb
                rp = rp.abs()
            else:
                rp = -torch.min(rp, torch.zeros_like(rp))
            max_exact = nb // 2
            large = max_exact + (torch.log(rp.float() / max_exact) / math.log(d.rel_max_dist / max_exact) * (nb - max_exact)).to(torch.long)
            large = torch.min(large, torch.full_like(large, nb - 1))
            bucket = ret + torch.where(rp < max_exact, rp, large)
            return w.float()[bucket].t().contiguous()
        return self._frozen(("rel", stack, L), build), L - 1

    def _t5_ff(self, h, k, tag):
        d = self.eng.t5dims
        x = ag.rms_norm(h, self.W(k["ln_ff"]), d.eps)
        pt = float(self.eng.config.text_config.dropout_rate)
        ab = ag.linear(x, self._t5_cat(tag + ("wi",), [k["wi0_w"], k["wi1_w"]]))
        return self._lin_res(self._drop(ag.gated_gelu(ab), pt), self.W(k["wo_w"]), None, h, pt)  # hf :118 dropout before wo, :140 after

    def t5_loss(self, emb: torch.Tensor, enc_mask: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        d = self.eng.t5dims
        B, Le, D = emb.shape
        H, I = d.heads, d.heads * d.d_kv
        km = enc_mask.to(emb.device, torch.int32).contiguous()
        pt = float(self.eng.config.text_config.dropout_rate)
        h = self._drop(emb.reshape(B * Le, D), pt)  # T5Stack :1012 dropout(inputs_embeds)
        rel = self._t5_rel("encoder", Le)
        for l in range(d.enc_layers):
            k = abi.t5_layer_keys("encoder", l)
            x = ag.rms_norm(h, self.W(k["ln_sa"]), d.eps)
            qkv = ag.linear(x, self._t5_cat(("enc", l, "qkv"), [k["q_w"], k["k_w"], k["v_w"]])).view(B, Le, 3 * I)
            ctx = ag.attention_packed(qkv, H, 1.0, causal=False, key_mask=km, rel=rel, drop=self._adrop(pt)).view(B * Le, I)
            h = self._lin_res(ctx, self.W(k["o_w"]), None, h, pt)
            h = self._t5_ff(h, k, ("enc", l))
        enc = self._drop(ag.rms_norm(h, self.W("language_model.encoder.final_layer_norm.weight"), d.eps), pt)  # (B * Le, D)

        t = self.eng.config.text_config
        lab = labels.to(emb.device)
        Lt = lab.shape[1]
        start = t.decoder_start_token_id if t.decoder_start_token_id is not None else t.pad_token_id
        dec_ids = torch.cat((torch.full_like(lab[:, :1], start), lab[:, :-1]), dim=1)  # T5._shift_right
        dec_ids = dec_ids.masked_fill(dec_ids == -100, t.pad_token_id)
        g = self._drop(self.W("language_model.shared.weight")[dec_ids].reshape(B * Lt, D), pt)  # frozen embedding rows
        rel = self._t5_rel("decoder", Lt)
        for l in range(d.dec_layers):
            k = abi.t5_layer_keys("decoder", l)
            x = ag.rms_norm(g, self.W(k["ln_sa"]), d.eps)
            qkv = ag.linear(x, self._t5_cat(("dec", l, "qkv"), [k["q_w"], k["k_w"], k["v_w"]])).view(B, Lt, 3 * I)
            ctx = ag.attention_packed(qkv, H, 1.0, causal=True, rel=rel, drop=self._adrop(pt)).view(B * Lt, I)
            g = self._lin_res(ctx, self.W(k["o_w"]), None, g, pt)
            x = ag.rms_norm(g, self.W(k["ln_ca"]), d.eps)
            q = ag.linear(x, self.W(k["cq_w"])).view(B, Lt, I)
            kv = ag.linear(enc, self._t5_cat(("dec", l, "ckv"), [k["ck_w"], k["cv_w"]])).view(B, Le, 2, I)
            ctx = ag.attention(q, kv[:, :, 0], kv[:, :, 1], H, 1.0, causal=False, key_mask=km, drop=self._adrop(pt)).view(B * Lt, I)
            g = self._lin_res(ctx, self.W(k["co_w"]), None, g, pt)
            g = self._t5_ff(g, k, ("dec", l))
        out = self._drop(ag.rms_norm(g, self.W("language_model.decoder.final_layer_norm.weight"), d.eps), pt)
        if d.scale_decoder_outputs:  # tied embeddings: hf :1037-1041
            out = out * (D ** -0.5)
        sel = lab.reshape(-1) >= 0
        head = self.params.get("language_model.lm_head.weight")
        if head is None:
            head = self.eng._keep.get("language_model.lm_head.weight", self.eng._keep["language_model.shared.weight"])
        self._last = (out.detach().view(B, Lt, D), head.detach())
        return ag.lm_head_ce(out[sel].contiguous(), head, lab.reshape(-1)[sel])

    @torch.no_grad()
    def logits(self) -> torch.Tensor:
        """Logits (B, L, vocab) bf16 of the batch `loss` just ran — `lm_head(hidden states)` of the SAME forward (dropout included in
        train() mode), outside the autograd graph: the reference's output object always carries them (ref:eilev/model/v2.py:239-252)."""
        hid, head = self._last
        B, L, D = hid.shape
        return ag.linear(hid.reshape(B * L, D).contiguous(), head).view(B, L, -1)

    def loss(self, input_ids, attention_mask, pixel_values, video_input_mask, labels) -> torch.Tensor:
        """Token cross-entropy (ignore_index -100; shifted for the decoder-only LM), differentiable w.r.t. ``params``."""
        eng = self.eng
        self._site = 0
        ag.new_step(eng.__dict__.setdefault("_train_frozen_t", {}))  # transposed frozen weights live and die with the engine
        dev = eng.device
        input_ids = input_ids.to(dev)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if pixel_values is not None:
            if video_input_mask is None:
                raise ValueError("video_input_mask is required with pixel_values")
            with torch.no_grad():
                img = eng.vit(pixel_values)  # frozen: inference kernels, nothing saved
            feats = ag.linear(self.qformer(img), self.W("language_projection.weight"), self.W("language_projection.bias"))
            emb = _EmbedScatter.apply(feats, eng, input_ids, video_input_mask)
        else:
            emb = eng.embed_scatter(input_ids, None, None)
        if eng.is_t5:
            return self.t5_loss(emb, attention_mask, labels)
        hid = self.opt_hidden(emb, attention_mask)
        self._last = (hid.detach(), self.W("language_model.model.decoder.embed_tokens.weight").detach())
        tgt = labels.to(dev)[:, 1:]
        sel = tgt >= 0  # position t predicts labels[t + 1]
        rows = hid[:, :-1][sel]
        return ag.lm_head_ce(rows.contiguous(), self.W("language_model.model.decoder.embed_tokens.weight"), tgt[sel])


def allreduce_gradients(params, group=None, bucket_bytes: int = 64 << 20) -> None:
    """Average ``.grad`` of the trainable parameters over the data-parallel ranks (what DDP / `accelerator.backward`
    does for ref:scripts/general/train_v2.py under torchrun): gradients are packed into flat fp32 buckets and each bucket
    is ONE all-reduce (RCCL over xGMI on the GPU box: 107 M parameters = 428 MB -> 7 buckets of 64 MiB, large enough to
    run at link bandwidth on the ring, small enough to start while later buckets are still being packed).
    ``params``: iterable of tensors or a dict; parameters without a gradient contribute zeros (a rank whose batch has no
    video still takes part in every collective)."""
    import torch.distributed as dist

    ps = [p for p in (params.values() if isinstance(params, dict) else params) if p.requires_grad]
    if not ps or not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    bucket, size = [], 0
    buckets = []
    for p in ps:
        nb = p.numel() * 4
        if bucket and size + nb > bucket_bytes:
            buckets.append(bucket)
            bucket, size = [], 0
        bucket.append(p)
        size += nb
    if bucket:
        buckets.append(bucket)
    for bk in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in bk])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for p in bk:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
