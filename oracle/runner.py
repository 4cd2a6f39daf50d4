"""numpy driver of the CPU oracle (oracle/libeilev_ref.so).

*** TEST INFRASTRUCTURE, NOT PRODUCT. ***  Imported only by tests/, __graft_entry__.smoke()
and the cpu_baseline leg of bench.py.  Nothing under eilev_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from eilev_amd import abi
from eilev_amd.statedict import state_dict_shapes
from eilev_amd.synth import synth_param

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libeilev_ref.so")


def build_oracle(force: bool = False) -> str:
    src = os.path.join(_HERE, "eilev_ref.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libeilev_ref.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_oracle()
        _lib = abi.load_library(LIB_PATH)
        assert _lib.eilev_backend() == b"cpu-oracle"
    return _lib


def synth_state_dict(config, mode: str = "fanin", seed: int = 0):
    """Deterministic fp32 numpy state dict (bf16-exact values), identical to tools/make_goldens.py."""
    return {k: synth_param(k, shp, mode, seed) for k, shp in state_dict_shapes(config).items()}


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleModel:
    """Runs the reference arithmetic on CPU through the oracle's C ABI."""

    def __init__(self, config, weights: dict, emulate_bf16: bool = False):
        self.config = config
        self.w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
        self.dims = abi.dims_from_config(config, emulate_bf16)
        self.is_t5 = getattr(config.text_config, "model_type", "opt") == "t5"
        self.t5dims = abi.t5_dims_from_config(config, emulate_bf16) if self.is_t5 else None
        self.pack = abi.WeightPack(self.dims, lambda k: self.w[k].ctypes.data, self.t5dims)
        self.lib = lib()
        self.tokens_per_frame = (self.dims.image_size // self.dims.patch_size) ** 2 + 1

    # ---- stages -------------------------------------------------------------------------
    def vit(self, pixels: np.ndarray, want_pooler: bool = False):
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        N, _, T = px.shape[:3]
        d = self.dims
        out = np.empty((N, T * self.tokens_per_frame, d.v_hidden), np.float32)
        pool = np.empty((N, T, d.v_hidden), np.float32) if want_pooler else None
        nbytes = self.lib.eilev_vit_workspace_bytes(C.byref(d), N, T)
        ws = np.empty(nbytes // 4 + 1, np.float32)
        abi.check(self.lib.eilev_vit_forward(C.byref(d), C.byref(self.pack.vit), _p(px), abi_f32(), N, T, _p(out),
                                             _p(pool), _p(ws), nbytes, None), "oracle vit")
        return (out, pool) if want_pooler else out

    def vit_debug(self, pixels: np.ndarray):
        """(last, pooler, hidden_states (L+1, N, T*tok, D), attentions (L, N, T, H, tok, tok)) — ref:eilev/model/v2.py:76-103."""
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        N, _, T = px.shape[:3]
        d = self.dims
        tok = self.tokens_per_frame
        out = np.empty((N, T * tok, d.v_hidden), np.float32)
        pool = np.empty((N, T, d.v_hidden), np.float32)
        hid = np.empty((d.v_layers + 1, N, T * tok, d.v_hidden), np.float32)
        att = np.empty((d.v_layers, N, T, d.v_heads, tok, tok), np.float32)
        nbytes = self.lib.eilev_vit_workspace_bytes(C.byref(d), N, T)
        ws = np.empty(nbytes // 4 + 1, np.float32)
        abi.check(self.lib.eilev_vit_forward_debug(C.byref(d), C.byref(self.pack.vit), _p(px), abi_f32(), N, T, _p(out), _p(pool), _p(hid),
                                                   _p(att), _p(ws), nbytes, None), "oracle vit debug")
        return out, pool, hid, att

    def qformer(self, image_embeds: np.ndarray):
        img = np.ascontiguousarray(image_embeds, dtype=np.float32)
        N, kv = img.shape[:2]
        d = self.dims
        out = np.empty((N, d.num_query, d.q_hidden), np.float32)
        nbytes = self.lib.eilev_qformer_workspace_bytes(C.byref(d), N, kv)
        ws = np.empty(nbytes // 4 + 1, np.float32)
        abi.check(self.lib.eilev_qformer_forward(C.byref(d), C.byref(self.pack.qf), _p(img), N, kv, _p(out), _p(ws),
                                                 nbytes, None), "oracle qformer")
        return out

    # ---- attention weights (`output_attentions=True` inside the full forward): the same composition as eilev_amd/engine.py, on the oracle's entries ----
    def _lin(self, x2d, wname, bname=None, resid=None):
        x = np.ascontiguousarray(x2d, np.float32)
        w, b = self.w[wname], (self.w[bname] if bname else None)
        out = np.empty((x.shape[0], w.shape[0]), np.float32)
        abi.check(self.lib.eilev_linear(_p(x), _p(w), _p(b), _p(resid), _p(out), x.shape[0], w.shape[0], x.shape[1], 0, 1, None), "oracle linear")
        return out

    def _ln(self, x2d, wname, bname, eps):
        x = np.ascontiguousarray(x2d, np.float32)
        out = np.empty_like(x)
        abi.check(self.lib.eilev_layernorm(_p(x), _p(self.w[wname]), _p(self.w[bname]), _p(out), x.shape[0], x.shape[1], C.c_float(eps), None), "oracle layernorm")
        return out

    def _probs(self, q, k, B, H, sq, skv, hd, scale, causal=False, key_mask=None, rel=None):
        out = np.empty((B, H, sq, skv), np.float32)
        km = None if key_mask is None else np.ascontiguousarray(key_mask, np.int32)
        tab, off = rel if rel is not None else (None, 0)
        abi.check(self.lib.eilev_attention_probs(_p(q), _p(k), _p(out), B, H, sq, skv, hd, q.shape[-1], k.shape[-1], C.c_float(scale), int(causal), _p(km),
                                                 None if tab is None else _p(tab), 0 if tab is None else tab.shape[1], off, 0 if tab is None else tab.shape[1],
                                                 None), "oracle attention_probs")
        return out

    def t5_rel_table(self, stack, L):
        """(heads, 2 L - 1) f32 bias over key - query and the offset L - 1: hf T5Attention._relative_position_bucket (modeling_t5.py) in numpy
        float32 (half the buckets per sign when bidirectional; exact below max_exact, logarithmic up to max_distance)."""
        d = self.t5dims
        w = self.w[f"language_model.{stack}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]  # (buckets, heads)
        rp = np.arange(-(L - 1), L, dtype=np.int64)
        nb = int(d.rel_buckets)
        ret = np.zeros_like(rp)
        if stack == "encoder":
            nb //= 2
            ret = ret + (rp > 0).astype(np.int64) * nb
            rp = np.abs(rp)
        else:
            rp = -np.minimum(rp, 0)
        max_exact = nb // 2
        with np.errstate(divide="ignore"):
            lg = np.log(rp.astype(np.float32) / np.float32(max_exact)) / np.float32(np.log(d.rel_max_dist / max_exact)) * np.float32(nb - max_exact)
        large = max_exact + np.where(np.isfinite(lg), lg, 0).astype(np.int64)
        large = np.minimum(large, nb - 1)
        bucket = ret + np.where(rp < max_exact, rp, large)
        return np.ascontiguousarray(w.astype(np.float32)[bucket].T), L - 1

    def _rms(self, x2d, wname, eps):
        x = np.ascontiguousarray(x2d, np.float32)
        out = np.empty_like(x)
        abi.check(self.lib.eilev_rmsnorm(_p(x), _p(self.w[wname]), _p(out), x.shape[0], x.shape[1], C.c_float(eps), None), "oracle rmsnorm")
        return out

    def t5_attentions(self, enc_hs, dec_hs, attn_mask, dec_mask=None):
        """hf T5Attention `attn_weights` of every block of both stacks from the block inputs (t5_forward_debug): encoder self (layers, B, H, L, L),
        decoder self (layers, B, H, T, T), cross (layers, B, H, T, L)."""
        d = self.t5dims
        H, hd = d.heads, d.d_kv
        I = H * hd
        B, L, D = enc_hs.shape[1:]
        T = dec_hs.shape[2]
        enc_rel, dec_rel = self.t5_rel_table("encoder", L), self.t5_rel_table("decoder", T)
        dm = None if dec_mask is None else np.ascontiguousarray(dec_mask, np.int32)
        enc_a, dec_a, cross_a = [], [], []
        for l in range(d.enc_layers):
            k_ = abi.t5_layer_keys("encoder", l)
            x = self._rms(enc_hs[l].reshape(B * L, D), k_["ln_sa"], d.eps)
            enc_a.append(self._probs(self._lin(x, k_["q_w"]), self._lin(x, k_["k_w"]), B, H, L, L, hd, 1.0, key_mask=attn_mask, rel=enc_rel))
        enc_out = np.ascontiguousarray(enc_hs[-1].reshape(B * L, D))
        for l in range(d.dec_layers):
            k_ = abi.t5_layer_keys("decoder", l)
            h = np.ascontiguousarray(dec_hs[l].reshape(B * T, D), np.float32)
            x = self._rms(h, k_["ln_sa"], d.eps)
            q, k, v = self._lin(x, k_["q_w"]), self._lin(x, k_["k_w"]), self._lin(x, k_["v_w"])
            dec_a.append(self._probs(q, k, B, H, T, T, hd, 1.0, causal=True, key_mask=dm, rel=dec_rel))
            ctx = np.empty_like(q)
            abi.check(self.lib.eilev_attention_rel(_p(q), _p(k), _p(v), _p(ctx), B, H, T, T, hd, I, I, I, C.c_float(1.0), 1, None if dm is None else _p(dm),
                                                   _p(dec_rel[0]), dec_rel[0].shape[1], dec_rel[1], dec_rel[0].shape[1], None), "oracle attention_rel")
            x2 = self._rms(self._lin(ctx, k_["o_w"], resid=h), k_["ln_ca"], d.eps)
            cross_a.append(self._probs(self._lin(x2, k_["cq_w"]), self._lin(enc_out, k_["ck_w"]), B, H, T, L, hd, 1.0, key_mask=attn_mask))
        return np.stack(enc_a), np.stack(dec_a), np.stack(cross_a)

    def lm_attentions(self, hidden_states, attn_mask):
        """hf OPT eager `attn_weights` of every block from the block inputs (prefill(hidden_states=True)): (layers, B, heads, L, L)."""
        d = self.dims
        Lyr, B, L, D = hidden_states.shape[0] - 1, *hidden_states.shape[1:]
        H, hd = d.t_heads, d.t_hidden // d.t_heads
        out = []
        for l in range(Lyr):
            p = abi.OPT_PREFIX.format(l)
            x = self._ln(hidden_states[l].reshape(B * L, D), p + "self_attn_layer_norm.weight", p + "self_attn_layer_norm.bias", d.t_eps)
            q = self._lin(x, p + "self_attn.q_proj.weight", p + "self_attn.q_proj.bias")
            k = self._lin(x, p + "self_attn.k_proj.weight", p + "self_attn.k_proj.bias")
            out.append(self._probs(q, k, B, H, L, L, hd, hd ** -0.5, causal=True, key_mask=attn_mask))
        return np.stack(out)

    def qformer_hidden_states(self, image_embeds):
        """embedding output + every block's output: the stack run with its first i blocks (the C ABI has no per-block export)."""
        d = self.dims
        img = np.ascontiguousarray(image_embeds, np.float32)
        N, kv = img.shape[:2]
        qt = np.ascontiguousarray(self.w["query_tokens"].reshape(d.num_query, d.q_hidden))
        outs = [np.broadcast_to(self._ln(qt, "qformer.layernorm.weight", "qformer.layernorm.bias", d.q_eps)[None], (N, d.num_query, d.q_hidden)).copy()]
        nbytes = self.lib.eilev_qformer_workspace_bytes(C.byref(d), N, kv)
        ws = np.empty(nbytes // 4 + 1, np.float32)
        for i in range(1, d.q_layers + 1):
            di = type(d).from_buffer_copy(d)
            di.q_layers = i
            out = np.empty((N, d.num_query, d.q_hidden), np.float32)
            abi.check(self.lib.eilev_qformer_forward(C.byref(di), C.byref(self.pack.qf), _p(img), N, kv, _p(out), _p(ws), nbytes, None), "oracle qformer")
            outs.append(out)
        return outs

    def qformer_attentions(self, image_embeds, hidden_states):
        """(self-attention weights per block, cross-attention weights of the blocks that have one) — hf Blip2QFormerLayer."""
        d = self.dims
        img = np.ascontiguousarray(image_embeds, np.float32)
        N, kv, Dv = img.shape
        nq, Dq, H = d.num_query, d.q_hidden, d.q_heads
        hd = Dq // H
        selfs, crosses = [], []
        for i in range(d.q_layers):
            p = f"qformer.encoder.layer.{i}."
            h = np.ascontiguousarray(hidden_states[i].reshape(N * nq, Dq), np.float32)
            q = self._lin(h, p + "attention.attention.query.weight", p + "attention.attention.query.bias")
            k = self._lin(h, p + "attention.attention.key.weight", p + "attention.attention.key.bias")
            selfs.append(self._probs(q, k, N, H, nq, nq, hd, hd ** -0.5))
            if i % d.q_cross_freq == 0:
                v = self._lin(h, p + "attention.attention.value.weight", p + "attention.attention.value.bias")
                ctx = np.empty_like(q)
                abi.check(self.lib.eilev_attention(_p(q), _p(k), _p(v), _p(ctx), N, H, nq, nq, hd, Dq, Dq, Dq, C.c_float(hd ** -0.5), 0, None, None), "oracle attention")
                ao = self._ln(self._lin(ctx, p + "attention.output.dense.weight", p + "attention.output.dense.bias", resid=h),
                              p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias", d.q_eps)
                qc = self._lin(ao, p + "crossattention.attention.query.weight", p + "crossattention.attention.query.bias")
                kc = self._lin(img.reshape(N * kv, Dv), p + "crossattention.attention.key.weight", p + "crossattention.attention.key.bias")
                crosses.append(self._probs(qc, kc, N, H, nq, kv, hd, hd ** -0.5))
        return selfs, crosses

    def project(self, query_out: np.ndarray):
        q = np.ascontiguousarray(query_out, dtype=np.float32).reshape(-1, self.dims.q_hidden)
        out = np.empty((q.shape[0], self.dims.t_hidden), np.float32)
        abi.check(self.lib.eilev_project_rows(C.byref(self.dims), self.pack.proj_w, self.pack.proj_b, _p(q), q.shape[0],
                                              _p(out), None), "oracle project")
        return out

    def embed_scatter(self, input_ids, video_mask, video_feats):
        ids = np.ascontiguousarray(input_ids, dtype=np.int64)
        B, L = ids.shape
        vm = None if video_mask is None else np.ascontiguousarray(video_mask != 0, dtype=np.uint8)
        vf = None if video_feats is None else np.ascontiguousarray(video_feats, dtype=np.float32)
        out = np.empty((B, L, self.dims.t_hidden), np.float32)
        abi.check(self.lib.eilev_embed_scatter(C.byref(self.dims), self.pack.embed_tokens, _p(ids), _p(vm), _p(vf),
                                               0 if vf is None else vf.shape[0], B, L, _p(out), None), "oracle embed_scatter")
        return out

    def encode(self, pixels, input_ids, video_mask):
        img = self.vit(pixels)
        feats = self.project(self.qformer(img))
        return self.embed_scatter(input_ids, video_mask, feats)

    def prefill(self, inputs_embeds, attn_mask, kv_capacity=None, all_logits=True, hidden_states=False):
        x = np.ascontiguousarray(inputs_embeds, dtype=np.float32)
        B, L, _ = x.shape
        d = self.dims
        cap = int(kv_capacity or L)
        am = np.ascontiguousarray(attn_mask, dtype=np.int32)
        kv = np.zeros(self.lib.eilev_opt_kv_cache_bytes(C.byref(d), B, cap) // 4, np.float32)
        last = np.empty((B, d.vocab), np.float32)
        alll = np.empty((B, L, d.vocab), np.float32) if all_logits else None
        nbytes = self.lib.eilev_opt_workspace_bytes(C.byref(d), B, L)
        ws = np.empty(nbytes // 4 + 1, np.float32)
        if hidden_states:  # hf output_hidden_states: every block's input, then the output of final_layer_norm
            hs = np.empty((d.t_layers + 1, B, L, d.t_hidden), np.float32)
            abi.check(self.lib.eilev_opt_prefill_debug(C.byref(d), C.byref(self.pack.opt), _p(x), _p(am), B, L, _p(kv), cap, _p(last),
                                                       _p(alll), _p(hs), _p(ws), nbytes, None), "oracle prefill_debug")
            return last, alll, kv, hs
        abi.check(self.lib.eilev_opt_prefill(C.byref(d), C.byref(self.pack.opt), _p(x), _p(am), B, L, _p(kv), cap, _p(last),
                                             _p(alll), _p(ws), nbytes, None), "oracle prefill")
        return last, alll, kv

    def forward_logits(self, pixels, input_ids, attn_mask, video_mask):
        """= reference forward(...).logits (ref:eilev/model/v2.py:132-252)."""
        emb = self.encode(pixels, input_ids, video_mask)
        _, alll, _ = self.prefill(emb, attn_mask)
        return alll

    def generate(self, pixels, input_ids, attn_mask, video_mask, max_new_tokens, eos_id=-1, pad_id=1,
                 return_logits=False):
        """Greedy = reference generate(num_beams=1, do_sample=False) (ref:eilev/model/v2.py:254-324)."""
        d = self.dims
        emb = self.encode(pixels, input_ids, video_mask)
        B, L, _ = emb.shape
        cap = L + max_new_tokens
        am = np.ascontiguousarray(attn_mask, dtype=np.int32)
        last, _, kv = self.prefill(emb, am, kv_capacity=cap, all_logits=False)
        state = np.zeros(2, np.int32)
        state[1] = B
        finished = np.zeros(B, np.uint8)
        tokens = np.zeros(B, np.int64)
        out = np.full((B, max_new_tokens), pad_id, np.int64)
        step_logits = [last.copy()]
        abi.check(self.lib.eilev_greedy_select(_p(last), B, d.vocab, _p(state), _p(finished), eos_id, pad_id, _p(tokens),
                                               _p(out), max_new_tokens, None), "oracle select")
        n_valid = am.sum(axis=1).astype(np.int32)
        nbytes = self.lib.eilev_opt_workspace_bytes(C.byref(d), B, 1)
        ws = np.empty(nbytes // 4 + 1, np.float32)
        logits = np.empty((B, d.vocab), np.float32)
        while state[0] < max_new_tokens and state[1] > 0:
            abi.check(self.lib.eilev_opt_decode_step(C.byref(d), C.byref(self.pack.opt), _p(tokens), _p(state), _p(am),
                                                     _p(n_valid), B, L, _p(kv), cap, _p(logits), _p(finished), eos_id,
                                                     pad_id, _p(out), max_new_tokens, _p(ws), nbytes, None), "oracle decode")
            step_logits.append(logits.copy())
        ids = out[:, : int(state[0])]
        return (ids, step_logits) if return_logits else ids


    def classify(self, pixels, input_ids, attn_mask, video_mask, class_ids, class_mask=None, class_batch_size=None):
        """= reference classify(...) (ref:eilev/model/v2.py:326-501): (B, num_classes) mean class log-likelihoods."""
        d = self.dims
        emb = self.encode(pixels, input_ids, video_mask)
        B, L, _ = emb.shape
        cls = np.asarray(class_ids, dtype=np.int64)
        n_cls, Lc = cls.shape
        cmask = np.ones_like(cls) if class_mask is None else np.asarray(class_mask, dtype=np.int64)
        pm = np.ascontiguousarray(attn_mask, dtype=np.int32)
        cap = L + Lc
        last, _, kv = self.prefill(emb, pm, kv_capacity=cap, all_logits=False)
        planes = 2 * d.t_layers
        step = n_cls if class_batch_size is None else int(class_batch_size)
        cols = []
        for i in range(0, n_cls, step):
            ids, msk = cls[i:i + step], cmask[i:i + step]
            nc = ids.shape[0]
            R = B * nc
            rows_ids = np.ascontiguousarray(np.broadcast_to(ids[None], (B, nc, Lc)).reshape(R, Lc))
            rows_msk = np.broadcast_to(msk[None], (B, nc, Lc)).reshape(R, Lc)
            full = np.ascontiguousarray(np.concatenate((np.repeat(pm, nc, axis=0), rows_msk.astype(np.int32)), axis=1))
            kv_rows = np.ascontiguousarray(np.repeat(kv.reshape(planes, B, -1), nc, axis=1))
            x = self.embed_scatter(rows_ids, None, None)
            logits = np.empty((R, Lc, d.vocab), np.float32)
            nbytes = self.lib.eilev_opt_workspace_bytes(C.byref(d), R, cap)
            ws = np.empty(nbytes // 4 + 1, np.float32)
            abi.check(self.lib.eilev_opt_extend(C.byref(d), C.byref(self.pack.opt), _p(x), _p(full), R, Lc, L, _p(kv_rows), cap,
                                                _p(logits), _p(ws), nbytes, None), "oracle extend")
            shift = np.concatenate((np.repeat(last, nc, axis=0)[:, None], logits[:, :-1]), axis=1).astype(np.float64)
            lse = np.log(np.exp(shift - shift.max(-1, keepdims=True)).sum(-1)) + shift.max(-1)
            tok = np.take_along_axis(shift, rows_ids[..., None], axis=-1)[..., 0]
            ll = np.where(rows_msk != 0, tok - lse, 0.0).reshape(B, nc, Lc).sum(-1)
            cols.append(ll / msk.sum(-1)[None].astype(np.float64))
        return np.concatenate(cols, axis=1).astype(np.float32)

    # ---- encoder-decoder LM (flan-t5) ----------------------------------------------------------------------
    def t5_encode(self, inputs_embeds, attn_mask):
        d = self.t5dims
        x = np.ascontiguousarray(inputs_embeds, dtype=np.float32)
        B, L, _ = x.shape
        am = np.ascontiguousarray(attn_mask, dtype=np.int32)
        out = np.empty_like(x)
        nb = self.lib.eilev_t5_workspace_bytes(C.byref(d), B, L, L)
        ws = np.empty(nb // 4 + 1, np.float32)
        abi.check(self.lib.eilev_t5_encode(C.byref(d), C.byref(self.pack.t5), _p(x), _p(am), B, L, _p(out), _p(ws), nb, None), "oracle t5 encode")
        return out

    def t5_cross_kv(self, enc_out):
        d = self.t5dims
        B, L, _ = enc_out.shape
        kv = np.zeros(self.lib.eilev_t5_cross_kv_bytes(C.byref(d), B, L) // 4, np.float32)
        abi.check(self.lib.eilev_t5_cross_kv(C.byref(d), C.byref(self.pack.t5), _p(np.ascontiguousarray(enc_out, np.float32)), B, L, _p(kv), None, 0, None),
                  "oracle t5 cross kv")
        return kv

    def t5_decode(self, dec_ids, enc_mask, past_len, self_kv, cap, cross_kv, enc_len):
        d = self.t5dims
        ids = np.ascontiguousarray(dec_ids, dtype=np.int64)
        B, T = ids.shape
        am = np.ascontiguousarray(enc_mask, dtype=np.int32)
        logits = np.empty((B, T, d.vocab), np.float32)
        nb = self.lib.eilev_t5_workspace_bytes(C.byref(d), B, T, max(enc_len, past_len + T))
        ws = np.empty(nb // 4 + 1, np.float32)
        abi.check(self.lib.eilev_t5_decode(C.byref(d), C.byref(self.pack.t5), _p(ids), _p(am), B, T, past_len, _p(self_kv), cap, _p(cross_kv),
                                           enc_len, _p(logits), _p(ws), nb, None), "oracle t5 decode")
        return logits

    def t5_forward_debug(self, pixels, input_ids, attn_mask, video_mask, decoder_input_ids, decoder_attention_mask=None):
        """Teacher-forced forward with decoder_attention_mask and the per-block tensors of both stacks (ref:eilev/model/v2.py:228-238 with
        output_hidden_states=True): logits, encoder hidden_states (layers + 1, B, L, D), decoder hidden_states (layers + 1, B, T, D)."""
        d = self.t5dims
        emb = np.ascontiguousarray(self.encode(pixels, input_ids, video_mask), dtype=np.float32)
        B, L, D = emb.shape
        am = np.ascontiguousarray(attn_mask, dtype=np.int32)
        enc = np.empty_like(emb)
        enc_hs = np.empty((d.enc_layers + 1, B, L, D), np.float32)
        nb = self.lib.eilev_t5_workspace_bytes(C.byref(d), B, L, L)
        ws = np.empty(nb // 4 + 1, np.float32)
        abi.check(self.lib.eilev_t5_encode_debug(C.byref(d), C.byref(self.pack.t5), _p(emb), _p(am), B, L, _p(enc), _p(enc_hs), _p(ws), nb, None),
                  "oracle t5 encode debug")
        ckv = self.t5_cross_kv(enc)
        ids = np.ascontiguousarray(decoder_input_ids, dtype=np.int64)
        T = ids.shape[1]
        dm = None if decoder_attention_mask is None else np.ascontiguousarray(decoder_attention_mask, dtype=np.int32)
        skv = np.zeros(self.lib.eilev_t5_self_kv_bytes(C.byref(d), B, T) // 4, np.float32)
        logits = np.empty((B, T, d.vocab), np.float32)
        dec_hs = np.empty((d.dec_layers + 1, B, T, D), np.float32)
        nb = self.lib.eilev_t5_workspace_bytes(C.byref(d), B, T, max(L, T))
        ws = np.empty(nb // 4 + 1, np.float32)
        abi.check(self.lib.eilev_t5_decode_debug(C.byref(d), C.byref(self.pack.t5), _p(ids), _p(am), None if dm is None else _p(dm), B, T, 0, _p(skv), T,
                                                 _p(ckv), L, _p(logits), _p(dec_hs), _p(ws), nb, None), "oracle t5 decode debug")
        return logits, enc_hs, dec_hs

    def t5_forward_logits(self, pixels, input_ids, attn_mask, video_mask, decoder_input_ids):
        """= reference forward(..., decoder_input_ids / labels).logits for the encoder-decoder LM (ref:eilev/model/v2.py:228-238);
        also returns the encoder's last hidden state."""
        d = self.t5dims
        emb = self.encode(pixels, input_ids, video_mask)
        enc = self.t5_encode(emb, attn_mask)
        ckv = self.t5_cross_kv(enc)
        B, T = np.asarray(decoder_input_ids).shape
        skv = np.zeros(self.lib.eilev_t5_self_kv_bytes(C.byref(d), B, T) // 4, np.float32)
        return self.t5_decode(decoder_input_ids, attn_mask, 0, skv, T, ckv, enc.shape[1]), enc

    def t5_generate(self, pixels, input_ids, attn_mask, video_mask, max_new_tokens, eos_id=1, pad_id=0, start_id=0):
        """Greedy = reference generate(num_beams=1, do_sample=False) for T5: returns decoder ids INCLUDING the start token."""
        d = self.t5dims
        emb = self.encode(pixels, input_ids, video_mask)
        enc = self.t5_encode(emb, attn_mask)
        ckv = self.t5_cross_kv(enc)
        B, L = enc.shape[:2]
        cap = max_new_tokens + 1
        skv = np.zeros(self.lib.eilev_t5_self_kv_bytes(C.byref(d), B, cap) // 4, np.float32)
        cur = np.full((B, 1), start_id, np.int64)
        out = [cur.copy()]
        done = np.zeros(B, bool)
        for t in range(max_new_tokens):
            logits = self.t5_decode(cur, attn_mask, t, skv, cap, ckv, L)[:, 0]
            nxt = logits.argmax(-1).astype(np.int64)
            nxt = np.where(done, pad_id, nxt)
            out.append(nxt[:, None])
            done |= nxt == eos_id
            cur = nxt[:, None]
            if done.all():
                break
        return np.concatenate(out, axis=1)

    def t5_generate_beam(self, pixels, input_ids, attn_mask, video_mask, max_new_tokens, num_beams, length_penalty=1.0, eos_id=1,
                         pad_id=0, start_id=0, early_stopping=False):
        """Beam search for the encoder-decoder LM: eilev_amd.beam (the HF-equivalent selection rule) over the oracle decoder."""
        import torch

        from eilev_amd.beam import beam_search

        if pad_id == 0:  # hf generation/utils.py:3319: `pad_token_id or eos_token_id[0]` — pad id 0 is falsy
            pad_id = eos_id if eos_id >= 0 else -1
        d = self.t5dims
        emb = self.encode(pixels, input_ids, video_mask)
        enc = self.t5_encode(emb, attn_mask)
        B, L = enc.shape[:2]
        R = B * num_beams
        planes = 2 * d.dec_layers
        ckv = np.ascontiguousarray(np.repeat(self.t5_cross_kv(enc).reshape(planes, B, -1), num_beams, axis=1))
        am = np.repeat(np.ascontiguousarray(attn_mask, dtype=np.int32), num_beams, axis=0)
        cap = max_new_tokens + 1
        skv = np.zeros(self.lib.eilev_t5_self_kv_bytes(C.byref(d), R, cap) // 4, np.float32)
        start = np.full((R, 1), start_id, np.int64)
        first = self.t5_decode(start, am, 0, skv, cap, ckv, L)[:, 0]
        steps = [0]

        def step(next_tokens, beam_src):
            nonlocal skv
            src = beam_src.numpy()
            skv = np.ascontiguousarray(skv.reshape(planes, R, -1)[:, src]).reshape(-1)
            steps[0] += 1
            logits = self.t5_decode(next_tokens.numpy().reshape(R, 1), am, steps[0], skv, cap, ckv, L)[:, 0]
            return torch.from_numpy(logits)

        ids = beam_search(step, torch.from_numpy(first[::num_beams].copy()), B, num_beams, max_new_tokens, length_penalty, eos_id, pad_id,
                          early_stopping).numpy()
        return np.concatenate((np.full((ids.shape[0], 1), start_id, np.int64), ids), axis=1)

    def generate_beam(self, pixels, input_ids, attn_mask, video_mask, max_new_tokens, num_beams, length_penalty=1.0, eos_id=-1,
                      pad_id=1, early_stopping=False, no_move=False, trace=None):
        """Beam search = reference generate(num_beams=k) with the oracle as the language model (eilev_amd.beam drives it).
        no_move=True runs the steps through eilev_opt_decode_step_beam (the cache is never reordered: prompt cache + generation cache +
        ancestor table, include/eilev.h) instead of reordering the cache rows like hf does; trace: a list that receives every step's logits."""
        import torch

        from eilev_amd.beam import beam_search

        d = self.dims
        emb = self.encode(pixels, input_ids, video_mask)
        B, L, _ = emb.shape
        R, cap = B * num_beams, L + max_new_tokens
        am = np.ascontiguousarray(attn_mask, dtype=np.int32)
        if no_move:
            last, _, kv_prompt = self.prefill(emb, am, kv_capacity=L, all_logits=False)
            gen_cap = max(1, max_new_tokens)
            kv_gen = np.zeros(int(self.lib.eilev_opt_kv_cache_bytes(C.byref(d), R, gen_cap)) // 4 + 1, np.float32)
            anc = np.zeros((gen_cap, R), np.int32)
            n_valid = np.repeat(am.sum(axis=1), num_beams).astype(np.int32)
            state, tokens = np.zeros(2, np.int32), np.zeros(R, np.int64)
            logits = np.empty((R, d.vocab), np.float32)
            nbytes = self.lib.eilev_opt_workspace_bytes(C.byref(d), R, 1)
            ws = np.empty(nbytes // 4 + 1, np.float32)
            steps = [0]

            def step_nm(next_tokens, beam_src):
                t = steps[0]
                if t > 0:
                    anc[:t] = anc[:t][:, beam_src.numpy()]
                anc[t] = np.arange(R, dtype=np.int32)
                steps[0] = t + 1
                state[0] = t + 1
                tokens[:] = next_tokens.numpy()
                abi.check(self.lib.eilev_opt_decode_step_beam(C.byref(d), C.byref(self.pack.opt), _p(tokens), _p(state), _p(am), _p(n_valid), R,
                                                              num_beams, L, _p(kv_prompt), _p(kv_gen), gen_cap, _p(anc), _p(logits), _p(ws), nbytes,
                                                              None), "oracle beam decode")
                assert state[0] == t + 2
                if trace is not None:
                    trace.append(logits.copy())
                return torch.from_numpy(logits.copy())

            return beam_search(step_nm, torch.from_numpy(last.copy()), B, num_beams, max_new_tokens, length_penalty, eos_id, pad_id,
                               early_stopping).numpy()
        last, _, kv_small = self.prefill(emb, am, kv_capacity=cap, all_logits=False)
        planes = 2 * d.t_layers
        kv = [np.repeat(kv_small.reshape(planes, B, -1), num_beams, axis=1).copy()]
        am_r = np.repeat(am, num_beams, axis=0).copy()
        n_valid = am_r.sum(axis=1).astype(np.int32)
        state = np.zeros(2, np.int32)
        finished = np.zeros(R, np.uint8)
        tokens = np.zeros(R, np.int64)
        out = np.zeros((R, max_new_tokens), np.int64)
        logits = np.empty((R, d.vocab), np.float32)
        nbytes = self.lib.eilev_opt_workspace_bytes(C.byref(d), R, 1)
        ws = np.empty(nbytes // 4 + 1, np.float32)
        steps = [0]

        def step(next_tokens, beam_src):
            kv[0] = np.ascontiguousarray(kv[0][:, beam_src.numpy()])
            steps[0] += 1
            state[0] = steps[0]
            tokens[:] = next_tokens.numpy()
            abi.check(self.lib.eilev_opt_decode_step(C.byref(d), C.byref(self.pack.opt), _p(tokens), _p(state), _p(am_r), _p(n_valid),
                                                     R, L, _p(kv[0]), cap, _p(logits), _p(finished), -1, pad_id, _p(out),
                                                     max_new_tokens, _p(ws), nbytes, None), "oracle decode")
            if trace is not None:
                trace.append(logits.copy())
            return torch.from_numpy(logits.copy())

        ids = beam_search(step, torch.from_numpy(last.copy()), B, num_beams, max_new_tokens, length_penalty, eos_id, pad_id, early_stopping)
        return ids.numpy()


def abi_f32():
    return 0


def shifted_ce_loss(logits: np.ndarray, labels: np.ndarray) -> float:
    """HF ForCausalLM loss: mean CE over shifted positions with label != -100."""
    lg = logits[:, :-1].astype(np.float64)
    lb = labels[:, 1:]
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    sel = lb != -100
    picked = np.take_along_axis(lg, np.where(sel, lb, 0)[..., None], -1)[..., 0]
    return float(((lse - picked) * sel).sum() / sel.sum())


def process_frames(video: np.ndarray, size: int = 224, lut: np.ndarray | None = None) -> np.ndarray:
    """Oracle for eilev_process_frames: uint8 (B, 3, T, H, W) -> fp32 (B, 3, T, size, size); coefficient tables from the
    oracle's own restatement of Pillow's precompute_coeffs (eilev_resample_coeffs)."""
    from eilev_amd.preprocess import normalize_lut

    L = lib()
    L.eilev_resample_coeffs.restype = C.c_int
    L.eilev_resample_coeffs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    video = np.ascontiguousarray(video, np.uint8)
    b, c, t, h, w = video.shape
    assert c == 3

    def table(n_in):
        if n_in == size:
            return None, None, 0
        ks = L.eilev_resample_coeffs(n_in, size, None, None)
        coef = np.zeros((size, ks), np.int32)
        bounds = np.zeros((size, 2), np.int32)
        L.eilev_resample_coeffs(n_in, size, coef.ctypes.data, bounds.ctypes.data)
        return coef, bounds, ks

    ch, bh, kh = table(w)
    cv, bv, kv = table(h)
    lut = np.ascontiguousarray(normalize_lut() if lut is None else lut, np.float32)
    out = np.empty((b, 3, t, size, size), np.float32)
    nb = L.eilev_process_workspace_bytes(b, t, h, size)
    ws = np.empty(max(nb, 1), np.uint8)
    pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    abi.check(L.eilev_process_frames(pp(video), b, t, h, w, size, size, pp(ch), pp(bh), kh, pp(cv), pp(bv), kv, pp(lut), pp(out), 0,
                                     pp(ws), nb, None), "eilev_process_frames (oracle)")
    return out
