"""torch.autograd bindings of the HIP kernels for the train_v2 path.

ref:scripts/general/train_v2.py:124-130 freezes the ViT and the language model and trains the Q-Former, the query
tokens and the language projection; `accelerator.backward(loss)` therefore needs parameter gradients for ~107 M
parameters and *activation* gradients through the whole frozen LM.  Autograd is kept as the bookkeeping (graph,
gradient accumulation into `.grad`), every gradient itself is computed by libeilev_hip.so:

  linear      y = x W^T + b (+ residual)     dX = dY . W         eilev_linear(dY, W^T)
                                             dW = dY^T . X       eilev_linear(dY^T, X^T) -> f32
                                             db = colsum(dY)     eilev_colsum
  layer_norm                                 eilev_layernorm_bwd (dx; dgamma/dbeta when trainable)
  attention   softmax(scale q k^T + mask) v  eilev_attention_bwd (P recomputed)
  act         GELU(erf) / ReLU               eilev_act_fwd / eilev_act_bwd
  lm_head_ce  mean CE of rows . E^T          eilev_linear (f32 logits) + eilev_ce_loss, dRows = dLogits . E

All activations are bf16; a trainable parameter may be an fp32 master copy (cast to bf16 for the kernels, gradient returned
in the parameter's dtype).  No fallback: tensors must live on the GPU and the HIP library must load.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import abi

_BF = torch.bfloat16
_tcache: dict = {}      # per-step transposes of activations shared by several linears (cleared by new_step())
_cur = {"frozen": {}}   # transposes of FROZEN weights: a dict owned by whoever owns those weights (the engine), see new_step()


def new_step(frozen_cache: dict | None = None) -> None:
    """Start a step.  ``frozen_cache`` holds the transposed copies of frozen weights; it must live and die with the tensors it
    mirrors (TrainGraph keeps it on its engine), because a freed weight's address can be reused by different values.  Every
    Function captures the dict at forward time, so interleaved graphs of different models cannot see each other's copies."""
    _tcache.clear()
    _cur["frozen"] = frozen_cache if frozen_cache is not None else {}


def _lib():
    return abi.load_hip()


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the training path runs on the GPU only (got a {t.device} tensor)")
    if t.dtype != _BF:
        raise TypeError(f"{what}: expected bf16 activations, got {t.dtype}")
    return t.contiguous()


def _bf(t: torch.Tensor) -> torch.Tensor:
    return (t if t.dtype == _BF else t.to(_BF)).contiguous()


def _padded_t(x2d: torch.Tensor, cache: bool) -> torch.Tensor:
    """(M, C) -> (C, ceil64(M)) with zero columns: the K-contiguous operand of a product contracting over rows."""
    key = (x2d.data_ptr(), x2d._version, tuple(x2d.shape))
    if cache and key in _tcache:
        return _tcache[key]
    M, Cc = x2d.shape
    Mp = (M + 63) // 64 * 64
    out = torch.zeros((Cc, Mp), dtype=_BF, device=x2d.device) if Mp != M else torch.empty((Cc, Mp), dtype=_BF, device=x2d.device)
    out[:, :M].copy_(x2d.t())
    if cache:
        _tcache[key] = out
    return out


def _weight_t(weight: torch.Tensor, w16: torch.Tensor, frozen: dict) -> torch.Tensor:
    """W^T (K, N) contiguous.  Frozen weights are transposed once per owner of ``frozen``."""
    if weight.requires_grad:
        return w16.t().contiguous()
    key = (weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = frozen.get(key)
    if hit is None:
        hit = frozen[key] = w16.t().contiguous()
    return hit


def _gemm(a, w, bias, resid, m, n, k, out_f32=False):
    out = torch.empty((m, n), dtype=torch.float32 if out_f32 else _BF, device=a.device)
    abi.check(_lib().eilev_linear(_p(a), _p(w), _p(bias), _p(resid), _p(out), m, n, k, 0, int(out_f32), _s()), "eilev_linear")
    return out


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        x2 = _need(x, "linear").reshape(-1, x.shape[-1])
        w16 = _bf(weight)
        N, K = w16.shape
        M = x2.shape[0]
        r2 = None if residual is None else _need(residual, "linear residual").reshape(M, N)
        y = _gemm(x2, w16, None if bias is None else _bf(bias), r2, M, N, K)
        ctx.save_for_backward(x2, weight, w16 if weight.requires_grad else None)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.has_resid = residual is not None
        ctx.xshape = x.shape
        ctx.frozen = _cur["frozen"]
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, w16 = ctx.saved_tensors
        if w16 is None:
            w16 = _bf(weight)
        N, K = w16.shape
        dy2 = _need(dy, "linear grad").reshape(-1, N)
        M = dy2.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gemm(dy2, _weight_t(weight, w16, ctx.frozen), None, None, M, K, N).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dyt = _padded_t(dy2, cache=False)
            xt = _padded_t(x2, cache=True)
            dw = _gemm(dyt, xt, None, None, N, K, dyt.shape[1], out_f32=True).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            acc = torch.zeros(N, dtype=torch.float32, device=dy2.device)
            abi.check(_lib().eilev_colsum(_p(dy2), _p(acc), M, N, _s()), "eilev_colsum")
            db = acc.to(ctx.bias_dtype)
        return dx, dw, db, (dy if ctx.has_resid else None)


class _LinearReLU(torch.autograd.Function):
    """relu(x W^T + b) with the activation in the GEMM epilogue.  The output doubles as the saved activation: relu'(pre) = (y > 0),
    so no pre-activation tensor is written (GELU cannot do this and keeps the separate act kernels)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = _need(x, "linear_relu").reshape(-1, x.shape[-1])
        w16 = _bf(weight)
        N, K = w16.shape
        M = x2.shape[0]
        y = torch.empty((M, N), dtype=_BF, device=x2.device)
        abi.check(_lib().eilev_linear(_p(x2), _p(w16), _p(None if bias is None else _bf(bias)), None, _p(y), M, N, K, 2, 0, _s()), "eilev_linear")
        ctx.save_for_backward(y, weight)
        ctx.xshape = x.shape
        ctx.frozen = _cur["frozen"]
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        y, weight = ctx.saved_tensors
        xshape, frozen = ctx.xshape, ctx.frozen
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise NotImplementedError("linear_relu is used for frozen layers (the OPT feed-forward); trainable layers use linear + relu")
        w16 = _bf(weight)
        N, K = w16.shape
        dy2 = _need(dy, "linear_relu grad").reshape(-1, N)
        dpre = torch.empty_like(dy2)
        abi.check(_lib().eilev_act_bwd(_p(y), _p(dy2), _p(dpre), dy2.numel(), 2, _s()), "eilev_act_bwd")
        dx = _gemm(dpre, _weight_t(weight, w16, frozen), None, None, dy2.shape[0], K, N).view(xshape)
        return dx, None, None


def linear_relu(x, weight, bias=None):
    return _LinearReLU.apply(x, weight, bias)


def linear(x, weight, bias=None, residual=None):
    return _Linear.apply(x, weight, bias, residual)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x2 = _need(x, "layer_norm").reshape(-1, x.shape[-1])
        g16, b16 = _bf(gamma), _bf(beta)
        y = torch.empty_like(x2)
        abi.check(_lib().eilev_layernorm(_p(x2), _p(g16), _p(b16), _p(y), x2.shape[0], x2.shape[1], float(eps), _s()), "eilev_layernorm")
        ctx.save_for_backward(x2, g16)
        ctx.eps = float(eps)
        ctx.pdtype = (gamma.dtype, beta.dtype)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g16 = ctx.saved_tensors
        rows, cols = x2.shape
        dy2 = _need(dy, "layer_norm grad").reshape(rows, cols)
        dx = torch.empty_like(x2)
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dg = db = st = None
        if want:
            dg = torch.zeros(cols, dtype=torch.float32, device=x2.device)
            db = torch.zeros_like(dg)
            st = torch.empty((rows, 2), dtype=torch.float32, device=x2.device)
        abi.check(_lib().eilev_layernorm_bwd(_p(x2), _p(g16), _p(dy2), _p(dx), _p(dg), _p(db), _p(st), rows, cols, ctx.eps, _s()),
                  "eilev_layernorm_bwd")
        return (dx.view(dy.shape), dg.to(ctx.pdtype[0]) if ctx.needs_input_grad[1] else None,
                db.to(ctx.pdtype[1]) if ctx.needs_input_grad[2] else None, None)


def layer_norm(x, gamma, beta, eps):
    return _LayerNorm.apply(x, gamma, beta, eps)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale, causal, key_mask, rel, drop):
        q, k, v = _need(q, "attention q"), _need(k, "attention k"), _need(v, "attention v")
        B, Sq, D = q.shape
        Skv = k.shape[1]
        hd = D // heads
        km = None if key_mask is None else key_mask.to(torch.int32).contiguous()
        o = torch.empty_like(q)
        tab, roff = _rel_args(rel, heads)
        dp, dseed = _drop_args(drop)
        abi.check(_lib().eilev_attention_dropout(_p(q), _p(k), _p(v), _p(o), B, heads, Sq, Skv, hd, D, D, D, float(scale), int(causal), _p(km),
                                                 _p(tab), 0 if tab is None else tab.shape[1], roff, 0 if tab is None else tab.shape[1], dp, dseed,
                                                 _s()), "eilev_attention_dropout")
        ctx.save_for_backward(q, k, v, o, km, tab)
        ctx.cfg = (heads, float(scale), int(causal), roff, dp, dseed)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, km, tab = ctx.saved_tensors
        heads, scale, causal, roff, dp, dseed = ctx.cfg
        B, Sq, D = q.shape
        Skv = k.shape[1]
        hd = D // heads
        d_o = _need(d_o, "attention grad")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty((2, B, heads, Sq), dtype=torch.float32, device=q.device)
        rn = 0 if tab is None else tab.shape[1]
        abi.check(_lib().eilev_attention_dropout_bwd(_p(q), _p(k), _p(v), _p(o), _p(d_o), _p(dq), _p(dk), _p(dv), _p(ws), B, heads, Sq, Skv, hd,
                                                     D, D, D, D, D, D, scale, causal, _p(km), _p(tab), rn, roff, rn, dp, dseed, _s()),
                  "eilev_attention_dropout_bwd")
        return dq, dk, dv, None, None, None, None, None, None


def _rel_args(rel, heads):
    """rel = (table, offset): f32 (heads, n) bias per relative distance (key - query) + offset; frozen (no gradient)."""
    if rel is None:
        return None, 0
    tab, off = rel
    if tab.dtype != torch.float32 or tab.dim() != 2 or tab.shape[0] != heads or not tab.is_contiguous() or tab.requires_grad:
        raise TypeError("relative position bias: a contiguous frozen f32 (heads, n) table is expected")
    return tab, int(off)


def _drop_args(drop):
    """drop = (p, seed) or None: dropout on the attention probabilities (include/eilev.h "dropout of the training graph")."""
    if drop is None or drop[0] <= 0.0:
        return 0.0, 0
    return float(drop[0]), int(drop[1]) & 0xFFFFFFFF


def attention(q, k, v, heads, scale, causal=False, key_mask=None, rel=None, drop=None):
    """q (B, Sq, heads*hd), k / v (B, Skv, heads*hd) -> (B, Sq, heads*hd); causal: key <= query + (Skv - Sq)."""
    return _Attention.apply(q, k, v, heads, scale, causal, key_mask, rel, drop)


class _AttentionPacked(torch.autograd.Function):
    """Self-attention on a packed q|k|v projection (B, S, 3 D): the kernels read the three thirds in place (row stride 3 D) and
    the backward writes dq|dk|dv into one (B, S, 3 D) tensor — the gradient of the fused projection, no slicing copies."""

    @staticmethod
    def forward(ctx, qkv, heads, scale, causal, key_mask, rel, drop):
        qkv = _need(qkv, "attention qkv")
        B, S, D3 = qkv.shape
        D = D3 // 3
        hd = D // heads
        km = None if key_mask is None else key_mask.to(torch.int32).contiguous()
        o = torch.empty((B, S, D), dtype=_BF, device=qkv.device)
        base = qkv.data_ptr()
        tab, roff = _rel_args(rel, heads)
        rn = 0 if tab is None else tab.shape[1]
        dp, dseed = _drop_args(drop)
        abi.check(_lib().eilev_attention_dropout(C.c_void_p(base), C.c_void_p(base + 2 * D), C.c_void_p(base + 4 * D), _p(o), B, heads, S, S, hd,
                                                 D3, D3, D3, float(scale), int(causal), _p(km), _p(tab), rn, roff, rn, dp, dseed, _s()),
                  "eilev_attention_dropout")
        ctx.save_for_backward(qkv, o, km, tab)
        ctx.cfg = (heads, float(scale), int(causal), roff, dp, dseed)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, km, tab = ctx.saved_tensors
        heads, scale, causal, roff, dp, dseed = ctx.cfg
        B, S, D3 = qkv.shape
        D = D3 // 3
        hd = D // heads
        d_o = _need(d_o, "attention grad")
        dqkv = torch.empty_like(qkv)
        ws = torch.empty((2, B, heads, S), dtype=torch.float32, device=qkv.device)
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        rn = 0 if tab is None else tab.shape[1]
        abi.check(_lib().eilev_attention_dropout_bwd(C.c_void_p(base), C.c_void_p(base + 2 * D), C.c_void_p(base + 4 * D), _p(o), _p(d_o),
                                                     C.c_void_p(dbase), C.c_void_p(dbase + 2 * D), C.c_void_p(dbase + 4 * D), _p(ws), B, heads, S,
                                                     S, hd, D3, D3, D3, D3, D3, D3, scale, causal, _p(km), _p(tab), rn, roff, rn, dp, dseed,
                                                     _s()), "eilev_attention_dropout_bwd")
        return dqkv, None, None, None, None, None, None


def attention_packed(qkv, heads, scale, causal=False, key_mask=None, rel=None, drop=None):
    """qkv (B, S, 3 * heads * hd) = [q | k | v] per row -> (B, S, heads * hd)."""
    return _AttentionPacked.apply(qkv, heads, scale, causal, key_mask, rel, drop)


class _DropoutAdd(torch.autograd.Function):
    """x * M / (1 - p) (+ residual): nn.Dropout on a hidden state followed by the residual add, mask recomputed in the backward."""

    @staticmethod
    def forward(ctx, x, residual, p, seed):
        x = _need(x, "dropout")
        r = None if residual is None else _need(residual, "dropout residual")
        y = torch.empty_like(x)
        abi.check(_lib().eilev_dropout_add(_p(x), _p(r), _p(y), x.numel(), float(p), int(seed) & 0xFFFFFFFF, _s()), "eilev_dropout_add")
        ctx.cfg = (float(p), int(seed) & 0xFFFFFFFF, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, has_r = ctx.cfg
        dy = _need(dy, "dropout grad")
        dx = torch.empty_like(dy)
        abi.check(_lib().eilev_dropout_add(_p(dy), None, _p(dx), dy.numel(), p, seed, _s()), "eilev_dropout_add")
        return dx, (dy if has_r else None), None, None


def dropout_add(x, residual, p, seed):
    """dropout(x) + residual (residual may be None); p == 0 is a plain add."""
    if p <= 0.0:
        return x if residual is None else x + residual
    return _DropoutAdd.apply(x, residual, p, seed)


class _RMSNorm(torch.autograd.Function):
    """T5LayerNorm with a frozen weight (hf modeling_t5.py:50-72)."""

    @staticmethod
    def forward(ctx, x, gamma, eps):
        x2 = _need(x, "rms_norm").reshape(-1, x.shape[-1])
        g16 = _bf(gamma)
        y = torch.empty_like(x2)
        abi.check(_lib().eilev_rmsnorm(_p(x2), _p(g16), _p(y), x2.shape[0], x2.shape[1], float(eps), _s()), "eilev_rmsnorm")
        ctx.save_for_backward(x2, g16)
        ctx.eps = float(eps)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("the T5 layer-norm weights are frozen on the train_v2 path")
        x2, g16 = ctx.saved_tensors
        dy2 = _need(dy, "rms_norm grad").reshape(x2.shape)
        dx = torch.empty_like(x2)
        abi.check(_lib().eilev_rmsnorm_bwd(_p(x2), _p(g16), _p(dy2), _p(dx), x2.shape[0], x2.shape[1], ctx.eps, _s()), "eilev_rmsnorm_bwd")
        return dx.view(dy.shape), None, None


def rms_norm(x, gamma, eps):
    return _RMSNorm.apply(x, gamma, eps)


class _GatedGelu(torch.autograd.Function):
    """gelu_new(a) * b on rows [a | b] (T5DenseGatedActDense, hf modeling_t5.py:97-124)."""

    @staticmethod
    def forward(ctx, ab):
        ab2 = _need(ab, "gated_gelu").reshape(-1, ab.shape[-1])
        F = ab2.shape[1] // 2
        out = torch.empty((ab2.shape[0], F), dtype=_BF, device=ab2.device)
        abi.check(_lib().eilev_gated_gelu(_p(ab2), _p(out), ab2.shape[0], F, _s()), "eilev_gated_gelu")
        ctx.save_for_backward(ab2)
        ctx.shape = ab.shape
        return out.view(*ab.shape[:-1], F)

    @staticmethod
    def backward(ctx, dy):
        (ab2,) = ctx.saved_tensors
        F = ab2.shape[1] // 2
        dy2 = _need(dy, "gated_gelu grad").reshape(-1, F)
        dab = torch.empty_like(ab2)
        abi.check(_lib().eilev_gated_gelu_bwd(_p(ab2), _p(dy2), _p(dab), ab2.shape[0], F, _s()), "eilev_gated_gelu_bwd")
        return dab.view(ctx.shape)


def gated_gelu(ab):
    return _GatedGelu.apply(ab)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre, kind):
        pre = _need(pre, "act")
        y = torch.empty_like(pre)
        abi.check(_lib().eilev_act_fwd(_p(pre), _p(y), pre.numel(), kind, _s()), "eilev_act_fwd")
        ctx.save_for_backward(pre)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (pre,) = ctx.saved_tensors
        dy = _need(dy, "act grad")
        dx = torch.empty_like(pre)
        abi.check(_lib().eilev_act_bwd(_p(pre), _p(dy), _p(dx), pre.numel(), ctx.kind, _s()), "eilev_act_bwd")
        return dx, None


def gelu(pre):
    return _Act.apply(pre, 1)


def relu(pre):
    return _Act.apply(pre, 2)


class _LMHeadCE(torch.autograd.Function):
    """mean_r CE(rows[r] . E^T, targets[r]) over the rows with target >= 0 (hf loss_utils.ForCausalLMLoss after the shift)."""

    @staticmethod
    def forward(ctx, rows, embed, targets):
        rows = _need(rows, "lm_head_ce")
        e16 = _bf(embed)
        R, D = rows.shape
        V = e16.shape[0]
        tg = targets.to(rows.device, torch.int64).contiguous()
        n_valid = int((tg >= 0).sum().item())
        if n_valid == 0:
            # no supervised position in the batch: F.cross_entropy's mean over zero rows is NaN (what the reference's loss is);
            # the gradient is defined as zero here so that one empty batch does not poison the parameters
            ctx.empty = True
            ctx.shape = (R, D)
            return torch.full((), float("nan"), dtype=torch.float32, device=rows.device)
        ctx.empty = False
        logits = _gemm(rows, e16, None, None, R, V, D, out_f32=True)
        row_loss = torch.empty(R, dtype=torch.float32, device=rows.device)
        # dlogits is the A operand of dRows = dlogits . E: its row length is padded to a multiple of 256 (zero columns, matched by
        # zero rows of the transposed embedding) so that product takes the DMA kernels instead of the K-tail kernel
        Vp = (V + 255) // 256 * 256
        dlogits = torch.zeros((R, Vp), dtype=_BF, device=rows.device) if Vp != V else torch.empty((R, V), dtype=_BF, device=rows.device)
        if Vp != V:  # the kernel writes rows of length V: run it on a dense (R, V) buffer, then place it
            dense = torch.empty((R, V), dtype=_BF, device=rows.device)
            abi.check(_lib().eilev_ce_loss(_p(logits), _p(tg), 1.0 / n_valid, _p(row_loss), _p(dense), R, V, _s()), "eilev_ce_loss")
            dlogits[:, :V].copy_(dense)
        else:
            abi.check(_lib().eilev_ce_loss(_p(logits), _p(tg), 1.0 / n_valid, _p(row_loss), _p(dlogits), R, V, _s()), "eilev_ce_loss")
        ctx.save_for_backward(dlogits, embed, e16 if embed.requires_grad else None)
        ctx.frozen = _cur["frozen"]
        return row_loss.sum() / n_valid

    @staticmethod
    def backward(ctx, g):
        if ctx.empty:
            return torch.zeros(ctx.shape, dtype=_BF, device=g.device), None, None
        dlogits, embed, e16 = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("the token embedding / lm_head is frozen on the train_v2 path")
        if e16 is None:
            e16 = _bf(embed)
        R, Vp = dlogits.shape
        V, D = e16.shape
        key = ("embed_t", embed.data_ptr(), embed._version, V, D)
        et = ctx.frozen.get(key)
        if et is None:  # E^T (D, Vp), zero beyond V; the embedding is frozen on this path
            et = ctx.frozen[key] = torch.zeros((D, Vp), dtype=_BF, device=e16.device)
            et[:, :V].copy_(e16.t())
        # f32 output: few output tiles and K = 50 432 -> the split-K launch (include/eilev.h eilev_linear)
        drows = _gemm(dlogits, et, None, None, R, D, Vp, out_f32=True)
        return (drows * g.float()).to(_BF), None, None


def lm_head_ce(rows, embed, targets):
    return _LMHeadCE.apply(rows, embed, targets)
