"""Gradient building blocks (include/eilev.h): oracle vs torch.autograd on CPU, HIP vs oracle on the GPU.

The oracle (fp32 C) is pinned against torch.autograd of the same fp32 op to 1e-4 relative (accumulation order only).
The HIP kernels store bf16 and feed bf16 P / dS into the MFMAs: gradients must agree with the oracle to 2e-2 of the
tensor's max |value| (bf16 has 2^-9 relative steps; P and dS are rounded once more before the second product).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from eilev_amd.synth import det_normal, round_bf16
from oracle import runner as orc

pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)


def _attn_inputs(b, h, sq, skv, hd, seed=0, pad=0):
    q = round_bf16(det_normal("q", (b, sq, h * hd), seed))
    k = round_bf16(det_normal("k", (b, skv, h * hd), seed))
    v = round_bf16(det_normal("v", (b, skv, h * hd), seed))
    do = round_bf16(det_normal("do", (b, sq, h * hd), seed))
    mask = None
    if pad:
        mask = np.ones((b, skv), np.int32)
        mask[0, :pad] = 0  # left padding, as the collator produces
    return q, k, v, do, mask


def _oracle_attn_bwd(q, k, v, do, b, h, sq, skv, hd, scale, causal, mask):
    L = orc.lib()
    o = np.empty((b, sq, h * hd), np.float32)
    assert L.eilev_attention(pp(q), pp(k), pp(v), pp(o), b, h, sq, skv, hd, h * hd, h * hd, h * hd, scale, causal, pp(mask), None) == 0
    dq, dk, dv = np.empty_like(q), np.empty_like(k), np.empty_like(v)
    ws = np.empty((2, b, h, sq), np.float32)
    assert L.eilev_attention_bwd(pp(q), pp(k), pp(v), pp(o), pp(do), pp(dq), pp(dk), pp(dv), pp(ws), b, h, sq, skv, hd, h * hd, h * hd,
                                 h * hd, h * hd, h * hd, h * hd, scale, causal, pp(mask), None) == 0
    return o, dq, dk, dv, ws


def _torch_attn_bwd(q, k, v, do, b, h, sq, skv, hd, scale, causal, mask):
    tq, tk, tv = (torch.tensor(x, requires_grad=True) for x in (q, k, v))
    Q = tq.view(b, sq, h, hd).transpose(1, 2)
    K = tk.view(b, skv, h, hd).transpose(1, 2)
    V = tv.view(b, skv, h, hd).transpose(1, 2)
    s = (Q @ K.transpose(-1, -2)) * scale
    ok = torch.ones(b, 1, sq, skv, dtype=torch.bool)
    if causal:
        i = torch.arange(sq)[:, None]
        j = torch.arange(skv)[None, :]
        ok = ok & (j <= i + (skv - sq))
    if mask is not None:
        ok = ok & torch.tensor(mask, dtype=torch.bool)[:, None, None, :]
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)  # rows without a visible key
    o = (p @ V).transpose(1, 2).reshape(b, sq, h * hd)
    o.backward(torch.tensor(do))
    return o.detach().numpy(), tq.grad.numpy(), tk.grad.numpy(), tv.grad.numpy()


ATTN_CASES = [
    # b, h, sq, skv, hd, causal, pad
    (2, 3, 40, 40, 16, 1, 5),      # causal + left padding (OPT)
    (1, 2, 33, 33, 80, 1, 0),      # OPT-2.7b head size
    (2, 2, 32, 32, 64, 0, 0),      # Q-Former self-attention
    (1, 2, 32, 150, 64, 0, 0),     # Q-Former cross-attention: few queries, long ragged keys
    (1, 1, 70, 130, 24, 1, 0),     # causal with a prefix (skv > sq), ragged tiles
]


@pytest.mark.parametrize("b,h,sq,skv,hd,causal,pad", ATTN_CASES)
def test_oracle_attention_bwd_vs_autograd(b, h, sq, skv, hd, causal, pad):
    q, k, v, do, mask = _attn_inputs(b, h, sq, skv, hd, pad=pad)
    scale = hd ** -0.5
    o, dq, dk, dv, _ = _oracle_attn_bwd(q, k, v, do, b, h, sq, skv, hd, scale, causal, mask)
    to, tdq, tdk, tdv = _torch_attn_bwd(q, k, v, do, b, h, sq, skv, hd, scale, causal, mask)
    valid = np.ones((b, sq), bool)
    if pad and causal:
        valid[0, :pad] = False  # fully masked rows: the forward value is unspecified (hf yields a uniform average)
    np.testing.assert_allclose(o[valid], to[valid], rtol=1e-4, atol=1e-5)
    if pad and causal:
        do = do.copy()
        do[~valid] = 0.0  # what autograd delivers to those rows in the model (no loss, no visible key)
        _, dq, dk, dv, _ = _oracle_attn_bwd(q, k, v, do, b, h, sq, skv, hd, scale, causal, mask)
        _, tdq, tdk, tdv = _torch_attn_bwd(q, k, v, do, b, h, sq, skv, hd, scale, causal, mask)
    for got, ref in ((dq, tdq), (dk, tdk), (dv, tdv)):
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_oracle_layernorm_act_ce_vs_autograd():
    L = orc.lib()
    rows, cols = 37, 96
    x = det_normal("x", (rows, cols), 1) * 2 + 0.3
    g = det_normal("g", (cols,), 1) * 0.2 + 1
    dy = det_normal("dy", (rows, cols), 1)
    dx = np.empty_like(x)
    dg, db, st = np.zeros(cols, np.float32), np.zeros(cols, np.float32), np.empty((rows, 2), np.float32)
    assert L.eilev_layernorm_bwd(pp(x), pp(g), pp(dy), pp(dx), pp(dg), pp(db), pp(st), rows, cols, 1e-5, None) == 0
    tx, tg, tb = torch.tensor(x, requires_grad=True), torch.tensor(g, requires_grad=True), torch.zeros(cols, requires_grad=True)
    torch.nn.functional.layer_norm(tx, (cols,), tg, tb, 1e-5).backward(torch.tensor(dy))
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dg, tg.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-4, atol=1e-4)
    cs = np.zeros(cols, np.float32)
    assert L.eilev_colsum(pp(dy), pp(cs), rows, cols, None) == 0
    np.testing.assert_allclose(cs, dy.sum(0), rtol=1e-5, atol=1e-5)
    for kind, fn in ((1, torch.nn.functional.gelu), (2, torch.relu)):
        pre = np.ascontiguousarray(x.reshape(-1)[:3552])
        gy = np.ascontiguousarray(dy.reshape(-1)[:3552])
        y, gx = np.empty_like(pre), np.empty_like(pre)
        assert L.eilev_act_fwd(pp(pre), pp(y), pre.size, kind, None) == 0
        assert L.eilev_act_bwd(pp(pre), pp(gy), pp(gx), pre.size, kind, None) == 0
        tp = torch.tensor(pre, requires_grad=True)
        ty = fn(tp)
        ty.backward(torch.tensor(gy))
        np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gx, tp.grad.numpy(), rtol=1e-5, atol=1e-6)
    vocab, n = 211, 9
    logits = det_normal("lg", (n, vocab), 2) * 3
    tgt = np.array([5, -100, 210, 0, 17, -100, 99, 100, 3], np.int64)
    nv = int((tgt >= 0).sum())
    rl, dl = np.empty(n, np.float32), np.empty((n, vocab), np.float32)
    assert L.eilev_ce_loss(pp(logits), pp(tgt), 1.0 / nv, pp(rl), pp(dl), n, vocab, None) == 0
    tl = torch.tensor(logits, requires_grad=True)
    loss = torch.nn.functional.cross_entropy(tl, torch.tensor(tgt), ignore_index=-100)
    loss.backward()
    np.testing.assert_allclose(rl.sum() / nv, loss.item(), rtol=1e-5)
    np.testing.assert_allclose(dl, tl.grad.numpy(), rtol=1e-4, atol=1e-7)


# ---- HIP vs oracle ----------------------------------------------------------------------------------------------------------
HIP_ATTN_CASES = ATTN_CASES + [
    (2, 4, 200, 200, 80, 1, 37),   # several tiles, OPT head size, padded sample next to a full one
    (2, 3, 32, 2056, 64, 0, 0),    # Q-Former cross-attention at the real key count
    (1, 2, 257, 257, 88, 0, 0),    # ViT-shaped (hd 88), tail of 1
    (1, 2, 130, 130, 128, 1, 0),   # widest supported head
]


@pytest.mark.gpu
@pytest.mark.parametrize("b,h,sq,skv,hd,causal,pad", HIP_ATTN_CASES)
def test_hip_attention_bwd(b, h, sq, skv, hd, causal, pad):
    from eilev_amd import abi
    from hip_utils import P, dev_bf16, host, stream_ptr

    hip = abi.load_hip()
    q, k, v, do, mask = _attn_inputs(b, h, sq, skv, hd, pad=pad)
    if pad and causal:
        do[0, :pad] = 0.0
    scale = hd ** -0.5
    dq_, dk_, dv_, d_o = dev_bf16(q), dev_bf16(k), dev_bf16(v), dev_bf16(do)
    dm = torch.from_numpy(mask).cuda() if mask is not None else None
    o = torch.empty((b, sq, h * hd), dtype=torch.bfloat16, device="cuda")
    W = h * hd
    assert hip.eilev_attention(P(dq_), P(dk_), P(dv_), P(o), b, h, sq, skv, hd, W, W, W, scale, causal, P(dm), stream_ptr()) == 0
    # the oracle differentiates at the bf16 output the kernel saved, so delta sees the same o
    o_np = host(o)
    L = orc.lib()
    rdq, rdk, rdv = np.empty_like(q), np.empty_like(k), np.empty_like(v)
    ws = np.empty((2, b, h, sq), np.float32)
    assert L.eilev_attention_bwd(pp(q), pp(k), pp(v), pp(o_np), pp(do), pp(rdq), pp(rdk), pp(rdv), pp(ws), b, h, sq, skv, hd, W, W, W, W, W, W,
                                 scale, causal, pp(mask), None) == 0
    gq, gk, gv = torch.empty_like(dq_), torch.empty_like(dk_), torch.empty_like(dv_)
    gws = torch.empty((2, b, h, sq), dtype=torch.float32, device="cuda")
    rc = hip.eilev_attention_bwd(P(dq_), P(dk_), P(dv_), P(o), P(d_o), P(gq), P(gk), P(gv), P(gws), b, h, sq, skv, hd, W, W, W, W, W, W, scale,
                                 causal, P(dm), stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    vis = ws[0] < 1e29  # rows with a visible key
    np.testing.assert_allclose(host(gws)[0][vis], ws[0][vis], rtol=0, atol=2e-2)
    np.testing.assert_allclose(host(gws)[1], ws[1], rtol=0, atol=2e-2 * max(1.0, np.abs(ws[1]).max()))
    for name, got, ref in (("dq", gq, rdq), ("dk", gk, rdk), ("dv", gv, rdv)):
        err = np.abs(host(got) - ref).max()
        assert err <= 2e-2 * np.abs(ref).max(), (name, err, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(37, 96), (544, 768), (1000, 1408), (130, 2560), (65, 4096)])
def test_hip_layernorm_bwd_and_colsum(rows, cols):
    from eilev_amd import abi
    from hip_utils import P, dev_bf16, host, stream_ptr

    hip = abi.load_hip()
    L = orc.lib()
    x = round_bf16(det_normal("x", (rows, cols), 1) * 2 + 0.3)
    g = round_bf16(det_normal("g", (cols,), 1) * 0.2 + 1)
    dy = round_bf16(det_normal("dy", (rows, cols), 1))
    rdx = np.empty_like(x)
    rdg, rdb, rst = np.zeros(cols, np.float32), np.zeros(cols, np.float32), np.empty((rows, 2), np.float32)
    assert L.eilev_layernorm_bwd(pp(x), pp(g), pp(dy), pp(rdx), pp(rdg), pp(rdb), pp(rst), rows, cols, 1e-5, None) == 0
    dx_, dg_, ddy = dev_bf16(x), dev_bf16(g), dev_bf16(dy)
    gdx = torch.empty_like(dx_)
    gdg = torch.zeros(cols, dtype=torch.float32, device="cuda")
    gdb = torch.zeros_like(gdg)
    gst = torch.empty((rows, 2), dtype=torch.float32, device="cuda")
    assert hip.eilev_layernorm_bwd(P(dx_), P(dg_), P(ddy), P(gdx), P(gdg), P(gdb), P(gst), rows, cols, 1e-5, stream_ptr()) == 0
    cs = torch.zeros(cols, dtype=torch.float32, device="cuda")
    assert hip.eilev_colsum(P(ddy), P(cs), rows, cols, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert np.abs(host(gdx) - rdx).max() <= 1e-2 * np.abs(rdx).max()
    np.testing.assert_allclose(host(gdg), rdg, rtol=1e-3, atol=1e-3 * np.abs(rdg).max())
    np.testing.assert_allclose(host(gdb), rdb, rtol=1e-3, atol=1e-3 * np.abs(rdb).max())
    np.testing.assert_allclose(host(cs), dy.sum(0), rtol=1e-3, atol=1e-3 * np.abs(rdb).max())
    # frozen LayerNorm (the language model): dx only
    gdx2 = torch.empty_like(dx_)
    assert hip.eilev_layernorm_bwd(P(dx_), P(dg_), P(ddy), P(gdx2), None, None, None, rows, cols, 1e-5, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(gdx, gdx2)


@pytest.mark.gpu
def test_hip_act_and_ce():
    from eilev_amd import abi
    from hip_utils import P, dev_bf16, host, stream_ptr

    hip = abi.load_hip()
    L = orc.lib()
    n = 8 * 4099
    pre = round_bf16(det_normal("pre", (n,), 3) * 2)
    gy = round_bf16(det_normal("gy", (n,), 3))
    for kind in (1, 2):
        ry, rx = np.empty_like(pre), np.empty_like(pre)
        assert L.eilev_act_fwd(pp(pre), pp(ry), n, kind, None) == 0
        assert L.eilev_act_bwd(pp(pre), pp(gy), pp(rx), n, kind, None) == 0
        dp, dg = dev_bf16(pre), dev_bf16(gy)
        y, gx = torch.empty_like(dp), torch.empty_like(dp)
        assert hip.eilev_act_fwd(P(dp), P(y), n, kind, stream_ptr()) == 0
        assert hip.eilev_act_bwd(P(dp), P(dg), P(gx), n, kind, stream_ptr()) == 0
        torch.cuda.synchronize()
        np.testing.assert_allclose(host(y), ry, rtol=8e-3, atol=2e-3)
        np.testing.assert_allclose(host(gx), rx, rtol=8e-3, atol=1e-4)
    vocab, rows = 50272, 7
    logits = det_normal("lg", (rows, vocab), 2) * 3
    tgt = np.array([5, -100, 50271, 0, 17, -100, 999], np.int64)
    nv = int((tgt >= 0).sum())
    rl, dl = np.empty(rows, np.float32), np.empty((rows, vocab), np.float32)
    assert L.eilev_ce_loss(pp(logits), pp(tgt), 1.0 / nv, pp(rl), pp(dl), rows, vocab, None) == 0
    dlg = torch.from_numpy(logits).cuda()
    dt = torch.from_numpy(tgt).cuda()
    grl = torch.empty(rows, dtype=torch.float32, device="cuda")
    gdl = torch.empty((rows, vocab), dtype=torch.bfloat16, device="cuda")
    assert hip.eilev_ce_loss(P(dlg), P(dt), 1.0 / nv, P(grl), P(gdl), rows, vocab, stream_ptr()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(host(grl), rl, rtol=1e-5, atol=1e-5)
    assert np.abs(host(gdl) - dl).max() <= 8e-3 * np.abs(dl).max()


# ---- encoder-decoder (T5) building blocks -------------------------------------------------------------------------------------
def _rel_table(h, n, seed=5):
    return (det_normal("rel", (h, n), seed) * 0.7).astype(np.float32)


REL_CASES = [
    # b, h, sq, skv, hd, causal, pad (right padding, as the T5 encoder sees it)
    (2, 3, 40, 40, 16, 0, 6),    # encoder: bidirectional + key mask + bias
    (2, 2, 37, 37, 64, 1, 0),    # decoder self-attention: causal + bias (flan-t5 d_kv = 64)
    (1, 4, 130, 130, 64, 1, 0),  # several tiles
]


def _rel_inputs(b, h, sq, skv, hd, pad):
    q, k, v, do, _ = _attn_inputs(b, h, sq, skv, hd, seed=7)
    mask = None
    if pad:
        mask = np.ones((b, skv), np.int32)
        mask[1, skv - pad:] = 0
    return q, k, v, do, mask


@pytest.mark.parametrize("b,h,sq,skv,hd,causal,pad", REL_CASES)
def test_oracle_attention_rel_vs_autograd(b, h, sq, skv, hd, causal, pad):
    L = orc.lib()
    q, k, v, do, mask = _rel_inputs(b, h, sq, skv, hd, pad)
    n = 2 * skv - 1
    tab = _rel_table(h, n)
    W = h * hd
    o = np.empty((b, sq, W), np.float32)
    assert L.eilev_attention_rel(pp(q), pp(k), pp(v), pp(o), b, h, sq, skv, hd, W, W, W, 1.0, causal, pp(mask), pp(tab), n, skv - 1, n, None) == 0
    dq, dk, dv = np.empty_like(q), np.empty_like(k), np.empty_like(v)
    ws = np.empty((2, b, h, sq), np.float32)
    assert L.eilev_attention_rel_bwd(pp(q), pp(k), pp(v), pp(o), pp(do), pp(dq), pp(dk), pp(dv), pp(ws), b, h, sq, skv, hd, W, W, W, W, W, W, 1.0,
                                     causal, pp(mask), pp(tab), n, skv - 1, n, None) == 0
    tq, tk, tv = (torch.tensor(x, requires_grad=True) for x in (q, k, v))
    Q, K, V = (t.view(b, -1, h, hd).transpose(1, 2) for t in (tq, tk, tv))
    i = torch.arange(sq)[:, None]
    j = torch.arange(skv)[None, :]
    bias = torch.tensor(tab)[:, (j - i) + skv - 1]  # (h, sq, skv)
    s = Q @ K.transpose(-1, -2) + bias[None]
    ok = torch.ones(b, 1, sq, skv, dtype=torch.bool)
    if causal:
        ok = ok & (j <= i)
    if mask is not None:
        ok = ok & torch.tensor(mask, dtype=torch.bool)[:, None, None, :]
    p = torch.softmax(s.masked_fill(~ok, float("-inf")), -1)
    to = (p @ V).transpose(1, 2).reshape(b, sq, W)
    to.backward(torch.tensor(do))
    np.testing.assert_allclose(o, to.detach().numpy(), rtol=1e-4, atol=1e-5)
    for got, ref in ((dq, tq.grad.numpy()), (dk, tk.grad.numpy()), (dv, tv.grad.numpy())):
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_oracle_rmsnorm_gated_gelu_vs_autograd():
    L = orc.lib()
    rows, cols, F = 29, 96, 64
    x = det_normal("x", (rows, cols), 1) * 2 + 0.3
    g = det_normal("g", (cols,), 1) * 0.2 + 1
    dy = det_normal("dy", (rows, cols), 1)
    y, dx = np.empty_like(x), np.empty_like(x)
    assert L.eilev_rmsnorm(pp(x), pp(g), pp(y), rows, cols, 1e-6, None) == 0
    assert L.eilev_rmsnorm_bwd(pp(x), pp(g), pp(dy), pp(dx), rows, cols, 1e-6, None) == 0
    tx = torch.tensor(x, requires_grad=True)
    ty = torch.tensor(g) * (tx * torch.rsqrt(tx.pow(2).mean(-1, keepdim=True) + 1e-6))  # T5LayerNorm
    ty.backward(torch.tensor(dy))
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=1e-4, atol=1e-5)
    ab = det_normal("ab", (rows, 2 * F), 2) * 1.5
    gy = det_normal("gy", (rows, F), 2)
    out, dab = np.empty((rows, F), np.float32), np.empty_like(ab)
    assert L.eilev_gated_gelu(pp(ab), pp(out), rows, F, None) == 0
    assert L.eilev_gated_gelu_bwd(pp(ab), pp(gy), pp(dab), rows, F, None) == 0
    tab_ = torch.tensor(ab, requires_grad=True)
    a, bb = tab_[:, :F], tab_[:, F:]
    tout = 0.5 * a * (1.0 + torch.tanh(0.7978845608028654 * (a + 0.044715 * a.pow(3)))) * bb  # NewGELUActivation * linear branch
    tout.backward(torch.tensor(gy))
    np.testing.assert_allclose(out, tout.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dab, tab_.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("b,h,sq,skv,hd,causal,pad", REL_CASES + [(2, 32, 200, 200, 64, 0, 30)])
def test_hip_attention_rel(b, h, sq, skv, hd, causal, pad):
    from eilev_amd import abi
    from hip_utils import P, dev_bf16, host, stream_ptr

    hip = abi.load_hip()
    L = orc.lib()
    q, k, v, do, mask = _rel_inputs(b, h, sq, skv, hd, pad)
    n = 2 * skv - 1
    tab = _rel_table(h, n)
    W = h * hd
    dq_, dk_, dv_, d_o = dev_bf16(q), dev_bf16(k), dev_bf16(v), dev_bf16(do)
    dm = torch.from_numpy(mask).cuda() if mask is not None else None
    dtab = torch.from_numpy(tab).cuda()
    o = torch.empty((b, sq, W), dtype=torch.bfloat16, device="cuda")
    assert hip.eilev_attention_rel(P(dq_), P(dk_), P(dv_), P(o), b, h, sq, skv, hd, W, W, W, 1.0, causal, P(dm), P(dtab), n, skv - 1, n,
                                   stream_ptr()) == 0
    ro = np.empty((b, sq, W), np.float32)
    assert L.eilev_attention_rel(pp(q), pp(k), pp(v), pp(ro), b, h, sq, skv, hd, W, W, W, 1.0, causal, pp(mask), pp(tab), n, skv - 1, n, None) == 0
    torch.cuda.synchronize()
    o_np = host(o)
    assert np.abs(o_np - ro).max() <= 1e-2 * np.abs(ro).max()
    rdq, rdk, rdv = np.empty_like(q), np.empty_like(k), np.empty_like(v)
    ws = np.empty((2, b, h, sq), np.float32)
    assert L.eilev_attention_rel_bwd(pp(q), pp(k), pp(v), pp(o_np), pp(do), pp(rdq), pp(rdk), pp(rdv), pp(ws), b, h, sq, skv, hd, W, W, W, W, W, W,
                                     1.0, causal, pp(mask), pp(tab), n, skv - 1, n, None) == 0
    gq, gk, gv = torch.empty_like(dq_), torch.empty_like(dk_), torch.empty_like(dv_)
    gws = torch.empty((2, b, h, sq), dtype=torch.float32, device="cuda")
    assert hip.eilev_attention_rel_bwd(P(dq_), P(dk_), P(dv_), P(o), P(d_o), P(gq), P(gk), P(gv), P(gws), b, h, sq, skv, hd, W, W, W, W, W, W, 1.0,
                                       causal, P(dm), P(dtab), n, skv - 1, n, stream_ptr()) == 0
    torch.cuda.synchronize()
    for name, got, ref in (("dq", gq, rdq), ("dk", gk, rdk), ("dv", gv, rdv)):
        err = np.abs(host(got) - ref).max()
        assert err <= 2e-2 * np.abs(ref).max(), (name, err, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,F", [(29, 96, 64), (700, 2048, 5120), (65, 512, 1024)])
def test_hip_rmsnorm_gated_gelu(rows, cols, F):
    from eilev_amd import abi
    from hip_utils import P, dev_bf16, host, stream_ptr

    hip = abi.load_hip()
    L = orc.lib()
    x = round_bf16(det_normal("x", (rows, cols), 1) * 2 + 0.3)
    g = round_bf16(det_normal("g", (cols,), 1) * 0.2 + 1)
    dy = round_bf16(det_normal("dy", (rows, cols), 1))
    ry, rdx = np.empty_like(x), np.empty_like(x)
    assert L.eilev_rmsnorm(pp(x), pp(g), pp(ry), rows, cols, 1e-6, None) == 0
    assert L.eilev_rmsnorm_bwd(pp(x), pp(g), pp(dy), pp(rdx), rows, cols, 1e-6, None) == 0
    dx_, dg_, ddy = dev_bf16(x), dev_bf16(g), dev_bf16(dy)
    y, gdx = torch.empty_like(dx_), torch.empty_like(dx_)
    assert hip.eilev_rmsnorm(P(dx_), P(dg_), P(y), rows, cols, 1e-6, stream_ptr()) == 0
    assert hip.eilev_rmsnorm_bwd(P(dx_), P(dg_), P(ddy), P(gdx), rows, cols, 1e-6, stream_ptr()) == 0
    ab = round_bf16(det_normal("ab", (rows, 2 * F), 2) * 1.5)
    gy = round_bf16(det_normal("gy", (rows, F), 2))
    rout, rdab = np.empty((rows, F), np.float32), np.empty_like(ab)
    assert L.eilev_gated_gelu(pp(ab), pp(rout), rows, F, None) == 0
    assert L.eilev_gated_gelu_bwd(pp(ab), pp(gy), pp(rdab), rows, F, None) == 0
    dab_, dgy = dev_bf16(ab), dev_bf16(gy)
    out = torch.empty((rows, F), dtype=torch.bfloat16, device="cuda")
    gdab = torch.empty_like(dab_)
    assert hip.eilev_gated_gelu(P(dab_), P(out), rows, F, stream_ptr()) == 0
    assert hip.eilev_gated_gelu_bwd(P(dab_), P(dgy), P(gdab), rows, F, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert np.abs(host(y) - ry).max() <= 1e-2 * np.abs(ry).max()
    assert np.abs(host(gdx) - rdx).max() <= 1e-2 * np.abs(rdx).max()
    np.testing.assert_allclose(host(out), rout, rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(host(gdab), rdab, rtol=1e-2, atol=2e-3)


# ---- dropout of the training graph -------------------------------------------------------------------------------------------
def _hash32(seed, idx):
    """include/eilev.h: splitmix64 finaliser of (seed, element index) (numpy uint64 wraps like the C code)."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(seed + 1)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(32)).astype(np.uint32)


def _keep(seed, idx, p):
    return _hash32(seed, idx) >= np.uint32(min(int(p * 4294967296.0), 4294967295))


def test_oracle_dropout_vs_autograd_with_the_same_mask():
    L = orc.lib()
    n, p, seed = 8 * 1201, 0.1, 12345
    x = det_normal("x", (n,), 4)
    r = det_normal("r", (n,), 5)
    y = np.empty_like(x)
    assert L.eilev_dropout_add(pp(x), pp(r), pp(y), n, p, seed, None) == 0
    keep = _keep(seed, np.arange(n), p)
    assert abs(keep.mean() - 0.9) < 0.02
    np.testing.assert_allclose(y, np.where(keep, x / np.float32(0.9), 0).astype(np.float32) + r, rtol=1e-6, atol=1e-6)
    # attention: the oracle's gradients = autograd of softmax(s) * M / (1 - p) @ v with M taken from the documented hash
    b, h, sq, skv, hd, causal = 2, 3, 37, 50, 16, 0
    q, k, v, do, _ = _attn_inputs(b, h, sq, skv, hd, seed=9)
    W = h * hd
    o = np.empty((b, sq, W), np.float32)
    assert L.eilev_attention_dropout(pp(q), pp(k), pp(v), pp(o), b, h, sq, skv, hd, W, W, W, 0.25, causal, None, None, 0, 0, 0, p, seed, None) == 0
    dq, dk, dv = np.empty_like(q), np.empty_like(k), np.empty_like(v)
    ws = np.empty((2, b, h, sq), np.float32)
    assert L.eilev_attention_dropout_bwd(pp(q), pp(k), pp(v), pp(o), pp(do), pp(dq), pp(dk), pp(dv), pp(ws), b, h, sq, skv, hd, W, W, W, W, W, W,
                                         0.25, causal, None, None, 0, 0, 0, p, seed, None) == 0
    idx = np.arange(b * h * sq * skv).reshape(b, h, sq, skv)
    M = torch.tensor(_keep(seed, idx, p).astype(np.float32) / 0.9)
    tq, tk, tv = (torch.tensor(a, requires_grad=True) for a in (q, k, v))
    Q, K, V = (t.view(b, -1, h, hd).transpose(1, 2) for t in (tq, tk, tv))
    to = ((torch.softmax(Q @ K.transpose(-1, -2) * 0.25, -1) * M) @ V).transpose(1, 2).reshape(b, sq, W)
    to.backward(torch.tensor(do))
    np.testing.assert_allclose(o, to.detach().numpy(), rtol=1e-4, atol=1e-5)
    for got, ref in ((dq, tq.grad.numpy()), (dk, tk.grad.numpy()), (dv, tv.grad.numpy())):
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("b,h,sq,skv,hd,causal,rel", [(2, 3, 37, 50, 16, 0, False), (17, 12, 32, 300, 64, 0, False), (2, 4, 70, 70, 64, 1, True)])
def test_hip_dropout(b, h, sq, skv, hd, causal, rel):
    from eilev_amd import abi
    from hip_utils import P, dev_bf16, host, stream_ptr

    hip = abi.load_hip()
    L = orc.lib()
    p, seed = 0.1, 777
    n = 8 * 4099
    x = round_bf16(det_normal("x", (n,), 4))
    r = round_bf16(det_normal("r", (n,), 5))
    ry = np.empty_like(x)
    assert L.eilev_dropout_add(pp(x), pp(r), pp(ry), n, p, seed, None) == 0
    dx, dr = dev_bf16(x), dev_bf16(r)
    y = torch.empty_like(dx)
    assert hip.eilev_dropout_add(P(dx), P(dr), P(y), n, p, seed, stream_ptr()) == 0
    y2 = torch.empty_like(dx)
    assert hip.eilev_dropout_add(P(dx), None, P(y2), n, p, seed, stream_ptr()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(host(y), ry, rtol=1e-2, atol=1e-2)
    assert np.array_equal(host(y2) == 0, ~_keep(seed, np.arange(n), p) | (x == 0))  # exactly the documented mask
    q, k, v, do, _ = _attn_inputs(b, h, sq, skv, hd, seed=9)
    W = h * hd
    nrel = 2 * skv - 1
    tab = _rel_table(h, nrel) if rel else None
    dtab = torch.from_numpy(tab).cuda() if rel else None
    ra = (pp(tab), nrel, skv - 1, nrel) if rel else (None, 0, 0, 0)
    ga = (P(dtab), nrel, skv - 1, nrel) if rel else (None, 0, 0, 0)
    scale = hd ** -0.5
    dq_, dk_, dv_, d_o = dev_bf16(q), dev_bf16(k), dev_bf16(v), dev_bf16(do)
    o = torch.empty((b, sq, W), dtype=torch.bfloat16, device="cuda")
    assert hip.eilev_attention_dropout(P(dq_), P(dk_), P(dv_), P(o), b, h, sq, skv, hd, W, W, W, scale, causal, None, *ga, p, seed, stream_ptr()) == 0
    ro = np.empty((b, sq, W), np.float32)
    assert L.eilev_attention_dropout(pp(q), pp(k), pp(v), pp(ro), b, h, sq, skv, hd, W, W, W, scale, causal, None, *ra, p, seed, None) == 0
    torch.cuda.synchronize()
    o_np = host(o)
    assert np.abs(o_np - ro).max() <= 1.5e-2 * np.abs(ro).max()
    rdq, rdk, rdv = np.empty_like(q), np.empty_like(k), np.empty_like(v)
    ws = np.empty((2, b, h, sq), np.float32)
    assert L.eilev_attention_dropout_bwd(pp(q), pp(k), pp(v), pp(o_np), pp(do), pp(rdq), pp(rdk), pp(rdv), pp(ws), b, h, sq, skv, hd, W, W, W, W, W,
                                         W, scale, causal, None, *ra, p, seed, None) == 0
    gq, gk, gv = torch.empty_like(dq_), torch.empty_like(dk_), torch.empty_like(dv_)
    gws = torch.empty((2, b, h, sq), dtype=torch.float32, device="cuda")
    assert hip.eilev_attention_dropout_bwd(P(dq_), P(dk_), P(dv_), P(o), P(d_o), P(gq), P(gk), P(gv), P(gws), b, h, sq, skv, hd, W, W, W, W, W, W,
                                           scale, causal, None, *ga, p, seed, stream_ptr()) == 0
    torch.cuda.synchronize()
    for name, got, ref in (("dq", gq, rdq), ("dk", gk, rdk), ("dv", gv, rdv)):
        err = np.abs(host(got) - ref).max()
        assert err <= 2e-2 * np.abs(ref).max(), (name, err, np.abs(ref).max())
