"""-m gpu: unit parity of the HIP building blocks (through the C ABI) against the CPU oracle.

Inputs are bf16-exact; the oracle computes in fp32.  Tolerances are bf16 output rounding (2^-9 relative)
plus fp32 accumulation-order noise: |err| <= 1e-2 * max|ref| for GEMM/attention outputs stored in bf16,
<= 2e-5 * K-scaled for fp32 logits-style outputs.  Every shape uses ASYMMETRIC random operands so a
transposed fragment mapping cannot pass.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from eilev_amd import abi
from eilev_amd.synth import det_normal, round_bf16
from oracle import runner as orc

from hip_utils import P, dev_bf16, host, stream_ptr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    return abi.load_hip()


def _lin_case(hip, m, n, k, epi, bias, resid, out_f32=False, seed=0):
    a = round_bf16(det_normal(f"A{m}x{k}", (m, k), seed))
    w = round_bf16(det_normal(f"W{n}x{k}", (n, k), seed) / np.sqrt(k))
    b = round_bf16(0.5 * det_normal("b", (n,), seed)) if bias else None
    r = round_bf16(det_normal("r", (m, n), seed)) if resid else None
    ref = np.empty((m, n), np.float32)
    pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    assert orc.lib().eilev_linear(pp(a), pp(w), pp(b), pp(r), pp(ref), m, n, k, epi, 0, None) == 0
    da, dw = dev_bf16(a), dev_bf16(w)
    db = dev_bf16(b) if bias else None
    dr = dev_bf16(r) if resid else None
    out = torch.empty((m, n), dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
    rc = hip.eilev_linear(P(da), P(dw), P(db), P(dr), P(out), m, n, k, epi, int(out_f32), stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = host(out)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    tol = (2e-4 if out_f32 else 1e-2) * scale
    assert err <= tol, (m, n, k, epi, err, tol)


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (257, 528, 176), (300, 200, 72), (1000, 1408, 1408), (4112, 768, 3072),
                                   (513, 4224, 1408), (2056, 6144, 640)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_linear_tiled(hip, m, n, k, epi):
    _lin_case(hip, m, n, k, epi, bias=True, resid=(epi == 0))


@pytest.mark.parametrize("m,n,k,epi,bias,resid", [
    # fewer than 4 rounds of 256x256 tiles -> one-wave-per-SIMD persistent kernel (w6)
    (8229, 2048, 320, 0, True, False),     # M tail of 37 rows, short K
    (8229, 2304, 1408, 0, False, False),   # N a multiple of 128 but not of 256, no bias
    (16500, 4224, 256, 0, True, False),    # minimum K (4 K-steps), ViT qkv width
    (8229, 2048, 1408, 1, True, False),    # GELU
    (8229, 2560, 2560, 2, True, True),     # ReLU + residual (rows staged through the epilogue unit)
    # >= 4 rounds -> ping-pong kernel (pp4) with the lean epilogue on interior tiles
    (70001, 4096, 256, 1, True, True),     # GELU + residual, M tail of 113 rows (stores dropped by the descriptor)
    (70001, 4224, 320, 0, True, False),    # last column tile half empty (general epilogue) next to lean tiles
    (90003, 1408, 256, 0, True, True),     # N = 1408: 5 lean column tiles + the half tile, residual
])
def test_linear_persistent_kernels(hip, m, n, k, epi, bias, resid):
    _lin_case(hip, m, n, k, epi, bias=bias, resid=resid)


@pytest.mark.parametrize("m,n,k", [(1, 160, 160), (8, 2560, 2560), (8, 7680, 2560), (3, 320, 160), (16, 1000, 10240), (8, 50272, 2560),
                                   (32, 2560, 2560), (17, 7680, 2560), (25, 2560, 10240), (20, 320, 160)])
def test_linear_skinny(hip, m, n, k):
    _lin_case(hip, m, n, k, 0, bias=True, resid=True)
    _lin_case(hip, m, n, k, 2, bias=True, resid=False)


def test_linear_split_k_weight_gradient_shape(hip):
    """dW = dY^T X of the cross-attention K/V projections: few output tiles, K = rows of the step -> split-K with f32 atomics."""
    _lin_case(hip, 768, 1408, 8768, 0, bias=False, resid=False, out_f32=True)     # 66 tiles x 7 slices
    _lin_case(hip, 200, 264, 16448, 0, bias=False, resid=False, out_f32=True, seed=3)   # ragged tiles, 16 slices


def test_linear_f32_logits(hip):
    _lin_case(hip, 300, 1000, 160, 0, bias=False, resid=False, out_f32=True)
    _lin_case(hip, 4, 50272, 2560, 0, bias=False, resid=False, out_f32=True)
    _lin_case(hip, 32, 50272, 2560, 0, bias=False, resid=False, out_f32=True)


def test_linear_rejects_unaligned_k(hip):
    a = torch.zeros((32, 20), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((32, 20), dtype=torch.bfloat16, device="cuda")
    o = torch.zeros((32, 32), dtype=torch.bfloat16, device="cuda")
    assert hip.eilev_linear(P(a), P(w), None, None, P(o), 32, 32, 20, 0, 0, stream_ptr()) == abi_unsupported()


def abi_unsupported():
    return -2


@pytest.mark.parametrize("rows,cols", [(7, 128), (1000, 1408), (33, 2560), (5, 176), (64, 768), (3, 4096)])
def test_layernorm(hip, rows, cols):
    x = round_bf16(2.0 * det_normal("lnx", (rows, cols)) + 0.5)
    g = round_bf16(1.0 + 0.1 * det_normal("lng", (cols,)))
    b = round_bf16(0.1 * det_normal("lnb", (cols,)))
    ref = np.empty_like(x)
    pp = lambda v: v.ctypes.data_as(C.c_void_p)
    assert orc.lib().eilev_layernorm(pp(x), pp(g), pp(b), pp(ref), rows, cols, 1e-5, None) == 0
    dx, dg, db = dev_bf16(x), dev_bf16(g), dev_bf16(b)
    out = torch.empty((rows, cols), dtype=torch.bfloat16, device="cuda")
    assert hip.eilev_layernorm(P(dx), P(dg), P(db), P(out), rows, cols, 1e-5, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert np.abs(host(out) - ref).max() <= 1e-2 * np.abs(ref).max()


@pytest.mark.parametrize("batch,heads,sq,skv,hd,causal,masked", [
    (3, 2, 17, 17, 88, 0, False),      # ViT mid config
    (2, 16, 257, 257, 88, 0, False),   # ViT-g frame (persistent frame kernel, one pair per workgroup)
    (40, 16, 257, 257, 88, 0, False),  # 640 (frame, head) pairs > 256 CUs: the frame kernel's K/V ring over several pairs
    (3, 2, 270, 270, 88, 0, False),    # other S in (256, 272]: partially filled last key tile
    (2, 12, 32, 32, 64, 0, False),     # Q-Former self
    (2, 12, 32, 2056, 64, 0, False),   # Q-Former cross over 8 frames
    (2, 4, 100, 100, 80, 1, True),     # OPT prefill, left padding
    (1, 32, 200, 200, 80, 1, False),   # OPT-2.7B heads
    (2, 2, 70, 130, 128, 1, True),     # causal with offset (sq < skv), d=128: attn_prefill_v2_kernel<4, 4> (round 6: 272-byte padded LDS rows)
    (2, 3, 333, 333, 128, 1, True),    # OPT-6.7B heads: ragged length, left padding, <8, 4>
    (1, 2, 960, 960, 128, 1, False),   # the configs[4] prefill length
    (2, 2, 64, 200, 128, 0, True),     # not causal, masked, the smallest query count the route takes
    (1, 2, 40, 40, 128, 1, False),     # below it: the round-1 kernel
])
def test_attention(hip, batch, heads, sq, skv, hd, causal, masked):
    D = heads * hd
    q = round_bf16(det_normal("q", (batch, sq, D)))
    k = round_bf16(det_normal("k", (batch, skv, D)))
    v = round_bf16(det_normal("v", (batch, skv, D)))
    scale = 1.0 / np.sqrt(hd)
    km = None
    if masked:
        km = np.ones((batch, skv), np.int32)
        km[0, :5] = 0  # left padding on row 0
        if batch > 1:
            km[1, :1] = 0
    ref = np.empty((batch, sq, D), np.float32)
    pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    assert orc.lib().eilev_attention(pp(q), pp(k), pp(v), pp(ref), batch, heads, sq, skv, hd, D, D, D, scale, causal, pp(km), None) == 0
    dq, dk, dv = dev_bf16(q), dev_bf16(k), dev_bf16(v)
    dkm = torch.from_numpy(km).cuda() if masked else None
    out = torch.empty((batch, sq, D), dtype=torch.bfloat16, device="cuda")
    rc = hip.eilev_attention(P(dq), P(dk), P(dv), P(out), batch, heads, sq, skv, hd, D, D, D, scale, causal, P(dkm), stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = host(out)
    # rows whose every key is masked are undefined in the reference (HF) and never read downstream
    valid = np.ones((batch, sq), bool)
    if masked and causal:
        for b in range(batch):
            first = int(np.argmax(km[b] != 0))
            for i in range(sq):
                if i + (skv - sq) < first:
                    valid[b, i] = False
    err = np.abs(got - ref)[valid].max()
    assert err <= 2e-2 * np.abs(ref).max(), err


def test_attention_online_softmax_rescale_branch(hip):
    """One late key dominates (score jump >> 8 between tiles): exercises the alpha rescale of O and l."""
    batch, heads, sq, skv, hd = 1, 1, 16, 200, 64
    q = round_bf16(det_normal("q2", (batch, sq, hd)))
    k = round_bf16(0.1 * det_normal("k2", (batch, skv, hd)))
    v = round_bf16(det_normal("v2", (batch, skv, hd)))
    k[0, 150] = round_bf16(8.0 * q[0, 3])  # spike: q_3 . k_150 is huge, in the third tile
    ref = np.empty((batch, sq, hd), np.float32)
    pp = lambda x: x.ctypes.data_as(C.c_void_p)
    assert orc.lib().eilev_attention(pp(q), pp(k), pp(v), pp(ref), 1, 1, sq, skv, hd, hd, hd, hd, 1.0, 0, None, None) == 0
    dq, dk, dv = dev_bf16(q), dev_bf16(k), dev_bf16(v)
    out = torch.empty((batch, sq, hd), dtype=torch.bfloat16, device="cuda")
    assert hip.eilev_attention(P(dq), P(dk), P(dv), P(out), 1, 1, sq, skv, hd, hd, hd, hd, 1.0, 0, None, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert np.abs(host(out) - ref).max() <= 2e-2 * np.abs(ref).max()


@pytest.mark.parametrize("m,n,k,epi", [(1000, 1408, 1408, 0), (700, 1408, 640, 1), (520, 1536, 128, 0), (300, 128, 256, 2)])
def test_linear_256x256_half_column_tile(probes, m, n, k, epi):
    """N % 256 <= 128 with the 256x256 kernel forced: the last column tile runs the 8-waves-as-4x2 half-tile path."""
    probes.eilev_debug_gemm_flags(1 << 4)
    try:
        _lin_case(probes, m, n, k, epi, bias=True, resid=(epi == 0))
    finally:
        probes.eilev_debug_gemm_flags(0)


def test_linear_operand_over_2gib(probes):
    """An A operand beyond the 32-bit buffer-offset range of the LDS-DMA kernels (Q-Former k|v projection of a whole bench
    step) is processed in row chunks: same result as the register-staged kernel, which addresses with 64-bit pointers."""
    hip = raw = probes
    m, n, k = 800_000, 256, 1408  # 2.25 GB of A
    torch.manual_seed(0)
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ref = torch.empty_like(out)
    assert hip.eilev_linear(P(a), P(w), P(b), None, P(out), m, n, k, 0, 0, stream_ptr()) == 0
    raw.eilev_debug_gemm_flags(4)
    try:
        assert hip.eilev_linear(P(a), P(w), P(b), None, P(ref), m, n, k, 0, 0, stream_ptr()) == 0
    finally:
        raw.eilev_debug_gemm_flags(0)
    torch.cuda.synchronize()
    for rows in (slice(0, 4096), slice(m // 2, m // 2 + 4096), slice(m - 4096, m)):
        assert (out[rows].float() - ref[rows].float()).abs().max().item() <= 2e-2
    assert torch.equal(out[::997], ref[::997]) or (out[::997].float() - ref[::997].float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize("batch", [1, 3, 18, 41])
def test_attention_frame_two_group_kernel(probes, batch):
    """`attn_frame3_kernel` (two wave groups one phase apart: two query tiles per wave share every K / V fragment, the CLS row split over
    the keys and merged from eight partial softmaxes; the default from 256 frames, forced here by probe flag 16) against the oracle and
    the single-tile kernel.  1 / 3 frames: workgroups with one pair (no next pair to prefetch); 18: 288 pairs > 256 CUs, the buffer
    ring and the deferred CLS merge run over two pairs; 41: the XCD-aware walk with a ragged last round."""
    hip = raw = probes
    heads, sq, hd = 16, 257, 88
    D = heads * hd
    q, k, v = (round_bf16(det_normal(n, (batch, sq, D))) for n in ("qj", "kj", "vj"))
    ref = np.empty((batch, sq, D), np.float32)
    pp = lambda x: x.ctypes.data_as(C.c_void_p)
    assert orc.lib().eilev_attention(pp(q), pp(k), pp(v), pp(ref), batch, heads, sq, sq, hd, D, D, D, hd ** -0.5, 0, None, None) == 0
    dq, dk, dv = dev_bf16(q), dev_bf16(k), dev_bf16(v)
    outs = []
    for flag in (32, 16):
        raw.eilev_debug_attn_v1(flag << 1)
        try:
            out = torch.full((batch, sq, D), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert hip.eilev_attention(P(dq), P(dk), P(dv), P(out), batch, heads, sq, sq, hd, D, D, D, hd ** -0.5, 0, None, stream_ptr()) == 0
            torch.cuda.synchronize()
        finally:
            raw.eilev_debug_attn_v1(0)
        outs.append(host(out))
    assert np.isfinite(outs[1]).all()
    assert np.abs(outs[1] - ref).max() <= 1e-2 * np.abs(ref).max()
    assert np.array_equal(outs[0][:, :256], outs[1][:, :256])           # same arithmetic for the 256 patch rows
    assert np.abs(outs[0][:, 256] - outs[1][:, 256]).max() <= 2.0 ** -7 * np.abs(ref[:, 256]).max()  # CLS row: another summation order


def test_attention_frame_default_route_from_512_frames(probes):
    """From 512 frames `eilev_attention` takes the two-group kernel on its own: the same bits as forcing it, the patch rows the same bits
    as the single-tile kernel, and a sample of frames against fp32 softmax attention."""
    hip = raw = probes
    batch, heads, sq, hd = 520, 16, 257, 88  # 32-33 pairs per workgroup
    D = heads * hd
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    qkv = (torch.randn((batch, sq, 3 * D), device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    outs = {}
    for flag in (0, 16, 32):
        raw.eilev_debug_attn_v1(flag << 1)
        try:
            out = torch.full((batch, sq, D), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert hip.eilev_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(qkv.data_ptr() + 2 * D), C.c_void_p(qkv.data_ptr() + 4 * D), P(out),
                                       batch, heads, sq, sq, hd, 3 * D, 3 * D, 3 * D, hd ** -0.5, 0, None, stream_ptr()) == 0
            torch.cuda.synchronize()
        finally:
            raw.eilev_debug_attn_v1(0)
        outs[flag] = out
    assert torch.equal(outs[0], outs[16])
    assert torch.equal(outs[0][:, :256], outs[32][:, :256])
    for b in (0, 259, 519):
        qf, kf, vf = (t.float().view(sq, heads, hd).transpose(0, 1) for t in qkv[b].split(D, dim=-1))
        ref = (torch.softmax(qf @ kf.transpose(-1, -2) * hd ** -0.5, -1) @ vf).transpose(0, 1).reshape(sq, D)
        assert (outs[0][b].float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("batch,heads,sq,skv,causal,masked,rel", [
    (2, 4, 960, 960, 0, False, True),    # flan-t5-xl encoder: L = 960, relative position bias (round 5: the hd = 64 form of the v2 kernel)
    (2, 3, 333, 333, 0, True, True),     # ragged length (partial last key tile and last query tile), left padding through the key mask
    (1, 2, 200, 200, 1, False, True),    # causal + bias (decoder self-attention, teacher-forced)
    (2, 2, 130, 300, 1, True, True),     # causal with offset (sq < skv) + bias + mask
    (2, 4, 256, 512, 0, False, False),   # hd = 64 without a bias (long cross-attention)
    (1, 2, 128, 64, 0, True, False),     # the smallest shapes the route takes
])
def test_attention_hd64_v2_route(hip, batch, heads, sq, skv, causal, masked, rel):
    """hd = 64, >= 128 query rows: attn_prefill_v2_kernel<*, 2, REL> (128-byte LDS rows, XOR-swizzled through the LDS-DMA's source side;
    the head's relative-position table in LDS) against the oracle's eilev_attention_rel (hf T5Attention: no 1 / sqrt(d) factor)."""
    hd = 64
    D = heads * hd
    q = round_bf16(0.35 * det_normal("q64", (batch, sq, D)))
    k = round_bf16(det_normal("k64", (batch, skv, D)))
    v = round_bf16(det_normal("v64", (batch, skv, D)))
    km = None
    if masked:
        km = np.ones((batch, skv), np.int32)
        km[0, :7] = 0
        if batch > 1:
            km[1, :1] = 0
            km[1, skv - 3:] = 0 if not causal else 1
    n = sq + skv - 1
    tab = (0.5 * det_normal("rel64", (heads, n))).astype(np.float32) if rel else None
    ref = np.empty((batch, sq, D), np.float32)
    pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    assert orc.lib().eilev_attention_rel(pp(q), pp(k), pp(v), pp(ref), batch, heads, sq, skv, hd, D, D, D, 1.0, causal, pp(km), pp(tab), n, skv - 1, n, None) == 0
    dq, dk, dv = dev_bf16(q), dev_bf16(k), dev_bf16(v)
    dkm = torch.from_numpy(km).cuda() if masked else None
    dtab = torch.from_numpy(tab).cuda() if rel else None
    out = torch.empty((batch, sq, D), dtype=torch.bfloat16, device="cuda")
    rc = hip.eilev_attention_rel(P(dq), P(dk), P(dv), P(out), batch, heads, sq, skv, hd, D, D, D, 1.0, causal, P(dkm), P(dtab), n, skv - 1, n, stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = host(out)
    valid = np.ones((batch, sq), bool)
    if masked and causal:
        for b in range(batch):
            first = int(np.argmax(km[b] != 0))
            for i in range(sq):
                if i + (skv - sq) < first:
                    valid[b, i] = False
    err = np.abs(got - ref)[valid].max()
    assert err <= 2e-2 * np.abs(ref).max(), err
