"""fp8 (e4m3) weight-only linear: format (oracle encode / decode vs torch's float8_e4m3fn cast), quantiser, and on the GPU the
HIP kernels (byte-streaming skinny kernel for M <= 32, expand + bf16 kernels above) against the oracle on the same bytes."""
import ctypes as C

import numpy as np
import pytest
import torch

from eilev_amd import quant
from oracle import runner as orc


def _olib():
    L = orc.lib()
    L.eilev_e4m3_decode.restype = C.c_float
    L.eilev_e4m3_decode.argtypes = [C.c_uint8]
    L.eilev_e4m3_encode.restype = C.c_uint8
    L.eilev_e4m3_encode.argtypes = [C.c_float]
    return L


def test_e4m3_decode_matches_torch_for_all_256_bytes():
    L = _olib()
    b = torch.arange(256, dtype=torch.uint8)
    ref = b.view(torch.float8_e4m3fn).to(torch.float32).numpy()
    got = np.array([L.eilev_e4m3_decode(int(i)) for i in range(256)], np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(nan, np.isnan(got)) and np.array_equal(ref[~nan], got[~nan])


def test_e4m3_encode_matches_torch_cast():
    L = _olib()
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.randn(20000, generator=g) * 100, torch.randn(20000, generator=g), torch.randn(20000, generator=g) * 0.01,
                   torch.tensor([0.0, -0.0, 448.0, -448.0, 447.9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10, 3 * 2.0 ** -10, 0.0625, 0.017578125])])
    x = x.clamp(-448, 448)
    ref = x.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = np.array([L.eilev_e4m3_encode(float(v)) for v in x.numpy()], np.uint8)
    # +0 / -0 of tiny inputs: compare decoded values and signs of non-zeros
    dec = lambda a: torch.from_numpy(a.copy()).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    assert np.array_equal(dec(ref), dec(got))


def test_quantiser_round_trip_error_and_scale():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 256, generator=g) * torch.rand(64, 1, generator=g) * 3
    w[5] = 0
    q, s = quant.quantize_e4m3_per_channel(w)
    assert q.dtype == torch.uint8 and s.dtype == torch.float32 and s[5] == 1
    dq = quant.dequantize(q, s)
    # e4m3 has 3 mantissa bits: relative error <= 2^-4 for normal values, absolute <= scale * 2^-10 near zero
    assert torch.all((dq - w).abs() <= w.abs() * 2.0 ** -4 + s[:, None] * 2.0 ** -10 + 1e-12)
    assert torch.allclose(dq.abs().amax(1)[s != 1], w.abs().amax(1)[s != 1])  # the row maximum is represented exactly (448 * scale)


def _case(m, n, k, epi, bias, resid, out_f32=False, seed=0):
    from eilev_amd import abi

    g = torch.Generator().manual_seed(seed)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    q, s = quant.quantize_e4m3_per_channel(w)
    b = (0.5 * torch.randn(n, generator=g)).to(torch.bfloat16) if bias else None
    r = torch.randn(m, n, generator=g).to(torch.bfloat16) if resid else None
    f = lambda t: None if t is None else np.ascontiguousarray(t.to(torch.float32).numpy())
    ref = np.empty((m, n), np.float32)
    pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    af, bf_, rf, qn, sn = f(a), f(b), f(r), q.numpy(), s.numpy()
    assert orc.lib().eilev_linear_w8(pp(af), pp(qn), pp(sn), pp(bf_), pp(rf), pp(ref), m, n, k, epi, 0, None, 0, None) == 0
    d = lambda t: None if t is None else t.cuda()
    got = quant.linear_w8(d(a), d(q), d(s), d(b), d(r), epilogue=epi, out_dtype=torch.float32 if out_f32 else None, lib=abi.load_hip())
    got = got.to(torch.float32).cpu().numpy()
    err = np.abs(got - ref).max()
    tol = (2e-4 if out_f32 else 1e-2) * np.abs(ref).max()
    assert err <= tol, (m, n, k, epi, err, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k", [(1, 160, 256), (8, 2560, 2560), (17, 7680, 2560), (32, 2560, 10240), (32, 1000, 512), (3, 50272, 2560)])
def test_linear_w8_decode_shapes(m, n, k):
    _case(m, n, k, 0, bias=True, resid=True)


@pytest.mark.gpu
def test_linear_w8_fp32_logits_and_activations():
    _case(8, 1024, 2560, 0, bias=False, resid=False, out_f32=True)
    _case(32, 2048, 2560, 2, bias=True, resid=False)
    _case(16, 768, 768, 1, bias=True, resid=False)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k,epi,resid", [(300, 200, 128, 0, True), (1000, 1408, 1408, 1, False), (960, 2560, 2560, 2, False),
                                              (8229, 2048, 320, 0, True), (40, 256, 192, 0, False)])
def test_linear_w8_prefill_shapes_expand_path(m, n, k, epi, resid):
    _case(m, n, k, epi, bias=True, resid=resid)
