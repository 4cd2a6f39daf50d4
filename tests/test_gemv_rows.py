"""Small-batch decode linears (eilev_amd/csrc/gemv.hip, include/eilev.h eilev_linear_rows): m <= 8 rows as row dot products with the
LayerNorm that feeds the linear in the same launch — what hf runs as nn.LayerNorm -> nn.Linear inside OPTDecoderLayer for one decode step
(modeling_opt.py:226-247) via ref:eilev/model/v2.py:318-322.

CPU: the oracle's entry equals its own layernorm + linear.  GPU: the HIP kernel against the oracle at the decode shapes of OPT-2.7B / 6.7B
(K = 2560 / 10240 / 4096 / 16384, N incl. the 50272-wide lm_head) for every m in 1..8, fp32 and bf16 outputs, bias / ReLU / residual; the
tolerance is the bf16 rounding of the normalised rows (the reference's own bf16 run rounds there too) + the bf16 output."""
import ctypes as C

import numpy as np
import pytest
import torch

from eilev_amd.synth import round_bf16
from oracle import runner as orc

pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)


def _rand(shape, seed, scale=1.0):
    return round_bf16((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _oracle(x, gamma, beta, w, bias, resid, epi, eps=1e-5):
    m, k = x.shape
    n = w.shape[0]
    out = np.empty((m, n), np.float32)
    assert orc.lib().eilev_linear_rows(pp(x), pp(gamma), pp(beta), C.c_float(eps), pp(w), pp(bias), pp(resid), pp(out), m, n, k, epi, 1, None) == 0
    return out


def test_oracle_linear_rows_is_layernorm_then_linear():
    m, k, n = 3, 512, 40
    x, gamma, beta = _rand((m, k), 0, 2.0) + 0.5, _rand(k, 1, 0.3) + 1.0, _rand(k, 2, 0.2)
    w, bias, resid = _rand((n, k), 3, k ** -0.5), _rand(n, 4), _rand((m, n), 5)
    got = _oracle(x, gamma, beta, w, bias, resid, 2)
    t = torch.from_numpy
    ref = torch.relu(torch.nn.functional.linear(torch.nn.functional.layer_norm(t(x), (k,), t(gamma), t(beta), 1e-5), t(w), t(bias))) + t(resid)
    assert np.allclose(got, ref.numpy(), rtol=1e-5, atol=1e-5)
    plain = _oracle(x, None, None, w, None, None, 0)
    assert np.allclose(plain, x @ w.T, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("n,k,ln,epi,resid,f32", [
    (7680, 2560, True, 0, False, False),    # self_attn_layer_norm + q|k|v
    (2560, 2560, False, 0, True, False),    # out_proj + residual
    (10240, 2560, True, 2, False, False),   # final_layer_norm + fc1 + ReLU
    (2560, 10240, False, 0, True, False),   # fc2 + residual: four K blocks of 2560
    (50272, 2560, True, 0, False, True),    # decoder final_layer_norm + lm_head, fp32 logits
    (4096, 16384, False, 0, True, False),   # OPT-6.7B fc2: four K blocks of 4096
    (1001, 512, True, 2, True, False),      # ragged N (last wave partly out of range), minimum K
])
def test_hip_linear_rows_vs_oracle(m, n, k, ln, epi, resid, f32):
    from eilev_amd import abi

    lib = abi.load_hip()
    x = _rand((m, k), 10 + m, 1.5) + 0.3
    gamma, beta = (_rand(k, 11, 0.3) + 1.0, _rand(k, 12, 0.2)) if ln else (None, None)
    w, bias = _rand((n, k), 13, k ** -0.5), _rand(n, 14, 0.5)
    r = _rand((m, n), 15, 2.0) if resid else None
    ref = _oracle(x, gamma, beta, w, bias, r, epi)
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda().to(torch.bfloat16).contiguous()
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    xd, gd, bd, wd, biasd, rd = dev(x), dev(gamma), dev(beta), dev(w), dev(bias), dev(r)
    out = torch.full((m, n), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    if m * k * 2 > 150 * 1024:  # the rows do not fit the LDS staging: refused, loudly (the decode step keeps the MFMA kernels there)
        rc = lib.eilev_linear_rows(P(xd), P(gd), P(bd), C.c_float(1e-5), P(wd), P(biasd), P(rd), P(out), m, n, k, epi, int(f32),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == -2  # EILEV_E_UNSUPPORTED
        return
    abi.check(lib.eilev_linear_rows(P(xd), P(gd), P(bd), C.c_float(1e-5), P(wd), P(biasd), P(rd), P(out), m, n, k, epi, int(f32),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "eilev_linear_rows")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    scale = np.abs(ref).max()
    # LN rows rounded to bf16 (2^-9 relative per element, averaging over K) + (bf16 | fp32) output
    tol = (6e-3 if ln else 2e-4) * scale + (0.0 if f32 else 2.0 ** -8 * scale)
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)
    rel = np.sqrt(((got - ref) ** 2).mean() / (ref ** 2).mean())
    assert rel <= (3e-3 if ln else 1e-4) + (0.0 if f32 else 2e-3)
    # rows are independent of how many share the launch: row 0 alone gives the same bits
    if m > 1:
        one = torch.empty((1, n), dtype=out.dtype, device="cuda")
        abi.check(lib.eilev_linear_rows(P(xd), P(gd), P(bd), C.c_float(1e-5), P(wd), P(biasd), P(rd), P(one), 1, n, k, epi, int(f32),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "eilev_linear_rows")
        torch.cuda.synchronize()
        assert torch.equal(one[0], out[0])
