"""-m gpu: gemm_a4_kernel (eilev_amd/csrc/gemm_a4.h) — one wave per SIMD, 128 x 128 per wave, the K loop hand-scheduled inline asm
(gen_a4_loop.py) — forced through the probe switch of the library on shapes far below its production threshold, so that full tiles,
half tiles (N % 256 = 128), ragged row tiles, every epilogue (bias / GELU / ReLU / residual / LayerNorm-statistics producer) and the
shortest K loops (3 and 4 K-steps: peeled first step + both tail steps, with and without a steady-state iteration) are all exercised.

Parity: against the fp32 CPU oracle's `eilev_linear` (hf nn.Linear of Blip2MLP / Blip2Attention via ref:eilev/model/v2.py:59-64) at the
bf16 output rounding, and BIT-identical to the library's default kernels for the same call wherever those share the K order (no GELU:
the small-tile kernels evaluate the degree-12 GELU, the persistent ones the degree-8 form)."""
import ctypes as C

import numpy as np
import pytest
import torch

from eilev_amd.synth import round_bf16
from oracle import runner as orc

pytestmark = pytest.mark.gpu
pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
FORCE_A4 = 10 << 4


def _rand(shape, seed, scale=1.0):
    return round_bf16((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _gpu():
    from eilev_amd import abi

    lib = abi.load_hip()
    raw = C.CDLL(abi.HIP_LIB_PATH)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda().to(torch.bfloat16).contiguous()
    return abi, lib, raw, st, P, dev


@pytest.mark.parametrize("m,n,k,epi,resid", [
    (300, 256, 192, 0, False),      # one full column tile, 3 K-steps (no steady-state iteration), ragged second row tile
    (520, 128, 256, 2, False),      # a single HALF tile column, 4 K-steps, ReLU
    (1000, 1408, 1408, 0, True),    # proj: 5 full + 1 half column tile, residual
    (777, 1408, 6144, 0, True),     # fc2: long K
    (2000, 6144, 1408, 1, False),   # fc1 + GELU
    (700, 4224, 1408, 0, False),    # qkv: 16 full + 1 half
    (257, 384, 320, 0, True),       # 1 full + 1 half, 5 K-steps
    (33, 512, 192, 0, False),       # fewer rows than one wave's 128
])
def test_a4_linear_vs_oracle_and_default_kernels(m, n, k, epi, resid):
    abi, lib, raw, st, P, dev = _gpu()
    a, w, b = _rand((m, k), 1), _rand((n, k), 2, k ** -0.5), _rand(n, 3, 0.5)
    r = _rand((m, n), 4, 2.0) if resid else None
    ref = np.empty((m, n), np.float32)
    assert orc.lib().eilev_linear(pp(a), pp(w), pp(b), pp(r), pp(ref), m, n, k, epi, 0, None) == 0
    ad, wd, bd, rd = dev(a), dev(w), dev(b), dev(r)
    outs = []
    try:
        for flags in (FORCE_A4, 0):
            raw.eilev_debug_gemm_flags(flags)
            o = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
            abi.check(lib.eilev_linear(P(ad), P(wd), P(bd), P(rd), P(o), m, n, k, epi, 0, st()), "eilev_linear")
            torch.cuda.synchronize()
            outs.append(o.float().cpu().numpy())
    finally:
        raw.eilev_debug_gemm_flags(0)
    got, dflt = outs
    assert np.isfinite(got).all()  # every cell written (half tiles, ragged rows)
    assert np.abs(got - ref).max() <= 2.0 ** -7 * np.abs(ref).max() + (2e-4 if epi == 1 else 0.0)
    if epi != 1:
        assert np.array_equal(got, dflt)


@pytest.mark.parametrize("m,n,k", [(300, 256, 192), (1000, 1408, 1408), (777, 1408, 6144), (130, 128, 256)])
def test_a4_statistics_producer_vs_oracle(m, n, k):
    """eilev_linear_stats on the a4 kernel: C = A . W^T + bias + residual and, per 64-column slot and row, (sum, sum of squares) of the
    fp32 values it rounds — every (slot, row) written, rows past M untouched, finalize identical to the default kernel's."""
    abi, lib, raw, st, P, dev = _gpu()
    a, w, b, r = _rand((m, k), 20), _rand((n, k), 21, k ** -0.5), _rand(n, 22, 0.5), round_bf16(_rand((m, n), 23, 3.0) + 0.7)
    slots = (n + 63) // 64
    c_r, st_r = np.empty((m, n), np.float32), np.empty((slots, m, 2), np.float32)
    assert orc.lib().eilev_linear_stats(pp(a), pp(w), pp(b), pp(r), pp(c_r), m, n, k, pp(st_r), None) == 0
    ad, wd, bd, rd = dev(a), dev(w), dev(b), dev(r)
    res = []
    try:
        for flags in (FORCE_A4, 0):
            raw.eilev_debug_gemm_flags(flags)
            c = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
            flat = torch.full((slots * m * 2 + 64,), float("nan"), dtype=torch.float32, device="cuda")  # 64 guard floats behind the last slot
            abi.check(lib.eilev_linear_stats(P(ad), P(wd), P(bd), P(rd), P(c), m, n, k, P(flat), st()), "linear_stats")
            torch.cuda.synchronize()
            assert bool(torch.isnan(flat[slots * m * 2:]).all())  # rows past M of the last slot are dropped, not written
            res.append((c.float().cpu().numpy(), flat[: slots * m * 2].view(slots, m, 2).cpu().numpy()))
    finally:
        raw.eilev_debug_gemm_flags(0)
    (got_c, got), (dc, dstat) = res
    assert np.abs(got_c - c_r).max() <= 2.0 ** -7 * np.abs(c_r).max()
    assert np.array_equal(got_c, dc)
    assert np.isfinite(got).all()  # every (slot, row) written
    assert np.allclose(got[..., 0], st_r[..., 0], rtol=1e-4, atol=1e-3 * np.abs(st_r[..., 0]).max())
    assert np.allclose(got[..., 1], st_r[..., 1], rtol=1e-4, atol=1e-4 * np.abs(st_r[..., 1]).max())
    assert np.allclose(got, dstat, rtol=1e-5, atol=1e-4 * np.abs(st_r[..., 1]).max())
