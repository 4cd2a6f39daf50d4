"""LayerNorm folded into the linear that consumes it (include/eilev.h ABI 9; VERDICT r1 item 9: "emit row sum / sum-of-squares from the
proj / fc2 residual epilogues so the next LN ... disappears into the qkv / fc1 A-load").

The reference runs nn.LayerNorm then nn.Linear (hf modeling_blip_2.py:390-399).  CPU: the oracle's restatement of the folded algebra
equals its own layernorm + linear.  GPU: every HIP stage against the oracle's, and the folded ViT (3 blocks, also at the ViT-g
widths) against the unfolded one and against the fp32 oracle — the folded path must be as close to fp32 as the unfolded one is."""
import ctypes as C

import numpy as np
import pytest
import torch

from eilev_amd.synth import round_bf16, synth_pixels
from oracle import runner as orc

pp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
f32 = lambda a: np.ascontiguousarray(a, np.float32)


def _rand(shape, seed, scale=1.0):
    return round_bf16((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _oracle_fold(w, gamma, beta, bias):
    n, k = w.shape
    wf, cs, bf = np.empty((n, k), np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
    assert orc.lib().eilev_fold_layernorm(pp(w), pp(gamma), pp(beta), pp(bias), n, k, pp(wf), pp(cs), pp(bf), None) == 0
    return wf, cs, bf


def _oracle_ln_linear(x, gamma, beta, w, bias, eps, act):
    m, k = x.shape
    n = w.shape[0]
    ln, out = np.empty((m, k), np.float32), np.empty((m, n), np.float32)
    lib = orc.lib()
    assert lib.eilev_layernorm(pp(x), pp(gamma), pp(beta), pp(ln), m, k, C.c_float(eps), None) == 0
    assert lib.eilev_linear(pp(ln), pp(w), pp(bias), None, pp(out), m, n, k, act, 0, None) == 0
    return out


def _oracle_rows(x, eps):
    """(rstd, -mean) per row through the oracle's own stats + finalize of an identity-free producer: x = 0 . w + x."""
    m, n = x.shape
    a, w = np.zeros((m, 8), np.float32), np.zeros((n, 8), np.float32)
    c, st = np.empty((m, n), np.float32), np.empty(((n + 63) // 64, m, 2), np.float32)
    rows = np.empty((m, 2), np.float32)
    lib = orc.lib()
    assert lib.eilev_linear_stats(pp(a), pp(w), None, pp(x), pp(c), m, n, 8, pp(st), None) == 0
    assert np.array_equal(c, x)
    assert lib.eilev_ln_finalize(pp(st), m, n, C.c_float(eps), pp(rows), None) == 0
    return rows


def test_oracle_folded_algebra_equals_layernorm_then_linear():
    m, k, n, eps = 70, 192, 136, 1e-6
    x = _rand((m, k), 0, 2.0) + 1.5  # a mean that matters
    gamma = np.exp2(np.random.default_rng(1).integers(-2, 2, k)).astype(np.float32)  # powers of two: gamma (.) W stays bf16-exact
    beta, w, bias = _rand(k, 2, 0.5), _rand((n, k), 3, k ** -0.5), _rand(n, 4, 0.5)
    wf, cs, bf = _oracle_fold(w, gamma, beta, bias)
    assert np.array_equal(wf, w * gamma)
    rows = _oracle_rows(x, eps)
    mean = x.astype(np.float64).mean(1)
    var = x.astype(np.float64).var(1)
    assert np.allclose(rows[:, 0], 1 / np.sqrt(var + eps), rtol=1e-5) and np.allclose(rows[:, 1], -mean, rtol=1e-5, atol=1e-6)
    for act in (0, 1):
        ref = _oracle_ln_linear(x, gamma, beta, w, bias, eps, act)
        out = np.empty((m, n), np.float32)
        assert orc.lib().eilev_linear_lnfold(pp(x), pp(wf), pp(bf), pp(cs), pp(rows), pp(out), m, n, k, act, None) == 0
        # the only difference left is the bf16 rounding of the folded bias
        assert np.abs(out - ref).max() <= 2.0 ** -8 * np.abs(bf).max() + 1e-4


# ---- GPU ------------------------------------------------------------------------------------------------------------------
def _gpu():
    from eilev_amd import abi

    lib = abi.load_hip()
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    dev = lambda a: torch.from_numpy(f32(a)).cuda().to(torch.bfloat16).contiguous()
    return abi, lib, st, P, dev


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(136, 192), (4224, 1408)])
def test_hip_fold_layernorm_vs_oracle(n, k):
    abi, lib, st, P, dev = _gpu()
    w, gamma, beta, bias = _rand((n, k), 10, k ** -0.5), round_bf16(_rand(k, 11, 0.3) + 1.0), _rand(k, 12, 0.5), _rand(n, 13, 0.5)
    wf_r, cs_r, bf_r = _oracle_fold(w, gamma, beta, bias)
    wf, bf = torch.empty((n, k), dtype=torch.bfloat16, device="cuda"), torch.empty(n, dtype=torch.bfloat16, device="cuda")
    cs = torch.empty(n, dtype=torch.float32, device="cuda")
    wd, gd, bd, bid = dev(w), dev(gamma), dev(beta), dev(bias)  # (named: the pointers must outlive the call)
    abi.check(lib.eilev_fold_layernorm(P(wd), P(gd), P(bd), P(bid), n, k, P(wf), P(cs), P(bf), st()), "fold")
    torch.cuda.synchronize()
    assert np.array_equal(wf.float().cpu().numpy(), wf_r)  # one fp32 product, one RNE: the same bits
    assert np.allclose(cs.cpu().numpy(), cs_r, rtol=1e-5, atol=1e-5)
    assert np.abs(bf.float().cpu().numpy() - bf_r).max() <= 2.0 ** -7 * np.abs(bf_r).max()  # fp32 vs double sum, then one bf16 ulp


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k", [(300, 192, 384), (1000, 1408, 1408), (777, 1408, 6144)])
def test_hip_residual_linear_emits_row_statistics(m, n, k):
    abi, lib, st, P, dev = _gpu()
    a, w, b, r = _rand((m, k), 20), _rand((n, k), 21, k ** -0.5), _rand(n, 22, 0.5), round_bf16(_rand((m, n), 23, 3.0) + 0.7)
    slots = (n + 63) // 64
    c_r, st_r = np.empty((m, n), np.float32), np.empty((slots, m, 2), np.float32)
    assert orc.lib().eilev_linear_stats(pp(a), pp(w), pp(b), pp(r), pp(c_r), m, n, k, pp(st_r), None) == 0
    c = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    stats = torch.full((slots, m, 2), float("nan"), dtype=torch.float32, device="cuda")
    ad, wd, bd, rd = dev(a), dev(w), dev(b), dev(r)
    abi.check(lib.eilev_linear_stats(P(ad), P(wd), P(bd), P(rd), P(c), m, n, k, P(stats), st()), "linear_stats")
    torch.cuda.synchronize()
    got_c, got = c.float().cpu().numpy(), stats.cpu().numpy()
    assert np.abs(got_c - c_r).max() <= 2.0 ** -7 * np.abs(c_r).max()
    assert np.isfinite(got).all()  # every (slot, row) written, also by the half tile of the last column and the partial last row tile
    assert np.allclose(got[..., 0], st_r[..., 0], rtol=1e-4, atol=1e-3 * np.abs(st_r[..., 0]).max())
    assert np.allclose(got[..., 1], st_r[..., 1], rtol=1e-4, atol=1e-4 * np.abs(st_r[..., 1]).max())
    # finalize: fixed slot order -> the same bits on every run, and the oracle's numbers
    rows = torch.empty((m, 2), dtype=torch.float32, device="cuda")
    rows2 = torch.empty_like(rows)
    for dst in (rows, rows2):
        abi.check(lib.eilev_ln_finalize(P(stats), m, n, C.c_float(1e-6), P(dst), st()), "finalize")
    torch.cuda.synchronize()
    assert torch.equal(rows, rows2)
    rows_r = np.empty((m, 2), np.float32)
    assert orc.lib().eilev_ln_finalize(pp(st_r), m, n, C.c_float(1e-6), pp(rows_r), None) == 0
    assert np.allclose(rows.cpu().numpy(), rows_r, rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k,act", [(300, 384, 192, 0), (300, 384, 192, 1), (1000, 4224, 1408, 0), (1300, 6144, 1408, 1), (257, 136, 64, 0)])
def test_hip_linear_lnfold_vs_oracle_and_vs_layernorm_linear(m, n, k, act):
    abi, lib, st, P, dev = _gpu()
    eps = 1e-6
    x = round_bf16(_rand((m, k), 30, 2.0) + 0.8)
    gamma, beta = round_bf16(_rand(k, 31, 0.3) + 1.0), _rand(k, 32, 0.5)
    w, bias = _rand((n, k), 33, k ** -0.5), _rand(n, 34, 0.5)
    wf, cs, bf = _oracle_fold(w, gamma, beta, bias)
    rows = _oracle_rows(x, eps)
    ref = np.empty((m, n), np.float32)
    assert orc.lib().eilev_linear_lnfold(pp(x), pp(wf), pp(bf), pp(cs), pp(rows), pp(ref), m, n, k, act, None) == 0
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    cs_d, rows_d = torch.from_numpy(cs).cuda(), torch.from_numpy(rows).cuda()
    xd, wfd, bfd = dev(x), dev(wf), dev(bf)
    abi.check(lib.eilev_linear_lnfold(P(xd), P(wfd), P(bfd), P(cs_d), P(rows_d), P(out), m, n, k, act, st()), "lnfold")
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2.0 ** -7 * scale + 1e-3  # same operands: fp32 summation order + the bf16 output (+ the GELU polynomial)
    # and against what the reference computes: layernorm, then linear (fp32): the distance is the bf16 rounding of gamma (.) W
    plain = _oracle_ln_linear(x, gamma, beta, w, bias, eps, act)
    rel = np.sqrt(((got - plain) ** 2).mean() / (plain ** 2).mean())
    assert rel < 6e-3, rel


def _vit_three_ways(cfg_name, frames):
    from hip_utils import models, rel_rms

    cfg, oracle, eng = models(cfg_name)
    eng.ensure_vit_fold()  # (built lazily by the first launch of >= 24576 token rows; these fixtures are smaller)
    assert bool(eng.pack.vit.layers_fold)
    px = synth_pixels(1, frames, cfg.vision_config.image_size)
    ref = oracle.vit(px)
    pxd = torch.from_numpy(px).cuda()
    try:  # EilevVitWeights.fold_min_rows (ABI 16): 1 = every launch, negative = never, 0 = the library's default
        eng.pack.vit.fold_min_rows = 1
        folded = eng.vit(pxd).float().cpu().numpy()
        folded2 = eng.vit(pxd).float().cpu().numpy()
        eng.pack.vit.fold_min_rows = -1
        plain = eng.vit(pxd).float().cpu().numpy()
    finally:
        eng.pack.vit.fold_min_rows = 0
    return ref, folded, folded2, plain, rel_rms


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,frames", [("fold_3l", 3), ("real_vit_3l", 2)])
def test_folded_vit_is_as_close_to_fp32_as_the_unfolded_one(cfg_name, frames):
    from hip_utils import record_parity

    ref, folded, folded2, plain, rel_rms = _vit_three_ways(cfg_name, frames)
    assert np.array_equal(folded, folded2)  # deterministic: no atomics in the statistics
    assert not np.array_equal(folded, plain)  # the knob really switches paths
    e_fold, e_plain = rel_rms(folded, ref), rel_rms(plain, ref)
    record_parity(f"ln_fold_vit[{cfg_name}]", folded_vs_fp32=e_fold, unfolded_vs_fp32=e_plain, folded_vs_unfolded=rel_rms(folded, plain))
    assert e_fold <= 1.25 * e_plain + 5e-4, (e_fold, e_plain)


@pytest.mark.gpu
def test_hip_linear_lnfold_with_a_large_row_mean_and_an_outlier_channel():
    """The statistics are E[x^2] - mean^2 over fp32 partial sums and the mean enters as two bf16 pieces of a rank-1 MFMA slice: rows
    whose mean is 15 standard deviations away from 0 and a channel 100x larger than the rest (the "massive activation" pattern of ViT
    residual streams) must still come out at the bf16 noise of layernorm-then-linear."""
    abi, lib, st, P, dev = _gpu()
    m, n, k, eps = 520, 1408, 1408, 1e-6
    x = _rand((m, k), 40, 2.0) + 30.0
    x[:, 77] *= 100.0
    x = round_bf16(x)
    gamma, beta = round_bf16(_rand(k, 41, 0.3) + 1.0), _rand(k, 42, 0.5)
    w, bias = _rand((n, k), 43, k ** -0.5), _rand(n, 44, 0.5)
    # the row statistics through the HIP producer itself: x = 0 . w0 + x
    xd = dev(x)
    zero_a, zero_w = torch.zeros((m, 64), dtype=torch.bfloat16, device="cuda"), torch.zeros((k, 64), dtype=torch.bfloat16, device="cuda")
    c = torch.empty((m, k), dtype=torch.bfloat16, device="cuda")
    stats = torch.empty(((k + 63) // 64, m, 2), dtype=torch.float32, device="cuda")
    rows = torch.empty((m, 2), dtype=torch.float32, device="cuda")
    abi.check(lib.eilev_linear_stats(P(zero_a), P(zero_w), None, P(xd), P(c), m, k, 64, P(stats), st()), "stats")
    abi.check(lib.eilev_ln_finalize(P(stats), m, k, C.c_float(eps), P(rows), st()), "finalize")
    torch.cuda.synchronize()
    assert torch.equal(c, xd)
    x64 = x.astype(np.float64)
    assert np.allclose(rows[:, 1].cpu().numpy(), -x64.mean(1), rtol=1e-5)
    assert np.allclose(rows[:, 0].cpu().numpy(), 1 / np.sqrt(x64.var(1) + eps), rtol=2e-4)
    wf, cs, bf = _oracle_fold(w, gamma, beta, bias)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    wfd, bfd, csd = dev(wf), dev(bf), torch.from_numpy(cs).cuda()
    abi.check(lib.eilev_linear_lnfold(P(xd), P(wfd), P(bfd), P(csd), P(rows), P(out), m, n, k, 0, st()), "lnfold")
    torch.cuda.synchronize()
    plain = _oracle_ln_linear(x, gamma, beta, w, bias, eps, 0)
    got = out.float().cpu().numpy()
    rel = np.sqrt(((got - plain) ** 2).mean() / (plain ** 2).mean())
    assert rel < 6e-3, rel
