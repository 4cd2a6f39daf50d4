"""CPU: the polynomial GELU of the persistent GEMM's fc1 epilogue (csrc/common.h gelu_erf_n, round 6: x * Phi(x) with Phi = 1/2 + xc q(xc^2)).

The coefficients are read from the header and the kernel's arithmetic is replayed in fp32 (same operation order: clamp, square, Horner in
s, one FMA, one product) against the exact erf form hf's `Blip2MLP` evaluates (ACT2FN["gelu"]): max abs error <= 2e-4, i.e. an order of
magnitude inside the bf16 rounding of the pre-activation the reference's own bf16 run applies GELU to; exact identity / zero tails."""
import os
import re

import numpy as np


def _header():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return open(os.path.join(root, "eilev_amd", "csrc", "common.h")).read()


def _phi_form():
    h = _header()
    c = float(re.search(r"#define EILEV_GELU_PHI_C ([0-9.]+)f", h).group(1))
    body = re.search(r"#define EILEV_GELU_PHI_COEFFS\s*\\\s*\n\s*\{([^}]*)\}", h).group(1)
    q = np.array([float(t.strip().rstrip("f")) for t in body.split(",")], np.float32)
    assert len(q) == 7
    return np.float32(c), q


def _gelu_kernel(x, c, q):
    x = x.astype(np.float32)
    xc = np.clip(x, -c, c).astype(np.float32)
    s = (xc * xc).astype(np.float32)
    p = np.full_like(s, q[6])
    for k in range(5, -1, -1):
        p = (p * s + q[k]).astype(np.float32)
    return (x * (xc * p + np.float32(0.5)).astype(np.float32)).astype(np.float32)


def test_phi_form_error_against_exact_erf_gelu():
    from math import erf, sqrt

    c, q = _phi_form()
    x = np.linspace(-12.0, 12.0, 480001)
    exact = np.array([0.5 * v * (1.0 + erf(v / sqrt(2.0))) for v in x])
    got = _gelu_kernel(x, c, q).astype(np.float64)
    err = np.abs(got - exact)
    assert err.max() <= 2.0e-4, (err.max(), x[err.argmax()])
    big = np.abs(exact) > 0.05
    assert (err[big] / np.abs(exact[big])).max() <= 3.0e-3


def test_phi_form_tails_are_identity_and_zero():
    c, q = _phi_form()
    x = np.array([4.0, 7.5, 100.0, 3.0e4], np.float32)
    assert np.allclose(_gelu_kernel(x, c, q), x, rtol=1e-6, atol=0)
    y = _gelu_kernel(-x, c, q)
    assert np.all(np.abs(y) <= 1e-6 * x + 1e-6), y  # the pinned endpoint: Phi(-c) = 0 up to fp32 rounding
    assert _gelu_kernel(np.zeros(1, np.float32), c, q)[0] == 0.0


def test_the_form_is_the_default():
    assert re.search(r"#define EILEV_GELU_PHI 1\b", _header())
