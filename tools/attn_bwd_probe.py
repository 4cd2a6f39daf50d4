#!/usr/bin/env python
"""Time eilev_attention (forward) and eilev_attention_bwd at the training-graph shapes."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eilev_amd import abi  # noqa: E402

SHAPES = {  # name: (batch, heads, sq, skv, hd, causal)
    "opt27_b1": (1, 32, 960, 960, 80, 1),
    "opt27_b8": (8, 32, 960, 960, 80, 1),
    "qf_self": (17, 12, 32, 32, 64, 0),
    "qf_cross": (17, 12, 32, 2056, 64, 0),
}


def main():
    hip = abi.load_hip()
    P = lambda t: C.c_void_p(t.data_ptr())
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, (b, h, sq, skv, hd, causal) in SHAPES.items():
        D = h * hd
        q, do = (torch.randn(b, sq, D, device="cuda").bfloat16() for _ in range(2))
        k, v = (torch.randn(b, skv, D, device="cuda").bfloat16() for _ in range(2))
        o, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(2, b, h, sq, device="cuda")
        fwd = lambda: hip.eilev_attention(P(q), P(k), P(v), P(o), b, h, sq, skv, hd, D, D, D, hd ** -0.5, causal, None, st())
        bwd = lambda: hip.eilev_attention_bwd(P(q), P(k), P(v), P(o), P(do), P(dq), P(dk), P(dv), P(ws), b, h, sq, skv, hd, D, D, D, D, D, D,
                                              hd ** -0.5, causal, None, st())
        res = []
        for f in (fwd, bwd):
            for _ in range(3):
                assert f() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1e3)
        flops = 4.0 * b * h * sq * skv * hd * (0.5 if causal else 1.0)
        print(f"{name:10s} fwd {res[0]:8.1f} us ({flops / res[0] / 1e6:6.1f} TF/s)   bwd {res[1]:8.1f} us ({2.5 * flops / res[1] / 1e6:6.1f} TF/s)   bwd/fwd {res[1] / res[0]:.1f}")


if __name__ == "__main__":
    main()
