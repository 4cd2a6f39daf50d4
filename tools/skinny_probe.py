#!/usr/bin/env python
"""GPU probe: decode-step (M = 8) linear layers: achieved weight-streaming bandwidth."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd import abi

lib = abi.load_hip()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, n, k in [("qkv", 7680, 2560), ("out", 2560, 2560), ("fc1", 10240, 2560), ("fc2", 2560, 10240), ("lm_head", 50272, 2560)]:
    # rotate over several weight copies so the 256 MiB infinity cache cannot serve the stream
    copies = max(2, int(600e6 // (n * k * 2)))
    ws = [(torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16) for _ in range(copies)]
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    for w in ws:
        lib.eilev_linear(P(a), P(w), P(b), None, P(o), M, n, k, 0, 0, st())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 4
    for _ in range(reps):
        for w in ws:
            lib.eilev_linear(P(a), P(w), P(b), None, P(o), M, n, k, 0, 0, st())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * copies)
    print(f"{name:8s} N={n:6d} K={k:6d}: {us:7.1f} us  {n*k*2/us/1e6:6.2f} TB/s", flush=True)
