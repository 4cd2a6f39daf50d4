#!/usr/bin/env python
"""GPU probe: decode-step linear layers (M rows): achieved weight-streaming bandwidth, bf16 weights and fp8 (e4m3) weights.

    python tools/skinny_probe.py [M]
"""
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
    # fp8 weights (eilev_linear_w8): half the bytes
    from eilev_amd import quant
    qs = [quant.quantize_e4m3_per_channel(w) for w in ws]
    nb = lib.eilev_linear_w8_scratch_bytes(M, n, k)
    scr = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
    for q, sc in qs:
        lib.eilev_linear_w8(P(a), P(q), P(sc), P(b), None, P(o), M, n, k, 0, 0, P(scr), nb, st())
    e0.record()
    for _ in range(reps):
        for q, sc in qs:
            lib.eilev_linear_w8(P(a), P(q), P(sc), P(b), None, P(o), M, n, k, 0, 0, P(scr), nb, st())
    e1.record(); torch.cuda.synchronize()
    us8 = e0.elapsed_time(e1) * 1e3 / (reps * copies)
    print(f"{name:8s} N={n:6d} K={k:6d}: bf16 {us:7.1f} us  {n*k*2/us/1e6:6.2f} TB/s | fp8 {us8:7.1f} us  {n*k/us8/1e6:6.2f} TB/s  ({us/us8:.2f}x)", flush=True)
