#!/usr/bin/env python
"""GPU probe (yardstick only, never on the product path): the vendor library's bf16 GEMM rate (torch -> hipBLASLt/rocBLAS)
on the ViT/OPT shapes, next to eilev_linear, so the remaining headroom of the hand-written kernel is known."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd import abi

lib = abi.load_hip()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("PROBE_M", 279616))
SHAPES = {"fc1": (M, 6144, 1408), "fc2": (M, 1408, 6144), "qkv": (M, 4224, 1408), "proj": (M, 1408, 1408),
          "opt_fc1": (30720, 10240, 2560), "opt_qkv": (30720, 7680, 2560), "opt_fc2": (30720, 2560, 10240)}


def timeit(fn, n=5, rounds=None):
    rounds = rounds or ROUNDS
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


if os.environ.get("YARD_SHAPES"):
    SHAPES = {k: SHAPES[k] for k in os.environ["YARD_SHAPES"].split(",")}
ROUNDS = int(os.environ.get("YARD_ROUNDS", 4))
for name, (m, n, k) in SHAPES.items():
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    f_lib = lambda: torch.nn.functional.linear(a, w, b)
    f_mm = lambda: torch.matmul(a, w.t(), out=o)
    f_own = lambda: lib.eilev_linear(P(a), P(w), P(b), None, P(o), m, n, k, 0, 0, st())
    for f in (f_lib, f_mm, f_own):
        f()
    torch.cuda.synchronize()
    fl = 2 * m * n * k / 1e9
    tl, tm, to = timeit(f_lib), timeit(f_mm), timeit(f_own)
    print(f"{name:8s} M={m} N={n} K={k}: vendor linear+bias {fl/tl:7.1f} TF/s | vendor matmul {fl/tm:7.1f} TF/s | eilev_linear(+bias) {fl/to:7.1f} TF/s", flush=True)
