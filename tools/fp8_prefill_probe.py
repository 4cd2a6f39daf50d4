#!/usr/bin/env python
"""GPU probe (BASELINE configs[4], OPT-6.7B prefill shapes, M = 32 samples x 960 tokens): the four linears as bf16 GEMMs (eilev_linear) vs
per-token e4m3 quantisation + fp8 MFMA (eilev_quant_rows_e4m3 + eilev_linear_a8w8).  Same persistent kernel, one fp8 MFMA where two bf16 were."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eilev_amd import abi
lib = abi.load_hip()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("PROBE_M", 32 * 960))
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)
tot = {"bf16": 0.0, "fp8": 0.0, "quant": 0.0}
for name, n, k, epi in [("qkv", 12288, 4096, 0), ("out_proj", 4096, 4096, 0), ("fc1", 16384, 4096, 2), ("fc2", 4096, 16384, 0)]:
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    a8 = torch.empty(M, k, device="cuda", dtype=torch.uint8); asc = torch.empty(M, device="cuda", dtype=torch.float32)
    w8 = torch.empty(n, k, device="cuda", dtype=torch.uint8); wsc = torch.empty(n, device="cuda", dtype=torch.float32)
    assert lib.eilev_quant_rows_e4m3(P(w), P(w8), P(wsc), n, k, st()) == 0
    t_b = timed(lambda: lib.eilev_linear(P(a), P(w), P(b), None, P(o), M, n, k, epi, 0, st()))
    t_q = timed(lambda: lib.eilev_quant_rows_e4m3(P(a), P(a8), P(asc), M, k, st()))
    t_8 = timed(lambda: lib.eilev_linear_a8w8(P(a8), P(asc), P(w8), P(wsc), P(b), None, P(o), M, n, k, epi, 0, st()))
    fl = 2.0 * M * n * k
    tot["bf16"] += t_b; tot["fp8"] += t_8; tot["quant"] += t_q
    print(f"{name:8s} M={M} N={n} K={k}: bf16 {t_b:8.1f} us ({fl / t_b / 1e6:6.0f} TF/s) | fp8 MFMA {t_8:8.1f} us ({fl / t_8 / 1e6:6.0f} TF/s, {t_b / t_8:.2f}x) | quantise rows {t_q:6.1f} us "
          f"-> with it {t_b / (t_8 + t_q):.2f}x", flush=True)
print(f"block: bf16 {tot['bf16']:.0f} us | fp8 {tot['fp8']:.0f} + quantisers {tot['quant']:.0f} us = {tot['bf16'] / (tot['fp8'] + tot['quant']):.2f}x (GEMMs alone {tot['bf16'] / tot['fp8']:.2f}x)")
