#!/usr/bin/env python
"""GPU probe (round 6): time eilev_attention on the PREFILL shapes (causal flash kernel attn_prefill_v2_kernel).

    python tools/attn_prefill_probe.py [opt27|opt67|t5] [reps]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd import abi

if os.environ.get("ATTN_DBG"):  # probe build: kernel debug bits (attn_prefill_v2_kernel: 1 = no compute, 2 = no LDS-DMA of the tiles after the first)
    abi.use_probes()
lib = abi.load_hip()
if os.environ.get("ATTN_DBG"):
    C.CDLL(abi.HIP_LIB_PATH).eilev_debug_attn_v1(int(os.environ["ATTN_DBG"]) << 1)
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = {"opt27": (32, 32, 960, 80, 1), "opt67": (32, 32, 1872, 128, 1), "t5": (32, 32, 960, 64, 0), "opt27_b1": (1, 32, 960, 80, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "opt27"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
b, h, L, hd, causal = SHAPES[name]
D = h * hd
qkv = torch.randn(b, L, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(b, L, D, device="cuda", dtype=torch.bfloat16)
call = lambda: lib.eilev_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(qkv.data_ptr() + 2 * D), C.c_void_p(qkv.data_ptr() + 4 * D),
                                   P(o), b, h, L, L, hd, 3 * D, 3 * D, 3 * D, hd ** -0.5, causal, None, st())
for _ in range(2):
    assert call() == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 4.0 * b * h * L * L * hd * (0.5 if causal else 1.0)
print(f"{name}: b={b} h={h} L={L} hd={hd} causal={causal}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s (causal flops counted once)", flush=True)
