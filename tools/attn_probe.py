#!/usr/bin/env python
"""GPU probe: time eilev_attention on the ViT / OPT prefill shapes, v1 vs v2 kernels."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd import abi

abi.use_probes()  # the eilev_debug_* switches live in the probe build only (libeilev_hip_probes.so)

lib = abi.load_hip()
raw = C.CDLL(abi.HIP_LIB_PATH)
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("vit", 544, 16, 257, 257, 88, 0)]
modes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "0"])]
for name, b, h, sq, skv, hd, causal in shapes:
    D = h * hd
    qkv = torch.randn(b, sq, 3 * D, device="cuda").to(torch.bfloat16)
    o = torch.empty(b, sq, D, device="cuda", dtype=torch.bfloat16)
    outs = {}
    for v1 in modes:
        raw.eilev_debug_attn_v1(v1)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
        call = lambda: lib.eilev_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(qkv.data_ptr() + 2 * D), C.c_void_p(qkv.data_ptr() + 4 * D),
                                           P(o), b, h, sq, skv, hd, 3 * D, 3 * D, 3 * D, hd ** -0.5, causal, None, st())
        for _ in range(3):
            assert call() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 4.0 * b * h * sq * skv * hd * (0.5 if causal else 1.0)
        outs[v1] = o.float().clone()
        print(f"{name:12s} mode {v1}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
    ks = list(outs)
    if len(ks) >= 2:
        print(f"   max |mode {ks[0]} - mode {ks[1]}| =", (outs[ks[0]] - outs[ks[1]]).abs().max().item())
raw.eilev_debug_attn_v1(0)
