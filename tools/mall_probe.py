#!/usr/bin/env python
"""GPU probe: eilev_linear at M = 32 (and 5) on the decode shapes with the weights COLD (rotated through > 600 MB: HBM) vs HOT (one copy, re-read
every launch: the 256-MiB Infinity Cache serves it).  What a weight prefetch one kernel ahead could buy at most.

    python tools/mall_probe.py
"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eilev_amd import abi
lib = abi.load_hip()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M in (32, 5):
    for name, n, k in [("qkv", 7680, 2560), ("out_proj", 2560, 2560), ("fc1", 10240, 2560), ("fc2", 2560, 10240)]:
        copies = max(2, int(600e6 // (n * k * 2)))
        ws = [(torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16) for _ in range(copies)]
        a = torch.randn(M, k, device="cuda").to(torch.bfloat16); b = torch.randn(n, device="cuda").to(torch.bfloat16)
        o = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        res = {}
        for rd in range(4):
            for tag, seq in (("cold", ws), ("hot", [ws[0]] * copies)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    for w in seq: lib.eilev_linear(P(a), P(w), P(b), None, P(o), M, n, k, 0, 0, st())
                e1.record(); torch.cuda.synchronize()
                if rd: res.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3 / (3 * copies))
        print(f"{name:8s} M={M} N={n} K={k}: " + " | ".join(f"{t}: {statistics.median(v):6.1f} us ({n*k*2/statistics.median(v)/1e6:4.2f} TB/s)" for t, v in res.items()), flush=True)
