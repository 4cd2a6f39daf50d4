#!/usr/bin/env python
"""GPU probe: eilev_linear at M = 32 on the decode shapes: round-3 kernels (flag 1 << 28) vs gemm_rows32_kernel vs the same kernel without
its activation loads / without its weight loads (variant libraries), weights rotated through > 256 MB so no cache serves them.

    python eilev_amd/csrc/build.py --force --variant nox -DROWS32_NOX=1     # (add -DROWS32_NOW=1 as well for the kernel with NO operand loads)
    python eilev_amd/csrc/build.py --force --variant now -DROWS32_NOW=1
    python tools/rows32_floor.py
"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eilev_amd import abi
libs = {n: abi.bind(C.CDLL(os.path.abspath(p))) for n, p in (("main", "eilev_amd/csrc/libeilev_hip.so"), ("nox", "eilev_amd/csrc/libeilev_hip_nox.so"), ("now", "eilev_amd/csrc/libeilev_hip_now.so"))}
raws = {n: C.CDLL(os.path.abspath(p)) for n, p in (("main", "eilev_amd/csrc/libeilev_hip.so"), ("nox", "eilev_amd/csrc/libeilev_hip_nox.so"), ("now", "eilev_amd/csrc/libeilev_hip_now.so"))}
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 32
for name, n, k in [("qkv", 7680, 2560), ("fc1", 10240, 2560), ("lm_head", 50272, 2560)]:
    copies = max(2, int(600e6 // (n * k * 2)))
    ws = [(torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16) for _ in range(copies)]
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16); b = torch.randn(n, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    res = {}
    cases = [("round3", "main", 1 << 28), ("rows32", "main", 0), ("rows32-no-x", "nox", 0), ("rows32-no-w", "now", 0)]
    for rd in range(4):
        for tag, lib, flag in cases:
            raws[lib].eilev_debug_gemm_flags(flag)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                for w in ws: libs[lib].eilev_linear(P(a), P(w), P(b), None, P(o), M, n, k, 0, 0, st())
            e1.record(); torch.cuda.synchronize()
            if rd: res.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3 / (3 * copies))
            raws[lib].eilev_debug_gemm_flags(0)
    print(f"{name:8s} M={M} N={n} K={k}: " + " | ".join(f"{t}: {statistics.median(v):6.1f} us ({n*k*2/statistics.median(v)/1e6:4.2f} TB/s)" for t, v in res.items()), flush=True)
