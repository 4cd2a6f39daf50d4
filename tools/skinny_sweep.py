#!/usr/bin/env python
"""GPU probe: the decode GEMV shapes with 1 / 2 / 4 weight blocks per workgroup (gemm_skinny_nb_kernel), interleaved.
    python tools/skinny_sweep.py [M]"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eilev_amd import abi

abi.use_probes()  # the eilev_debug_* switches live in the probe build only (libeilev_hip_probes.so)
lib = abi.load_hip(); raw = C.CDLL(abi.HIP_LIB_PATH)
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for name, n, k in [("qkv", 7680, 2560), ("out", 2560, 2560), ("fc1", 10240, 2560), ("fc2", 2560, 10240), ("lm_head", 50272, 2560), ("qkv67", 12288, 4096), ("fc1_67", 16384, 4096)]:
    copies = max(2, int(600e6 // (n * k * 2)))
    ws = [(torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16) for _ in range(copies)]
    a = torch.randn(M, k, device="cuda").to(torch.bfloat16); b = torch.randn(n, device="cuda").to(torch.bfloat16)
    scr = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    outs = {}
    res = {}
    for nbsel in (1, 2, 4, 7):
        raw.eilev_debug_gemm_flags(nbsel << 26)
        o = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        for w in ws: lib.eilev_linear(P(a), P(w), P(b), None, P(o), M, n, k, 0, 0, st())
        torch.cuda.synchronize(); outs[nbsel] = o.clone()
    for rd in range(4):
        for nbsel in (1, 2, 4, 7):
            raw.eilev_debug_gemm_flags(nbsel << 26)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                for w in ws: lib.eilev_linear(P(a), P(w), P(b), None, P(o), M, n, k, 0, 0, st())
            e1.record(); torch.cuda.synchronize()
            if rd: res.setdefault(nbsel, []).append(e0.elapsed_time(e1) * 1e3 / (3 * copies))
    raw.eilev_debug_gemm_flags(0)
    same = all(torch.equal(outs[1], outs[x]) for x in (2, 4, 7))
    print(f"{name:8s} M={M} N={n:6d} K={k:6d}: " + " | ".join(f"{('old' if x == 3 else 'NB=' + str(x))}: {statistics.median(res[x]):6.1f} us {n*k*2/statistics.median(res[x])/1e6:5.2f} TB/s" for x in (1, 2, 4, 7)) + f" | identical: {same}", flush=True)
