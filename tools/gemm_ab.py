#!/usr/bin/env python
"""GPU probe: A/B two builds of the library on the ViT GEMM shapes, interleaved in ONE process on ONE box (boxes differ by up
to 8 % on identical kernels, so cross-call comparisons are useless).

    python tools/gemm_ab.py <libA.so> <libB.so> [rows] [rounds]
"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd import abi

libs = [abi.bind(C.CDLL(os.path.abspath(p))) for p in sys.argv[1:3]]
m = int(sys.argv[3]) if len(sys.argv) > 3 else 139808
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 7
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = {"fc1": (6144, 1408, 1, False), "fc1_noact": (6144, 1408, 0, False), "fc2": (1408, 6144, 0, True), "qkv": (4224, 1408, 0, False),
          "proj": (1408, 1408, 0, True),
          # what the bench's folded-LayerNorm ViT blocks launch (eilev_linear_lnfold / eilev_linear_stats)
          "fc1_ln": (6144, 1408, 1, False), "qkv_ln": (4224, 1408, 0, False), "fc2_st": (1408, 6144, 0, True), "proj_st": (1408, 1408, 0, True)}
# the OPT-2.7B prefill linears of a bench step (32 samples x 960 tokens): AB_SHAPES=opt_qkv,... (rows from AB_MOPT, default 30720)
SHAPES.update({"opt_x1280": (1280, 6144, 0, True), "opt_x1408": (1408, 6144, 0, True), "opt_x2560": (2560, 6144, 0, True), "opt_x2560k10": (2560, 10240, 0, True),
               "opt_t5_wo": (2048, 5120, 0, True), "opt_t5_o": (2048, 2048, 0, True), "opt_67_fc2": (4096, 16384, 0, True), "opt_67_out": (4096, 4096, 0, True),
               "opt_67_qkv": (12288, 4096, 0, False),  # (round 6: where does the three-deep A ring start to pay?  flan-t5-xl wo / o, OPT-6.7B linears)
               "opt_qkv": (7680, 2560, 0, False), "opt_fc1": (10240, 2560, 2, False), "opt_fc2": (2560, 10240, 0, True), "opt_out": (2560, 2560, 0, True)})
only = os.environ.get("AB_SHAPES")
for name, (n, k, epi, resid) in SHAPES.items():
    if (only and name not in only.split(",")) or (not only and name.startswith("opt_")):
        continue
    if name.startswith("opt_"):
        m = int(os.environ.get("AB_MOPT", 30720))
    elif len(sys.argv) > 3:
        m = int(sys.argv[3])
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    r = torch.randn(m, n, device="cuda").to(torch.bfloat16) if resid else None
    outs = [torch.empty(m, n, device="cuda", dtype=torch.bfloat16) for _ in libs]
    if name.endswith("_ln"):
        cs = torch.randn(n, device="cuda")
        rows = torch.stack([torch.rand(m, device="cuda") + 0.5, torch.randn(m, device="cuda") * 0.1], 1).contiguous()
        call = lambda lib, o: lib.eilev_linear_lnfold(P(a), P(w), P(b), P(cs), P(rows), P(o), m, n, k, epi, st())
    elif name.endswith("_st"):
        stats = torch.empty(((n + 63) // 64, m, 2), device="cuda")
        call = lambda lib, o: lib.eilev_linear_stats(P(a), P(w), P(b), P(r), P(o), m, n, k, P(stats), st())
    else:
        call = lambda lib, o: lib.eilev_linear(P(a), P(w), P(b), P(r), P(o), m, n, k, epi, 0, st())
    for lib, o in zip(libs, outs):
        call(lib, o)
    torch.cuda.synchronize()
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    times = [[] for _ in libs]
    for rd in range(rounds + 1):
        for i, (lib, o) in enumerate(zip(libs, outs)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call(lib, o)
            e1.record()
            torch.cuda.synchronize()
            if rd:
                times[i].append(e0.elapsed_time(e1) / 5)
    med = [statistics.median(t) for t in times]
    tf = [2 * m * n * k / x / 1e9 for x in med]
    print(f"{name:10s} M={m} A {med[0]*1e3:7.1f} us {tf[0]:7.1f} TF/s | B {med[1]*1e3:7.1f} us {tf[1]:7.1f} TF/s | B/A speed {tf[1]/tf[0]:.3f} | max |A-B| {d:.4g}", flush=True)
