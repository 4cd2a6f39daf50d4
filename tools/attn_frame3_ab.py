#!/usr/bin/env python
"""GPU A/B of attn_frame3_kernel between two builds of the library in ONE process on ONE box (alternating rounds):
    AB_LIBS=eilev_amd/csrc/libeilev_hip_fa3old.so,eilev_amd/csrc/libeilev_hip.so python tools/attn_frame3_ab.py [frames]
"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eilev_amd import abi
paths = os.environ.get("AB_LIBS", "eilev_amd/csrc/libeilev_hip_fa3old.so,eilev_amd/csrc/libeilev_hip.so").split(",")
libs = [abi.bind(C.CDLL(os.path.abspath(p))) for p in paths]
raws = [C.CDLL(os.path.abspath(p)) for p in paths]
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
h, sq, hd = 16, 257, 88
D = h * hd
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1088
reps = int(os.environ.get("REPS", "20"))
torch.manual_seed(0)
qkv = (torch.randn(b, sq, 3 * D, device="cuda") * 1.5).to(torch.bfloat16)
outs = [torch.empty((b, sq, D), device="cuda", dtype=torch.bfloat16) for _ in libs]
times = [[] for _ in libs]
for rd in range(7):
    for i, lib in enumerate(libs):
        raws[i].eilev_debug_attn_v1(32)
        call = lambda: lib.eilev_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(qkv.data_ptr() + 2 * D), C.c_void_p(qkv.data_ptr() + 4 * D),
                                           C.c_void_p(outs[i].data_ptr()), b, h, sq, sq, hd, 3 * D, 3 * D, 3 * D, hd ** -0.5, 0, None, st())
        assert call() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        if rd: times[i].append(e0.elapsed_time(e1) / reps * 1e3)
        raws[i].eilev_debug_attn_v1(0)
for p, t in zip(paths, times):
    print(f"{os.path.basename(p):32s} median {statistics.median(t):8.1f} us  (min {min(t):.1f} max {max(t):.1f})")
print("max |A - B| =", (outs[0].float() - outs[1].float()).abs().max().item(), " equal:", torch.equal(outs[0], outs[1]))
