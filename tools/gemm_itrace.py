#!/usr/bin/env python
"""GPU probe: the eight phase edges of ONE K-step inside the persistent ping-pong GEMM (gemm_pp4_kernel), from a library built with
-DEILEV_PROBES -DEILEV_PP4_ITRACE=<k-step> (eilev_amd/csrc/build.py --variant itrace -DEILEV_PROBES -DEILEV_PP4_ITRACE=8).

    python tools/gemm_itrace.py <lib.so> [fc1_ln|fc2_st|qkv_ln|proj_st|fc2|fc1_noact] [rows]

Per wave group (early / late) the median shader-clock cycles of: read phase of half 0 (issue -> fragments landed), wait at the
barrier, MFMA phase 0, wait, read phase of half 1 (+ the late group's DMA wait / issue), wait, MFMA phase 1 — and the K-step period."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eilev_amd import abi

raw = C.CDLL(os.path.abspath(sys.argv[1]))
lib = abi.bind(raw)
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = {"fc2": (1408, 6144, 0, True), "fc1_noact": (6144, 1408, 0, False), "fc1_ln": (6144, 1408, 1, False), "qkv_ln": (4224, 1408, 0, False),
          "proj_st": (1408, 1408, 0, True), "fc2_st": (1408, 6144, 0, True), "fc2_n1536": (1536, 6144, 0, True)}
name = sys.argv[2] if len(sys.argv) > 2 else "fc1_ln"
m = int(sys.argv[3]) if len(sys.argv) > 3 else 279616
n, k, epi, resid = SHAPES[name]
a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
b = torch.randn(n, device="cuda").to(torch.bfloat16)
r = torch.randn(m, n, device="cuda").to(torch.bfloat16) if resid else None
o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
TILES, WG = 64, 256
if name.endswith("_ln"):
    cs = torch.randn(n, device="cuda")
    rows = torch.stack([torch.rand(m, device="cuda") + 0.5, torch.randn(m, device="cuda") * 0.1], 1).contiguous()
    run = lambda: lib.eilev_linear_lnfold(P(a), P(w), P(b), P(cs), P(rows), P(o), m, n, k, epi, st())
elif name.endswith("_st"):
    stats = torch.empty(((n + 63) // 64, m, 2), device="cuda")
    run = lambda: lib.eilev_linear_stats(P(a), P(w), P(b), P(r), P(o), m, n, k, P(stats), st())
else:
    run = lambda: lib.eilev_linear(P(a), P(w), P(b), P(r), P(o), m, n, k, epi, 0, st())
for _ in range(3):
    assert run() == 0
buf = torch.zeros(WG * 2 * TILES * 8, dtype=torch.int64, device="cuda")
raw.eilev_debug_gemm_trace(C.c_void_p(buf.data_ptr()), TILES)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
raw.eilev_debug_gemm_trace(None, 0)
ms = e0.elapsed_time(e1)
t = buf.cpu().numpy().reshape(WG, 2, TILES, 8).astype(np.float64)
print(f"{name}: M={m} N={n} K={k}  {ms * 1e3:.0f} us  {2 * m * n * k / ms / 1e9:.0f} TFLOP/s (traced launch)")
names = ["R0 (reads of half 0 [+ early group's 8 DMA pieces])", "wait at barrier", "M0 (16 MFMAs)", "wait at barrier",
         "R1 (reads of half 1 [+ late group's DMA wait + 8 pieces])", "wait at barrier", "M1 (16 MFMAs [+ early group's DMA wait])"]
for grp, gname in ((0, "early waves"), (1, "late waves")):
    x = t[:, grp]
    valid = (x[:, :, 7] > x[:, :, 0]) & (x[:, :, 0] > 0)
    valid[:, 0] = False
    sel = x[valid]
    d = np.diff(sel, axis=1)
    print(f"  {gname}: {sel.shape[0]} K-steps traced; cycles median (p10 .. p90)")
    for i, nm in enumerate(names):
        q = np.percentile(d[:, i], [10, 50, 90])
        print(f"    {nm:72s} {q[1]:7.0f}  ({q[0]:.0f} .. {q[2]:.0f})")
    tot = sel[:, 7] - sel[:, 0]
    q = np.percentile(tot, [10, 50, 90])
    print(f"    {'edge 0 -> edge 7 (a K-step minus its last barrier wait)':72s} {q[1]:7.0f}  ({q[0]:.0f} .. {q[2]:.0f})   ideal 4 x 512 = 2048 MFMA-pipe cycles per K-step")
