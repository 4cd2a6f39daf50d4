#!/usr/bin/env python
"""GPU probe: where a tile's time goes inside the persistent ping-pong GEMM (gemm_pp4_kernel).

    python tools/gemm_trace.py [fc1|fc2|qkv|proj|fc1_noact] [rows]

One wave of each wave group of every workgroup stamps s_memrealtime (100 MHz) at: loop top, K-loop start, K-loop end, epilogue
start, epilogue end — and s_memtime (shader clock) at top / end, which gives the clock the CU actually ran at.  Prints the mean
phase durations in microseconds over all workgroups and tiles (first and last tile of a workgroup excluded)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eilev_amd import abi

abi.use_probes()  # the eilev_debug_* switches live in the probe build only (libeilev_hip_probes.so)

lib = abi.load_hip()
raw = C.CDLL(abi.HIP_LIB_PATH)
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = {"fc1": (6144, 1408, 1, False), "fc2": (1408, 6144, 0, True), "qkv": (4224, 1408, 0, False), "proj": (1408, 1408, 0, True),
          "fc1_noact": (6144, 1408, 0, False), "fc1_ln": (6144, 1408, 1, False), "qkv_ln": (4224, 1408, 0, False),
          "proj_st": (1408, 1408, 0, True), "fc2_st": (1408, 6144, 0, True),
          "fc2_n1280": (1280, 6144, 0, True), "fc2_n1536": (1536, 6144, 0, True)}  # fc2 with whole column tiles only (half-tile diagnosis, r4)  # _ln: folded-LayerNorm consumer, _st: statistics producer
name = sys.argv[1] if len(sys.argv) > 1 else "fc1"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 139808
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
if os.environ.get("TRACE_FLAGS"):
    raw.eilev_debug_gemm_flags(int(os.environ["TRACE_FLAGS"]))
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
for grp, gname in ((0, "early waves"), (1, "late waves")):
    x = t[:, grp]
    valid = (x[:, :, 4] > 0)
    valid[:, 0] = False
    nt = valid.sum(1)
    for wg in range(WG):  # drop the last tile of each workgroup too
        if nt[wg] > 0:
            valid[wg, int(np.nonzero(valid[wg])[0][-1])] = False
    sel = x[valid]
    us = lambda a_, b_: float((sel[:, b_] - sel[:, a_]).mean()) * 0.01
    clk = float(((sel[:, 6] - sel[:, 5]) / np.maximum(sel[:, 4] - sel[:, 0], 1)).mean()) * 100.0
    nxt = []
    for wg in range(WG):
        idx = np.nonzero(valid[wg])[0]
        for i in idx[:-1]:
            if valid[wg, i + 1]:
                nxt.append(x[wg, i + 1, 0] - x[wg, i, 0])
    kl = (sel[:, 2] - sel[:, 1]) * 0.01  # K-loop durations: whole tiles and half (or tall) tiles separate into two modes
    q = np.percentile(kl, [5, 25, 50, 75, 95])
    lo = kl[kl < 0.75 * q[2]]
    hi_ = kl[kl > 1.25 * q[2]]
    print(f"  {gname}: K loop us  p5 {q[0]:.1f}  p25 {q[1]:.1f}  p50 {q[2]:.1f}  p75 {q[3]:.1f}  p95 {q[4]:.1f} | < 0.75 median: {len(lo)} tiles, mean "
          f"{lo.mean() if len(lo) else 0:.1f} | > 1.25 median: {len(hi_)} tiles, mean {hi_.mean() if len(hi_) else 0:.1f}")
    print(f"  {gname}: tiles {len(sel)}  top->kloop {us(0, 1):.2f}  kloop {us(1, 2):.2f}  kend->epi {us(2, 3):.2f}  epilogue {us(3, 4):.2f}  "
          f"tile period {np.mean(nxt) * 0.01:.2f} us  shader clock {clk:.0f} MHz")
