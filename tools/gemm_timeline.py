#!/usr/bin/env python
"""GPU probe (round 6): the TIMELINE of a persistent-GEMM launch — who runs which tile when.

    python tools/gemm_timeline.py [fc2_st|proj_st|fc2_n1536|...] [rows]

Same stamps as tools/gemm_trace.py (s_memrealtime, 100 MHz, one clock for the chip), read per workgroup instead of averaged:
  * duration of whole tiles and of half tiles (the last column tile of N = 1408), from the tile coordinates of the static stride;
  * when each workgroup finishes its last tile (the launch ends with the slowest): balance;
  * per round of the static stride, the spread of the tile START times inside an XCD (32 workgroups that share an L2): drift —
    tiles that share an A row panel only meet in the L2 if they walk K within a few K-steps of each other."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eilev_amd import abi

abi.use_probes()
lib = abi.load_hip()
raw = C.CDLL(abi.HIP_LIB_PATH)
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = {"fc2": (1408, 6144, True), "proj": (1408, 1408, True), "proj_st": (1408, 1408, True), "fc2_st": (1408, 6144, True),
          "fc2_n1280": (1280, 6144, True), "fc2_n1536": (1536, 6144, True), "fc1_noact": (6144, 1408, False)}
name = sys.argv[1] if len(sys.argv) > 1 else "fc2_st"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 279616
n, k, resid = SHAPES[name]
a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
b = torch.randn(n, device="cuda").to(torch.bfloat16)
r = torch.randn(m, n, device="cuda").to(torch.bfloat16) if resid else None
o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
TILES, WG = 64, 256
if name.endswith("_st"):
    stats = torch.empty(((n + 63) // 64, m, 2), device="cuda")
    run = lambda: lib.eilev_linear_stats(P(a), P(w), P(b), P(r), P(o), m, n, k, P(stats), st())
else:
    run = lambda: lib.eilev_linear(P(a), P(w), P(b), P(r), P(o), m, n, k, 0, 0, st())
if os.environ.get("TRACE_FLAGS"):
    raw.eilev_debug_gemm_flags(int(os.environ["TRACE_FLAGS"]))
for _ in range(3):
    assert run() == 0
buf = torch.zeros(WG * 2 * TILES * 8, dtype=torch.int64, device="cuda")
raw.eilev_debug_gemm_trace(C.c_void_p(buf.data_ptr()), TILES)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
raw.eilev_debug_gemm_trace(None, 0)
ms = e0.elapsed_time(e1)
t = buf.cpu().numpy().reshape(WG, 2, TILES, 8).astype(np.float64)[:, 0] * 0.01  # early waves, microseconds
print(f"{name}: M={m} N={n} K={k}  {ms * 1e3:.0f} us  {2 * m * n * k / ms / 1e9:.0f} TFLOP/s (traced launch: the stamps cost ~10 %)")
tiles_m, tiles_n = (m + 255) // 256, (n + 255) // 256
ntiles = tiles_m * tiles_n
half_col = (n % 256) != 0 and (n % 256) <= 128
gm = 4 if tiles_n <= 8 else 8


def coords(tt):  # gemm_common.h tile_coords
    xcd, q, rr = tt & 7, ntiles >> 3, ntiles & 7
    o_ = (xcd * (q + 1) if xcd < rr else rr * (q + 1) + (xcd - rr) * q) + (tt >> 3)
    width = gm * tiles_n
    group = o_ // width
    first = group * gm
    gsz = min(tiles_m - first, gm)
    inn = o_ - group * width
    return first + inn % gsz, inn // gsz


t0 = t[:, 0, 0][t[:, 0, 0] > 0].min()
start, end = t[:, :, 0] - t0, t[:, :, 4] - t0  # loop top .. epilogue end of every (workgroup, tile slot)
valid = t[:, :, 4] > 0
dur_full, dur_half, rounds = [], [], {}
for wg in range(WG):
    for i in range(TILES):
        if not valid[wg, i]:
            continue
        tt = wg + i * WG
        if tt >= ntiles:
            continue
        tm, tn = coords(tt)
        d = end[wg, i] - start[wg, i]
        (dur_half if (half_col and tn == tiles_n - 1) else dur_full).append(d)
        rounds.setdefault(i, []).append((wg & 7, start[wg, i], end[wg, i]))
pf = lambda x: f"n={len(x)} mean {np.mean(x):.1f} p5 {np.percentile(x, 5):.1f} p50 {np.percentile(x, 50):.1f} p95 {np.percentile(x, 95):.1f}" if len(x) else "n=0"
print(f"  whole tiles (us): {pf(dur_full)}")
print(f"  half  tiles (us): {pf(dur_half)}" + (f"   -> half / whole = {np.mean(dur_half) / np.mean(dur_full):.2f}" if dur_half else ""))
last_end = np.array([end[wg][valid[wg]].max() if valid[wg].any() else 0.0 for wg in range(WG)])
ntile_wg = valid.sum(1)
print(f"  workgroup finish times (us): min {last_end.min():.0f}  p25 {np.percentile(last_end, 25):.0f}  p50 {np.percentile(last_end, 50):.0f}  "
      f"p75 {np.percentile(last_end, 75):.0f}  max {last_end.max():.0f}   -> mean idle at the end {100 * (1 - last_end.mean() / last_end.max()):.1f} % of the launch")
busy = np.array([(end[wg][valid[wg]] - start[wg][valid[wg]]).sum() for wg in range(WG)])
print(f"  tiles per workgroup {ntile_wg.min()}..{ntile_wg.max()}; busy time per workgroup (us): min {busy.min():.0f} mean {busy.mean():.0f} max {busy.max():.0f}")
# per XCD (block b runs on XCD b % 8): are some XCDs / CUs systematically slower?
for x in range(8):
    idx = [wg for wg in range(WG) if (wg & 7) == x]
    fe = last_end[idx]
    mt = np.array([np.mean([end[wg, i] - start[wg, i] for i in range(1, TILES) if valid[wg, i] and (wg + i * WG) < ntiles and not (half_col and coords(wg + i * WG)[1] == tiles_n - 1)]) for wg in idx])
    print(f"  XCD {x}: finish min {fe.min():.0f} mean {fe.mean():.0f} max {fe.max():.0f} | mean WHOLE-tile time per workgroup: min {mt.min():.1f} mean {mt.mean():.1f} max {mt.max():.1f} us")
print("  round: spread of tile START times inside an XCD (max - min over its 32 workgroups; mean over the 8 XCDs), and of the END times")
for i in sorted(rounds):
    by = {}
    for x, s_, e_ in rounds[i]:
        by.setdefault(x, []).append((s_, e_))
    ss = np.mean([max(v)[0] - min(v)[0] for v in by.values() if len(v) > 1]) if by else 0
    ee = np.mean([max(q[1] for q in v) - min(q[1] for q in v) for v in by.values() if len(v) > 1]) if by else 0
    if i < 4 or i % 4 == 0 or i >= max(rounds) - 1:
        print(f"    round {i:2d}: {len(rounds[i]):3d} tiles, start spread {ss:7.1f} us, end spread {ee:7.1f} us, mean start {np.mean([q[1] for q in rounds[i]]):8.1f}")
