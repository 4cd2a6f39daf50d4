#!/usr/bin/env python
"""GPU probe: per-phase s_memtime stamps of workgroup 0 of the frame attention kernel (ViT shape)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eilev_amd import abi

lib = abi.load_hip()
raw = C.CDLL(abi.HIP_LIB_PATH)
b, h, sq, hd = 544, 16, 257, 88
D = h * hd
qkv = torch.randn(b, sq, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(b, sq, D, device="cuda", dtype=torch.bfloat16)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
new = False  # (the round-2 attn_frame2_kernel experiment was removed from the library in round 3: git history)
raw.eilev_debug_attn_v1((512 if new else 256) << 1)
for _ in range(3):
    assert lib.eilev_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(qkv.data_ptr() + 2 * D), C.c_void_p(qkv.data_ptr() + 4 * D),
                               C.c_void_p(o.data_ptr()), b, h, sq, sq, hd, 3 * D, 3 * D, 3 * D, hd ** -0.5, 0, None, st) == 0
torch.cuda.synchronize()
ts = np.zeros((9, 8, 16), np.uint64)
assert raw.eilev_debug_attn_ts(ts.ctypes.data_as(C.c_void_p)) == 0
t0 = ts[:, 2, 1].min()
names = (["top", "bar1", "S", "sm", "clsS", "bar2", "PV", "store", "end"] + ["-"] * 6) if new else ["top", "bar1", "S0", "sm0", "vwait", "bar2", "PV0", "S1", "sm1", "PV1", "-", "-", "-", "-", "-"]
for w in range(8 if new else 9):
    for it in (2, 3):
        row = ts[w, it].astype(np.int64) - int(t0)
        print(f"wave {w} pair {it}: " + " ".join(f"{names[e]}={row[e]}" for e in range(15) if names[e] != "-" and ts[w, it, e]))
