#!/usr/bin/env python
"""GPU probe: per-phase s_memtime stamps of workgroup 0 of the frame attention kernel (ViT shape)."""
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
b, h, sq, hd = 544, 16, 257, 88
D = h * hd
qkv = torch.randn(b, sq, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(b, sq, D, device="cuda", dtype=torch.bfloat16)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
new = len(sys.argv) > 1 and sys.argv[1] == "3"  # attn_frame3_kernel (two wave groups) instead of attn_frame_kernel
raw.eilev_debug_attn_v1((512 if new else 256) << 1)
for _ in range(3):
    assert lib.eilev_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(qkv.data_ptr() + 2 * D), C.c_void_p(qkv.data_ptr() + 4 * D),
                               C.c_void_p(o.data_ptr()), b, h, sq, sq, hd, 3 * D, 3 * D, 3 * D, hd ** -0.5, 0, None, st) == 0
torch.cuda.synchronize()
ts = np.zeros((9, 8, 16), np.uint64)
assert raw.eilev_debug_attn_ts(ts.ctypes.data_as(C.c_void_p)) == 0
t0 = ts[:, 2, 1].min()
names = (["top", "ph0", "bar1", "ph1", "bar2", "ph2", "pvmain", "stored", "pvstart"] + ["-"] * 6) if new else ["top", "bar1", "S0", "sm0", "vwait", "bar2", "PV0", "S1", "sm1", "PV1", "-", "-", "-", "-", "-"]
for w in range(8 if new else 9):
    for it in (2, 3):
        row = ts[w, it].astype(np.int64) - int(t0)
        print(f"wave {w} pair {it}: " + " ".join(f"{names[e]}={row[e]}" for e in range(15) if names[e] != "-" and ts[w, it, e]))
if new:
    for w, (a0, a1) in ((0, (4, 5)), (4, (0, 1))):
        d = ts[w, 2:7].astype(np.int64)
        print(f"wave {w} PV phase: issue Q/V {(d[:, 8] - d[:, a0]).tolist()} main loop {(d[:, 6] - d[:, 8]).tolist()} stores {(d[:, 7] - d[:, 6]).tolist()} "
              f"CLS part {(d[:, a1] - d[:, 7]).tolist()}")
    print("group E (waves 0-3): ph0 = S, ph1 = softmax, ph2 = PV + stores; group L (waves 4-7): ph0 = PV(p-1) + stores, ph1 = S, ph2 = softmax")
    for w in (0, 4):
        d = ts[w, 2:7].astype(np.int64)
        per = np.diff(d[:, 0])
        print(f"wave {w}: period {per.tolist()}  slot0 work {(d[:, 1] - d[:, 0]).tolist()} wait {(d[:, 2] - d[:, 1]).tolist()}  "
              f"slot1 work {(d[:, 3] - d[:, 2]).tolist()} wait {(d[:, 4] - d[:, 3]).tolist()}  slot2 work {(d[:, 5] - d[:, 4]).tolist()} wait {(d[1:, 0] - d[:-1, 5]).tolist()}")
