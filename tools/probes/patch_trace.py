#!/usr/bin/env python
"""GPU probe: phase cycles of the fused patch-embed + LayerNorm kernel (s_memtime stamps of thread 0 of every workgroup)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from eilev_amd import abi
from hip_utils import models
raw = C.CDLL(abi.HIP_LIB_PATH)
cfg, _, eng = models("real_1l")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1088
px = torch.randn((frames // 8, 3, 8, 224, 224), device="cuda").clamp_(-2.5, 2.5).to(torch.bfloat16)
eng.vit(px); torch.cuda.synchronize()
buf = torch.zeros(frames * 4 * 8, dtype=torch.int64, device="cuda")
raw.eilev_debug_patch_trace(C.c_void_p(buf.data_ptr()))
eng.vit(px); torch.cuda.synchronize()
raw.eilev_debug_patch_trace(None)
t = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
names = ["stage im2col", "K loop", "x = acc+bias+pos", "LN stats", "flush stores", "(end)"]
for i in range(5):
    print(f"{names[i]:20s} {np.mean(t[:, i + 1] - t[:, i]):10.0f} cycles")
print("total", np.mean(t[:, 5] - t[:, 0]), "cycles per workgroup;", len(t), "workgroups")
