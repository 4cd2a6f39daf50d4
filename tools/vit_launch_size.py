#!/usr/bin/env python
"""GPU probe: ViT encode of one bench step (544 clips x 8 frames = 4352 frames) at different launch sizes.
Measured (two boxes): 1088 frames per launch 2169 ms, 2176 -> 2201, 4352 -> 2253; 272 -> 2271, 544 -> 2228, 816 -> 2241, 1088 -> 2229:
the engine's 1088 frames (279 616 token rows) per launch sits on the flat optimum."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine

cfg = blip2_config("opt27")
dev = torch.device("cuda")
w = bench.random_weights(cfg, dev)
eng = HipEngine(cfg, w, device=dev, parts=("vit",))
g = torch.Generator(device="cuda"); g.manual_seed(1)
px = torch.randn((544, 3, 8, 224, 224), device="cuda", generator=g).clamp_(-2.5, 2.5).to(torch.bfloat16)
res = {}
for rd in range(3):
    for frames in (272, 544, 816, 1088):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = eng.vit(px, max_frames_per_call=frames)
        e1.record(); torch.cuda.synchronize()
        if rd: res.setdefault(frames, []).append(e0.elapsed_time(e1))
        del out
for f, t in res.items():
    print(f"{f} frames per launch: {statistics.median(t):.1f} ms for 4352 frames")
