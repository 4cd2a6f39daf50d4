#!/usr/bin/env python
"""GPU probe: the ViT encode of one bench launch (1088 frames) with the LayerNorms folded into qkv / fc1 vs with LayerNorm kernels,
interleaved in one process (same box, same clocks).    python tools/ln_fold_ab.py [layers] [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd.configs import CONFIGS, blip2_config
from eilev_amd.engine import HipEngine

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
CONFIGS["_ab"] = dict(CONFIGS["real_vit_3l"], vision_config=dict(CONFIGS["real_vit_3l"]["vision_config"], num_hidden_layers=layers))
cfg = blip2_config("_ab")
from bench import random_weights  # noqa: E402

w = random_weights(cfg, torch.device("cuda"))
eng = HipEngine(cfg, w, device="cuda", parts=("vit",))
g = torch.Generator(device="cuda")
g.manual_seed(1)
px = torch.randn((136, 3, 8, 224, 224), device="cuda", generator=g).clamp_(-2.5, 2.5).to(torch.bfloat16)
times = {1: [], -1: []}  # EilevVitWeights.fold_min_rows (ABI 16): 1 = every launch, -1 = never
outs = {}
for rd in range(rounds + 1):
    for knob in times:
        eng.pack.vit.fold_min_rows = knob
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs[knob] = eng.vit(px)
        e1.record()
        torch.cuda.synchronize()
        if rd:
            times[knob].append(e0.elapsed_time(e1))
eng.pack.vit.fold_min_rows = 0
a, b = outs[1].float(), outs[-1].float()
rel = float(((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item())
f, u = statistics.median(times[1]), statistics.median(times[-1])
print(f"{layers} blocks x 1088 frames: folded {f:.2f} ms | LayerNorm kernels {u:.2f} ms | per block {f / layers:.3f} vs {u / layers:.3f} ms | "
      f"folded/unfolded time {f / u:.4f} | rel-RMS folded vs unfolded {rel:.2e}")
