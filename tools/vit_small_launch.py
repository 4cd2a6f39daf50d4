#!/usr/bin/env python
"""GPU probe: ViT encode of ONE 136-frame launch (17 clips x 8 frames = one rank's share of the N = 8 strong-scaling step, and the latency mode) with
the LayerNorm fold off (default below 65 536 rows: separate LayerNorm kernels, gemm_w6 / small-tile GEMMs) and forced on (persistent kernel + fold)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eilev_amd import abi

abi.use_probes()  # the eilev_debug_* switches live in the probe build only (libeilev_hip_probes.so)
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine

cfg = blip2_config("opt27")
dev = torch.device("cuda")
w = bench.random_weights(cfg, dev)
eng = HipEngine(cfg, w, device=dev, parts=("vit",))
eng.ensure_vit_fold()
g = torch.Generator(device="cuda"); g.manual_seed(1)
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 17
px = torch.randn((clips, 3, 8, 224, 224), device="cuda", generator=g).clamp_(-2.5, 2.5).to(torch.bfloat16)
res, outs = {}, {}
for rd in range(4):
    for tag, rows in (("fold off", -1), ("fold on", 1)):
        eng.pack.vit.fold_min_rows = rows  # EilevVitWeights.fold_min_rows (ABI 16)
        eng.vit(px); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): out = eng.vit(px)
        e1.record(); torch.cuda.synchronize()
        if rd: res.setdefault(tag, []).append(e0.elapsed_time(e1) / 3)
        outs[tag] = out[0].float() if isinstance(out, (tuple, list)) else out.float()
eng.pack.vit.fold_min_rows = 0
for t, v in res.items():
    print(f"{clips} clips ({clips * 8} frames), {t}: {statistics.median(v):.2f} ms")
a, b = outs["fold off"], outs["fold on"]
print("rel-RMS fold on vs off:", float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()))
