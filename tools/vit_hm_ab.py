#!/usr/bin/env python
"""GPU probe (round 5): one ViT-g launch of 1088 frames with the q|k|v rows head-major (default) against row-major (probe switch), same box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from eilev_amd import abi

if os.environ.get("PROBE_LIB"):  # A/B against another build of the probe library
    abi.PROBES_LIB_PATH = os.path.abspath(os.environ["PROBE_LIB"])
abi.use_probes()
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine

cfg = blip2_config("opt27")
dev = torch.device("cuda")
w = bench.random_weights(cfg, dev)
eng = HipEngine(cfg, w, device=dev, parts=("vit",))
raw = C.CDLL(abi.HIP_LIB_PATH)
px = torch.randn((136, 3, 8, 224, 224), device=dev).clamp_(-2.5, 2.5).to(torch.bfloat16)
eng.vit(px)
outs = {}
for rd in range(6):
    hm = int(os.environ['HM']) if 'HM' in os.environ else 1 - rd % 2  # HM=0 / 1: one form only (for a kernel trace)
    raw.eilev_debug_vit_head_major(hm)
    eng.vit(px)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        out = eng.vit(px)
    e1.record()
    torch.cuda.synchronize()
    outs[hm] = out
    print(f"round {rd}: q|k|v {'head-major' if hm else 'row-major '}: {e0.elapsed_time(e1) / 3:.2f} ms per 1088-frame launch", flush=True)
raw.eilev_debug_vit_head_major(1)
if len(outs) == 2:
    print("outputs bit-identical:", bool(torch.equal(outs[0], outs[1])))
