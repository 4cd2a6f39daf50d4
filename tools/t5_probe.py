#!/usr/bin/env python
"""GPU probe: flan-t5-xl sized language model on the HIP path (random weights): encoder, cross K/V, greedy decode timings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine
from eilev_amd.statedict import state_dict_shapes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 960
NEW = 32
cfg = blip2_config("t5xl")
g = torch.Generator(device="cuda"); g.manual_seed(0)
named = {}
for k, shp in state_dict_shapes(cfg).items():
    if not k.startswith("language_model"):
        continue
    if "layer_norm" in k:
        t = torch.ones(shp, device="cuda")
    else:
        std = 0.02 if ".q.weight" not in k else 0.0025
        t = torch.randn(shp, device="cuda", generator=g) * std
    named[k] = t.to(torch.bfloat16)
eng = HipEngine(cfg, named, device="cuda", parts=("t5",))
emb = (0.5 * torch.randn(B, L, 2048, device="cuda", generator=g)).to(torch.bfloat16)
am = torch.ones(B, L, dtype=torch.int32, device="cuda")


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms_enc, enc = timed(lambda: eng.t5_encode(emb, am))
ms_ckv, ckv = timed(lambda: eng.t5_cross_kv(enc))
ms_gen, ids = timed(lambda: eng.t5_greedy(emb, am, NEW, eos_id=-1), n=1)
fl_enc = B * L * 24 * 2 * (4 * 2048 * 2048 + 3 * 2048 * 5120) + 24 * B * 32 * 4 * L * L * 64
print(f"B={B} L={L}: encoder {ms_enc:.1f} ms ({fl_enc / ms_enc / 1e9:.0f} TFLOP/s), cross K/V {ms_ckv:.1f} ms, "
      f"generate({NEW}) total {ms_gen:.1f} ms -> decode ~{(ms_gen - ms_enc - ms_ckv) / NEW:.2f} ms/token")
