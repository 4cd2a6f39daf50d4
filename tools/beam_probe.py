#!/usr/bin/env python
"""GPU probe: beam search (the sample script's default: num_beams=5, length_penalty=-1, max_new_tokens=32) on the OPT-2.7B language model
at the 16-shot prompt length L = 960, random-init weights: ms per generated token of eilev_amd.engine.HipEngine.beam_decode
(eilev_opt_decode_step_beam: no cache moves, step captured into a hipGraph), next to a plain greedy decode of the same number of rows.

    python tools/beam_probe.py [beams=5] [batch=1] [new=32]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import random_weights
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine

beams = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
new = int(sys.argv[3]) if len(sys.argv) > 3 else 32
L = 960
dev = torch.device("cuda", 0)
cfg = blip2_config("opt27")
w = {k: v for k, v in random_weights(cfg, dev).items() if k.startswith("language_model.")}
eng = HipEngine(cfg, w, device=dev, parts=("opt",))
emb = (0.02 * torch.randn(B, L, cfg.text_config.hidden_size, device=dev)).to(torch.bfloat16)
am = torch.ones(B, L, dtype=torch.int32, device=dev)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


t_pre, _ = timed(lambda: eng.prefill(emb, am))
for graph in (True, False):
    t, ids = timed(lambda: eng.beam_decode(emb, am, new, beams, length_penalty=-1.0, eos_id=-1, pad_id=1, use_graph=graph))
    print(f"beam search  beams={beams} batch={B} L={L} new={ids.shape[1]} graph={graph}: {1e3 * t:8.1f} ms total, prefill {1e3 * t_pre:6.1f} ms, "
          f"{1e3 * (t - t_pre) / max(1, ids.shape[1] - 1):6.3f} ms per generated token (host selection included)", flush=True)
rows = torch.cat([emb] * beams)[: beams * B]
amr = torch.cat([am] * beams)[: beams * B]
t_pre_r, _ = timed(lambda: eng.prefill(rows, amr))
t, ids = timed(lambda: eng.greedy_decode(rows, amr, new, eos_id=-1, pad_id=1, use_graph=True))
print(f"greedy       rows={beams * B} L={L} new={new} graph=True: {1e3 * t:8.1f} ms total, prefill {1e3 * t_pre_r:6.1f} ms, "
      f"{1e3 * (t - t_pre_r) / (new - 1):6.3f} ms per token")
