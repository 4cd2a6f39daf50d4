#!/usr/bin/env python
"""GPU probe: batch-32 greedy decode of OPT-2.7B at L = 960 with / without the Infinity-Cache weight prefetch on a parallel graph branch
(eilev_debug_decode_prefetch): ms per token under hipGraph, variants interleaved.   python tools/decode_prefetch_probe.py [batch=32]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import random_weights
from eilev_amd import abi
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L, new = 960, 32
dev = torch.device("cuda", 0)
cfg = blip2_config("opt27")
w = {k: v for k, v in random_weights(cfg, dev).items() if k.startswith("language_model.")}
eng = HipEngine(cfg, w, device=dev, parts=("opt",))
raw = C.CDLL(abi.HIP_LIB_PATH)
emb = (0.02 * torch.randn(B, L, cfg.text_config.hidden_size, device=dev)).to(torch.bfloat16)
am = torch.ones(B, L, dtype=torch.int32, device=dev)


def run():
    eng._dec_cache = None  # re-capture the step with the current switch
    eng.greedy_decode(emb, am, new, eos_id=-1, pad_id=1, use_graph=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ids = eng.greedy_decode(emb, am, new, eos_id=-1, pad_id=1, use_graph=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, ids


t0 = time.perf_counter(); eng.prefill(emb, am); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.prefill(emb, am); torch.cuda.synchronize(); t_pre = time.perf_counter() - t0
ref = None
for rd in range(2):
    for wg in (0, 64, 128, 256, 512):
        raw.eilev_debug_decode_prefetch(wg)
        t, ids = run()
        if ref is None:
            ref = ids
        print(f"round {rd} prefetch workgroups {wg:4d}: {1e3 * (t - t_pre) / (new - 1):6.3f} ms per token   ids equal: {bool(torch.equal(ids, ref))}", flush=True)
raw.eilev_debug_decode_prefetch(0)
