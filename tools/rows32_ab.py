#!/usr/bin/env python
"""GPU probe: decode step at batch PROBE_B (default 32; hipGraph, L = 960) with gemm_rows32_kernel (default) vs the round-3 kernels
(probe flag 1 << 28), interleaved in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from eilev_amd import abi

abi.use_probes()  # the eilev_debug_* switches live in the probe build only (libeilev_hip_probes.so)
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine
cfg = blip2_config("opt27"); dev = torch.device("cuda")
eng = HipEngine(cfg, bench.random_weights(cfg, dev), device=dev, parts=("opt",))
raw = C.CDLL(abi.HIP_LIB_PATH)
B, L, NEW = int(os.environ.get("PROBE_B", "32")), 960, 32
emb = (torch.randn(B, L, cfg.text_config.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
am = torch.ones(B, L, dtype=torch.int32, device=dev)
outs = {}
for rd in range(6):
    flag = (1 << 28) if rd % 2 else 0
    if os.environ.get("PROBE_SWITCH") == "reduce_ln":  # A/B of the split-K reduce + LayerNorm kernel instead (odd rounds: one wave per row)
        pass  # (the one-wave reduce kernel was removed in round 5)
    else:
        raw.eilev_debug_gemm_flags(flag)
    eng._dec_cache = None
    e0, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.timing = []; e0.record()
    out = eng.greedy_decode(emb, am, NEW, eos_id=-1, pad_id=1, use_graph=True)
    e2.record(); torch.cuda.synchronize()
    pre = e0.elapsed_time(dict(eng.timing)["prefill_done"])
    outs[flag] = out.cpu()
    print(f"round {rd} ({'round-3 kernels' if flag else 'rows32'}): decode {(e0.elapsed_time(e2) - pre) / (NEW - 1):.3f} ms/token", flush=True)
raw.eilev_debug_gemm_flags(0)
print("ids equal between the two:", float((outs[0] == outs[1 << 28]).float().mean()))
