#!/usr/bin/env python
"""GPU probe: ms per decode step (hipGraph) at batch PROBE_B (default 32) / L = 960 with the bench's OPT-2.7B, several rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine

from eilev_amd import abi

if os.environ.get("PROBE_LIB"):  # A/B against another build of the probe library
    abi.PROBES_LIB_PATH = os.path.abspath(os.environ["PROBE_LIB"])
abi.use_probes()  # the eilev_debug_* switches live in the probe build only (libeilev_hip_probes.so)
cfg = blip2_config("opt27")
dev = torch.device("cuda")
w = bench.random_weights(cfg, dev)
eng = HipEngine(cfg, w, device=dev, parts=("opt",))
B, L, NEW = int(os.environ.get("PROBE_B", "32")), 960, 32
emb = (torch.randn(B, L, cfg.text_config.hidden_size, device=dev) * 0.02).to(torch.bfloat16)
am = torch.ones(B, L, dtype=torch.int32, device=dev)
import ctypes as C
from eilev_amd import abi

raw = C.CDLL(abi.HIP_LIB_PATH)
for rd in range(6):
    flag = int(os.environ.get("PROBE_FLAG", "536870912")) if rd % 2 else 0  # odd rounds: the probe flag (default: split-K reduce and LayerNorm as two launches)
    raw.eilev_debug_gemm_flags(flag)
    if os.environ.get("PROBE_CALL"):  # e.g. eilev_debug_attn_part32: called with 1 on odd rounds, 0 on even ones
        getattr(raw, os.environ["PROBE_CALL"])(int(os.environ.get("PROBE_CALL_ON", "1")) if rd % 2 else int(os.environ.get("PROBE_CALL_OFF", "0")))
    eng._dec_cache = None  # re-capture the decode graph under this setting
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    eng.timing = []
    e0.record()
    out = eng.greedy_decode(emb, am, NEW, eos_id=-1, pad_id=1, use_graph=True)
    e2.record()
    torch.cuda.synchronize()
    marks = dict(eng.timing)
    pre = e0.elapsed_time(marks["prefill_done"])
    print(f"round {rd} ({'probe flag ' + str(flag) if flag else 'default'}): prefill {pre:.1f} ms, decode {(e0.elapsed_time(e2) - pre) / (NEW - 1):.3f} ms/token", flush=True)
raw.eilev_debug_gemm_flags(0)
