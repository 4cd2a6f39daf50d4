import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from bench import random_weights
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine
dev = torch.device("cuda", 0)
cfg = blip2_config("opt27")
w = {k: v for k, v in random_weights(cfg, dev).items() if k.startswith("language_model.")}
eng = HipEngine(cfg, w, device=dev, parts=("opt",))
L = 960
emb = (0.02 * torch.randn(1, L, cfg.text_config.hidden_size, device=dev)).to(torch.bfloat16)
am = torch.ones(1, L, dtype=torch.int32, device=dev)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
for dev_loop in (False, True):
    eng.beam_device_loop = dev_loop
    for graph in (True, False):
        ts = {T: timed(lambda: eng.beam_decode(emb, am, T, 5, length_penalty=-1.0, eos_id=-1, pad_id=1, use_graph=graph)) for T in (16, 48)}
        print(f"device_loop={dev_loop} graph={graph}: T=16 {1e3*ts[16]:.1f} ms, T=48 {1e3*ts[48]:.1f} ms -> {1e3*(ts[48]-ts[16])/32:.3f} ms per extra token, fixed {1e3*(ts[16]-16*(ts[48]-ts[16])/32):.1f} ms", flush=True)
