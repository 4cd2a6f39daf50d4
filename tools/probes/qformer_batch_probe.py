import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine
cfg = blip2_config("opt27"); dev = torch.device("cuda")
w = bench.random_weights(cfg, dev)
eng = HipEngine(cfg, w, device=dev)
Dv = cfg.vision_config.hidden_size
def t(n_clips, reps=3):
    img = (torch.randn(n_clips, 8 * 257, Dv, device=dev) * 0.5).to(torch.bfloat16)
    for _ in range(2): eng.project(eng.qformer(img))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.project(eng.qformer(img))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
a = t(136); b = t(544)
print(f"Q-Former + projection: 136 clips {a:.2f} ms (x4 = {4*a:.2f}), 544 clips {b:.2f} ms")
