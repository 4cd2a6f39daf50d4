import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from eilev_amd.configs import blip2_config
from eilev_amd.engine import HipEngine
from eilev_amd.synth import synth_pixels
from oracle.runner import synth_state_dict
cfg = blip2_config("mid")
dev = torch.device("cuda", 0)
for mode, seed in (("fanin", 0), ("varied", 176)):
    sd = synth_state_dict(cfg, mode, seed)
    eng = HipEngine(cfg, {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}, device=dev)
    for clips, frames in ((3, 2), (4, 2), (2, 2), (4, 1), (8, 2)):
        px = synth_pixels(clips, frames, cfg.vision_config.image_size)
        try:
            f = eng.encode_clips(torch.from_numpy(px).to(dev))
            torch.cuda.synchronize()
            print(mode, clips, frames, "ok", tuple(f.shape), flush=True)
        except Exception as e:
            print(mode, clips, frames, "FAIL", e, flush=True)
