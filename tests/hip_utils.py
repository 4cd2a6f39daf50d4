"""Helpers for the -m gpu parity tests: build the HIP engine and the CPU oracle on the same weights."""
import ctypes as C
import json
import os

import numpy as np
import torch

from eilev_amd import abi
from eilev_amd.configs import blip2_config
from eilev_amd.synth import round_bf16, synth_pixels
from oracle.runner import OracleModel, synth_state_dict

_cache = {}


def models(cfg_name, mode="fanin", emu=False, seed=0):
    """(config, oracle model, HIP engine) sharing one deterministic bf16-exact state dict."""
    from eilev_amd.engine import HipEngine

    key = (cfg_name, mode, seed)
    if key not in _cache:
        cfg = blip2_config(cfg_name)
        sd = synth_state_dict(cfg, mode, seed)
        eng = HipEngine(cfg, {k: torch.from_numpy(v).cuda() for k, v in sd.items()}, device="cuda")
        _cache[key] = (cfg, sd, eng, {})
    cfg, sd, eng, oracles = _cache[key]
    if emu not in oracles:
        oracles[emu] = OracleModel(cfg, sd, emulate_bf16=emu)
    return cfg, oracles[emu], eng


def load_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    nclips = sum(sum(c) for c, _ in meta["rows"])
    px = synth_pixels(nclips, meta["frames"], cfg.vision_config.image_size)
    return g, meta, px


def dev_bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(torch.bfloat16).contiguous()


def host(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def rel_rms(a, ref):
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.sqrt(((a - ref) ** 2).mean()) / (np.sqrt((ref ** 2).mean()) + 1e-30))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def record_parity(section: str, **numbers):
    """Append measured parity distances to gpurun_out/parity_r06.json (merged back from the GPU box; the copy under
    profiles/ is the tracked one).  Numbers only: what HIP-vs-reference distances actually are, next to the tolerance."""
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_r06.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        with open(path) as fh:
            data = json.load(fh)
    except (OSError, ValueError):
        data = {}
    data.setdefault(section, {}).update({k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v)
                                         for k, v in numbers.items()})
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)
