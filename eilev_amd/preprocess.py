"""Device-side image half of ``process()``: uint8 frames -> CLIP-normalised ``pixel_values`` on the GPU.

Host side of ``eilev_process_frames`` (include/eilev.h).  What the reference does on the CPU
[ref:eilev/model/utils.py:5-26 -> Blip2Processor -> BlipImageProcessor, hf:models/blip/image_processing_blip.py]:
PIL BICUBIC resize of every uint8 frame to ``size x size``, ``x * (1/255)`` and ``(x - mean) / std``.  The tables built
here depend only on the sizes / constants and follow Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc``
(src/libImaging/Resample.c) and the HF float pipeline operation by operation, so the HIP kernels (integer
multiply-accumulate, shift, clip, table look-up) reproduce the reference bit for bit.  No CPU fallback: without the HIP
library / a CUDA tensor this raises.
"""
from __future__ import annotations

import ctypes as C
import functools
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # hf:utils/constants.py OPENAI_CLIP_MEAN / _STD (Blip2 preprocessor_config)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=64)
def resample_coeffs(in_size: int, out_size: int):
    """(coef int32 [out_size, ksize], bounds int32 [out_size, 2]) of one axis, as Pillow computes them (Python floats are
    C doubles; same operation order, no fused multiply-add)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            coef[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return coef, bounds


def normalize_lut(mean=CLIP_MEAN, std=CLIP_STD, rescale_factor: float = 1 / 255) -> np.ndarray:
    """fp32 (3, 256): what the HF processor turns byte x of channel c into — ``x * rescale`` in float64 (uint8 array times a
    Python float) cast to float32, then ``(t - mean) / std`` in float32."""
    t = (np.arange(256, dtype=np.uint8) * rescale_factor).astype(np.float32)
    m = np.asarray(mean, np.float32)[:, None]
    s = np.asarray(std, np.float32)[:, None]
    return ((t[None, :] - m) / s).astype(np.float32)


_DEVICE_TABLES: dict = {}


def process_frames(video, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD, rescale_factor: float = 1 / 255, dtype=None, lib=None):
    """uint8 CUDA tensor (B, 3, T, H, W) or (3, T, H, W) -> pixel_values (B, 3, T, size, size) (float32 unless ``dtype`` is
    torch.bfloat16), on the tensor's device and the current stream."""
    import torch

    from . import abi

    if video.dim() == 4:
        video = video[None]
    if video.dim() != 5 or video.shape[1] != 3 or video.dtype != torch.uint8:
        raise ValueError("process_frames expects a uint8 tensor (batch, 3, time, height, width)")
    if not video.is_cuda:
        raise RuntimeError("process_frames runs on the GPU only (HIP library): move the uint8 frames to the device first")
    lib = lib or abi.load_hip()
    video = video.contiguous()
    b, _, t, h_in, w_in = video.shape
    dev = video.device
    out_dtype = torch.float32 if dtype in (None, torch.float32) else dtype
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("dtype must be torch.float32 or torch.bfloat16")

    def table(n_in):
        if n_in == size:
            return None, None, 0
        key = ("axis", int(n_in), int(size), str(dev))
        if key not in _DEVICE_TABLES:  # the tables depend only on the sizes: one upload per (size pair, device)
            coef, bounds = resample_coeffs(int(n_in), int(size))
            _DEVICE_TABLES[key] = (torch.from_numpy(coef).to(dev), torch.from_numpy(bounds).to(dev), coef.shape[1])
        return _DEVICE_TABLES[key]

    ch, bh, kh = table(w_in)
    cv, bv, kv = table(h_in)
    lkey = ("lut", tuple(mean), tuple(std), float(rescale_factor), str(dev))
    if lkey not in _DEVICE_TABLES:
        _DEVICE_TABLES[lkey] = torch.from_numpy(normalize_lut(tuple(mean), tuple(std), rescale_factor)).to(dev)
    lut = _DEVICE_TABLES[lkey]
    out = torch.empty((b, 3, t, size, size), dtype=out_dtype, device=dev)
    nb = lib.eilev_process_workspace_bytes(b, t, h_in, size) if ch is not None else 0
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    abi.check(lib.eilev_process_frames(p(video), b, t, h_in, w_in, size, size, p(ch), p(bh), kh, p(cv), p(bv), kv, p(lut), p(out),
                                       0 if out_dtype == torch.float32 else 1, p(ws), nb, stream), "eilev_process_frames")
    return out
