"""``process()``: video tensor -> Blip2Processor -> (B, C, T, H, W) pixel_values (host side).

Same contract as ref:eilev/model/utils.py:5-26 (the sample script imports it): frames are flattened into
the image batch for the HF image processor (bicubic resize to 224, 1/255, CLIP mean/std) and folded back."""
from __future__ import annotations

import torch


def process(processor, video: torch.Tensor | None = None, text=None):
    """video: (batch, channel, time, height, width) or (channel, time, height, width)."""
    shape = None
    if video is not None:
        if video.dim() == 4:
            video = video[None]
        b, c, t = video.shape[:3]
        shape = (b, t, c)
        video = video.transpose(1, 2).reshape(b * t, c, *video.shape[3:])
    inputs = processor(images=video, text=text, return_tensors="pt")
    if shape is not None:
        b, t, c = shape
        px = inputs.pixel_values
        inputs["pixel_values"] = px.reshape(b, t, c, *px.shape[2:]).transpose(1, 2)
    return inputs
