"""``process()``: video tensor -> Blip2Processor -> (B, C, T, H, W) pixel_values.

Same contract as ref:eilev/model/utils.py:5-26 (the sample script imports it): frames are flattened into
the image batch for the HF image processor (bicubic resize to 224, 1/255, CLIP mean/std) and folded back.
A uint8 video that already lives on the GPU takes the device path instead (eilev_amd/preprocess.py ->
eilev_process_frames): the same PIL-bicubic / rescale / normalise arithmetic, bit-exact, without the trip through
host memory and PIL."""
from __future__ import annotations

import torch


def process(processor, video: torch.Tensor | None = None, text=None):
    """video: (batch, channel, time, height, width) or (channel, time, height, width)."""
    if video is not None and video.is_cuda and video.dtype == torch.uint8:
        return _process_on_device(processor, video, text)
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


def _process_on_device(processor, video: torch.Tensor, text):
    from transformers import BatchEncoding

    from ..preprocess import CLIP_MEAN, CLIP_STD, process_frames

    ip = getattr(processor, "image_processor", None)
    size, mean, std, rescale = 224, CLIP_MEAN, CLIP_STD, 1 / 255
    if ip is not None:
        sz = getattr(ip, "size", None)
        h = sz["height"] if isinstance(sz, dict) else getattr(sz, "height", 224)
        w = sz["width"] if isinstance(sz, dict) else getattr(sz, "width", 224)
        flags = (getattr(ip, "do_resize", True), getattr(ip, "do_rescale", True), getattr(ip, "do_normalize", True))
        if h != w or not all(flags) or int(getattr(ip, "resample", 3)) != 3:
            raise NotImplementedError("device-side process(): only the Blip2 recipe (square BICUBIC resize, rescale, normalize)")
        size, mean, std = int(h), tuple(ip.image_mean), tuple(ip.image_std)
        rescale = float(getattr(ip, "rescale_factor", 1 / 255))
    inputs = processor(text=text, return_tensors="pt") if text is not None else BatchEncoding({})
    inputs["pixel_values"] = process_frames(video, size=size, mean=mean, std=std, rescale_factor=rescale)
    return inputs
