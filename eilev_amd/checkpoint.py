"""On-disk weight format of the released checkpoints, read without the HF loader: ``config.json`` + ``model.safetensors``
(or ``model.safetensors.index.json`` + shards) -> tensors straight onto the GPU -> ``HipEngine``.

The safetensors container: 8-byte little-endian header length N, N bytes of JSON ``{name: {"dtype", "shape",
"data_offsets": [begin, end]}, "__metadata__": {...}}``, then the raw little-endian tensor bytes (offsets relative to the
end of the header).  Files are memory-mapped; every tensor is uploaded once (bf16 bytes stay bf16: no float32 detour).
State-dict keys are the reference class's (SURVEY §8 a-W); tied weights that the file omits (``lm_head.weight``) are
resolved by the engine's weight pack as in the HF path.
"""
from __future__ import annotations

import json
import mmap
import os
import struct

import numpy as np

_NP = {"F64": np.float64, "F32": np.float32, "F16": np.float16, "BF16": np.uint16, "I64": np.int64, "I32": np.int32, "I16": np.int16,
       "I8": np.int8, "U8": np.uint8, "BOOL": np.bool_}


def read_header(path: str):
    """(header dict without __metadata__, data start offset, metadata dict)."""
    with open(path, "rb") as fh:
        raw = fh.read(8)
        if len(raw) != 8:
            raise ValueError(f"{path}: not a safetensors file (shorter than its length prefix)")
        (n,) = struct.unpack("<Q", raw)
        if n <= 0 or n > 100 * 1024 * 1024:
            raise ValueError(f"{path}: implausible safetensors header length {n}")
        hdr = json.loads(fh.read(n).decode("utf-8"))
    meta = hdr.pop("__metadata__", {}) or {}
    size = os.path.getsize(path)
    for name, e in hdr.items():
        b, end = e["data_offsets"]
        if e["dtype"] not in _NP:
            raise ValueError(f"{path}: tensor {name!r} has unsupported dtype {e['dtype']}")
        want = int(np.prod(e["shape"], dtype=np.int64)) * np.dtype(_NP[e["dtype"]]).itemsize
        if not (0 <= b <= end and 8 + n + end <= size and end - b == want):
            raise ValueError(f"{path}: tensor {name!r} has inconsistent offsets / shape")
    return hdr, 8 + n, meta


def iter_tensors(path: str):
    """Yield (name, numpy view, is_bf16) for every tensor of one file; bf16 comes as its uint16 bit pattern."""
    hdr, start, _ = read_header(path)
    with open(path, "rb") as fh:
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    for name, e in hdr.items():
        b, end = e["data_offsets"]
        arr = np.frombuffer(mm, dtype=_NP[e["dtype"]], count=(end - b) // np.dtype(_NP[e["dtype"]]).itemsize, offset=start + b)
        yield name, arr.reshape(e["shape"]), e["dtype"] == "BF16"


def checkpoint_files(model_dir: str):
    single = os.path.join(model_dir, "model.safetensors")
    index = os.path.join(model_dir, "model.safetensors.index.json")
    if os.path.exists(single):
        return [single]
    if os.path.exists(index):
        with open(index) as fh:
            wm = json.load(fh)["weight_map"]
        return [os.path.join(model_dir, f) for f in sorted(set(wm.values()))]
    raise FileNotFoundError(f"{model_dir}: neither model.safetensors nor model.safetensors.index.json")


def _from_readonly(arr):
    import warnings

    import torch

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)  # "non-writable array": the tensor is only read (copied to the device)
        return torch.from_numpy(arr)


def load_state_dict(model_dir: str, device="cpu", dtype=None):
    """{key: torch tensor on ``device``}.  ``dtype`` (e.g. torch.bfloat16) converts floating tensors on the device."""
    import torch

    out = {}
    for path in checkpoint_files(model_dir):
        for name, arr, is_bf16 in iter_tensors(path):
            # CPU: own the bytes (the mapping is read-only); GPU: upload straight from the mapping
            t = torch.from_numpy(arr.copy()) if str(device) == "cpu" else _from_readonly(arr).to(device)
            if is_bf16:
                t = t.view(torch.bfloat16)
            if dtype is not None and t.is_floating_point() and t.dtype != dtype:
                t = t.to(dtype)
            out[name] = t
    return out


def engine_from_checkpoint(model_dir: str, device="cuda", parts=None):
    """``HipEngine`` for a checkpoint directory (config.json + safetensors) without instantiating the HF model class."""
    from transformers import Blip2Config

    from .engine import HipEngine

    config = Blip2Config.from_pretrained(model_dir)
    return HipEngine(config, load_state_dict(model_dir, device=device), device=device, parts=parts)
