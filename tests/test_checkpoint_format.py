"""On-disk format (safetensors) read without the HF loader: header parsing, dtypes incl. bf16, sharded checkpoints, corrupt files."""
import json
import os
import struct

import numpy as np
import pytest
import torch

from eilev_amd import checkpoint


def _write(path, tensors):  # reference writer: the safetensors library itself
    from safetensors.torch import save_file

    save_file(tensors, path, metadata={"format": "pt"})


def test_single_file_round_trip_matches_safetensors_library(tmp_path):
    from safetensors.torch import load_file

    g = torch.Generator().manual_seed(0)
    tensors = {"a.weight": torch.randn(5, 7, generator=g).to(torch.bfloat16), "b.bias": torch.randn(9, generator=g),
               "ids": torch.arange(6, dtype=torch.int64).reshape(2, 3), "h": torch.randn(3, 2, generator=g).to(torch.float16),
               "scalar": torch.tensor(3.5)}
    p = str(tmp_path / "model.safetensors")
    _write(p, tensors)
    hdr, start, meta = checkpoint.read_header(p)
    assert meta == {"format": "pt"} and set(hdr) == set(tensors) and start > 8
    got = checkpoint.load_state_dict(str(tmp_path))
    ref = load_file(p)
    assert set(got) == set(ref)
    for k in ref:
        assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape and torch.equal(got[k], ref[k]), k
    as_bf16 = checkpoint.load_state_dict(str(tmp_path), dtype=torch.bfloat16)
    assert as_bf16["b.bias"].dtype == torch.bfloat16 and as_bf16["ids"].dtype == torch.int64
    assert torch.equal(as_bf16["b.bias"], ref["b.bias"].to(torch.bfloat16))


def test_sharded_checkpoint_with_index(tmp_path):
    a = {"x.weight": torch.ones(4, 4, dtype=torch.bfloat16)}
    b = {"y.weight": torch.full((2, 3), 2.0)}
    _write(str(tmp_path / "model-00001-of-00002.safetensors"), a)
    _write(str(tmp_path / "model-00002-of-00002.safetensors"), b)
    with open(tmp_path / "model.safetensors.index.json", "w") as fh:
        json.dump({"metadata": {}, "weight_map": {"x.weight": "model-00001-of-00002.safetensors", "y.weight": "model-00002-of-00002.safetensors"}}, fh)
    got = checkpoint.load_state_dict(str(tmp_path))
    assert torch.equal(got["x.weight"], a["x.weight"]) and torch.equal(got["y.weight"], b["y.weight"])


def test_corrupt_files_are_rejected(tmp_path):
    p = str(tmp_path / "model.safetensors")
    with open(p, "wb") as fh:
        fh.write(b"\x01\x02")
    with pytest.raises(ValueError):
        checkpoint.read_header(p)
    hdr = json.dumps({"w": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}).encode()
    with open(p, "wb") as fh:
        fh.write(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 8)  # data shorter than the offsets claim
    with pytest.raises(ValueError):
        checkpoint.read_header(p)
    hdr = json.dumps({"w": {"dtype": "F32", "shape": [3], "data_offsets": [0, 16]}}).encode()
    with open(p, "wb") as fh:
        fh.write(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 16)  # shape does not match the byte range
    with pytest.raises(ValueError):
        checkpoint.read_header(p)
    with pytest.raises(FileNotFoundError):
        checkpoint.checkpoint_files(str(tmp_path / "nope"))


@pytest.mark.gpu
def test_engine_from_checkpoint_equals_engine_from_model(tmp_path):
    from eilev_amd import configs
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration

    cfg = configs.blip2_config("tiny")
    model = VideoBlipForConditionalGeneration(cfg).to(torch.bfloat16)
    model.save_pretrained(str(tmp_path), safe_serialization=True)
    eng = checkpoint.engine_from_checkpoint(str(tmp_path), device="cuda")
    ref = model.to("cuda").engine()
    px = torch.randn(2, 3, 2, cfg.vision_config.image_size, cfg.vision_config.image_size, device="cuda").to(torch.bfloat16)
    a, b = eng.vit(px), ref.vit(px)
    a, b = (a[0] if isinstance(a, tuple) else a), (b[0] if isinstance(b, tuple) else b)
    assert torch.equal(a, b)
