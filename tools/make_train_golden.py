#!/usr/bin/env python
"""Golden loss + gradients of the train_v2 step, from the REFERENCE model (run in the build container only).

    python tools/make_train_golden.py

Builds ``eilev.model.v2.VideoBlipForConditionalGeneration`` from /root/reference (loaded by path, never copied) with the
deterministic weights / inputs of ``eilev_amd.synth``, freezes the ViT and the language model exactly as
ref:scripts/general/train_v2.py:124-130 does (including ``enable_input_require_grads``), runs
``loss = model(**batch, labels=...).loss; loss.backward()`` in fp32 on the CPU and stores the loss, every trainable
gradient's norm and a few gradients in full under tests/golden/train_<case>.npz.  Dropout is off (``model.eval()``): the
HIP training graph is the deterministic function (eilev_amd/train.py).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from make_goldens import RefModel, build_inputs, load_det_weights  # noqa: E402

from eilev_amd.configs import blip2_config  # noqa: E402

CASES = {
    # name: (config, frames, rows, padding side)
    "tiny_b2": ("tiny", 1, [([1, 1], [4, 6]), ([2], [3])], "left"),
    "mid_b2": ("mid", 2, [([1, 1], [6, 7]), ([1, 1], [3, 3])], "left"),
    # the training collator pads on the RIGHT (ref:eilev/data/utils.py:35-66 with the OPT tokenizer's default padding side)
    "mid_b2_right": ("mid", 2, [([1, 1], [6, 7]), ([1], [3])], "right"),
}
FULL = ["query_tokens", "language_projection.weight", "language_projection.bias", "qformer.layernorm.weight",
        "qformer.encoder.layer.0.crossattention.attention.key.weight", "qformer.encoder.layer.0.crossattention.attention.value.bias",
        "qformer.encoder.layer.0.attention.attention.query.weight", "qformer.encoder.layer.1.intermediate_query.dense.weight",
        "qformer.encoder.layer.1.output_query.LayerNorm.weight", "qformer.encoder.layer.1.output_query.LayerNorm.bias"]


T5_CASES = {
    # name: (config, frames, rows, target length) — encoder batches are right padded, the last target of row 1 is padding (-100)
    "tiny_t5_b2": ("tiny_t5", 1, [([1, 1], [4, 6]), ([2], [3])], 5),
    "mid_t5_b2": ("mid_t5", 2, [([1, 1], [6, 7]), ([1, 1], [3, 3])], 6),
}
T5_FULL = ["query_tokens", "language_projection.weight", "language_projection.bias",
           "qformer.encoder.layer.0.crossattention.attention.key.weight", "qformer.encoder.layer.1.output_query.LayerNorm.weight"]


def right_pad(a, attn, fill):
    """Move the left padding of build_inputs to the right of every row."""
    out = np.full_like(a, fill)
    for b in range(a.shape[0]):
        n = int(attn[b].sum())
        out[b, :n] = a[b, a.shape[1] - n:]
    return out


def run(name):
    cfg_name, frames, rows, side = CASES[name]
    cfg = blip2_config(cfg_name)
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    for p in model.vision_model.parameters():
        p.requires_grad = False
    for p in model.language_model.parameters():
        p.requires_grad = False
    model.enable_input_require_grads()
    pixels, input_ids, attn, vmask, labels = build_inputs(cfg_name, frames, rows)
    if side == "right":
        input_ids, attn, vmask, labels = (right_pad(a, attn, fill) for a, fill in ((input_ids, 1), (attn, 0), (vmask, 0), (labels, -100)))
    t = lambda a: torch.from_numpy(a)
    out = model(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=t(pixels), video_input_mask=t(vmask), labels=t(labels),
                return_dict=True)
    out.loss.backward()
    grads = {k: p.grad.numpy() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    unused = [k for k, p in model.named_parameters() if p.requires_grad and p.grad is None]
    save = {"loss": np.asarray(float(out.loss), np.float64),
            "meta": json.dumps({"config": cfg_name, "frames": frames, "rows": rows, "unused": unused, "pad": side}),
            "norm_keys": np.array(sorted(grads)), "norms": np.array([float(np.linalg.norm(grads[k])) for k in sorted(grads)], np.float64)}
    for k in FULL:
        if k in grads:
            save["grad::" + k] = grads[k].astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", f"train_{name}.npz")
    np.savez_compressed(path, **save)
    print(name, "loss", float(out.loss), "trainable grads", len(grads), "unused", len(unused), os.path.getsize(path), "bytes")


def run_t5(name):
    """flan-t5 family: `model(..., labels=targets).loss.backward()` (ref:eilev/model/v2.py:228-238)."""
    cfg_name, frames, rows, tgt_len = T5_CASES[name]
    cfg = blip2_config(cfg_name)
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    for p in model.vision_model.parameters():
        p.requires_grad = False
    for p in model.language_model.parameters():
        p.requires_grad = False
    model.enable_input_require_grads()
    pixels, input_ids, attn, vmask, _ = build_inputs(cfg_name, frames, rows)
    input_ids, vmask, attn = right_pad(input_ids, attn, 0), right_pad(vmask, attn, 0), right_pad(attn, attn, 0)
    B = input_ids.shape[0]
    rng = np.random.default_rng(11)
    labels = rng.integers(2, cfg.text_config.vocab_size, size=(B, tgt_len)).astype(np.int64)
    if B > 1:
        labels[1, tgt_len - 2:] = -100
    t = lambda a: torch.from_numpy(a)
    out = model(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=t(pixels), video_input_mask=t(vmask), labels=t(labels),
                return_dict=True)
    out.loss.backward()
    grads = {k: p.grad.numpy() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    save = {"loss": np.asarray(float(out.loss.detach()), np.float64), "labels": labels,
            "meta": json.dumps({"config": cfg_name, "frames": frames, "rows": rows, "pad": "right", "t5": True}),
            "norm_keys": np.array(sorted(grads)), "norms": np.array([float(np.linalg.norm(grads[k])) for k in sorted(grads)], np.float64)}
    for k in T5_FULL:
        if k in grads:
            save["grad::" + k] = grads[k].astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", f"train_{name}.npz")
    np.savez_compressed(path, **save)
    print(name, "loss", float(out.loss.detach()), "trainable grads", len(grads), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    for n in CASES:
        run(n)
    for n in T5_CASES:
        run_t5(n)
