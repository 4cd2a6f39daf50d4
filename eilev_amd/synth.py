"""Deterministic synthetic weights and inputs.

There are no checkpoints, datasets or tokenizers on the build or GPU machines,
so every test, the smoke check and the benchmark draw their tensors from this
counter-based generator.  It is pure integer arithmetic (splitmix64 over
``fnv1a64(name) ^ seed + index``) followed by an exact int->float conversion,
so the same (name, shape, seed) yields bit-identical values on any machine,
any numpy version and any thread count.  ``tools/make_goldens.py`` feeds the
reference model with exactly these tensors; the golden fixtures under
``tests/golden/`` therefore only need to store the *outputs*.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    # x is uint64; numpy wraps on overflow for arrays.
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def det_normal(name: str, shape, seed: int = 0) -> np.ndarray:
    """Approximately N(0,1) float32 tensor (Irwin-Hall of four 16-bit uniforms)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + base
    z = _splitmix64(idx)
    s = np.zeros(n, dtype=np.int64)
    for k in range(4):
        s += ((z >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.int64)
    # sum of four U{0..65535}: mean 2*65535, variance 4*(65536^2-1)/12
    x = (s - 2 * 65535).astype(np.float64) / np.sqrt(4.0 * (65536.0**2 - 1.0) / 12.0)
    return x.astype(np.float32).reshape(shape)


def det_uniform_int(name: str, shape, lo: int, hi: int, seed: int = 0) -> np.ndarray:
    """Integers uniform in [lo, hi) as int64."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + base
    z = _splitmix64(idx) >> np.uint64(11)
    return (lo + (z % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest bfloat16 (ties to even), return as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def synth_param(name: str, shape, mode: str = "fanin", seed: int = 0) -> np.ndarray:
    """One parameter tensor of the VideoBLIP state dict, by its HF name.

    ``mode='fanin'``: matrices ~ N(0, 1/fan_in) (activations stay O(1), attention is
    non-trivial: good for numerics tests).  ``mode='hf'``: matrices ~ N(0, 0.02) as
    HF ``initializer_range`` (what the benchmark uses).  LayerNorm weights are
    1 + 0.1 n, LayerNorm biases 0.05 n, linear biases 0.02 n; every value is
    representable in bf16 so fp32 and bf16 models hold identical weights.
    ``mode='varied'``: 'fanin' with a SMALL token embedding (std 0.06) and the usual 0.5 for position embeddings.  With the tied
    lm_head of OPT, 'fanin' makes a random model repeat one token for ever (the residual stream is dominated by the last token's own
    embedding, whose logit is its squared norm); with the token embedding small the stream is dominated by position and block outputs,
    greedy / beam outputs change from step to step, and a decode step that used a wrong position or a stale KV slot shows up in the ids.
    """
    shape = tuple(int(s) for s in shape)
    n = det_normal(name, shape, seed)
    low = name.lower()
    if "layernorm" in low or "layer_norm" in low:
        out = 1.0 + 0.1 * n if name.endswith("weight") else 0.05 * n
    elif name.endswith(".bias"):
        out = 0.02 * n
    elif len(shape) >= 2 and name.endswith("weight") and "embed" not in low:
        fan_in = int(np.prod(shape[1:]))
        std = (1.0 / np.sqrt(fan_in)) if mode in ("fanin", "varied") else 0.02
        if mode in ("fanin", "varied") and low.endswith("attention.q.weight"):
            # T5 attention has no 1/sqrt(d_kv) factor (trained checkpoints carry it in q): without it the synthetic softmax
            # saturates and the model amplifies bf16 noise chaotically
            std *= 0.125
        out = std * n
    else:
        # embeddings, query_tokens, class/position embeddings
        std = 0.02 if mode == "hf" else (0.06 if mode == "varied" and "embed_tokens" in low else 0.5)
        out = std * n
    return round_bf16(out.astype(np.float32))


def synth_pixels(num_clips: int, frames: int, image_size: int, seed: int = 1234) -> np.ndarray:
    """(N, 3, T, H, W) float32 ~ N(0,1) clipped to +-2.5 (CLIP-normalised range), bf16-exact."""
    x = det_normal("pixel_values", (num_clips, 3, frames, image_size, image_size), seed)
    return round_bf16(np.clip(x, -2.5, 2.5))


def synth_interleaved_ids(
    clips_per_block, text_lens, num_query_tokens: int, vocab: int,
    bos: int = 2, pad: int = 1, newline: int | None = 50118, seed: int = 1,
):
    """Integer restatement of the decoder-only branch of the reference's
    ``generate_input_ids_and_labels_from_interleaved`` (ref:eilev/data/utils.py:170-190)
    with synthetic text tokens: [bos] + per block (clips * ([pad]*nq + [nl]) + text).

    ``text_lens[i]`` counts the tokens of block i *including* its trailing newline
    (all blocks but the last end with one).  Returns (input_ids, video_input_mask).
    """
    nl = newline if newline is not None and newline < vocab else min(3, vocab - 1)
    lo = 4
    ids, mask = [bos], [0]
    nblocks = len(text_lens)
    for i, (nclip, tl) in enumerate(zip(clips_per_block, text_lens)):
        for _ in range(nclip):
            ids += [pad] * num_query_tokens + [nl]
            mask += [1] * num_query_tokens + [0]
        ntext = tl - (0 if i == nblocks - 1 else 1)
        toks = det_uniform_int(f"text_block_{i}", (ntext,), lo, max(lo + 1, min(vocab, 50000)), seed).tolist()
        if i != nblocks - 1:
            toks.append(nl)
        ids += toks
        mask += [0] * len(toks)
    return np.asarray(ids, dtype=np.int64), np.asarray(mask, dtype=np.int64)


# ---- the same generator on a torch device (bit-identical; tests/test_synth_torch.py) ---------------------------------------------------
# The full-depth fixture (tests/golden/full_c1.npz) needs the 3.8 G parameters of eilev-blip2-opt-2.7b by recipe on the GPU box: numpy
# takes ~6 minutes for them (page faults on GB-sized temporaries), the device a few seconds.  Integer arithmetic wraps identically in
# int64; the float steps are single correctly-rounded IEEE operations in the same order and types as the numpy code above.
def _to_i64(v: int) -> int:
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def det_normal_torch(name: str, shape, seed: int = 0, device="cpu"):
    import torch

    n = 1
    for s_ in shape:
        n *= int(s_)
    lsr = lambda z, k: (z >> k) & ((1 << (64 - k)) - 1)
    base = (fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    z = torch.arange(n, dtype=torch.int64, device=device) + _to_i64(base) + _to_i64(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * _to_i64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _to_i64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    s = (z & 0xFFFF) + (lsr(z, 16) & 0xFFFF) + (lsr(z, 32) & 0xFFFF) + (lsr(z, 48) & 0xFFFF)
    x = (s - 2 * 65535).to(torch.float64) / float(np.sqrt(4.0 * (65536.0**2 - 1.0) / 12.0))
    return x.to(torch.float32).reshape(tuple(int(s_) for s_ in shape))


def round_bf16_torch(x):
    import torch

    u = x.contiguous().view(torch.int32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & -65536
    return r.view(torch.float32)


def synth_param_torch(name: str, shape, mode: str = "fanin", seed: int = 0, device="cpu"):
    """`synth_param` evaluated on a torch device: float32 tensor holding bf16-exact values, bit-identical to the numpy version."""
    import torch

    shape = tuple(int(s_) for s_ in shape)
    n = det_normal_torch(name, shape, seed, device)
    low = name.lower()
    if "layernorm" in low or "layer_norm" in low:
        out = 1.0 + 0.1 * n if name.endswith("weight") else 0.05 * n
    elif name.endswith(".bias"):
        out = 0.02 * n
    elif len(shape) >= 2 and name.endswith("weight") and "embed" not in low:
        fan_in = int(np.prod(shape[1:]))
        if mode in ("fanin", "varied"):
            std = 1.0 / np.sqrt(fan_in)            # np.float64: the numpy product is evaluated in float64 ...
            if low.endswith("attention.q.weight"):
                std = std * 0.125
            out = (n.to(torch.float64) * float(std)).to(torch.float32)   # ... and cast back: same here
        else:
            out = 0.02 * n                          # python float: float32 arithmetic in both
    else:
        std = 0.02 if mode == "hf" else (0.06 if mode == "varied" and "embed_tokens" in low else 0.5)
        out = std * n
    return round_bf16_torch(out.to(torch.float32))
