#!/usr/bin/env python
"""Time one train_v2 step (loss + backward) of the HIP training graph at the headline configuration.

    python tools/train_probe.py [--samples 1] [--steps 3] [--profile]

Random-init weights of Salesforce/blip2-opt-2.7b's architecture, 16-shot samples of bench.py's shape (17 clips x 8 frames,
960 tokens per sample), labels on the text positions; ViT and LM frozen, Q-Former + query tokens + projection (107 M
parameters, fp32 masters) trainable as ref:scripts/general/train_v2.py:124-130 sets them.  Prints per-stage times
(ViT forward / graph forward / backward) and the peak memory.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from eilev_amd.configs import blip2_config  # noqa: E402
from eilev_amd.engine import HipEngine  # noqa: E402
from eilev_amd.train import TrainGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--config", default="opt27")
    ap.add_argument("--dropout", action="store_true", help="apply the configuration's dropout (train() mode of the reference)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = blip2_config(args.config)
    w = bench.random_weights(cfg, dev)
    eng = HipEngine(cfg, w, device=dev)
    params = {k: v.float().requires_grad_(True) for k, v in w.items() if k.startswith(("qformer.", "query_tokens", "language_projection."))}
    n_par = sum(p.numel() for p in params.values())
    px, ids, vm, am = bench.build_inputs(cfg, args.samples, dev)
    if eng.is_t5:  # encoder-decoder LM: the targets are the 14 tokens of the last narration (decoder side), ids stay below the vocabulary
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        ids = ids.clamp(max=cfg.text_config.vocab_size - 1)
        labels = torch.randint(4, cfg.text_config.vocab_size, (args.samples, 14), device=dev, generator=g)
    else:
        labels = torch.where(vm == 0, ids, torch.full_like(ids, -100))
    opt = torch.optim.AdamW(list(params.values()), lr=1e-5)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for step in range(args.steps + 1):
        torch.cuda.reset_peak_memory_stats()
        e = [ev() for _ in range(5)]
        e[0].record()
        with torch.no_grad():
            eng.vit(px)  # timed alone; loss() below runs it again
        e[1].record()
        loss = TrainGraph(eng, params, dropout=args.dropout, seed=step).loss(ids, am, px, vm, labels)
        e[2].record()
        loss.backward()
        e[3].record()
        opt.step()
        opt.zero_grad(set_to_none=True)
        e[4].record()
        torch.cuda.synchronize()
        vit, fwd, bwd, ost = (e[i].elapsed_time(e[i + 1]) for i in range(4))
        tag = "warm-up" if step == 0 else f"step {step}"
        last = dict(step_ms=round(fwd + bwd + ost, 2), vit_ms=round(vit, 2), graph_forward_ms=round(fwd - vit, 2), backward_ms=round(bwd, 2),
                    optimizer_ms=round(ost, 2), peak_gib=round(torch.cuda.max_memory_allocated() / 2**30, 2))
        print(f"{tag}: loss {float(loss.detach()):.4f}  vit {vit:.1f} ms  forward(vit+graph) {fwd:.1f} ms  backward {bwd:.1f} ms  adamw {ost:.1f} ms  "
              f"step {fwd + bwd + ost:.1f} ms  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  ({n_par / 1e6:.1f} M trainable, "
              f"{args.samples * 17} clips, {ids.shape[1]} tokens/sample)", flush=True)
    import json

    print(json.dumps({"metric": "train_v2 step (ViT + LM frozen, Q-Former trained), one MI355X", "config": args.config, "samples_per_step": args.samples,
                      "clips_per_step": args.samples * 17, "dropout": bool(args.dropout), "trainable_params": n_par,
                      "clips_per_s": round(args.samples * 17 / (last["step_ms"] / 1e3), 1), **last}))


if __name__ == "__main__":
    main()
