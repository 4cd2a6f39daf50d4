"""-m gpu: the second named caller of the path — `transformers.Trainer` as ref:scripts/general/train_v2.py:207-217 drives it.

The model class is handed to the stock HF Trainer exactly as the reference script does (freeze ViT + LM, enable_input_require_grads,
bf16 autocast, gradient accumulation, weight decay, grad clipping); every forward / backward runs on the HIP training graph.
Datasets / tokenizer are out of scope: samples are synthetic dicts of the collator's output format.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class _Samples(torch.utils.data.Dataset):
    def __init__(self, meta, n):
        from test_train_golden import _batch

        pixels, input_ids, attn, vmask, labels = _batch(meta)
        # one sample = row 0 of the golden batch (its clips come first in pixel order), repeated with different pixels
        nclips0 = sum(meta["rows"][0][0])
        self.items = []
        rng = np.random.default_rng(0)
        for i in range(n):
            px = pixels[:nclips0] + 0.05 * rng.standard_normal(pixels[:nclips0].shape).astype(np.float32)
            self.items.append(dict(pixel_values=torch.from_numpy(px), input_ids=torch.from_numpy(input_ids[0]),
                                   attention_mask=torch.from_numpy(attn[0]), video_input_mask=torch.from_numpy(vmask[0]),
                                   labels=torch.from_numpy(labels[0])))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _collator():
    """The SHIPPED collator, constructed as train_v2 does (ref:scripts/general/train_v2.py:207-216: tokenizer,
    pad_to_multiple_of=8 under bf16).  The fixture rows were LEFT padded when they were made, so the padding the collator adds
    (right side, training convention) comes on top: ids get pad 1, labels -100, the masks 0."""
    from eilev_amd.data.utils import DataCollatorForInterleavedVideoSeq2Seq
    from tok_utils import tiny_opt_like_tokenizer

    return DataCollatorForInterleavedVideoSeq2Seq(tiny_opt_like_tokenizer("right"), pad_to_multiple_of=8)


def _collate(samples):
    return _collator()([dict(s) for s in samples])


def test_hf_trainer_trains_the_qformer_on_the_hip_graph(tmp_path):
    import transformers

    from eilev_amd.configs import blip2_config
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration
    from oracle.runner import synth_state_dict

    g = np.load(os.path.join(GOLD, "train_tiny_b2.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    model = VideoBlipForConditionalGeneration(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg).items()}, strict=False)
    # ref:scripts/general/train_v2.py:124-130
    for p in model.vision_model.parameters():
        p.requires_grad = False
    for p in model.language_model.parameters():
        p.requires_grad = False
    model.enable_input_require_grads()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    args = transformers.TrainingArguments(
        output_dir=str(tmp_path), per_device_train_batch_size=1, gradient_accumulation_steps=2, max_steps=4, learning_rate=1e-3,
        weight_decay=0.05, warmup_steps=0, bf16=True, remove_unused_columns=False, report_to=[], save_strategy="no", logging_steps=1,
        dataloader_num_workers=0, optim="adamw_torch", seed=0)
    trainer = transformers.Trainer(model=model, args=args, train_dataset=_Samples(meta, 8), data_collator=_collator())
    result = trainer.train()
    losses = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    assert result.global_step == 4 and len(losses) == 4 and all(np.isfinite(losses))
    assert losses[-1] < losses[0], losses  # eight near-identical samples, lr 1e-3: the loss must come down
    after = model.state_dict()
    changed = [k for k in before if not torch.equal(before[k].to(after[k].device), after[k])]
    assert changed and all(k.startswith(("qformer.", "query_tokens", "language_projection.")) for k in changed), changed[:5]
    assert len(changed) >= 40  # every trainable tensor moved (47 at this configuration, a few biases may round to no change)
    # evaluation loop of the Trainer: eval() mode -> the inference route, loss without a graph
    metrics = trainer.evaluate(eval_dataset=_Samples(meta, 2))
    assert np.isfinite(metrics["eval_loss"])


def test_ddp_wrapper_reduces_the_hip_graph_gradients():
    """Under torchrun the Trainer wraps the model in DistributedDataParallel (RCCL).  One-rank process group here: the reducer's hooks
    must fire for every trainable parameter of the HIP graph (`ddp_find_unused_parameters False`, ref:README.md:155)."""
    import socket

    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    from eilev_amd.configs import blip2_config
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration
    from oracle.runner import synth_state_dict

    g = np.load(os.path.join(GOLD, "train_tiny_b2.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    model = VideoBlipForConditionalGeneration(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg).items()}, strict=False)
    model = model.cuda().train()
    model.hip_train_dropout = False
    for p in list(model.vision_model.parameters()) + list(model.language_model.parameters()):
        p.requires_grad = False
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    except Exception as e:  # environment without a usable RCCL transport: nothing to say about the graph
        pytest.skip(f"RCCL process group unavailable here: {e}")
    try:
        ddp = DDP(model, device_ids=[0], find_unused_parameters=False)
        batch = _collate([_Samples(meta, 1)[0]])
        batch = {k: v.cuda() for k, v in batch.items()}
        out = ddp(**batch)
        out.loss.backward()
        torch.cuda.synchronize()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)
        ref = model(**batch)
        assert abs(float(ref.loss.detach()) - float(out.loss.detach())) <= 1e-5 * abs(float(ref.loss.detach()))
    finally:
        dist.destroy_process_group()
