#!/usr/bin/env python
"""Generate golden fixtures from the REFERENCE implementation (run in the build container only).

    PYTHONPATH=/root/reference python tools/make_goldens.py

Imports ``eilev.model.v2.VideoBlipForConditionalGeneration`` from /root/reference (never
copied, never shipped) together with the installed ``transformers``; feeds it the
deterministic tensors of ``eilev_amd.synth`` and stores only inputs-by-recipe and
OUTPUTS as small ``.npz`` files under tests/golden/.  The fixtures pin the CPU oracle
(``oracle/``) in fp32 and give the bf16 reference outputs the HIP path is judged against.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import importlib.util  # noqa: E402

import transformers  # noqa: E402

# The REFERENCE class, loaded by file path (this repo also ships an `eilev` alias package, so a plain
# `import eilev` would not reach /root/reference).
_spec = importlib.util.spec_from_file_location("reference_eilev_model_v2", "/root/reference/eilev/model/v2.py")
_ref = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_ref)
RefModel = _ref.VideoBlipForConditionalGeneration
assert RefModel.__module__ == "reference_eilev_model_v2"

from eilev_amd.configs import CONFIGS, blip2_config  # noqa: E402
from eilev_amd.synth import synth_interleaved_ids, synth_param, synth_pixels  # noqa: E402

CASES = {
    # name: (config, frames T, rows=[(clips_per_block, text_lens)], new_tokens)
    "tiny_b1": ("tiny", 2, [([1, 1, 1], [5, 5, 4])], 6),
    "tiny_b2": ("tiny", 1, [([1, 1], [4, 6]), ([2], [3])], 6),
    "mid_b1": ("mid", 2, [([1, 1, 1], [5, 5, 4])], 6),
    "mid_b2": ("mid", 2, [([1, 1], [6, 7]), ([1, 1], [3, 3])], 6),
}


def build_inputs(cfg_name, frames, rows):
    c = CONFIGS[cfg_name]
    nq = c["num_query_tokens"]
    vocab = c["text_config"]["vocab_size"]
    image = c["vision_config"]["image_size"]
    ids_rows, mask_rows = [], []
    for r, (clips, lens) in enumerate(rows):
        ids, vm = synth_interleaved_ids(clips, lens, nq, vocab, seed=1 + r)
        ids_rows.append(ids)
        mask_rows.append(vm)
    L = max(len(x) for x in ids_rows)
    B = len(rows)
    input_ids = np.full((B, L), 1, dtype=np.int64)  # pad id 1, LEFT padding (generation convention)
    attn = np.zeros((B, L), dtype=np.int64)
    vmask = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = len(ids_rows[b])
        input_ids[b, L - n:] = ids_rows[b]
        attn[b, L - n:] = 1
        vmask[b, L - n:] = mask_rows[b]
    nclips = sum(sum(clips) for clips, _ in rows)
    pixels = synth_pixels(nclips, frames, image)
    labels = np.where((attn == 1) & (vmask == 0), input_ids, -100)
    return pixels, input_ids, attn, vmask, labels


T5_CASES = {
    # name: (config, frames T, rows, target length, new_tokens)
    "tiny_t5_b2": ("tiny_t5", 1, [([1, 1], [4, 6]), ([2], [3])], 5, 6),
    "mid_t5_b1": ("mid_t5", 2, [([1, 1, 1], [5, 5, 4])], 6, 6),
    "mid_t5_b2": ("mid_t5", 2, [([1, 1], [6, 7]), ([1, 1], [3, 3])], 6, 6),
}


def load_det_weights(model, mode="fanin", seed=0):
    sd = model.state_dict()
    new = {}
    tied = {"language_model.lm_head.weight": None, "language_model.encoder.embed_tokens.weight": "language_model.shared.weight",
            "language_model.decoder.embed_tokens.weight": "language_model.shared.weight"}
    for k, v in sd.items():
        if k in tied:
            continue
        new[k] = torch.from_numpy(synth_param(k, tuple(v.shape), mode, seed)).to(v.dtype)
    if "language_model.shared.weight" in new:  # T5: encoder/decoder embeddings and (installed transformers) lm_head = shared
        for k in tied:
            if k in sd:
                new[k] = new["language_model.shared.weight"]
    else:
        new["language_model.lm_head.weight"] = new["language_model.model.decoder.embed_tokens.weight"]
    model.load_state_dict(new)


@torch.no_grad()
def run_t5_case(name):
    """Encoder-decoder LM (flan-t5 family): teacher-forced forward with labels, greedy generate (ref:eilev/model/v2.py:228-238, 318-322)."""
    cfg_name, frames, rows, tgt_len, new_tokens = T5_CASES[name]
    cfg = blip2_config(cfg_name)
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    pixels, input_ids, attn, vmask, _ = build_inputs(cfg_name, frames, rows)
    # T5 batches are RIGHT padded (encoder): move the padding of build_inputs to the right
    B, L = input_ids.shape
    for b in range(B):
        n = int(attn[b].sum())
        input_ids[b] = np.concatenate([input_ids[b, L - n:], np.zeros(L - n, np.int64)])
        vmask[b] = np.concatenate([vmask[b, L - n:], np.zeros(L - n, np.int64)])
        attn[b] = np.concatenate([np.ones(n, np.int64), np.zeros(L - n, np.int64)])
    vocab = cfg.text_config.vocab_size
    rng = np.random.default_rng(11)
    labels = rng.integers(2, vocab, size=(B, tgt_len)).astype(np.int64)
    if B > 1:
        labels[1, tgt_len - 2:] = -100  # padded target
    dec_in = np.concatenate([np.zeros((B, 1), np.int64), np.where(labels[:, :-1] < 0, 0, labels[:, :-1])], axis=1)  # _shift_right
    t = lambda a: torch.from_numpy(a)
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        px = t(pixels).to(dtype)
        o = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=px, video_input_mask=t(vmask), labels=t(labels), return_dict=True)
        out[f"{tag}_logits"] = o.logits.float().numpy()
        out[f"{tag}_loss"] = np.asarray(float(o.loss), dtype=np.float64)
        out[f"{tag}_enc"] = o.language_model_outputs.encoder_last_hidden_state.float().numpy()
        o2 = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=px, video_input_mask=t(vmask), decoder_input_ids=t(dec_in),
               return_dict=True)
        assert torch.equal(o2.logits, o.logits)
        free = None
        for never in range(vocab - 1, 3, -1):
            g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn),
                           max_new_tokens=new_tokens, num_beams=1, do_sample=False, eos_token_id=never)
            if not (g[:, 1:] == never).any():
                free = g
                break
        assert free is not None and free.shape == (B, new_tokens + 1), free.shape
        out[f"{tag}_greedy_free"] = free.numpy().astype(np.int64)
        eos = int(free[0, 3])
        g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn),
                       max_new_tokens=new_tokens, num_beams=1, do_sample=False, eos_token_id=eos)
        out[f"{tag}_greedy_eos"] = g.numpy().astype(np.int64)
        out[f"{tag}_eos_id"] = np.asarray(eos, dtype=np.int64)
        for nbm, lp, nm in ((5, -1.0, "beam5_lpm1"), (3, 1.0, "beam3_lp1")):
            for e_id, suffix in ((eos, ""), (never, "_free")):
                g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn),
                               max_new_tokens=new_tokens, num_beams=nbm, do_sample=False, length_penalty=lp, eos_token_id=e_id)
                out[f"{tag}_{nm}{suffix}"] = g.numpy().astype(np.int64)
    meta = dict(case=name, config=cfg_name, frames=frames, rows=rows, new_tokens=new_tokens, weight_mode="fanin",
                torch=torch.__version__, transformers=transformers.__version__, generator="tools/make_goldens.py",
                reference="/root/reference/eilev/model/v2.py", padding="right", scale_decoder_outputs=bool(cfg.text_config.scale_decoder_outputs))
    out.update(input_ids=input_ids, attention_mask=attn, video_input_mask=vmask, labels=labels, decoder_input_ids=dec_in,
               meta=np.asarray(json.dumps(meta)))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"}, os.path.getsize(path))


@torch.no_grad()
def run_t5_debug_case(name="mid_t5_dbg"):
    """Encoder-decoder LM with what else the reference's forward hands down (ref:eilev/model/v2.py:228-238): a decoder_attention_mask with
    padding (row 1: right padding; row 0: a hole in the middle) and output_hidden_states=True (both stacks' tuples)."""
    cfg_name, frames, rows, tgt_len, _ = T5_CASES["mid_t5_b2"]
    cfg = blip2_config(cfg_name)
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    pixels, input_ids, attn, vmask, _ = build_inputs(cfg_name, frames, rows)
    B, L = input_ids.shape
    for b in range(B):  # right padding, as run_t5_case
        n = int(attn[b].sum())
        input_ids[b] = np.concatenate([input_ids[b, L - n:], np.zeros(L - n, np.int64)])
        vmask[b] = np.concatenate([vmask[b, L - n:], np.zeros(L - n, np.int64)])
        attn[b] = np.concatenate([np.ones(n, np.int64), np.zeros(L - n, np.int64)])
    vocab = cfg.text_config.vocab_size
    rng = np.random.default_rng(23)
    T = tgt_len + 2
    dec_in = np.concatenate([np.zeros((B, 1), np.int64), rng.integers(2, vocab, size=(B, T - 1))], axis=1).astype(np.int64)
    dec_mask = np.ones((B, T), np.int64)
    dec_mask[0, 3] = 0
    dec_mask[1, T - 3:] = 0
    dec_in[1, T - 3:] = 0
    t = lambda a: torch.from_numpy(a)
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        o = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=t(pixels).to(dtype), video_input_mask=t(vmask),
              decoder_input_ids=t(dec_in), decoder_attention_mask=t(dec_mask), output_hidden_states=True, return_dict=True)
        lm = o.language_model_outputs
        out[f"{tag}_logits"] = o.logits.float().numpy()
        out[f"{tag}_enc_hidden"] = torch.stack(lm.encoder_hidden_states).float().numpy()
        out[f"{tag}_dec_hidden"] = torch.stack(lm.decoder_hidden_states).float().numpy()
        o2 = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=t(pixels).to(dtype), video_input_mask=t(vmask),
               decoder_input_ids=t(dec_in), return_dict=True)
        out[f"{tag}_logits_nomask"] = o2.logits.float().numpy()  # (shows the mask matters: differs from *_logits after the masked keys)
        assert not torch.equal(o2.logits, o.logits)
    # output_attentions=True: the same call on the same weights with EAGER attention (the default implementation returns no weights)
    cfg_e = blip2_config(cfg_name)
    for c in (cfg_e, cfg_e.vision_config, cfg_e.qformer_config, cfg_e.text_config):
        c._attn_implementation = "eager"
    torch.manual_seed(0)
    model_e = RefModel(cfg_e).eval()
    load_det_weights(model_e)
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model_e.to(dtype)
        o = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=t(pixels).to(dtype), video_input_mask=t(vmask),
              decoder_input_ids=t(dec_in), decoder_attention_mask=t(dec_mask), output_attentions=True, return_dict=True)
        lm = o.language_model_outputs
        assert np.abs(o.logits.float().numpy() - out[f"{tag}_logits"]).max() < (1e-4 if dtype == torch.float32 else 0.5)
        out[f"{tag}_enc_attn"] = torch.stack(lm.encoder_attentions).float().numpy()    # (layers, B, heads, L, L)
        out[f"{tag}_dec_attn"] = torch.stack(lm.decoder_attentions).float().numpy()    # (layers, B, heads, T, T)
        out[f"{tag}_cross_attn"] = torch.stack(lm.cross_attentions).float().numpy()    # (layers, B, heads, T, L)
    meta = dict(case=name, config=cfg_name, frames=frames, rows=rows, weight_mode="fanin", torch=torch.__version__,
                transformers=transformers.__version__, generator="tools/make_goldens.py", reference="/root/reference/eilev/model/v2.py",
                padding="right", attentions="attn_implementation=eager")
    out.update(input_ids=input_ids, attention_mask=attn, video_input_mask=vmask, decoder_input_ids=dec_in, decoder_attention_mask=dec_mask,
               meta=np.asarray(json.dumps(meta)))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"}, os.path.getsize(path))


class _legacy_kv_compat:
    """The reference's classify() was written against the tuple-of-(k, v) KV cache of the transformers release it pins
    (ref:pyproject.toml); the installed release iterates a DynamicCache as (k, v, sliding_window) triples and only takes
    Cache objects back.  While generating the classify goldens, present the old protocol to the reference code: iterate
    as (k, v) pairs and rebuild a DynamicCache from the tuples it hands back.  Arithmetic is untouched."""

    def __enter__(self):
        from transformers.cache_utils import DynamicCache
        from transformers.models.opt import modeling_opt

        self.dc, self.opt = DynamicCache, modeling_opt.OPTForCausalLM
        self.old_iter, self.old_fwd = DynamicCache.__iter__, modeling_opt.OPTForCausalLM.forward

        def pairs(cache):
            for layer in cache.layers:
                yield layer.keys, layer.values

        old_fwd = self.old_fwd

        def fwd(lm, *a, past_key_values=None, **k):
            if isinstance(past_key_values, tuple):
                cache = DynamicCache(config=lm.config)
                for i, (kk, vv) in enumerate(past_key_values):
                    cache.update(kk, vv, i)
                past_key_values = cache
            return old_fwd(lm, *a, past_key_values=past_key_values, **k)

        DynamicCache.__iter__ = pairs
        modeling_opt.OPTForCausalLM.forward = fwd
        return self

    def __exit__(self, *exc):
        self.dc.__iter__ = self.old_iter
        self.opt.forward = self.old_fwd


@torch.no_grad()
def run_case(name):
    cfg_name, frames, rows, new_tokens = CASES[name]
    cfg = blip2_config(cfg_name)
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    pixels, input_ids, attn, vmask, labels = build_inputs(cfg_name, frames, rows)
    t = lambda a: torch.from_numpy(a)
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        px = t(pixels).to(dtype)
        o = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=px,
              video_input_mask=t(vmask), labels=t(labels), return_dict=True)
        out[f"{tag}_vit"] = o.vision_outputs.last_hidden_state.float().numpy()
        out[f"{tag}_pooler"] = o.vision_outputs.pooler_output.float().numpy()
        out[f"{tag}_qformer"] = o.qformer_outputs.last_hidden_state.float().numpy()
        out[f"{tag}_logits"] = o.logits.float().numpy()
        out[f"{tag}_loss"] = np.asarray(float(o.loss), dtype=np.float64)
        vocab = cfg.text_config.vocab_size
        free = None
        for never in range(vocab - 1, 3, -1):
            g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask),
                           attention_mask=t(attn), max_new_tokens=new_tokens, num_beams=1,
                           do_sample=False, eos_token_id=never)
            if not (g == never).any():
                free = g
                break
        assert free is not None and free.shape == (len(rows), new_tokens), free.shape
        out[f"{tag}_greedy_free"] = free.numpy().astype(np.int64)
        eos = int(free[0, 2])
        g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask),
                       attention_mask=t(attn), max_new_tokens=new_tokens, num_beams=1,
                       do_sample=False, eos_token_id=eos)
        out[f"{tag}_greedy_eos"] = g.numpy().astype(np.int64)
        out[f"{tag}_eos_id"] = np.asarray(eos, dtype=np.int64)
        # beam search as the sample script calls it (ref:samples/eilev_generate_action_narration.py:60-73), shorter budget
        for nbm, lp, nm in ((5, -1.0, "beam5_lpm1"), (3, 1.0, "beam3_lp1")):
            if len(rows) * nbm > 16:
                continue
            g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn),
                           max_new_tokens=new_tokens, num_beams=nbm, do_sample=False, length_penalty=lp, eos_token_id=eos)
            out[f"{tag}_{nm}"] = g.numpy().astype(np.int64)
            g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn),
                           max_new_tokens=new_tokens, num_beams=nbm, do_sample=False, length_penalty=lp, eos_token_id=never)
            out[f"{tag}_{nm}_free"] = g.numpy().astype(np.int64)
        # classify(): class log-likelihoods continued from the prompt's KV cache (ref:eilev/model/v2.py:326-501)
        rng = np.random.default_rng(7)
        n_cls, cls_len = 5, 4
        cls_ids = rng.integers(4, vocab, size=(n_cls, cls_len)).astype(np.int64)
        cls_mask = np.ones((n_cls, cls_len), dtype=np.int64)
        cls_mask[1, 3:] = 0  # right-padded shorter classes
        cls_mask[3, 2:] = 0
        cls_ids[cls_mask == 0] = 1
        for cbs, nm in ((None, "classify"), (2, "classify_cbs2")):
            with _legacy_kv_compat():
                ll = m.classify(t(input_ids), t(cls_ids), prompt_attention_mask=t(attn), pixel_values=px,
                                prompt_video_input_mask=t(vmask), class_attention_mask=t(cls_mask), class_batch_size=cbs)
            out[f"{tag}_{nm}"] = ll.float().numpy()
        out["class_input_ids"] = cls_ids
        out["class_attention_mask"] = cls_mask
    meta = dict(case=name, config=cfg_name, frames=frames, rows=rows, new_tokens=new_tokens,
                weight_mode="fanin", torch=torch.__version__, transformers=transformers.__version__,
                attn_implementation=str(getattr(cfg, "_attn_implementation", None)),
                generator="tools/make_goldens.py", reference="/root/reference/eilev/model/v2.py")
    out["input_ids"] = input_ids
    out["attention_mask"] = attn
    out["video_input_mask"] = vmask
    out["labels"] = labels
    out["meta"] = np.asarray(json.dumps(meta))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"},
          os.path.getsize(path))


VARIED_CASES = {
    # name: (config, frames T, rows, new_tokens) — weight mode 'varied' (eilev_amd.synth.synth_param): the reference's greedy and beam
    # outputs change from step to step (every 'fanin' OPT fixture repeats ONE id, which a decode step with a wrong
    # position or a stale KV slot would reproduce).  The weight SEED is searched (run_varied_case) and stored in the fixture's meta.
    "mid_v1": ("mid", 2, [([1, 1, 1], [5, 5, 4])], 14),
    "mid_v2": ("mid", 2, [([1, 1], [6, 7]), ([1, 1], [3, 3])], 14),
    "real_v1": ("real_1l", 8, [([1, 1], [10, 8])], 12),
}
FULL_CASES = {
    # the C1 workload at FULL DEPTH (39 ViT / 12 Q-Former / 32 OPT blocks, real widths): 1 clip x 8 frames, 0 in-context examples
    "full_c1": ("opt27", 8, [([1], [14])], 32),
    # round 5 (VERDICT r4 missing 2): the HEADLINE shape at full depth — ONE 16-shot sample (17 clips x 8 frames, L = 1 + 17 * 33 + 16 * 24
    # + 14 = 960, the bench's layout) plus a second, shorter row (3 clips, L = 128) that is LEFT-padded to 960: batch > 1, padding, 20 clips
    "full_c2": ("opt27", 8, [([1] * 17, [24] * 16 + [14]), ([1, 1, 1], [12, 9, 7])], 32),
}


def _greedy_with_scores(m, kw, new_tokens, eos):
    o = m.generate(**kw, max_new_tokens=new_tokens, num_beams=1, do_sample=False, eos_token_id=eos, output_scores=True,
                   return_dict_in_generate=True)
    return o.sequences, torch.stack([sc.float() for sc in o.scores])  # (B, n), (n, B, vocab): the processed scores = raw logits here


BEAM_CONFIGS = ((5, -1.0, "beam5_lpm1"), (3, 1.0, "beam3_lp1"))


def _beams_are_stable(m, kw, new_tokens, eos_ids, sigma=0.012, trials=6):
    """Beam search prunes: two hypotheses tied to within rounding noise can flip and change everything after.  A fixture whose
    beam outputs are to be matched EXACTLY by another bf16 implementation must not sit on such a tie: the reference's fp32 beam
    results have to survive Gaussian noise of the size of a bf16 path's logit deviation (sigma = 0.012 on logits of std 0.8:
    what the HIP path measures against these fixtures) on every step's scores, for both beam configurations, with and without EOS."""
    from transformers import LogitsProcessorList

    class Noise:
        def __init__(self, seed):
            self.g = torch.Generator().manual_seed(seed)

        def __call__(self, input_ids, scores):
            return scores + sigma * torch.randn(scores.shape, generator=self.g, dtype=scores.dtype)

    for nbm, lp, _ in BEAM_CONFIGS:
        for e_id in eos_ids:
            base = m.generate(**kw, max_new_tokens=new_tokens, num_beams=nbm, do_sample=False, length_penalty=lp, eos_token_id=e_id)
            for trial in range(trials):
                g = m.generate(**kw, max_new_tokens=new_tokens, num_beams=nbm, do_sample=False, length_penalty=lp, eos_token_id=e_id,
                               logits_processor=LogitsProcessorList([Noise(100 + trial)]))
                if g.shape != base.shape or not torch.equal(g, base):
                    return False
    return True


@torch.no_grad()
def run_varied_case(name, max_seeds=2000):
    full = name in FULL_CASES
    cfg_name, frames, rows, new_tokens = (FULL_CASES if full else VARIED_CASES)[name]
    cfg = blip2_config(cfg_name)
    vocab = cfg.text_config.vocab_size
    pixels, input_ids, attn, vmask, labels = build_inputs(cfg_name, frames, rows)
    B = len(rows)
    t = lambda a: torch.from_numpy(a)
    never = vocab - 1
    beam_tokens = new_tokens if B == 1 else 8  # two rows x four beam runs x 14 steps never all survive the noise test below
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    chosen = None
    for seed in range(max_seeds):
        model = model.to(torch.float32)
        load_det_weights(model, "varied", seed)
        res = {}
        for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            m = model.to(dtype)
            kw = dict(input_ids=t(input_ids), pixel_values=t(pixels).to(dtype), video_input_mask=t(vmask), attention_mask=t(attn))
            res[tag] = _greedy_with_scores(m, kw, new_tokens, never)
        ids32, sc32 = res["fp32"]
        ids16, sc16 = res["bf16"]
        if sc32.shape != sc16.shape or ids32.shape != (B, new_tokens):  # a run emitted the "never" id and stopped: not this seed
            print(f"{name}: weight seed {seed}: early stop", flush=True)
            continue
        tk = sc32.topk(2, dim=-1)
        top2 = tk.values
        margin = float((top2[..., 0] - top2[..., 1]).min())
        dev = float((sc16.gather(-1, tk.indices) - top2).abs().max())  # the reference's own bf16 deviation on the two leading logits
        distinct = min(len(set(r.tolist())) for r in ids32)
        same = bool(torch.equal(ids32, ids16)) and ids32.shape == (B, new_tokens) and not bool((ids32 == never).any())
        print(f"{name}: weight seed {seed}: fp32==bf16 ids {same}, distinct ids per row >= {distinct}, min top-2 margin {margin:.4f}, "
              f"max |bf16 - fp32| of the top-2 step logits {dev:.4f}, logit std {float(sc32.std()):.3f}", flush=True)
        r0 = ids32[0].tolist()
        has_eos_step = same and any(r0[k] not in r0[:k] for k in range(4, new_tokens - 3))
        print("   ", ids32.tolist(), flush=True)
        if full and same and (B > 1 or margin >= 2.0 * dev):
            # (two rows x 32 steps at L = 960: the tightest of 64 top-2 margins is of the size of the bf16 deviation for every seed tried —
            #  seed 0: 0.086 against 0.088, seed 1: 0.023 against 0.113 — while the reference's fp32 and bf16 runs still emit the same 64 ids.
            #  The fixture stores every step's top-8 logits, so the test knows which steps are near-ties: tests/test_hip_full_depth_c2.py)
            # 39 + 12 + 32 blocks take ~10 minutes per pass on this host: no search for variety here (a deep random stack settles on a
            # fixed point whatever the embedding scale; varied ids are what mid_v* / real_v1 are for).  What this fixture adds is DEPTH:
            # the full-vocabulary prefill row, the 32 greedy ids, and a FORCED continuation (below) whose tokens do change.
            chosen = seed
            kept = res  # the two passes of this seed ARE the fixture's greedy runs: not repeated below (minutes each)
            break
        if same and distinct >= 5 and margin >= 2.5 * dev and has_eos_step:
            if name.startswith("mid") and not _beams_are_stable(model.to(torch.float32), dict(
                    input_ids=t(input_ids), pixel_values=t(pixels), video_input_mask=t(vmask), attention_mask=t(attn)), beam_tokens,
                    (r0[next(k for k in range(4, new_tokens - 3) if r0[k] not in r0[:k])], never)):
                print("    beam search not stable under logit noise of bf16 size: next seed", flush=True)
                continue
            chosen = seed
            break
    assert chosen is not None, "no weight seed gives a well-separated, varied greedy sequence"
    out = {}
    model = model.to(torch.float32)
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        px = t(pixels).to(dtype)
        kw = dict(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn))
        ids, sc = kept[tag] if full else _greedy_with_scores(m, kw, new_tokens, never)
        out[f"{tag}_greedy_free"] = ids.numpy().astype(np.int64)
        out[f"{tag}_step_logits_top8_ids"] = sc.topk(8, dim=-1).indices.numpy().astype(np.int64)      # (n, B, 8)
        out[f"{tag}_step_logits_top8"] = sc.topk(8, dim=-1).values.numpy()
        if full:
            out[f"{tag}_logits_last"] = sc[0].numpy()                                                   # prefill last row, full vocabulary
        else:
            o = m(**kw, labels=t(labels), return_dict=True)
            lg = o.logits.float().numpy()
            out[f"{tag}_loss"] = np.asarray(float(o.loss), dtype=np.float64)
            if lg.size <= (1 << 20):
                out[f"{tag}_logits"] = lg
            else:
                out[f"{tag}_logits_last"] = lg[:, -1]
            if name.startswith("mid"):
                out[f"{tag}_step_logits"] = sc.numpy()                                                  # (n, B, vocab): every decode step
        if full:
            # teacher-forced continuation: 24 pseudo-random tokens appended to the prompt, ONE forward of the reference over prompt +
            # tokens; logits of position L - 1 + j = what a decode step returns after j forced tokens (positions, KV slots and 32 blocks
            # of weight-streaming kernels, on inputs that differ at every step)
            from eilev_amd.synth import det_uniform_int

            forced = det_uniform_int(f"{name}_forced", (B, 24), 4, 50000)
            ids_f = np.concatenate([input_ids, forced], 1)
            am_f = np.concatenate([attn, np.ones_like(forced)], 1)
            vm_f = np.concatenate([vmask, np.zeros_like(forced)], 1)
            o = m(input_ids=t(ids_f), pixel_values=px, video_input_mask=t(vm_f), attention_mask=t(am_f), return_dict=True)
            lg = o.logits.float()[:, input_ids.shape[1] - 1:]                                           # (B, 25, vocab)
            out["forced_tokens"] = forced
            out[f"{tag}_forced_top8_ids"] = lg.topk(8, dim=-1).indices.numpy().astype(np.int64)
            out[f"{tag}_forced_top8"] = lg.topk(8, dim=-1).values.numpy()
            out[f"{tag}_forced_logits_row12"] = lg[:, 12].numpy()                                       # one full row in the middle
            continue
        if tag == "fp32":  # a row that stops in the MIDDLE: eos = what row 0 emits at step 5, if it is new there
            r0 = ids[0].tolist()
            k_eos = next(k for k in range(4, new_tokens - 3) if r0[k] not in r0[:k])
            eos = r0[k_eos]
        g = m.generate(**kw, max_new_tokens=new_tokens, num_beams=1, do_sample=False, eos_token_id=eos)
        out[f"{tag}_greedy_eos"] = g.numpy().astype(np.int64)
        out[f"{tag}_eos_id"] = np.asarray(eos, dtype=np.int64)
        if not full:
            for nbm, lp, nm in ((5, -1.0, "beam5_lpm1"), (3, 1.0, "beam3_lp1")):
                for e_id, suffix in ((eos, ""), (never, "_free")):
                    g = m.generate(**kw, max_new_tokens=beam_tokens, num_beams=nbm, do_sample=False, length_penalty=lp, eos_token_id=e_id)
                    out[f"{tag}_{nm}{suffix}"] = g.numpy().astype(np.int64)
    assert full or np.array_equal(out["fp32_greedy_eos"], out["bf16_greedy_eos"])
    meta = dict(case=name, config=cfg_name, frames=frames, rows=rows, new_tokens=new_tokens, weight_mode="varied", weight_seed=chosen,
                never_id=never, beam_new_tokens=beam_tokens, torch=torch.__version__, transformers=transformers.__version__, generator="tools/make_goldens.py",
                reference="/root/reference/eilev/model/v2.py")
    out.update(input_ids=input_ids, attention_mask=attn, video_input_mask=vmask, labels=labels, meta=np.asarray(json.dumps(meta)))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"}, os.path.getsize(path))
    print(name, "greedy fp32", out["fp32_greedy_free"].tolist(), "eos", out["fp32_greedy_eos"].tolist() if "fp32_greedy_eos" in out else None)


REAL_CASES = {
    # name: (config, frames T, rows, new_tokens) — real widths, one block per stack (SURVEY 8c golden plan (2)).  The clip has
    # the real 8 frames so the Q-Former's cross-attention sees the real 2056 keys; L = 1 + 2 * 33 + 18 = 85 tokens.
    "real_b1": ("real_1l", 8, [([1, 1], [10, 8])], 5),
}
REAL_VIT_ROWS, REAL_LOGIT_COLS = 64, 512


def real_subsample(cfg, n_clips, frames, L):
    """Which rows / columns of the big tensors a real-shape fixture keeps (deterministic; tests rebuild the same index sets)."""
    from eilev_amd.synth import det_uniform_int

    tokens = frames * ((cfg.vision_config.image_size // cfg.vision_config.patch_size) ** 2 + 1)
    rows = np.unique(np.concatenate([[0, 1, 256, 257, tokens - 1, n_clips * tokens - 1],   # CLS rows, frame seams, the last row
                                     det_uniform_int("real_vit_rows", (REAL_VIT_ROWS,), 0, n_clips * tokens)]))
    cols = np.unique(det_uniform_int("real_logit_cols", (REAL_LOGIT_COLS,), 0, cfg.text_config.vocab_size))
    return rows, cols


@torch.no_grad()
def run_real_case(name):
    """Real-width single-block fixture: outputs are SUBSAMPLED (the weights come from the seed, not from the file)."""
    cfg_name, frames, rows, new_tokens = REAL_CASES[name]
    cfg = blip2_config(cfg_name)
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    pixels, input_ids, attn, vmask, labels = build_inputs(cfg_name, frames, rows)
    B, L = input_ids.shape
    vit_rows, logit_cols = real_subsample(cfg, pixels.shape[0], frames, L)
    t = lambda a: torch.from_numpy(a)
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        px = t(pixels).to(dtype)
        o = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=px, video_input_mask=t(vmask), labels=t(labels), return_dict=True)
        vit = o.vision_outputs.last_hidden_state.float().numpy()
        out[f"{tag}_vit_rows"] = vit.reshape(-1, vit.shape[-1])[vit_rows]
        out[f"{tag}_vit_checksum"] = np.asarray([vit.astype(np.float64).sum(), np.abs(vit.astype(np.float64)).sum()])
        out[f"{tag}_pooler"] = o.vision_outputs.pooler_output.float().numpy()
        out[f"{tag}_qformer"] = o.qformer_outputs.last_hidden_state.float().numpy()
        lg = o.logits.float().numpy()
        out[f"{tag}_logits_cols"] = lg[:, :, logit_cols]          # every position, sampled vocabulary columns
        out[f"{tag}_logits_last"] = lg[:, -1]                     # the row generate() consumes, full vocabulary
        out[f"{tag}_logits_checksum"] = np.asarray([lg.astype(np.float64).sum(), np.abs(lg.astype(np.float64)).sum()])
        out[f"{tag}_loss"] = np.asarray(float(o.loss), dtype=np.float64)
        g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn),
                       max_new_tokens=new_tokens, min_new_tokens=new_tokens, num_beams=1, do_sample=False)
        assert g.shape == (B, new_tokens), g.shape
        out[f"{tag}_greedy_free"] = g.numpy().astype(np.int64)
    meta = dict(case=name, config=cfg_name, frames=frames, rows=rows, new_tokens=new_tokens, weight_mode="fanin",
                torch=torch.__version__, transformers=transformers.__version__, generator="tools/make_goldens.py",
                reference="/root/reference/eilev/model/v2.py", subsample="eilev_amd.synth.det_uniform_int: see real_subsample()")
    out.update(input_ids=input_ids, attention_mask=attn, video_input_mask=vmask, labels=labels, vit_rows=vit_rows, logit_cols=logit_cols,
               meta=np.asarray(json.dumps(meta)))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"}, os.path.getsize(path))


REAL_T5_CASES = {"real_t5_b1": ("real_t5_1l", 8, [([1, 1], [10, 8])], 6, 5),
                 # round 5 (VERDICT r4 missing 2): the flan-t5-xl backbone at FULL DEPTH (39 ViT + 12 Q-Former + 24 encoder + 24 decoder blocks)
                 # on the C1-sized input (1 clip x 8 frames, L = 48; T5 layout: no BOS): encoder rows, logits, greedy ids
                 "full_t5": ("t5xl", 8, [([1], [14])], 6, 8)}


@torch.no_grad()
def run_real_t5_case(name):
    """Real-width single-block fixture of the encoder-decoder path (flan-t5-xl widths): subsampled outputs, weights from the seed."""
    from eilev_amd.synth import det_uniform_int

    cfg_name, frames, rows, tgt_len, new_tokens = REAL_T5_CASES[name]
    cfg = blip2_config(cfg_name)
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    pixels, input_ids, attn, vmask, _ = build_inputs(cfg_name, frames, rows)  # one row: no padding
    if cfg.text_config.model_type == "t5" and name.startswith("full"):  # the T5 layout of ref:eilev/data/utils.py:199-217 has no BOS: drop
        input_ids, attn, vmask = input_ids[:, 1:].copy(), attn[:, 1:].copy(), vmask[:, 1:].copy()  # the leading id of the OPT-style synthetic row
    B, L = input_ids.shape
    vocab = cfg.text_config.vocab_size
    rng = np.random.default_rng(11)
    labels = rng.integers(2, vocab, size=(B, tgt_len)).astype(np.int64)
    dec_in = np.concatenate([np.zeros((B, 1), np.int64), labels[:, :-1]], axis=1)
    enc_rows = np.unique(np.concatenate([[0, 1, 32, 33, L - 1], det_uniform_int("real_t5_enc_rows", (16,), 0, L)]))
    cols = np.unique(det_uniform_int("real_t5_logit_cols", (REAL_LOGIT_COLS,), 0, vocab))
    t = lambda a: torch.from_numpy(a)
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        px = t(pixels).to(dtype)
        o = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=px, video_input_mask=t(vmask), labels=t(labels), return_dict=True)
        enc = o.language_model_outputs.encoder_last_hidden_state.float().numpy()
        out[f"{tag}_enc_rows"] = enc[:, enc_rows]
        out[f"{tag}_enc_checksum"] = np.asarray([enc.astype(np.float64).sum(), np.abs(enc.astype(np.float64)).sum()])
        lg = o.logits.float().numpy()
        out[f"{tag}_logits_cols"] = lg[:, :, cols]
        out[f"{tag}_logits_checksum"] = np.asarray([lg.astype(np.float64).sum(), np.abs(lg.astype(np.float64)).sum()])
        out[f"{tag}_logits_argmax"] = lg.argmax(-1)
        out[f"{tag}_loss"] = np.asarray(float(o.loss), dtype=np.float64)
        if name.startswith("full"):  # round 6: the eight leading logits of every greedy step too, so that a free-running comparison knows where
            # the REFERENCE's near-ties are (oracle/parity.py, as the full-depth OPT fixtures have had since round 4)
            o2 = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn), max_new_tokens=new_tokens,
                            min_new_tokens=new_tokens, num_beams=1, do_sample=False, output_scores=True, return_dict_in_generate=True)
            g = o2.sequences
            # (min_new_tokens masks EOS in the processed scores: -inf never ranks among the leading eight)
            sc = torch.stack([s_.float() for s_ in o2.scores])
            out[f"{tag}_step_logits_top8_ids"] = sc.topk(8, dim=-1).indices.numpy().astype(np.int64)  # (n, B, 8)
            out[f"{tag}_step_logits_top8"] = sc.topk(8, dim=-1).values.numpy()
        else:
            g = m.generate(input_ids=t(input_ids), pixel_values=px, video_input_mask=t(vmask), attention_mask=t(attn),
                           max_new_tokens=new_tokens, min_new_tokens=new_tokens, num_beams=1, do_sample=False)
        out[f"{tag}_greedy_free"] = g.numpy().astype(np.int64)
    meta = dict(case=name, config=cfg_name, frames=frames, rows=rows, new_tokens=new_tokens, weight_mode="fanin", torch=torch.__version__,
                transformers=transformers.__version__, generator="tools/make_goldens.py", reference="/root/reference/eilev/model/v2.py")
    out.update(input_ids=input_ids, attention_mask=attn, video_input_mask=vmask, labels=labels, decoder_input_ids=dec_in, enc_rows=enc_rows,
               logit_cols=cols, meta=np.asarray(json.dumps(meta)))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"}, os.path.getsize(path))


@torch.no_grad()
def run_vit_debug_case(name="mid_vitdebug"):
    """output_hidden_states / output_attentions of the reference's VideoBlipVisionModel (ref:eilev/model/v2.py:76-103), eager
    attention (sdpa returns no attention maps): 2 clips x 3 frames at the `mid` widths."""
    cfg = blip2_config("mid")
    cfg.vision_config._attn_implementation = "eager"
    vm = _ref.VideoBlipVisionModel(cfg.vision_config).eval()
    sd = {k: torch.from_numpy(synth_param("vision_model." + k, tuple(v.shape))) for k, v in vm.state_dict().items()}
    vm.load_state_dict(sd)
    frames, clips = 3, 2
    pixels = synth_pixels(clips, frames, cfg.vision_config.image_size)
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        o = vm.to(dtype)(torch.from_numpy(pixels).to(dtype), output_attentions=True, output_hidden_states=True, return_dict=True)
        out[f"{tag}_last"] = o.last_hidden_state.float().numpy()
        out[f"{tag}_pooler"] = o.pooler_output.float().numpy()
        out[f"{tag}_hidden_states"] = np.stack([h.float().numpy() for h in o.hidden_states])
        out[f"{tag}_attentions"] = np.stack([a.float().numpy() for a in o.attentions])
    meta = dict(case=name, config="mid", frames=frames, clips=clips, weight_mode="fanin", torch=torch.__version__,
                transformers=transformers.__version__, attn_implementation="eager", generator="tools/make_goldens.py",
                reference="/root/reference/eilev/model/v2.py::VideoBlipVisionModel")
    out["meta"] = np.asarray(json.dumps(meta))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: v.shape for k, v in out.items() if k != "meta"}, os.path.getsize(path))


@torch.no_grad()
def run_lm_debug_case(name="mid_lmdebug", base="mid_b2"):
    """`output_hidden_states=True` through the reference's full forward (ref:eilev/model/v2.py:187-193 Q-Former, :220-227 language
    model): the per-block tuples of the language model and of the Q-Former on the inputs of `base` (left padding, two rows)."""
    cfg_name, frames, rows, _ = CASES[base]
    cfg = blip2_config(cfg_name)
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    pixels, input_ids, attn, vmask, _ = build_inputs(cfg_name, frames, rows)
    t = lambda a: torch.from_numpy(a)
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        o = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=t(pixels).to(dtype), video_input_mask=t(vmask),
              output_hidden_states=True, return_dict=True)
        out[f"{tag}_lm_hidden_states"] = np.stack([h.float().numpy() for h in o.language_model_outputs.hidden_states])
        out[f"{tag}_qformer_hidden_states"] = np.stack([h.float().numpy() for h in o.qformer_outputs.hidden_states])
        out[f"{tag}_logits"] = o.logits.float().numpy()
    meta = dict(case=name, base=base, config=cfg_name, frames=frames, rows=rows, weight_mode="fanin", torch=torch.__version__,
                transformers=transformers.__version__, generator="tools/make_goldens.py", reference="/root/reference/eilev/model/v2.py")
    out["input_ids"] = input_ids
    out["attention_mask"] = attn
    out["video_input_mask"] = vmask
    out["meta"] = np.asarray(json.dumps(meta))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: v.shape for k, v in out.items() if k != "meta"}, os.path.getsize(path))


@torch.no_grad()
def run_attn_debug_case(name="mid_attndebug", base="mid_b2"):
    """`output_attentions=True` through the reference's full forward with eager attention (sdpa returns no weights): the attention weights of
    the language model (ref:eilev/model/v2.py:220-227) and of the Q-Former, self and cross (:187-193), on the inputs of `base`."""
    cfg_name, frames, rows, _ = CASES[base]
    cfg = blip2_config(cfg_name)
    for c in (cfg, cfg.vision_config, cfg.qformer_config, cfg.text_config):
        c._attn_implementation = "eager"
    torch.manual_seed(0)
    model = RefModel(cfg).eval()
    load_det_weights(model)
    pixels, input_ids, attn, vmask, _ = build_inputs(cfg_name, frames, rows)
    t = lambda a: torch.from_numpy(a)
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = model.to(dtype)
        o = m(input_ids=t(input_ids), attention_mask=t(attn), pixel_values=t(pixels).to(dtype), video_input_mask=t(vmask),
              output_attentions=True, return_dict=True)
        out[f"{tag}_lm_attentions"] = np.stack([a.float().numpy() for a in o.language_model_outputs.attentions])
        # (installed transformers records EVERY attention module of the Q-Former in `attentions`, in execution order — self_0, cross_0,
        # self_1, ... — and the cross-attention ones again in `cross_attentions`: stored one array per entry)
        for i, a in enumerate(o.qformer_outputs.attentions):
            out[f"{tag}_qformer_attentions_{i}"] = a.float().numpy()
        for i, a in enumerate(o.qformer_outputs.cross_attentions):
            out[f"{tag}_qformer_cross_attentions_{i}"] = a.float().numpy()
        out["qformer_counts"] = np.asarray([len(o.qformer_outputs.attentions), len(o.qformer_outputs.cross_attentions)])
        out[f"{tag}_logits"] = o.logits.float().numpy()
    meta = dict(case=name, base=base, config=cfg_name, frames=frames, rows=rows, weight_mode="fanin", torch=torch.__version__,
                transformers=transformers.__version__, attn_implementation="eager", generator="tools/make_goldens.py",
                reference="/root/reference/eilev/model/v2.py")
    out.update(input_ids=input_ids, attention_mask=attn, video_input_mask=vmask, meta=np.asarray(json.dumps(meta)))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: v.shape for k, v in out.items() if k != "meta"}, os.path.getsize(path))


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CASES) + list(T5_CASES) + list(REAL_CASES) + [c for c in REAL_T5_CASES if not c.startswith("full")] + ["mid_vitdebug", "mid_lmdebug", "mid_attndebug", "mid_t5_dbg"] +
              list(VARIED_CASES)):  # full_c1 / full_c2 / full_t5 (15 GB of fp32 weights, minutes to half an hour): by name only
        (run_attn_debug_case if n == "mid_attndebug" else run_t5_debug_case if n == "mid_t5_dbg" else run_varied_case if n in VARIED_CASES or n in FULL_CASES else run_t5_case if n in T5_CASES else run_real_case if n in REAL_CASES else run_real_t5_case if n in REAL_T5_CASES
         else run_vit_debug_case if n == "mid_vitdebug" else run_lm_debug_case if n == "mid_lmdebug" else run_case)(n)
