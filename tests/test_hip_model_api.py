"""-m gpu: the drop-in class surface (forward / generate / vision model) on the HIP path vs the reference's goldens."""
import numpy as np
import pytest
import torch

from eilev_amd.configs import blip2_config
from hip_utils import host, load_case, rel_rms
from oracle.runner import synth_state_dict

pytestmark = pytest.mark.gpu


def build(cfg_name, dtype):
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration

    cfg = blip2_config(cfg_name)
    m = VideoBlipForConditionalGeneration(cfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg).items()}
    sd["language_model.lm_head.weight"] = sd["language_model.model.decoder.embed_tokens.weight"]
    m.load_state_dict(sd)
    return m.to(dtype).to("cuda")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["mid_b1", "mid_b2"])
def test_forward_and_generate_like_the_reference(golden_dir, name, dtype):
    g, meta, px = load_case(golden_dir, name)
    m = build(meta["config"], dtype)
    t = lambda a: torch.from_numpy(a).cuda()
    out = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype),
            video_input_mask=t(g["video_input_mask"]), labels=t(g["labels"]), return_dict=True)
    assert out.logits.dtype == dtype and out.logits.shape == g["fp32_logits"].shape
    valid = g["attention_mask"] == 1
    assert rel_rms(host(out.logits)[valid], g["fp32_logits"][valid]) <= 1.2e-2
    assert abs(float(out.loss) - float(g["fp32_loss"])) <= 2e-2 * abs(float(g["fp32_loss"]))
    assert out.vision_outputs.last_hidden_state.shape == g["fp32_vit"].shape
    assert out.vision_outputs.pooler_output.shape == g["fp32_pooler"].shape
    assert out.qformer_outputs.last_hidden_state.shape == g["fp32_qformer"].shape
    tup = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype),
            video_input_mask=t(g["video_input_mask"]), return_dict=False)
    assert torch.equal(tup[0], out.logits)
    # output_hidden_states: the Q-Former's tuple = embedding output + one tensor per block, the last one its last_hidden_state (bit for bit)
    dbg = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype),
            video_input_mask=t(g["video_input_mask"]), output_hidden_states=True, return_dict=True)
    qh = dbg.qformer_outputs.hidden_states
    assert len(qh) == m.config.qformer_config.num_hidden_layers + 1 and all(h.shape == dbg.qformer_outputs.last_hidden_state.shape for h in qh)
    assert torch.equal(qh[-1], dbg.qformer_outputs.last_hidden_state) and torch.equal(dbg.logits, out.logits)
    assert not torch.equal(qh[0], qh[1]) and len(dbg.vision_outputs.hidden_states) == m.config.vision_config.num_hidden_layers + 1
    n = meta["new_tokens"]
    ids = m.generate(input_ids=t(g["input_ids"]), pixel_values=t(px).to(dtype), video_input_mask=t(g["video_input_mask"]),
                     attention_mask=t(g["attention_mask"]), max_new_tokens=n, num_beams=1, do_sample=False,
                     eos_token_id=int(g["fp32_eos_id"]))
    assert np.array_equal(ids.cpu().numpy(), g["fp32_greedy_eos"])
    ids = m.generate(input_ids=t(g["input_ids"]), pixel_values=t(px).to(dtype), video_input_mask=t(g["video_input_mask"]),
                     attention_mask=t(g["attention_mask"]), max_new_tokens=n, min_new_tokens=n)
    assert np.array_equal(ids.cpu().numpy(), g["fp32_greedy_free"])


def test_vision_model_standalone_and_text_only(golden_dir):
    from eilev_amd.model.v2 import VideoBlipVisionModel

    g, meta, px = load_case(golden_dir, "mid_b1")
    cfg = blip2_config("mid")
    vm = VideoBlipVisionModel(cfg.vision_config)
    sd = {k[len("vision_model."):]: torch.from_numpy(v) for k, v in synth_state_dict(cfg).items() if k.startswith("vision_model.")}
    vm.load_state_dict(sd)
    vm = vm.cuda()
    o = vm(torch.from_numpy(px).cuda(), return_dict=True)
    assert rel_rms(host(o.last_hidden_state), g["fp32_vit"]) <= 1e-2
    assert rel_rms(host(o.pooler_output), g["fp32_pooler"]) <= 1e-2
    with pytest.raises(ValueError):
        vm(None)
    # text-only call (no pixel_values): plain OPT forward
    m = build("mid", torch.float32)
    ids = torch.randint(4, 500, (2, 9)).cuda()
    out = m(input_ids=ids)
    assert out.logits.shape == (2, 9, cfg.text_config.vocab_size) and out.vision_outputs is None


def test_engine_tracks_parameter_updates():
    m = build("mid", torch.float32)
    ids = torch.randint(4, 500, (1, 6)).cuda()
    a = m(input_ids=ids).logits
    with torch.no_grad():
        m.language_model.model.decoder.final_layer_norm.bias.add_(0.5)
    b = m(input_ids=ids).logits
    assert not torch.equal(a, b)


@pytest.mark.parametrize("name", ["mid_b1", "mid_b2"])
@pytest.mark.parametrize("tag,nb,lp", [("beam5_lpm1", 5, -1.0), ("beam3_lp1", 3, 1.0)])
def test_beam_search_like_the_sample_script(golden_dir, name, tag, nb, lp):
    """generate(num_beams=5, length_penalty=-1, eos_token_id=...) as ref:samples/eilev_generate_action_narration.py calls it."""
    g, meta, px = load_case(golden_dir, name)
    m = build(meta["config"], torch.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    kw = dict(input_ids=t(g["input_ids"]), pixel_values=t(px), video_input_mask=t(g["video_input_mask"]),
              attention_mask=t(g["attention_mask"]), max_new_tokens=meta["new_tokens"], num_beams=nb, do_sample=False, length_penalty=lp)
    ids = m.generate(**kw, eos_token_id=int(g["fp32_eos_id"]))
    assert np.array_equal(ids.cpu().numpy(), g[f"fp32_{tag}"]), (ids, g[f"fp32_{tag}"])
    # without a reachable EOS every hypothesis runs to the budget; bf16 arithmetic may reorder near-tied beams, so only
    # shape and the best first token are pinned here
    free = m.generate(**kw, eos_token_id=meta_never(g))
    assert free.shape == g[f"fp32_{tag}_free"].shape


def meta_never(g):
    return 511


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["mid_b1", "mid_b2"])
def test_classify_like_the_reference(golden_dir, name, dtype):
    """classify() on the HIP path (prefill once, eilev_opt_extend per class chunk) vs the reference's own classify()."""
    g, meta, px = load_case(golden_dir, name)
    m = build(meta["config"], dtype)
    t = lambda a: torch.from_numpy(a).cuda()
    kw = dict(prompt_attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype),
              prompt_video_input_mask=t(g["video_input_mask"]), class_attention_mask=t(g["class_attention_mask"]))
    ll = m.classify(t(g["input_ids"]), t(g["class_input_ids"]), **kw)
    assert ll.dtype == dtype and ll.shape == g["fp32_classify"].shape
    truth, ref_bf16 = g["fp32_classify"], g["bf16_classify"]
    # judged like the logits: no further from the fp32 truth than 1.5x the reference's own bf16 run (+ slack)
    budget = 1.5 * np.abs(ref_bf16 - truth).max() + 2e-2
    assert np.abs(host(ll) - truth).max() <= budget, (host(ll), truth)
    ll2 = m.classify(t(g["input_ids"]), t(g["class_input_ids"]), class_batch_size=2, **kw)
    # chunking only changes GEMM shapes; a bf16 model returns bf16 log-likelihoods, whose spacing at |ll| ~ 30 is 0.125: allow one ulp
    ulp = 2.0 ** -7 * np.abs(host(ll)).max() if dtype == torch.bfloat16 else 0.0
    assert np.abs(host(ll2) - host(ll)).max() <= max(2e-2, ulp)
    if dtype == torch.float32:
        assert np.array_equal(host(ll).argmax(-1), truth.argmax(-1)) or np.sort(truth, -1)[:, -1].min() - np.sort(truth, -1)[:, -2].max() < 5e-2


def test_vision_model_debug_outputs_like_the_reference(golden_dir):
    """VideoBlipVisionModel(output_hidden_states=True, output_attentions=True): shapes of ref:tests/model/test_model_v2.py:57-83,
    values against the reference's own run (tests/golden/mid_vitdebug.npz, eager attention)."""
    import json
    import os

    import numpy as np

    from eilev_amd.configs import blip2_config
    from eilev_amd.model.v2 import VideoBlipVisionModel
    from eilev_amd.synth import synth_pixels
    from hip_utils import rel_rms
    from oracle.runner import synth_state_dict

    g = np.load(os.path.join(golden_dir, "mid_vitdebug.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    vc = cfg.vision_config
    vm = VideoBlipVisionModel(vc)
    sd = synth_state_dict(cfg)
    vm.load_state_dict({k[len("vision_model."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("vision_model.")})
    vm = vm.cuda().eval()
    px = torch.from_numpy(synth_pixels(meta["clips"], meta["frames"], vc.image_size)).cuda()
    N, T, L = meta["clips"], meta["frames"], vc.num_hidden_layers
    tok = (vc.image_size // vc.patch_size) ** 2 + 1
    o = vm(px, output_attentions=True, output_hidden_states=True, return_dict=True)
    assert o.last_hidden_state.shape == (N, T * tok, vc.hidden_size) and o.pooler_output.shape == (N, T, vc.hidden_size)
    assert len(o.hidden_states) == L + 1 and all(h.shape == (N, T * tok, vc.hidden_size) for h in o.hidden_states)
    assert len(o.attentions) == L and all(a.shape == (N, T, vc.num_attention_heads, tok, tok) for a in o.attentions)
    hid = np.stack([h.float().cpu().numpy() for h in o.hidden_states])
    att = np.stack([a.float().cpu().numpy() for a in o.attentions])
    ref_dev = np.abs(g["bf16_hidden_states"] - g["fp32_hidden_states"]).max()
    assert np.abs(hid - g["fp32_hidden_states"]).max() <= 1.5 * ref_dev + 1e-3 and rel_rms(hid, g["fp32_hidden_states"]) <= 1e-2
    assert np.abs(att - g["fp32_attentions"]).max() <= 1.5 * np.abs(g["bf16_attentions"] - g["fp32_attentions"]).max() + 4e-3
    # the debug path and the fused path agree on what both return
    fused = vm(px, return_dict=True)
    assert torch.equal(fused.last_hidden_state, o.last_hidden_state) and torch.equal(fused.pooler_output, o.pooler_output)
    tup = vm(px, output_hidden_states=True, return_dict=False)
    assert len(tup) == 4 and tup[3] is None and len(tup[2]) == L + 1  # the fixed 4-tuple of ref:eilev/model/v2.py:103 (None placeholders)


def test_c1_workload_through_the_model_class():
    """BASELINE configs[0] (C1) on the GPU path through the drop-in class: ONE clip x 8 frames of 224 x 224, no in-context examples, a
    prompt of 48 positions (32 video slots + text), 32 greedy tokens — at the real ViT-g / Q-Former / OPT-2.7B widths (one block per
    stack: `real_1l`), checked against the CPU oracle on the same weights: ids exact, logits at the bf16 noise level."""
    from eilev_amd.synth import synth_interleaved_ids, synth_pixels
    from oracle.runner import OracleModel

    cfg = blip2_config("real_1l")
    m = build("real_1l", torch.bfloat16)
    sd = synth_state_dict(cfg)
    ora = OracleModel(cfg, sd)
    nq = cfg.num_query_tokens
    ids, vm = synth_interleaved_ids([1], [48 - 1 - nq - 1], nq, cfg.text_config.vocab_size)
    assert ids.shape[0] == 48
    ids, vm = ids[None], vm[None]
    am = np.ones_like(ids)
    px = synth_pixels(1, 8, cfg.vision_config.image_size)
    t = lambda a: torch.from_numpy(a).cuda()
    out = m(input_ids=t(ids), attention_mask=t(am), pixel_values=t(px).to(torch.bfloat16), video_input_mask=t(vm), return_dict=True)
    ref_logits = ora.forward_logits(px, ids, am, vm)
    assert rel_rms(host(out.logits), ref_logits) <= 1.2e-2
    got = m.generate(input_ids=t(ids), pixel_values=t(px).to(torch.bfloat16), video_input_mask=t(vm), attention_mask=t(am),
                     max_new_tokens=32, min_new_tokens=32, num_beams=1, do_sample=False)
    ref = ora.generate(px, ids, am, vm, 32, eos_id=-1)
    assert got.shape == (1, 32) and np.array_equal(got.cpu().numpy(), ref)


def test_generate_stopping_rules_and_return_sequences(golden_dir):
    """generate() kwargs of round 3 on the HIP decode step: several eos ids and 0 < min_new_tokens < max_new_tokens (host-side stopping
    rule over the same decode step: must agree with the captured greedy path wherever both apply), num_return_sequences with sampling
    (hf `_expand_inputs_for_generation`: rows of one prompt adjacent), and HF's errors."""
    g, meta, px = load_case(golden_dir, "mid_b2")
    m = build(meta["config"], torch.bfloat16)
    t = lambda a: torch.from_numpy(a).cuda()
    kw = dict(input_ids=t(g["input_ids"]), pixel_values=t(px).to(torch.bfloat16), video_input_mask=t(g["video_input_mask"]),
              attention_mask=t(g["attention_mask"]))
    n = meta["new_tokens"]
    eos = int(g["fp32_eos_id"])
    base = m.generate(**kw, max_new_tokens=n, eos_token_id=eos).cpu().numpy()
    assert np.array_equal(base, g["fp32_greedy_eos"])
    # a list whose extra id never occurs changes nothing (host rule == device rule)
    never = int(np.setdiff1d(np.arange(3, 200), g["fp32_greedy_free"].ravel())[0])
    assert np.array_equal(m.generate(**kw, max_new_tokens=n, eos_token_id=[eos, never]).cpu().numpy(), base)
    # two live ids: every row stops at its first occurrence of either, pads after it
    free = g["fp32_greedy_free"]
    second = int(free[0, 1])
    both = m.generate(**kw, max_new_tokens=n, eos_token_id=[eos, second], pad_token_id=1).cpu().numpy()
    for r in range(free.shape[0]):
        hits = np.nonzero(np.isin(free[r], [eos, second]))[0]
        stop = (int(hits[0]) + 1) if len(hits) else n
        assert np.array_equal(both[r, :min(stop, both.shape[1])], free[r, :min(stop, both.shape[1])])
        assert (both[r, stop:] == 1).all()
    # min_new_tokens: an eos that greedy would emit earlier cannot fire before that many tokens
    first = int(free[0, 0])
    late = m.generate(**kw, max_new_tokens=n, min_new_tokens=3, eos_token_id=first).cpu().numpy()
    assert late.shape[1] >= 3 and not (late[:, :3] == first).any()
    # sampling: N sequences per prompt, rows of a prompt adjacent; greedy refuses N > 1 like HF
    gen = torch.Generator(device="cuda").manual_seed(5)
    smp = m.generate(**kw, max_new_tokens=4, min_new_tokens=4, do_sample=True, top_k=1, num_return_sequences=3, generator=gen).cpu().numpy()
    assert smp.shape == (3 * free.shape[0], 4)
    for r in range(free.shape[0]):  # top_k = 1 is greedy: the three rows of a prompt are that prompt's greedy continuation
        assert (smp[3 * r: 3 * r + 3] == free[r, :4]).all()
    with pytest.raises(ValueError):
        m.generate(**kw, max_new_tokens=4, num_return_sequences=2)
    m.hf_device_map = {"": 0}  # accelerate's hook is a no-op on the single-device HIP engine
    assert np.array_equal(m.generate(**kw, max_new_tokens=n, eos_token_id=eos).cpu().numpy(), base)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hidden_states_of_the_language_model_and_qformer_like_the_reference(golden_dir, dtype):
    """`forward(output_hidden_states=True)` [ref:eilev/model/v2.py:187-193, 220-227]: the language model's tuple (every block's input, then
    the output of final_layer_norm: hf OPTDecoder.forward) and the Q-Former's, against the reference's own tuples
    (tests/golden/mid_lmdebug.npz, made by tools/make_goldens.py from the inputs of mid_b2: left padding, two rows)."""
    import json
    import os

    from eilev_amd.synth import synth_pixels

    g = np.load(os.path.join(golden_dir, "mid_lmdebug.npz"))
    meta = json.loads(str(g["meta"]))
    m = build(meta["config"], dtype)
    px = synth_pixels(sum(sum(c) for c, _ in meta["rows"]), meta["frames"], m.config.vision_config.image_size)
    t = lambda a: torch.from_numpy(a).cuda()
    out = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype),
            video_input_mask=t(g["video_input_mask"]), output_hidden_states=True, return_dict=True)
    hs = out.language_model_outputs.hidden_states
    ref = g["fp32_lm_hidden_states"]
    assert len(hs) == m.config.text_config.num_hidden_layers + 1 == ref.shape[0]
    assert all(h.dtype == dtype and tuple(h.shape) == ref.shape[1:] for h in hs)
    valid = g["attention_mask"] == 1  # hf leaves the rows of left-pad positions undefined
    for i, h in enumerate(hs):
        ref_dev = rel_rms(g["bf16_lm_hidden_states"][i][valid], ref[i][valid])  # what bf16 costs the reference itself
        assert rel_rms(host(h)[valid], ref[i][valid]) <= max(1.5 * ref_dev, 2e-3) + 2e-3, i
    qh, qref = out.qformer_outputs.hidden_states, g["fp32_qformer_hidden_states"]
    assert len(qh) == qref.shape[0]
    for i, h in enumerate(qh):
        assert rel_rms(host(h), qref[i]) <= 1e-2, i
    # the export changes nothing else
    plain = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype),
              video_input_mask=t(g["video_input_mask"]), return_dict=True)
    assert torch.equal(plain.logits, out.logits) and plain.language_model_outputs.hidden_states is None
    assert rel_rms(host(out.logits)[valid], g["fp32_logits"][valid]) <= 1.2e-2


def test_generate_passes_logits_processors_and_stopping_criteria(golden_dir):
    """hf hands every generate kwarg on (ref:eilev/model/v2.py:318-322): repetition_penalty / no_repeat_ngram_size / logits_processor /
    stopping_criteria / max_time run in the host loops over the HIP decode step (tests/test_generate_rules.py pins those loops to
    transformers token for token on the CPU)."""
    from transformers import LogitsProcessorList, StoppingCriteriaList

    g, meta, px = load_case(golden_dir, "mid_b2")
    m = build(meta["config"], torch.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    kw = dict(input_ids=t(g["input_ids"]), pixel_values=t(px), video_input_mask=t(g["video_input_mask"]), attention_mask=t(g["attention_mask"]),
              max_new_tokens=10, do_sample=False, eos_token_id=511)
    plain = m.generate(**kw, num_beams=1)
    assert (plain[:, 1:] == plain[:, :-1]).any()  # the 'fanin' model repeats itself ...
    ng = m.generate(**kw, num_beams=1, no_repeat_ngram_size=1)
    assert ng.shape == plain.shape and all(len(set(r.tolist())) == r.numel() for r in ng)  # ... and may not with the 1-gram ban
    rp = m.generate(**kw, num_beams=1, repetition_penalty=50.0)
    assert not torch.equal(rp, plain)
    bm = m.generate(**kw, num_beams=3, no_repeat_ngram_size=2)
    for r in bm.tolist():
        grams = list(zip(r, r[1:]))
        assert len(set(grams)) == len(grams)

    class Ban:
        def __call__(self, input_ids, scores):
            scores = scores.clone()
            scores[:, int(plain[0, 0])] = float("-inf")
            return scores

    class Stop3:
        def __call__(self, input_ids, scores, **kw):
            return torch.full((input_ids.shape[0],), input_ids.shape[1] >= 3, dtype=torch.bool, device=input_ids.device)

    banned = m.generate(**kw, num_beams=1, logits_processor=LogitsProcessorList([Ban()]))
    assert not (banned == plain[0, 0]).any()
    short = m.generate(**kw, num_beams=1, stopping_criteria=StoppingCriteriaList([Stop3()]))
    assert short.shape[1] == 3 and torch.equal(short, plain[:, :3])
    assert m.generate(**kw, num_beams=1, max_time=1e-9).shape[1] == 1  # the time budget is checked after the first token, as in hf


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_output_attentions_of_the_qformer_and_the_language_model(golden_dir, dtype):
    """`forward(output_attentions=True)` [ref:eilev/model/v2.py:187-193, 220-227]: Q-Former self / cross and OPT attention weights on the HIP
    path against the reference's eager run (tests/golden/mid_attndebug.npz); as close to its fp32 weights as its own bf16 run (+ slack)."""
    import json
    import os

    g = np.load(os.path.join(golden_dir, "mid_attndebug.npz"))
    meta = json.loads(str(g["meta"]))
    from eilev_amd.synth import synth_pixels

    cfg = blip2_config(meta["config"])
    px = synth_pixels(sum(sum(c) for c, _ in meta["rows"]), meta["frames"], cfg.vision_config.image_size)
    m = build(meta["config"], dtype)
    t = lambda a: torch.from_numpy(a).cuda()
    o = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype), video_input_mask=t(g["video_input_mask"]),
          output_attentions=True, return_dict=True)
    n_all, n_cross = [int(x) for x in g["qformer_counts"]]
    qa, qx = o.qformer_outputs.attentions, o.qformer_outputs.cross_attentions
    assert len(qa) == n_all and len(qx) == n_cross and all(a.dtype == dtype for a in qa)
    for i, a in enumerate(qa):
        ref, ref16 = g[f"fp32_qformer_attentions_{i}"], g[f"bf16_qformer_attentions_{i}"]
        assert a.shape == ref.shape
        assert np.abs(host(a) - ref).max() <= 1.5 * np.abs(ref16 - ref).max() + 4e-3, i
    for i, a in enumerate(qx):
        assert torch.equal(a, qa[1 + 2 * i * 0]) or a.shape == g[f"fp32_qformer_cross_attentions_{i}"].shape
        assert np.abs(host(a) - g[f"fp32_qformer_cross_attentions_{i}"]).max() <= 1.5 * np.abs(g[f"bf16_qformer_cross_attentions_{i}"] - g[f"fp32_qformer_cross_attentions_{i}"]).max() + 4e-3
    la = o.language_model_outputs.attentions
    ref, ref16 = g["fp32_lm_attentions"], g["bf16_lm_attentions"]
    assert len(la) == ref.shape[0] and la[0].shape == ref.shape[1:]
    valid = g["attention_mask"] == 1
    for l, a in enumerate(la):
        for b in range(a.shape[0]):
            got = host(a)[b][:, valid[b]]
            assert np.abs(got - ref[l, b][:, valid[b]]).max() <= 1.5 * np.abs(ref16[l, b][:, valid[b]] - ref[l, b][:, valid[b]]).max() + 4e-3
            assert np.abs(got.sum(-1) - 1.0).max() < 2e-2
    # logits unchanged by the debug outputs
    plain = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype), video_input_mask=t(g["video_input_mask"]))
    assert torch.equal(plain.logits, o.logits)


def test_generate_return_dict_with_scores_like_the_reference(golden_dir):
    """`generate(return_dict_in_generate=True, output_scores=True)` — hf hands the kwargs to GenerationMixin (ref:eilev/model/v2.py:318-322):
    a ModelOutput with `sequences` and one (rows, vocab) fp32 score tensor per generated token.  tests/golden/mid_v2.npz holds exactly what
    the REFERENCE returns for that call (tools/make_goldens.py::_greedy_with_scores: ids + every step's scores, two rows, left padding)."""
    g, meta, px = load_case(golden_dir, "mid_v2")
    from eilev_amd.configs import blip2_config
    from oracle.runner import synth_state_dict

    cfg = blip2_config(meta["config"])
    sd = synth_state_dict(cfg, meta["weight_mode"], meta["weight_seed"])
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration

    m = VideoBlipForConditionalGeneration(cfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    sd["language_model.lm_head.weight"] = sd["language_model.model.decoder.embed_tokens.weight"]
    m.load_state_dict(sd)
    m = m.to(torch.bfloat16).to("cuda")
    t = lambda a: torch.from_numpy(a).cuda()
    kw = dict(input_ids=t(g["input_ids"]), pixel_values=t(px).to(torch.bfloat16), video_input_mask=t(g["video_input_mask"]),
              attention_mask=t(g["attention_mask"]))
    n = meta["new_tokens"]
    out = m.generate(**kw, max_new_tokens=n, eos_token_id=meta["never_id"], return_dict_in_generate=True, output_scores=True, output_logits=True)
    assert np.array_equal(out.sequences.cpu().numpy(), g["fp32_greedy_free"])
    assert len(out.scores) == n == len(out.logits) and all(s_.dtype == torch.float32 and tuple(s_.shape) == g["fp32_step_logits"].shape[1:] for s_ in out.scores)
    ref, ref16 = g["fp32_step_logits"], g["bf16_step_logits"]
    for k in range(n):
        err, dev = rel_rms(host(out.scores[k]), ref[k]), rel_rms(ref16[k], ref[k])
        assert err <= 1.5 * dev + 2e-3, (k, err, dev)
        assert torch.equal(out.scores[k], out.logits[k])
    # sequences only: any decoding mode; scores without the dict flag change nothing (hf semantics)
    seq = m.generate(**kw, max_new_tokens=n, eos_token_id=meta["never_id"], return_dict_in_generate=True)
    assert np.array_equal(seq.sequences.cpu().numpy(), g["fp32_greedy_free"]) and seq.scores is None
    plain = m.generate(**kw, max_new_tokens=n, eos_token_id=meta["never_id"], output_scores=True)
    assert torch.is_tensor(plain) and np.array_equal(plain.cpu().numpy(), g["fp32_greedy_free"])
    beams = m.generate(**kw, max_new_tokens=6, num_beams=3, eos_token_id=meta["never_id"], return_dict_in_generate=True)
    assert beams.sequences.shape[0] == g["input_ids"].shape[0]
    with pytest.raises(NotImplementedError):
        m.generate(**kw, max_new_tokens=4, num_beams=3, return_dict_in_generate=True, output_scores=True)
