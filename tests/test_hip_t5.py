"""-m gpu: the encoder-decoder LM path (flan-t5 family, BASELINE configs[3]) on the HIP kernels vs the reference goldens.

Judged like the OPT path (tests/test_hip_stages.py): error against the reference's fp32 run no larger than 1.5x the
reference's own bf16 run (+ slack), greedy ids exact."""
import numpy as np
import pytest
import torch

from hip_utils import host, load_case, models, rel_rms

pytestmark = pytest.mark.gpu

CASES = ["mid_t5_b1", "mid_t5_b2"]


def _encode(eng, g, px):
    t = lambda a: torch.from_numpy(a).cuda()
    feats = eng.encode_clips(t(px))
    return eng.embed_scatter(t(g["input_ids"]), t(g["video_input_mask"]), feats)


@pytest.mark.parametrize("name", CASES)
def test_t5_logits_vs_reference(golden_dir, name):
    g, meta, px = load_case(golden_dir, name)
    cfg, oracle, eng = models(meta["config"])
    t = lambda a: torch.from_numpy(a).cuda()
    emb = _encode(eng, g, px)
    logits, enc = eng.t5_forward(emb, t(g["attention_mask"]), t(g["decoder_input_ids"]))
    valid = g["attention_mask"] == 1
    truth_e, ref_e = g["fp32_enc"][valid], g["bf16_enc"][valid]
    assert np.abs(host(enc)[valid] - truth_e).max() <= 1.5 * np.abs(ref_e - truth_e).max() + 1e-3
    # T5 keeps an un-normalised bf16 residual stream: the reference's own bf16 run is 3-4e-2 off its fp32 run here
    assert rel_rms(host(enc)[valid], truth_e) <= 1.2 * rel_rms(ref_e, truth_e) + 2e-3
    keep = g["labels"] >= 0
    truth, ref = g["fp32_logits"], g["bf16_logits"]
    assert np.abs(host(logits) - truth)[keep].max() <= 2.0 * np.abs(ref - truth)[keep].max() + 1e-3  # max of a noisy field
    assert rel_rms(host(logits)[keep], truth[keep]) <= 1.2 * rel_rms(ref[keep], truth[keep]) + 2e-3
    # and against the CPU oracle (fp32) on the same inputs
    lo, _ = oracle.t5_forward_logits(px, g["input_ids"], g["attention_mask"], g["video_input_mask"], g["decoder_input_ids"])
    assert rel_rms(host(logits)[keep], lo[keep]) <= 1.2 * rel_rms(ref[keep], truth[keep]) + 2e-3


@pytest.mark.parametrize("name", CASES)
def test_t5_greedy_ids(golden_dir, name):
    """Greedy decoding.  With random weights the T5 top-2 logit margins (0.01-0.1) are below the bf16 noise of the model
    (the reference's own bf16 run picks other tokens than its fp32 run on these fixtures), so ids are checked two ways:
    a row must reproduce the reference's fp32 OR bf16 ids exactly, or every token it picked must be within the bf16 noise
    budget of the fp32 oracle's best token when the oracle is teacher-forced on the generated prefix."""
    g, meta, px = load_case(golden_dir, name)
    cfg, oracle, eng = models(meta["config"])
    t = lambda a: torch.from_numpy(a).cuda()
    emb = _encode(eng, g, px)
    n = meta["new_tokens"]
    budget = 1.5 * np.abs(g["bf16_logits"] - g["fp32_logits"]).max() + 1e-2
    for eos, key in ((-1, "greedy_free"), (int(g["fp32_eos_id"]), "greedy_eos")):
        ids = eng.t5_greedy(emb, t(g["attention_mask"]), n, eos_id=eos).cpu().numpy()
        assert ids.shape[0] == g["input_ids"].shape[0] and (ids[:, 0] == 0).all()
        lo, _ = oracle.t5_forward_logits(px, g["input_ids"], g["attention_mask"], g["video_input_mask"], ids[:, :-1])
        for b in range(ids.shape[0]):
            exact = any(r.shape[1] >= ids.shape[1] and np.array_equal(r[b, : ids.shape[1]], ids[b]) for r in (g[f"fp32_{key}"], g[f"bf16_{key}"]))
            if exact:
                continue
            # the two reference runs agree on this row: the margins are above the bf16 noise and so must we
            assert not np.array_equal(g[f"fp32_{key}"][b], g[f"bf16_{key}"][b]), (ids[b], g[f"fp32_{key}"][b])
            done = False
            for i in range(ids.shape[1] - 1):
                tok = ids[b, i + 1]
                if done:
                    assert tok == 0  # pad after EOS
                    continue
                assert lo[b, i, tok] >= lo[b, i].max() - budget, (b, i, tok)
                done = eos >= 0 and tok == eos


def test_t5_decode_step_equals_teacher_forcing(golden_dir):
    """Size-independent property: feeding the target prefix one token at a time through the self-attention cache gives the
    logits of one teacher-forced pass (causal mask + relative bias of a single query row == row of the full bias)."""
    g, meta, px = load_case(golden_dir, "mid_t5_b2")
    cfg, oracle, eng = models(meta["config"])
    t = lambda a: torch.from_numpy(a).cuda()
    import ctypes as C
    emb = _encode(eng, g, px)
    am = t(g["attention_mask"])
    dec = t(g["decoder_input_ids"])
    full, enc = eng.t5_forward(emb, am, dec)
    ckv = eng.t5_cross_kv(enc)
    B, T = dec.shape
    skv = torch.zeros(int(eng.lib.eilev_t5_self_kv_bytes(C.byref(eng.t5dims), B, T)), dtype=torch.uint8, device="cuda")
    for i in range(T):
        step = eng.t5_decode(dec[:, i:i + 1], am, i, skv, T, ckv, enc.shape[1])
        assert rel_rms(host(step[:, 0]), host(full[:, i])) <= 2e-2, i


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_t5_model_class_like_the_reference(golden_dir, dtype):
    """The drop-in class with an encoder-decoder text config: forward(labels=...) / forward(decoder_input_ids=...) / generate()."""
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration
    from oracle.runner import synth_state_dict

    g, meta, px = load_case(golden_dir, "mid_t5_b2")
    cfg = models(meta["config"])[0]
    m = VideoBlipForConditionalGeneration(cfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(("lm_head.weight", "embed_tokens.weight")) for k in missing), (missing, unexpected)
    m = m.to(dtype).to("cuda")
    t = lambda a: torch.from_numpy(a).cuda()
    out = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype),
            video_input_mask=t(g["video_input_mask"]), labels=t(g["labels"]), return_dict=True)
    keep = g["labels"] >= 0
    truth, ref = g["fp32_logits"], g["bf16_logits"]
    assert out.logits.dtype == dtype and out.logits.shape == truth.shape
    assert rel_rms(host(out.logits)[keep], truth[keep]) <= 1.5 * rel_rms(ref[keep], truth[keep]) + 4e-3
    assert abs(float(out.loss) - float(g["fp32_loss"])) <= 2e-2 * abs(float(g["fp32_loss"]))
    out2 = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(dtype),
             video_input_mask=t(g["video_input_mask"]), decoder_input_ids=t(g["decoder_input_ids"]), return_dict=True)
    assert torch.equal(out2.logits, out.logits)
    n = meta["new_tokens"]
    ids = m.generate(input_ids=t(g["input_ids"]), pixel_values=t(px).to(dtype), video_input_mask=t(g["video_input_mask"]),
                     attention_mask=t(g["attention_mask"]), max_new_tokens=n, num_beams=1, do_sample=False, eos_token_id=int(g["fp32_eos_id"]))
    ref_ids = g["fp32_greedy_eos"]
    assert np.array_equal(ids.cpu().numpy()[:, : ref_ids.shape[1]], ref_ids)


def test_t5_decoder_mask_and_hidden_states(golden_dir):
    """decoder_attention_mask with padding + both stacks' hidden_states (eilev_t5_encode_debug / eilev_t5_decode_debug) against the
    reference's runs of the same call (tests/golden/mid_t5_dbg.npz) and the oracle; then the same through the drop-in class."""
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration
    from oracle.runner import synth_state_dict

    g, meta, px = load_case(golden_dir, "mid_t5_dbg")
    cfg, oracle, eng = models(meta["config"])
    t = lambda a: torch.from_numpy(a).cuda()
    emb = _encode(eng, g, px)
    logits, enc, enc_hs, dec_hs = eng.t5_forward_debug(emb, t(g["attention_mask"]), t(g["decoder_input_ids"]), t(g["decoder_attention_mask"]), True)
    truth, ref = g["fp32_logits"], g["bf16_logits"]
    assert rel_rms(host(logits), truth) <= 1.2 * rel_rms(ref, truth) + 2e-3
    assert np.abs(host(logits) - truth).max() <= 2.0 * np.abs(ref - truth).max() + 1e-3
    # the mask is really applied: the unmasked run is far away from both
    assert rel_rms(host(logits), g["fp32_logits_nomask"]) > 5 * rel_rms(host(logits), truth)
    valid = g["attention_mask"] == 1
    for got, key, sel in ((enc_hs, "enc_hidden", (slice(None), valid)), (dec_hs, "dec_hidden", (slice(None),))):
        tr, rf = g[f"fp32_{key}"][sel], g[f"bf16_{key}"][sel]
        assert host(got).shape == g[f"fp32_{key}"].shape
        for l in range(tr.shape[0]):
            assert rel_rms(host(got)[sel][l], tr[l]) <= 1.2 * rel_rms(rf[l], tr[l]) + 2e-3, (key, l)
    assert torch.equal(enc_hs[-1], enc)
    lo, eo, do = oracle.t5_forward_debug(px, g["input_ids"], g["attention_mask"], g["video_input_mask"], g["decoder_input_ids"],
                                         g["decoder_attention_mask"])
    assert rel_rms(host(logits), lo) <= 1.2 * rel_rms(ref, truth) + 2e-3
    # output_attentions: T5Attention weights of every block (encoder self, decoder self with the mask, cross) vs the reference's eager runs
    enc_a, dec_a, cross_a = eng.t5_attentions(enc_hs, dec_hs, t(g["attention_mask"]), t(g["decoder_attention_mask"]))
    for got, key in ((enc_a, "enc_attn"), (dec_a, "dec_attn"), (cross_a, "cross_attn")):
        tr, rf = g[f"fp32_{key}"], g[f"bf16_{key}"]
        assert host(got).shape == tr.shape
        sel = np.broadcast_to(valid[None, :, None, :, None], tr.shape) if key == "enc_attn" else np.ones_like(tr, bool)  # (padded encoder query rows: any)
        assert np.abs(host(got) - tr)[sel].max() <= 1.5 * np.abs(rf - tr)[sel].max() + 4e-3, key  # probabilities: bf16 output grid 2^-9
    oe, od, oc = oracle.t5_attentions(eo, do, g["attention_mask"], g["decoder_attention_mask"])
    assert np.abs(host(dec_a) - od).max() <= 1.5 * np.abs(g["bf16_dec_attn"] - g["fp32_dec_attn"]).max() + 4e-3
    assert float(dec_a[:, 1, :, :, 5:].abs().max()) == 0.0 and float(dec_a[:, 0, :, :, 3].abs().max()) == 0.0  # the masked target keys
    # the plain entries are the same arithmetic when nothing is masked
    l0, e0 = eng.t5_forward(emb, t(g["attention_mask"]), t(g["decoder_input_ids"]))
    l1, e1, _, _ = eng.t5_forward_debug(emb, t(g["attention_mask"]), t(g["decoder_input_ids"]), torch.ones_like(t(g["decoder_attention_mask"])), False)
    assert torch.equal(l0, l1) and torch.equal(e0, e1)
    with pytest.raises(NotImplementedError):
        bad = g["decoder_attention_mask"].copy()
        bad[0, 0] = 0
        eng.t5_forward_debug(emb, t(g["attention_mask"]), t(g["decoder_input_ids"]), t(bad), False)
    # ---- the model class: ref:eilev/model/v2.py:228-238
    m = VideoBlipForConditionalGeneration(cfg).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg).items()}, strict=False)
    m = m.to(torch.bfloat16).to("cuda")
    out = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), pixel_values=t(px).to(torch.bfloat16),
            video_input_mask=t(g["video_input_mask"]), decoder_input_ids=t(g["decoder_input_ids"]),
            decoder_attention_mask=t(g["decoder_attention_mask"]), output_hidden_states=True, output_attentions=True, return_dict=True)
    lm = out.language_model_outputs
    assert len(lm.encoder_attentions) == cfg.text_config.num_layers and len(lm.decoder_attentions) == len(lm.cross_attentions) == cfg.text_config.num_decoder_layers
    assert lm.cross_attentions[0].shape == g["fp32_cross_attn"].shape[1:] and lm.decoder_attentions[0].dtype == torch.bfloat16
    assert np.abs(host(torch.stack(lm.cross_attentions)) - g["fp32_cross_attn"]).max() <= 1.5 * np.abs(g["bf16_cross_attn"] - g["fp32_cross_attn"]).max() + 4e-3
    assert len(lm.encoder_hidden_states) == cfg.text_config.num_layers + 1 and len(lm.decoder_hidden_states) == cfg.text_config.num_decoder_layers + 1
    assert rel_rms(host(out.logits), truth) <= 1.5 * rel_rms(ref, truth) + 4e-3
    assert rel_rms(host(torch.stack(lm.decoder_hidden_states)), g["fp32_dec_hidden"]) <= 1.5 * rel_rms(g["bf16_dec_hidden"], g["fp32_dec_hidden"]) + 4e-3
    assert torch.equal(lm.encoder_hidden_states[-1], lm.encoder_last_hidden_state)
    assert out.qformer_outputs.hidden_states is not None and out.vision_outputs.hidden_states is not None


@pytest.mark.parametrize("B", [3, 16])
def test_t5_xl_widths_decode_equals_teacher_forcing(B):
    """flan-t5-xl widths (d_model 2048, 32 heads x 64, d_ff 5120, vocab 32128; 2 + 2 layers), L = 300 with right padding:
    size-independent properties — cached single-step decoding == teacher forcing, and padded encoder positions do not
    influence the logits.  B = 16 (round 5): 16 rows x 32 heads = 2 workgroups per CU — the cross-attention of a decode step runs on
    attn_decode_loop_kernel<8, 8, 256> (two key ranges, the second ragged, one row with masked encoder positions); the graph-replayed greedy
    loop (new keys / values stored by the self-attention kernel itself) gives the ids of the rows decoded one at a time."""
    import ctypes as C

    from eilev_amd.configs import blip2_config
    from eilev_amd.engine import HipEngine
    from eilev_amd.statedict import state_dict_shapes
    from eilev_amd.synth import synth_param

    cfg = blip2_config("t5xl")
    cfg.text_config.num_layers = 2
    cfg.text_config.num_decoder_layers = 2
    named = {k: torch.from_numpy(synth_param(k, shp, "fanin")).to(torch.bfloat16).cuda()
             for k, shp in state_dict_shapes(cfg).items() if k.startswith("language_model")}
    eng = HipEngine(cfg, named, device="cuda", parts=("t5",))
    torch.manual_seed(3)
    L, T, D = 300, 5, 2048
    emb = (0.5 * torch.randn(B, L, D, device="cuda")).to(torch.bfloat16)
    am = torch.ones(B, L, dtype=torch.int32, device="cuda")
    am[1, 250:] = 0
    dec = torch.randint(2, 32128, (B, T), device="cuda")
    dec[:, 0] = 0
    full, enc = eng.t5_forward(emb, am, dec)
    emb2 = emb.clone()
    emb2[1, 250:] = 7.0  # garbage in the padded encoder positions
    full2, _ = eng.t5_forward(emb2, am, dec)
    assert torch.equal(full2[1], full[1])
    ckv = eng.t5_cross_kv(enc)
    skv = torch.zeros(int(eng.lib.eilev_t5_self_kv_bytes(C.byref(eng.t5dims), B, T)), dtype=torch.uint8, device="cuda")
    for i in range(T):
        step = eng.t5_decode(dec[:, i:i + 1], am, i, skv, T, ckv, L)
        assert rel_rms(host(step[:, 0]), host(full[:, i])) <= 1e-2, i
    if B >= 16:
        ids = eng.t5_greedy(emb, am, 6, eos_id=-1).cpu().numpy()
        for r in (0, 1, B - 1):
            one = eng.t5_greedy(emb[r:r + 1], am[r:r + 1], 6, eos_id=-1).cpu().numpy()
            assert np.array_equal(ids[r], one[0]), (r, ids[r], one[0])


@pytest.mark.parametrize("nm,nb,lp", [("beam5_lpm1", 5, -1.0), ("beam3_lp1", 3, 1.0)])
@pytest.mark.parametrize("name", CASES)
def test_t5_beam_search(golden_dir, name, nm, nb, lp):
    """generate(num_beams=k) for the encoder-decoder LM on the HIP path vs the reference's beam outputs (exact when the
    reference's fp32 and bf16 runs agree, else the fp32 or the bf16 ids)."""
    g, meta, px = load_case(golden_dir, name)
    cfg, oracle, eng = models(meta["config"])
    t = lambda a: torch.from_numpy(a).cuda()
    emb = _encode(eng, g, px)
    n = meta["new_tokens"]
    for eos, suffix in ((int(g["fp32_eos_id"]), ""), (-1, "_free")):
        ids = eng.t5_beam(emb, t(g["attention_mask"]), n, nb, lp, eos_id=eos).cpu().numpy()
        cands = [g[f"fp32_{nm}{suffix}"], g[f"bf16_{nm}{suffix}"]]
        if any(ids.shape == c.shape and np.array_equal(ids, c) for c in cands):
            continue
        # Beam search over a random-weight model has near-ties between hypotheses (the reference's own fp32 and bf16 runs
        # disagree on some of these cases).  A differing result is accepted only if it is such a tie: under the HIP model's
        # own teacher-forced log-probabilities the returned hypothesis must score within 2 % of a reference hypothesis.
        am = t(g["attention_mask"])

        def score(seq_rows):
            out = []
            for b, row in enumerate(seq_rows):
                toks = [int(x) for x in row[1:]]
                if eos >= 0 and eos in toks:
                    toks = toks[: toks.index(eos) + 1]
                dec = torch.tensor([[0] + toks[:-1]], device="cuda")
                logits, _ = eng.t5_forward(emb[b:b + 1], am[b:b + 1], dec)
                lp_ = torch.log_softmax(logits[0].float(), -1)
                out.append(float(sum(lp_[i, tok] for i, tok in enumerate(toks))) / len(toks) ** lp)
            return np.array(out)

        mine = score(ids)
        assert any(np.all(np.abs(mine - score(c)) <= 2e-2 * np.abs(score(c)) + 1e-3) for c in cands), (ids, cands, mine)
