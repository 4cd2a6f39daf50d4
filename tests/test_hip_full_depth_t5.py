"""-m gpu: the encoder-decoder backbone at FULL DEPTH — BASELINE configs[3] (eilev-blip2-flan-t5-xl: 39 ViT-g + 12 Q-Former + 24 T5 encoder
+ 24 T5 decoder blocks at the real widths) on the C1-sized input (1 clip x 8 frames, L = 47 in the T5 token layout), against the
reference's own fp32 and bf16 runs (tests/golden/full_t5.npz, produced by `tools/make_goldens.py full_t5` from /root/reference in the
build container; weights by recipe: eilev_amd.synth 'fanin' — 4 G parameters are generated here on the device, not stored).

Until round 5 the T5 path was pinned to the reference at ONE block per stack only (real_t5_b1; VERDICT r4 missing 2).  Checked here:
sampled rows of the text encoder's output, sampled columns of the teacher-forced logits and their argmax, the loss-free greedy ids
(the reference's two precisions agree on all of them).  Tolerances as in tests/test_hip_real_shapes.py::test_real_width_t5_path."""
import json
import os

import numpy as np
import pytest
import torch

from eilev_amd.configs import blip2_config
from eilev_amd.statedict import state_dict_shapes
from eilev_amd.synth import synth_param_torch, synth_pixels
from hip_utils import host, record_parity, rel_rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_t5(golden_dir):
    from eilev_amd.engine import HipEngine

    g = np.load(os.path.join(golden_dir, "full_t5.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    sd = {}
    for k, shp in state_dict_shapes(cfg).items():  # the recipe evaluated ON the device (bit-identical to numpy: tests/test_synth_torch.py)
        sd[k] = synth_param_torch(k, shp, meta["weight_mode"], 0, device="cuda").to(torch.bfloat16)
    eng = HipEngine(cfg, sd, device="cuda")
    del sd
    px = torch.from_numpy(synth_pixels(1, meta["frames"], cfg.vision_config.image_size)).cuda()
    t = lambda k: torch.from_numpy(g[k]).cuda()
    emb = eng.embed_scatter(t("input_ids"), t("video_input_mask"), eng.encode_clips(px))
    return g, meta, eng, emb, t("attention_mask")


def test_full_depth_t5_encoder_and_logits(full_t5):
    g, meta, eng, emb, am = full_t5
    logits, enc = eng.t5_forward(emb, am, torch.from_numpy(g["decoder_input_ids"]).cuda())
    e = host(enc)[:, g["enc_rows"]]
    ref_dev = float(np.abs(g["bf16_enc_rows"] - g["fp32_enc_rows"]).max())
    record_parity("full_depth[t5]", enc_hip_vs_fp32_maxabs=float(np.abs(e - g["fp32_enc_rows"]).max()), enc_refbf16_vs_fp32_maxabs=ref_dev,
                  enc_hip_vs_fp32_relrms=rel_rms(e, g["fp32_enc_rows"]), enc_refbf16_vs_fp32_relrms=rel_rms(g["bf16_enc_rows"], g["fp32_enc_rows"]))
    assert np.abs(e - g["fp32_enc_rows"]).max() <= 1.5 * ref_dev + 2e-3 * float(np.abs(g["fp32_enc_rows"]).max())
    assert rel_rms(e, g["fp32_enc_rows"]) <= 1.5 * rel_rms(g["bf16_enc_rows"], g["fp32_enc_rows"]) + 2e-3
    lg = host(logits)[:, :, g["logit_cols"]]
    record_parity("full_depth[t5]", logits_hip_vs_fp32_relrms=rel_rms(lg, g["fp32_logits_cols"]),
                  logits_refbf16_vs_fp32_relrms=rel_rms(g["bf16_logits_cols"], g["fp32_logits_cols"]),
                  logits_hip_vs_refbf16_relrms=rel_rms(lg, g["bf16_logits_cols"]))
    assert np.abs(lg - g["fp32_logits_cols"]).max() <= 2.0 * np.abs(g["bf16_logits_cols"] - g["fp32_logits_cols"]).max() + 2e-3 * float(np.abs(g["fp32_logits_cols"]).max())
    assert rel_rms(lg, g["fp32_logits_cols"]) <= 1.5 * rel_rms(g["bf16_logits_cols"], g["fp32_logits_cols"]) + 2e-3
    # the argmax of every teacher-forced position (full vocabulary), where the reference's two precisions agree
    agree = g["fp32_logits_argmax"] == g["bf16_logits_argmax"]
    mine = host(logits).argmax(-1)
    assert np.array_equal(mine[agree], g["fp32_logits_argmax"][agree]), (mine, g["fp32_logits_argmax"])


@pytest.mark.parametrize("use_graph", [True, False])
def test_full_depth_t5_greedy_ids(full_t5, use_graph):
    g, meta, eng, emb, am = full_t5
    assert np.array_equal(g["fp32_greedy_free"], g["bf16_greedy_free"])
    ids = eng.t5_greedy(emb, am, meta["new_tokens"], eos_id=-1, use_graph=use_graph).cpu().numpy()
    assert ids[0, 0] == g["fp32_greedy_free"][0, 0]  # decoder start token
    # Round 6: free-running ids are compared like the full-depth OPT fixtures' — equal up to the first NEAR-TIE OF THE REFERENCE, where the id must
    # be the reference's runner-up (oracle/parity.py).  The fixture now stores the reference's eight leading logits of every step
    # (tools/make_goldens.py full_t5; the older keys regenerated bit-identically): at step 2 the reference's own fp32 run separates 30692 from
    # 30568 by 0.0003 (4.0008 vs 4.0005; its bf16 run moves those logits by 0.03) — a coin flip that the degree-8 GELU of rounds 2-5 happened to
    # land on the reference's side of and round 6's x * Phi(x) form (one bf16 ulp away on some fc1 outputs) does not.
    from oracle.parity import greedy_ids_vs_reference

    view = {k: g[k] for k in ("fp32_step_logits_top8", "fp32_step_logits_top8_ids", "bf16_step_logits_top8", "bf16_step_logits_top8_ids")}
    view["fp32_greedy_free"] = g["fp32_greedy_free"][:, 1:]
    verdict = greedy_ids_vs_reference(ids[:, 1:], view)
    record_parity("full_depth[t5]", **{f"greedy_graph{int(use_graph)}_ids_equal_before_first_flip": verdict["ids_equal_before_first_flip"],
                                       f"greedy_graph{int(use_graph)}_flips": len(verdict["flips"])})
    assert verdict["ok"], (verdict, ids, g["fp32_greedy_free"])
    assert verdict["ids_equal_before_first_flip"] >= 2
