"""Pins the CPU oracle at the REAL widths of eilev-blip2-opt-2.7b (tests/golden/real_b1.npz: one ViT block at 1408 / 6144 /
16 heads on 224 x 224 frames, a Q-Former block pair with cross-attention over the real 2056 keys, one OPT-2.7B block with
the 50272-token vocabulary; produced by tools/make_goldens.py from the reference).  Outputs are subsampled in the fixture;
tolerances as in test_oracle_golden.py (fp32, summation order only)."""
import json
import os

import numpy as np
import pytest

from eilev_amd.configs import blip2_config
from eilev_amd.synth import synth_pixels
from oracle.runner import OracleModel, synth_state_dict


@pytest.fixture(scope="module")
def case(golden_dir):
    g = np.load(os.path.join(golden_dir, "real_b1.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    px = synth_pixels(sum(sum(c) for c, _ in meta["rows"]), meta["frames"], cfg.vision_config.image_size)
    return g, meta, cfg, px, OracleModel(cfg, synth_state_dict(cfg))


def test_real_width_stages_match_reference(case):
    g, meta, cfg, px, m = case
    img, pool = m.vit(px, want_pooler=True)
    rows = img.reshape(-1, img.shape[-1])[g["vit_rows"]]
    assert np.abs(rows - g["fp32_vit_rows"]).max() < 2e-4
    assert abs(img.astype(np.float64).sum() - g["fp32_vit_checksum"][0]) < 1e-6 * g["fp32_vit_checksum"][1]
    assert np.abs(pool - g["fp32_pooler"]).max() < 2e-4
    q = m.qformer(img)
    assert np.abs(q - g["fp32_qformer"]).max() < 2e-4
    logits = m.forward_logits(px, g["input_ids"], g["attention_mask"], g["video_input_mask"])
    assert np.abs(logits[:, :, g["logit_cols"]] - g["fp32_logits_cols"]).max() < 5e-4
    assert np.abs(logits[:, -1] - g["fp32_logits_last"]).max() < 5e-4
    assert abs(logits.astype(np.float64).sum() - g["fp32_logits_checksum"][0]) < 1e-6 * g["fp32_logits_checksum"][1]


def test_real_width_greedy_ids_match_reference(case):
    g, meta, cfg, px, m = case
    ids = m.generate(px, g["input_ids"], g["attention_mask"], g["video_input_mask"], meta["new_tokens"], eos_id=-1)
    assert np.array_equal(ids, g["fp32_greedy_free"])


def test_real_width_t5_matches_reference(golden_dir):
    """flan-t5-xl widths (d_model 2048, 32 heads x 64, d_ff 5120, vocab 32128), one block per stack — tests/golden/real_t5_b1.npz."""
    g = np.load(os.path.join(golden_dir, "real_t5_b1.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    px = synth_pixels(sum(sum(c) for c, _ in meta["rows"]), meta["frames"], cfg.vision_config.image_size)
    m = OracleModel(cfg, synth_state_dict(cfg))
    logits, enc = m.t5_forward_logits(px, g["input_ids"], g["attention_mask"], g["video_input_mask"], g["decoder_input_ids"])
    assert np.abs(enc[:, g["enc_rows"]] - g["fp32_enc_rows"]).max() < 5e-4
    assert abs(enc.astype(np.float64).sum() - g["fp32_enc_checksum"][0]) < 1e-6 * g["fp32_enc_checksum"][1]
    assert np.abs(logits[:, :, g["logit_cols"]] - g["fp32_logits_cols"]).max() < 1e-3
    assert abs(logits.astype(np.float64).sum() - g["fp32_logits_checksum"][0]) < 1e-6 * g["fp32_logits_checksum"][1]
    assert np.array_equal(logits.argmax(-1), g["fp32_logits_argmax"])
    ids = m.t5_generate(px, g["input_ids"], g["attention_mask"], g["video_input_mask"], meta["new_tokens"], eos_id=-1)
    assert np.array_equal(ids, g["fp32_greedy_free"])


def test_real_width_varied_ids_match_reference(golden_dir):
    """tests/golden/real_v1.npz: the same real-width single-block model with weight mode 'varied' (small token embedding), whose
    reference greedy / beam outputs change from step to step and include a row stopped by EOS in the middle (12 tokens)."""
    g = np.load(os.path.join(golden_dir, "real_v1.npz"))
    meta = json.loads(str(g["meta"]))
    cfg = blip2_config(meta["config"])
    px = synth_pixels(sum(sum(c) for c, _ in meta["rows"]), meta["frames"], cfg.vision_config.image_size)
    m = OracleModel(cfg, synth_state_dict(cfg, meta["weight_mode"], meta["weight_seed"]))
    args = (px, g["input_ids"], g["attention_mask"], g["video_input_mask"])
    n = meta["new_tokens"]
    ids, steps = m.generate(*args, n, eos_id=-1, return_logits=True)
    assert len(set(g["fp32_greedy_free"][0].tolist())) >= 4 and np.array_equal(g["fp32_greedy_free"], g["bf16_greedy_free"])
    assert np.array_equal(ids, g["fp32_greedy_free"]), (ids, g["fp32_greedy_free"])
    assert np.abs(steps[0] - g["fp32_logits_last"]).max() < 5e-4
    for k in range(n):  # the eight leading logits of every step: ids and values
        top = np.argsort(-steps[k], axis=-1)[:, :8]
        assert np.array_equal(top[:, :2], g["fp32_step_logits_top8_ids"][k][:, :2]), k
        assert np.abs(np.take_along_axis(steps[k], g["fp32_step_logits_top8_ids"][k], -1) - g["fp32_step_logits_top8"][k]).max() < 5e-4, k
    eos = m.generate(*args, n, eos_id=int(g["fp32_eos_id"]))
    assert np.array_equal(eos, g["fp32_greedy_eos"]) and eos.shape[1] < n
    beam = m.generate_beam(*args, n, 3, 1.0, eos_id=-1, no_move=True)
    assert np.array_equal(beam, g["fp32_beam3_lp1_free"]), (beam, g["fp32_beam3_lp1_free"])
