"""The stock-transformers CPU baseline harness (oracle/hf_baseline.py, measurement infrastructure) computes the same function
as the oracle: a second, live pin of the oracle against the classes the reference instantiates (ref:eilev/model/v2.py:111-127)."""
import numpy as np
import torch

from eilev_amd.configs import blip2_config
from eilev_amd.synth import synth_interleaved_ids, synth_pixels
from oracle.hf_baseline import build_hf_modules, hf_encode, hf_generate
from oracle.runner import OracleModel, synth_state_dict


def test_stock_hf_composition_equals_oracle():
    cfg = blip2_config("mid")
    sd = synth_state_dict(cfg)
    vit, qf, lm, proj, qt = build_hf_modules(cfg, sd)
    px = synth_pixels(3, 2, cfg.vision_config.image_size)
    ids, vm = synth_interleaved_ids([1, 1, 1], [5, 5, 4], cfg.num_query_tokens, cfg.text_config.vocab_size)
    feats = hf_encode(vit, qf, proj, qt, torch.from_numpy(px))
    ora = OracleModel(cfg, sd)
    assert np.abs(ora.project(ora.qformer(ora.vit(px))) - feats.numpy()).max() < 2e-4
    out = hf_generate(lm, feats, torch.from_numpy(ids)[None], torch.from_numpy(vm)[None], 6)
    ref = ora.generate(px, ids[None], np.ones_like(ids)[None], vm[None], 6, eos_id=-1)
    assert np.array_equal(out.numpy(), ref)
