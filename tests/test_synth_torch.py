"""The torch evaluation of the deterministic weight recipe (eilev_amd.synth.synth_param_torch: what the GPU box uses to build the 3.8 G
parameters of the full-depth fixture in seconds) is bit-identical to the numpy one the fixtures were generated with."""
import numpy as np
import pytest
import torch

from eilev_amd.synth import synth_param, synth_param_torch

NAMES = [
    ("vision_model.encoder.layers.3.self_attn.qkv.weight", (96, 160)),
    ("vision_model.encoder.layers.3.layer_norm1.weight", (160,)),
    ("vision_model.encoder.layers.3.layer_norm1.bias", (160,)),
    ("vision_model.encoder.layers.0.mlp.fc1.bias", (320,)),
    ("vision_model.embeddings.class_embedding", (1, 1, 176)),
    ("vision_model.embeddings.patch_embedding.weight", (176, 3, 14, 14)),
    ("qformer.encoder.layer.1.attention.attention.query.weight", (128, 128)),
    ("query_tokens", (1, 8, 128)),
    ("language_model.model.decoder.embed_tokens.weight", (512, 160)),
    ("language_model.model.decoder.embed_positions.weight", (130, 160)),
    ("language_model.model.decoder.layers.1.fc2.weight", (160, 320)),
    ("language_model.encoder.block.0.layer.0.SelfAttention.q.weight", (64, 64)),  # T5: the 1/8 factor
    ("language_projection.weight", (160, 128)),
]


@pytest.mark.parametrize("mode", ["fanin", "varied", "hf"])
@pytest.mark.parametrize("seed", [0, 176])
def test_torch_recipe_equals_numpy_recipe(mode, seed):
    for name, shape in NAMES:
        a = synth_param(name, shape, mode, seed)
        b = synth_param_torch(name, shape, mode, seed).numpy()
        assert a.dtype == b.dtype == np.float32 and a.shape == b.shape
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, mode, seed, np.abs(a - b).max())


@pytest.mark.gpu
def test_device_recipe_equals_numpy_recipe():
    for name, shape in NAMES:
        for mode, seed in (("varied", 0), ("fanin", 3)):
            a = synth_param(name, shape, mode, seed)
            b = synth_param_torch(name, shape, mode, seed, device="cuda").cpu().numpy()
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, mode, seed)
