"""Named model configurations (plain dicts accepted by ``transformers.Blip2Config``).

``opt27`` is the benchmark model (eilev-blip2-opt-2.7b: ViT-g/14 39L, Q-Former 12L,
OPT-2.7B 32L; dims from ref:SURVEY §8 / hf Blip2Config defaults).  ``mid`` keeps the
awkward head sizes of the real model (ViT 88, Q-Former 64, OPT 80) at toy widths so
that every padded-head code path of the HIP kernels is exercised by the golden
fixtures.  ``tiny`` mirrors the shape regime of the reference's own unit tests
(ref:tests/model/test_model_v2.py:93-140) and is used to pin the CPU oracle only.
"""
from __future__ import annotations

CONFIGS = {
    "tiny": dict(
        vision_config=dict(hidden_size=16, intermediate_size=32, num_hidden_layers=2,
                           num_attention_heads=2, patch_size=8, image_size=32),
        qformer_config=dict(hidden_size=16, num_hidden_layers=2, num_attention_heads=2,
                            intermediate_size=32, encoder_hidden_size=16),
        text_config=dict(model_type="opt", hidden_size=16, num_hidden_layers=2, ffn_dim=32,
                         num_attention_heads=2, vocab_size=128, max_position_embeddings=64,
                         word_embed_proj_dim=16),
        num_query_tokens=4,
    ),
    "mid": dict(
        vision_config=dict(hidden_size=176, intermediate_size=352, num_hidden_layers=2,
                           num_attention_heads=2, patch_size=14, image_size=56),
        qformer_config=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                            intermediate_size=256, encoder_hidden_size=176),
        text_config=dict(model_type="opt", hidden_size=160, num_hidden_layers=2, ffn_dim=320,
                         num_attention_heads=2, vocab_size=512, max_position_embeddings=128,
                         word_embed_proj_dim=160),
        num_query_tokens=8,
    ),
    "opt27": dict(
        vision_config=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=39,
                           num_attention_heads=16, patch_size=14, image_size=224),
        qformer_config=dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                            intermediate_size=3072, encoder_hidden_size=1408),
        text_config=dict(model_type="opt", hidden_size=2560, num_hidden_layers=32, ffn_dim=10240,
                         num_attention_heads=32, vocab_size=50272, max_position_embeddings=2048,
                         word_embed_proj_dim=2560),
        num_query_tokens=32,
    ),
    # the REAL widths of eilev-blip2-opt-2.7b (ViT-g/14 at 224x224, Q-Former, OPT-2.7B incl. the 50272-token vocabulary) with
    # one ViT block, one Q-Former block pair (cross-attention on block 0) and one OPT block: the configuration of the
    # real-shape golden fixtures (SURVEY 8c golden plan (2)) — every kernel variant that depends on a WIDTH or a head size
    # (strip im2col at 224/14, frame attention 257 x 88, cross-attention over 2056 keys, hd 80 causal attention, K = 1408 /
    # 6144 / 2560 / 10240 GEMMs, the 50272-wide lm_head + argmax) is exercised against the reference at its true size
    "real_1l": dict(
        vision_config=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=1,
                           num_attention_heads=16, patch_size=14, image_size=224),
        qformer_config=dict(hidden_size=768, num_hidden_layers=2, num_attention_heads=12,
                            intermediate_size=3072, encoder_hidden_size=1408),
        text_config=dict(model_type="opt", hidden_size=2560, num_hidden_layers=1, ffn_dim=10240,
                         num_attention_heads=32, vocab_size=50272, max_position_embeddings=2048,
                         word_embed_proj_dim=2560),
        num_query_tokens=32,
    ),
    # vision towers whose widths are multiples of 64 and that have 3 blocks: what the LayerNorm-folded ViT path needs to be exercised
    # on every block boundary (block 0's layer_norm1 comes from the patch kernel, fc2 -> next block's qkv needs a next block) —
    # a small one, and one at the true ViT-g widths (1408 / 6144 / 16 heads, 257 tokens)
    "fold_3l": dict(
        vision_config=dict(hidden_size=192, intermediate_size=384, num_hidden_layers=3,
                           num_attention_heads=3, patch_size=14, image_size=56),
        qformer_config=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                            intermediate_size=256, encoder_hidden_size=192),
        text_config=dict(model_type="opt", hidden_size=160, num_hidden_layers=1, ffn_dim=320,
                         num_attention_heads=2, vocab_size=512, max_position_embeddings=128,
                         word_embed_proj_dim=160),
        num_query_tokens=8,
    ),
    "real_vit_3l": dict(
        vision_config=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=3,
                           num_attention_heads=16, patch_size=14, image_size=224),
        qformer_config=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                            intermediate_size=256, encoder_hidden_size=1408),
        text_config=dict(model_type="opt", hidden_size=160, num_hidden_layers=1, ffn_dim=320,
                         num_attention_heads=2, vocab_size=512, max_position_embeddings=128,
                         word_embed_proj_dim=160),
        num_query_tokens=8,
    ),
    # `mid` with a text model whose K dimensions are multiples of 128 (hidden 256 = 2 heads x 128, ffn 512): the smallest
    # configuration that takes the fp8-MFMA prefill path (eilev_linear_a8w8 needs k % 128 == 0)
    "mid_k128": dict(
        vision_config=dict(hidden_size=176, intermediate_size=352, num_hidden_layers=2,
                           num_attention_heads=2, patch_size=14, image_size=56),
        qformer_config=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                            intermediate_size=256, encoder_hidden_size=176),
        text_config=dict(model_type="opt", hidden_size=256, num_hidden_layers=2, ffn_dim=512,
                         num_attention_heads=2, vocab_size=512, max_position_embeddings=128,
                         word_embed_proj_dim=256),
        num_query_tokens=8,
    ),
    # the REAL widths of eilev-blip2-flan-t5-xl with one block per stack (ViT 1, Q-Former pair, T5 encoder 1 + decoder 1): the
    # real-shape fixture of the encoder-decoder path (BASELINE configs[3])
    "real_t5_1l": dict(
        vision_config=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=1,
                           num_attention_heads=16, patch_size=14, image_size=224),
        qformer_config=dict(hidden_size=768, num_hidden_layers=2, num_attention_heads=12,
                            intermediate_size=3072, encoder_hidden_size=1408),
        text_config=dict(model_type="t5", d_model=2048, d_kv=64, num_heads=32, d_ff=5120, num_layers=1, num_decoder_layers=1,
                         vocab_size=32128, feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0),
        num_query_tokens=32,
    ),
    "opt67": dict(  # blip2-opt-6.7b backbone (BASELINE configs[4]): hidden 4096, 32 heads x 128, ffn 16384
        vision_config=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=39,
                           num_attention_heads=16, patch_size=14, image_size=224),
        qformer_config=dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                            intermediate_size=3072, encoder_hidden_size=1408),
        text_config=dict(model_type="opt", hidden_size=4096, num_hidden_layers=32, ffn_dim=16384,
                         num_attention_heads=32, vocab_size=50272, max_position_embeddings=2048,
                         word_embed_proj_dim=4096),
        num_query_tokens=32,
    ),
    # encoder-decoder language model (flan-t5 family: gated-gelu FFN, RMSNorm, relative position bias, no biases)
    "tiny_t5": dict(
        vision_config=dict(hidden_size=16, intermediate_size=32, num_hidden_layers=2,
                           num_attention_heads=2, patch_size=8, image_size=32),
        qformer_config=dict(hidden_size=16, num_hidden_layers=2, num_attention_heads=2,
                            intermediate_size=32, encoder_hidden_size=16),
        text_config=dict(model_type="t5", d_model=32, d_kv=8, num_heads=4, d_ff=64, num_layers=2, num_decoder_layers=2,
                         vocab_size=128, feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0),
        num_query_tokens=4,
    ),
    "mid_t5": dict(
        vision_config=dict(hidden_size=176, intermediate_size=352, num_hidden_layers=2,
                           num_attention_heads=2, patch_size=14, image_size=56),
        qformer_config=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                            intermediate_size=256, encoder_hidden_size=176),
        text_config=dict(model_type="t5", d_model=192, d_kv=64, num_heads=2, d_ff=320, num_layers=2, num_decoder_layers=2,
                         vocab_size=512, feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0),
        num_query_tokens=8,
    ),
    "t5xl": dict(  # eilev-blip2-flan-t5-xl
        vision_config=dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=39,
                           num_attention_heads=16, patch_size=14, image_size=224),
        qformer_config=dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                            intermediate_size=3072, encoder_hidden_size=1408),
        text_config=dict(model_type="t5", d_model=2048, d_kv=64, num_heads=32, d_ff=5120, num_layers=24, num_decoder_layers=24,
                         vocab_size=32128, feed_forward_proj="gated-gelu", tie_word_embeddings=False, decoder_start_token_id=0),
        num_query_tokens=32,
    ),
}


def blip2_config(name: str):
    """Build a ``transformers.Blip2Config`` for a named configuration."""
    from transformers import Blip2Config

    c = CONFIGS[name]
    return Blip2Config(
        vision_config=dict(c["vision_config"]),
        qformer_config=dict(c["qformer_config"]),
        text_config=dict(c["text_config"]),
        num_query_tokens=c["num_query_tokens"],
    )
