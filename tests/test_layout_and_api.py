"""CPU tests: token layout against the reference's own known-answer vectors, model API surface."""
import json
import os

import pytest
import torch

from eilev_amd.configs import blip2_config
from eilev_amd.data.utils import clean_narration_text, generate_input_ids_and_labels_from_interleaved
from eilev_amd.statedict import state_dict_shapes


class _Enc(dict):
    @property
    def input_ids(self):
        return self["input_ids"]


class StubOPTTokenizer:
    """GPT-2 BPE restricted to the strings of ref:tests/data/test_utils.py:112-460 (ids are the real OPT ids)."""
    bos_token_id, pad_token_id, eos_token_id = 2, 1, 2
    padding_side = "right"
    table = {"A": [250], " A": [83], " prompt": [14302], " text": [2788], "Prompt": [35396, 3320], " 1": [112], " 2": [132],
             " 3": [155], " Text": [14159], "\n": [50118]}

    def __call__(self, text, add_special_tokens=True, return_attention_mask=False):
        ids = [self.bos_token_id] if add_special_tokens else []
        i = 0
        while i < len(text):
            for piece in sorted(self.table, key=len, reverse=True):
                if text.startswith(piece, i):
                    ids += self.table[piece]
                    i += len(piece)
                    break
            else:
                raise KeyError(text[i:])
        return _Enc(input_ids=ids)


class StubT5Tokenizer:
    """SentencePiece restricted to the strings of ref:tests/data/test_utils.py:463-671 (real Flan-T5 ids)."""
    pad_token_id, eos_token_id = 0, 1
    bos_token_id = None
    padding_side = "right"
    table = {"A": [71], "prompt": [9005], "text": [1499], "Prompt": [749, 1167, 17], "Text": [5027], "1": [209], "2": [204], "3": [220]}

    def __call__(self, text, add_special_tokens=True, return_attention_mask=False):
        ids = []
        for w in text.split():
            ids += self.table[w]
        if text.endswith("\n"):
            ids.append(3)  # every whitespace maps to the same piece (ref:eilev/data/utils.py:168-169)
        if add_special_tokens:
            ids.append(self.eos_token_id)
        return _Enc(input_ids=ids)


with open(os.path.join(os.path.dirname(__file__), "golden", "layout_cases.json")) as fh:
    LAYOUT = json.load(fh)


@pytest.mark.parametrize("case", LAYOUT["test_generate_input_ids_and_labels_from_interleaved_decoder_only"])
def test_interleaved_layout_opt(case):
    out = generate_input_ids_and_labels_from_interleaved(StubOPTTokenizer(), [tuple(p) for p in case["prompts"]], case["text"],
                                                         case["num_query_tokens"], True)
    for k, v in case["expected"].items():
        assert out[k].tolist() == v, k


@pytest.mark.parametrize("case", LAYOUT["test_generate_input_ids_and_labels_from_interleaved_seq2seq"])
def test_interleaved_layout_t5(case):
    out = generate_input_ids_and_labels_from_interleaved(StubT5Tokenizer(), [tuple(p) for p in case["prompts"]], case["text"],
                                                         case["num_query_tokens"], False)
    for k, v in case["expected"].items():
        assert out[k].tolist() == v, k


@pytest.mark.parametrize("raw,clean", [
    ("#C C drops the plate", "The camera wearer drops the plate."), ("#c c drops the plate", "The camera wearer drops the plate."),
    ("#C C drops the plate<|eos|>", "The camera wearer drops the plate."), ("#C C drops the #unsure.", "The camera wearer drops the."),
    ("#C C drops #unsure in the sink", "The camera wearer drops something in the sink."), ("  ", ""),
    ("#C C drops the plate!", "The camera wearer drops the plate!")])
def test_clean_narration_text(raw, clean):
    assert clean_narration_text(raw) == clean


def test_synthetic_layout_matches_layout_function():
    """eilev_amd.synth.synth_interleaved_ids (used by bench/goldens) has the layout of the real function."""
    from eilev_amd.synth import synth_interleaved_ids

    ids, vm = synth_interleaved_ids([1, 2], [4, 3], 2, 50272)
    assert ids[0] == 2 and vm.tolist() == [0, 1, 1, 0] + [0] * 4 + [1, 1, 0, 1, 1, 0] + [0] * 3
    assert ids[3] == 50118 and ids[7] == 50118 and len(ids) == 1 + 3 + 4 + 6 + 3


def test_model_state_dict_names_and_roundtrip(tmp_path):
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration

    cfg = blip2_config("tiny")
    m = VideoBlipForConditionalGeneration(cfg)
    sd = m.state_dict()
    want = state_dict_shapes(cfg)
    assert set(sd) - {"language_model.lm_head.weight"} == set(want)
    for k, shp in want.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert sd["language_model.lm_head.weight"].data_ptr() == sd["language_model.model.decoder.embed_tokens.weight"].data_ptr()
    m.save_pretrained(tmp_path)
    assert os.path.exists(tmp_path / "model.safetensors") and os.path.exists(tmp_path / "config.json")
    m2 = VideoBlipForConditionalGeneration.from_pretrained(tmp_path)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert m.config.num_query_tokens == 4 and m.config.use_decoder_only_language_model
    assert m.get_input_embeddings() is m.language_model.get_input_embeddings()


def test_no_cpu_fallback():
    from eilev_amd.model.v2 import VideoBlipForConditionalGeneration, VideoBlipVisionModel

    cfg = blip2_config("tiny")
    m = VideoBlipForConditionalGeneration(cfg)
    with pytest.raises(RuntimeError):
        m.generate(torch.ones(1, 4, dtype=torch.long), max_new_tokens=2)
    with pytest.raises(ValueError):
        VideoBlipVisionModel(cfg.vision_config)(None)
    with pytest.raises(RuntimeError):
        m.generate(torch.ones(1, 4, dtype=torch.long), num_beams=5)
    with pytest.raises(RuntimeError):
        m.classify(torch.ones(1, 4, dtype=torch.long), torch.ones(2, 3, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        m.generate(torch.ones(1, 4, dtype=torch.long), penalty_alpha=0.1, top_k=2)  # contrastive search


def test_c_abi_library_exports_every_declared_symbol():
    import ctypes
    import re

    from eilev_amd import abi

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "eilev.h")).read()
    declared = set(re.findall(r"\b(eilev_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(abi.EXPORTS)
    if os.path.exists(abi.HIP_LIB_PATH):
        lib = ctypes.CDLL(abi.HIP_LIB_PATH)  # loads without a GPU
        for sym in declared:
            assert hasattr(lib, sym), sym
        # round 5: the product library carries no probe switch (they exist in the -DEILEV_PROBES build only)
        for sw in ("eilev_debug_gemm_flags", "eilev_debug_gemm_trace", "eilev_debug_attn_v1", "eilev_debug_attn_ts", "eilev_debug_decode_rows",
                   "eilev_debug_beam_part", "eilev_debug_fused_patch", "eilev_debug_decode_prefetch", "eilev_debug_reduce_ln_wave"):
            assert not hasattr(lib, sw), sw
        # round 6: the dynamic symbol table IS the header — nothing else leaves the library (csrc/exports.map: no C++ launcher, no kernel stub)
        import subprocess

        nm = subprocess.run(["nm", "-D", "--defined-only", abi.HIP_LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
        assert exported == declared, (sorted(exported - declared)[:8], sorted(declared - exported)[:8])
    from oracle.runner import lib as oracle_lib

    for sym in declared:
        assert hasattr(oracle_lib(), sym), sym


def test_abi16_vit_weights_layout():
    """ABI 16 (round 6): EilevVitWeights ends with `int64_t fold_min_rows` behind its nine pointers (include/eilev.h) — the ctypes mirror the
    engine passes must have exactly that layout, and the header, the binding and the built library must agree on the version."""
    import ctypes
    import re

    from eilev_amd import abi

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "eilev.h")).read()
    assert int(re.search(r"#define EILEV_ABI_VERSION (\d+)", hdr).group(1)) == 16
    assert [f[0] for f in abi.VitWeights._fields_][-2:] == ["layers_fold_hm", "fold_min_rows"]
    assert abi.VitWeights.fold_min_rows.offset == 9 * ctypes.sizeof(ctypes.c_void_p) and ctypes.sizeof(abi.VitWeights) == 80
    body = hdr[hdr.index("typedef struct EilevVitWeights {"):hdr.index("} EilevVitWeights;")]
    assert body.rstrip().endswith("int64_t fold_min_rows;")
    assert "eilev_debug_ln_fold_min_rows" not in " ".join(abi.EXPORTS)
    if os.path.exists(abi.HIP_LIB_PATH):
        assert ctypes.CDLL(abi.HIP_LIB_PATH).eilev_abi_version() == 16
