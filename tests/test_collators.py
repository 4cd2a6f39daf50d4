"""DataCollatorForInterleavedVideoSeq2Seq / DataCollatorForVideoSeq2Seq (the SHIPPED classes, eilev_amd/data/utils.py) against
the reference's own known-answer vectors (ref:tests/data/test_utils.py:674-862, kept as data in tests/golden/collator_cases.json)."""
import json
import os

import pytest
import torch

from tok_utils import tiny_opt_like_tokenizer

with open(os.path.join(os.path.dirname(__file__), "golden", "collator_cases.json")) as fh:
    CASES = json.load(fh)["cases"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("alias", [False, True])
def test_interleaved_collator_reference_vectors(case, alias):
    if alias:  # the import path the reference's callers use (ref:scripts/general/train_v2.py:22-26)
        from eilev.data.utils import DataCollatorForInterleavedVideoSeq2Seq
    else:
        from eilev_amd.data.utils import DataCollatorForInterleavedVideoSeq2Seq
    tok = tiny_opt_like_tokenizer(case["padding_side"])
    collator = DataCollatorForInterleavedVideoSeq2Seq(tok, pad_to_multiple_of=case["pad_to_multiple_of"])
    datapoints = [{"pixel_values": torch.ones(s["clips"], 1, 1, 1, 1), "input_ids": torch.ones(len(s["video_input_mask"])).long(),
                   "video_input_mask": torch.tensor(s["video_input_mask"])} for s in case["samples"]]
    out = collator(datapoints)
    assert out["pixel_values"].equal(torch.ones(case["expected_clips"], 1, 1, 1, 1))
    assert out["video_input_mask"].equal(torch.tensor(case["expected_video_input_mask"]))
    # ids / attention mask are padded by the HF base class on the same side, to the same width
    assert out["input_ids"].shape == out["video_input_mask"].shape == out["attention_mask"].shape
    n = [len(s["video_input_mask"]) for s in case["samples"]]
    for b, k in enumerate(n):
        row = out["attention_mask"][b].tolist()
        assert row == ([0] * (len(row) - k) + [1] * k if case["padding_side"] == "left" else [1] * k + [0] * (len(row) - k))


def test_interleaved_collator_pads_labels_with_ignore_index():
    from eilev_amd.data.utils import DataCollatorForInterleavedVideoSeq2Seq

    tok = tiny_opt_like_tokenizer("right")
    c = DataCollatorForInterleavedVideoSeq2Seq(tok, pad_to_multiple_of=8)  # what train_v2 passes under bf16 (ref:scripts/general/train_v2.py:207-216)
    out = c([{"pixel_values": torch.zeros(2, 3, 2, 4, 4), "input_ids": torch.tensor([2, 1, 1, 3]), "labels": torch.tensor([-100, -100, -100, 3]),
              "video_input_mask": torch.tensor([0, 1, 1, 0])},
             {"pixel_values": torch.zeros(1, 3, 2, 4, 4), "input_ids": torch.tensor([2, 1, 4, 5, 3, 2]), "labels": torch.tensor([-100, -100, 4, 5, 3, 2]),
              "video_input_mask": torch.tensor([0, 1, 0, 0, 0, 0])}])
    assert out["pixel_values"].shape == (3, 3, 2, 4, 4)
    assert out["labels"].tolist() == [[-100, -100, -100, 3, -100, -100, -100, -100], [-100, -100, 4, 5, 3, 2, -100, -100]]
    assert out["input_ids"].tolist() == [[2, 1, 1, 3, 1, 1, 1, 1], [2, 1, 4, 5, 3, 2, 1, 1]]
    assert out["video_input_mask"].tolist() == [[0, 1, 1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0, 0]]


def test_video_collator_stacks_pixel_values():
    """ref:eilev/data/utils.py:19-32."""
    from eilev_amd.data.utils import DataCollatorForVideoSeq2Seq

    tok = tiny_opt_like_tokenizer("right")
    out = DataCollatorForVideoSeq2Seq(tok)([{"pixel_values": torch.ones(3, 2, 4, 4), "input_ids": torch.tensor([2, 3])},
                                            {"pixel_values": torch.zeros(3, 2, 4, 4), "input_ids": torch.tensor([2, 3, 4])}])
    assert out["pixel_values"].shape == (2, 3, 2, 4, 4) and out["input_ids"].tolist() == [[2, 3, 1], [2, 3, 4]]
