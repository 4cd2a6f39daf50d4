"""CPU: oracle/parity.py — generated ids against a reference-made full-depth fixture, compared up to the first near-tie of the REFERENCE."""
import os

import numpy as np

from oracle.parity import greedy_ids_vs_reference


def _fixture(golden_dir):
    return np.load(os.path.join(golden_dir, "full_c2.npz"))


def test_reference_runs_agree_with_each_other(golden_dir):
    g = _fixture(golden_dir)
    v = greedy_ids_vs_reference(g["bf16_greedy_free"], g)
    assert v["ok"] and v["ids_equal_before_first_flip"] == v["ids_total"] == 64 and not v["flips"]


def test_flip_at_a_near_tie_to_the_runner_up_is_accepted_and_ends_the_comparison(golden_dir):
    g = _fixture(golden_dir)
    ids = g["fp32_greedy_free"].copy()
    margins = g["fp32_step_logits_top8"][:, 0, 0] - g["fp32_step_logits_top8"][:, 0, 1]
    k = int(np.argmin(margins))  # the tightest step of row 0: a near-tie by construction of the threshold
    ids[0, k] = g["fp32_step_logits_top8_ids"][k, 0, 1]
    ids[0, k + 1:] = 7  # whatever follows a legitimate flip is another trajectory: not compared
    v = greedy_ids_vs_reference(ids, g)
    assert v["ok"] and v["flips"] == [{"row": 0, "step": k, "ref_top2_margin": round(float(margins[k]), 4), "is_runner_up": True, "legit": True}]
    assert v["ids_equal_before_first_flip"] == k + 32


def test_flip_at_a_clear_step_or_to_another_token_is_rejected(golden_dir):
    g = _fixture(golden_dir)
    margins = g["fp32_step_logits_top8"][:, 0, 0] - g["fp32_step_logits_top8"][:, 0, 1]
    k = int(np.argmax(margins))  # a step the reference decides by a wide margin
    ids = g["fp32_greedy_free"].copy()
    ids[0, k] = g["fp32_step_logits_top8_ids"][k, 0, 1]
    assert not greedy_ids_vs_reference(ids, g)["ok"]
    k2 = int(np.argmin(margins))
    ids = g["fp32_greedy_free"].copy()
    ids[0, k2] = g["fp32_step_logits_top8_ids"][k2, 0, 5]  # a near-tie step, but not the runner-up
    assert not greedy_ids_vs_reference(ids, g)["ok"]
