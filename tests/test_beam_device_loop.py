"""The device-resident beam loop (eilev_amd/beam.py::beam_search_device: step index on the device, in-place hypotheses, one captured graph
per step on the GPU) against the host loop `beam_search` (pinned to transformers and the reference's beam goldens elsewhere): same
hypotheses, token for token, on a synthetic language model whose logits are a deterministic function of the hypothesis."""
import pytest
import torch

from eilev_amd.beam import beam_search, beam_search_device


def _toy_lm(B, nb, V, T, seed, eos, peaked):
    """step(tokens, beam_src) -> logits: each row carries a hash state of its hypothesis (reordered by beam_src like a KV cache)."""
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(V, V, generator=g) * (3.0 if peaked else 0.7)  # next-token logits by last token
    mix = torch.randn(64, V, generator=g) * 0.5
    if eos is not None:
        table[:, eos] += 1.5  # EOS fires now and then
    state = {"h": torch.zeros(B * nb, dtype=torch.int64)}

    def logits_of(tokens, h):
        return table[tokens] + mix[h % 64]

    def step(tokens, beam_src):
        h = state["h"][beam_src]
        h = (h * 31 + tokens + 7) % 1000003
        state["h"] = h
        return logits_of(tokens, h)

    first = torch.randn(B, V, generator=g)
    return step, first, state


CASES = [(1, 5, 64, 12, -1.0, 3, False, False), (2, 3, 50, 9, 1.0, 7, False, True), (2, 4, 40, 16, 0.0, None, False, False),
         (1, 5, 64, 20, -1.0, 3, True, False), (3, 2, 30, 6, 2.0, 5, "never", True), (1, 5, 48, 5, -1.0, [3, 9], False, False)]


@pytest.mark.parametrize("B,nb,V,T,lp,eos,early,peaked", CASES)
@pytest.mark.parametrize("nret", [1, 2])
def test_device_loop_equals_host_loop(B, nb, V, T, lp, eos, early, peaked, nret):
    eos_arg = -1 if eos is None else eos
    e0 = None if eos is None else (eos if isinstance(eos, int) else eos[0])
    step, first, st = _toy_lm(B, nb, V, T, 11, e0, peaked)
    want = beam_search(step, first, B, nb, T, lp, eos_arg, 1, early, nret)
    step2, first2, st2 = _toy_lm(B, nb, V, T, 11, e0, peaked)
    buf = torch.empty(B * nb, V)

    def step_dev(tokens, beam_src):
        buf.copy_(step2(tokens.clone(), beam_src.clone()))

    got = beam_search_device(step_dev, buf, first2, B, nb, T, lp, eos_arg, 1, early, nret, use_graph=False, check_every=1)
    assert torch.equal(got, want), (got, want)
    # exit checks every 4th step: extra steps change nothing that is returned
    step3, first3, _ = _toy_lm(B, nb, V, T, 11, e0, peaked)
    buf3 = torch.empty(B * nb, V)
    got4 = beam_search_device(lambda t, s: buf3.copy_(step3(t.clone(), s.clone())), buf3, first3, B, nb, T, lp, eos_arg, 1, early, nret,
                              use_graph=False, check_every=4)
    assert torch.equal(got4, want)
