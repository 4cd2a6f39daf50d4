"""The device-resident beam loop (eilev_amd/beam.py::beam_search_device: step index on the device, in-place hypotheses, one captured graph
per step on the GPU) against the host loop `beam_search` (pinned to transformers and the reference's beam goldens elsewhere): same
hypotheses, token for token, on a synthetic language model whose logits are a deterministic function of the hypothesis."""
import ctypes as C

import numpy as np
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


def _oracle_topk(keep):
    from oracle import runner

    lib = runner.lib()

    def fn(buf, run_score):
        x = np.ascontiguousarray(buf.numpy(), np.float32)
        sc = np.ascontiguousarray(run_score.reshape(-1).numpy(), np.float32)
        R, V = x.shape
        val, idx = np.empty((R, keep), np.float32), np.empty((R, keep), np.int32)
        rc = lib.eilev_topk_logprob(x.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), R, V, keep, val.ctypes.data_as(C.c_void_p),
                                    idx.ctypes.data_as(C.c_void_p), None)
        assert rc == 0
        return torch.from_numpy(val), torch.from_numpy(idx)
    return fn


def test_topk_logprob_restatement_vs_torch():
    """oracle eilev_topk_logprob = torch.log_softmax + row score + torch.topk per row (values to fp32 rounding, ids exact)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(7, 1000, generator=g) * 4
    sc = torch.randn(7, generator=g)
    val, idx = _oracle_topk(6)(x, sc)
    want_v, want_i = torch.topk(torch.log_softmax(x, -1) + sc[:, None], 6, dim=1)
    assert torch.equal(idx.long(), want_i)
    assert (val - want_v).abs().max() < 2e-6


@pytest.mark.parametrize("B,nb,V,T,lp,eos,early,peaked", CASES[:4])
def test_device_loop_with_row_topk_equals_host_loop(B, nb, V, T, lp, eos, early, peaked):
    """the two-stage selection (per-row top 2K, then the merge over a sample's rows) picks what the top 2K over beams x vocabulary picks"""
    eos_arg = -1 if eos is None else eos
    e0 = None if eos is None else (eos if isinstance(eos, int) else eos[0])
    step, first, _ = _toy_lm(B, nb, V, T, 11, e0, peaked)
    want = beam_search(step, first, B, nb, T, lp, eos_arg, 1, early, 1)
    step2, first2, _ = _toy_lm(B, nb, V, T, 11, e0, peaked)
    buf = torch.empty(B * nb, V)
    keep = max(2, 1 + (0 if eos is None else (1 if isinstance(eos, int) else len(eos)))) * nb
    got = beam_search_device(lambda t, s: buf.copy_(step2(t.clone(), s.clone())), buf, first2, B, nb, T, lp, eos_arg, 1, early, 1, use_graph=False,
                             check_every=1, topk_fn=_oracle_topk(keep))
    assert torch.equal(got, want), (got, want)


def _oracle_advance(B, nb, T, state):
    from oracle import runner

    lib = runner.lib()
    P = lambda t: C.c_void_p(t.data_ptr())

    def fn(row_lp, row_tok, st):
        eos = st["eos"]
        eos_arr = (C.c_int64 * max(1, len(eos)))(*eos)
        keep = st["keep"]
        tokens = st.setdefault("tokens", torch.zeros(B * nb, dtype=torch.int64))
        rc = lib.eilev_beam_advance(P(row_lp), P(row_tok), B, nb, keep, T, P(state), eos_arr, len(eos), P(st["pow_tab"]), int(st["reciprocal"]),
                                    int(st["early"]), P(st["run_seq"]), P(st["run_score"]), P(st["fin_seq"]), P(st["fin_score"]), P(st["fin_len"]),
                                    P(st["finished"]), P(st["can_improve"]), P(tokens), P(st["anc"]), T, None, 0, None)
        assert rc == 0, rc
    return fn


@pytest.mark.parametrize("B,nb,V,T,lp,eos,early,peaked", CASES)
@pytest.mark.parametrize("nret", [1, 2])
def test_fused_advance_restatement_equals_host_loop(B, nb, V, T, lp, eos, early, peaked, nret):
    """eilev_beam_advance (oracle restatement: the whole bookkeeping of a step + tokens + ancestor table) inside the device loop returns
    the hypotheses of `beam_search`; the ancestor table it maintains names, for every row, the slots of its own hypothesis."""
    eos_arg = -1 if eos is None else eos
    e0 = None if eos is None else (eos if isinstance(eos, int) else eos[0])
    n_eos = 0 if eos is None else (1 if isinstance(eos, int) else len(eos))
    step, first, _ = _toy_lm(B, nb, V, T, 11, e0, peaked)
    want = beam_search(step, first, B, nb, T, lp, eos_arg, 1, early, nret)
    # the toy model driven the way the engine drives the decode step: hidden state per SLOT (step, row), read through the ancestor table
    g = torch.Generator().manual_seed(11)
    table = torch.randn(V, V, generator=g) * (3.0 if peaked else 0.7)
    mix = torch.randn(64, V, generator=g) * 0.5
    if e0 is not None:
        table[:, e0] += 1.5
    first2 = torch.randn(B, V, generator=g)
    assert torch.equal(first2, first)
    R = B * nb
    state = torch.ones(2, dtype=torch.int32)  # state[0] = 1 before the first step, incremented by every step
    anc = torch.zeros((T, R), dtype=torch.int32)
    slot_tok = torch.zeros((T, R), dtype=torch.int64)
    buf = torch.empty(R, V)
    keep = max(2, 1 + n_eos) * nb
    adv = _oracle_advance(B, nb, T, state)
    holder = {}

    def advance(row_lp, row_tok, st):
        st["anc"] = anc
        adv(row_lp, row_tok, st)
        holder["tokens"] = st["tokens"]

    def step_dev(_t, _s):
        t = int(state[0]) - 1  # tokens fed before this one
        toks = holder["tokens"]
        slot_tok[t] = toks
        h = torch.zeros(R, dtype=torch.int64)
        for gi in range(t + 1):  # replay the hypothesis of every row from its ancestor slots
            h = (h * 31 + slot_tok[gi, anc[gi].long()] + 7) % 1000003
        buf.copy_(table[toks] + mix[h % 64])
        state[0] += 1

    got = beam_search_device(step_dev, buf, first2, B, nb, T, lp, eos_arg, 1, early, nret, use_graph=False, check_every=1, topk_fn=_oracle_topk(keep),
                             advance_fn=advance)
    assert torch.equal(got, want), (got, want)
