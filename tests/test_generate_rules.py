"""Stopping rules of `generate` that the captured device step does not cover — several `eos_token_id`s and
`0 < min_new_tokens < max_new_tokens` — run through the host loops of eilev_amd/sampling.py (greedy / sampling) and eilev_amd/beam.py.
CPU: both loops are driven by a tiny random `transformers` OPT (the language model the reference wraps, ref:eilev/model/v2.py:318-322)
and must reproduce that model's own `generate()` token for token."""
import pytest
import torch

from eilev_amd.beam import beam_search
from eilev_amd.sampling import eos_list, sample_loop


@pytest.fixture(scope="module")
def tiny_opt():
    from transformers import OPTConfig, OPTForCausalLM

    torch.manual_seed(0)
    cfg = OPTConfig(vocab_size=40, hidden_size=32, num_hidden_layers=2, ffn_dim=64, num_attention_heads=4, max_position_embeddings=64,
                    word_embed_proj_dim=32, pad_token_id=1, bos_token_id=2, eos_token_id=3, do_layer_norm_before=True)
    m = OPTForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(8.0)  # sharper distributions: EOS ids actually win some steps
    return m


def _stepper(model, prompt, rows):
    """step(tokens, row_src) over a stateless full forward: keeps every row's sequence, reorders it like a KV cache would be."""
    state = {"seq": prompt.repeat_interleave(rows // prompt.shape[0], dim=0)}

    @torch.no_grad()
    def step(tokens, src):
        state["seq"] = torch.cat((state["seq"].index_select(0, src), tokens.view(-1, 1)), dim=1)
        return model(state["seq"]).logits[:, -1].float()

    with torch.no_grad():
        first = model(prompt).logits[:, -1].float()
    return step, first


def _hf(model, prompt, **kw):
    with torch.no_grad():
        out = model.generate(prompt, attention_mask=torch.ones_like(prompt), pad_token_id=1, **kw)
    return out[:, prompt.shape[1]:]


def _eq(ours, hf, pad=1):
    n = max(ours.shape[1], hf.shape[1])
    f = lambda t: torch.nn.functional.pad(t, (0, n - t.shape[1]), value=pad)
    assert torch.equal(f(ours), f(hf)), (ours.tolist(), hf.tolist())


def test_eos_list_forms():
    assert eos_list(None) == [] and eos_list(-1) == [] and eos_list(5) == [5] and eos_list([5, 7]) == [5, 7] and eos_list((4,)) == [4]


@pytest.mark.parametrize("eos,min_new", [([3, 7, 11], 0), (3, 4), ([5, 9], 3), ([3, 7, 11, 13, 17, 19, 23, 29], 0)])
def test_greedy_host_rules_equal_transformers(tiny_opt, eos, min_new):
    torch.manual_seed(1)
    prompt = torch.randint(4, 40, (3, 5))
    step, first = _stepper(tiny_opt, prompt, 3)
    ours = sample_loop(step, first, 10, eos_id=eos, pad_id=1, greedy=True, min_new_tokens=min_new)
    hf = _hf(tiny_opt, prompt, max_new_tokens=10, do_sample=False, num_beams=1, eos_token_id=eos, min_new_tokens=min_new or None)
    _eq(ours, hf)
    if min_new:
        e = torch.tensor(eos_list(eos))
        assert not torch.isin(ours[:, :min_new], e).any()


@pytest.mark.parametrize("eos,min_new,nb,lp", [([3, 7, 11], 0, 3, 1.0), (3, 4, 4, -1.0), ([5, 9, 12, 30], 2, 3, 1.0), (3, 0, 5, -1.0)])
def test_beam_host_rules_equal_transformers(tiny_opt, eos, min_new, nb, lp):
    torch.manual_seed(2)
    prompt = torch.randint(4, 40, (2, 6))
    step, first = _stepper(tiny_opt, prompt, 2 * nb)
    ours = beam_search(step, first, 2, nb, 9, lp, eos, 1, False, 1, min_new_tokens=min_new)
    hf = _hf(tiny_opt, prompt, max_new_tokens=9, do_sample=False, num_beams=nb, length_penalty=lp, eos_token_id=eos,
             min_new_tokens=min_new or None, early_stopping=False)
    _eq(ours, hf)


# ---- round 4: logits processors and stopping criteria through the same host loops (generate(repetition_penalty=, no_repeat_ngram_size=,
# logits_processor=, stopping_criteria=, max_time=) on the HIP path; hf's own processor classes, hf's order) ------------------------------
class _BanEven:
    """a user LogitsProcessor: even token ids above 9 are forbidden"""

    def __call__(self, input_ids, scores):
        scores = scores.clone()
        scores[:, 10::2] = float("-inf")
        return scores


class _StopOnLength:
    """a user StoppingCriteria: rows are done once they hold `n` ids"""

    def __init__(self, n):
        self.n = n

    def __call__(self, input_ids, scores, **kw):
        return torch.full((input_ids.shape[0],), input_ids.shape[1] >= self.n, dtype=torch.bool, device=input_ids.device)


def _procs(rp=None, ngram=None, user=False):
    from transformers import LogitsProcessorList, NoRepeatNGramLogitsProcessor, RepetitionPenaltyLogitsProcessor

    lst = LogitsProcessorList()
    if rp:
        lst.append(RepetitionPenaltyLogitsProcessor(penalty=rp))
    if ngram:
        lst.append(NoRepeatNGramLogitsProcessor(ngram))
    if user:
        lst.append(_BanEven())
    return lst


@pytest.mark.parametrize("rp,ngram,user", [(1.3, None, False), (None, 2, False), (1.2, 3, True), (None, None, True)])
def test_greedy_with_logits_processors_equals_transformers(tiny_opt, rp, ngram, user):
    from transformers import LogitsProcessorList

    torch.manual_seed(3)
    prompt = torch.randint(4, 40, (3, 6))
    step, first = _stepper(tiny_opt, prompt, 3)
    ours = sample_loop(step, first, 12, eos_id=3, pad_id=1, greedy=True, processors=_procs(rp, ngram, user), prefix=prompt)
    hf = _hf(tiny_opt, prompt, max_new_tokens=12, do_sample=False, num_beams=1, eos_token_id=3, repetition_penalty=rp, no_repeat_ngram_size=ngram,
             logits_processor=LogitsProcessorList([_BanEven()]) if user else None)
    _eq(ours, hf)
    if user:
        assert not ((ours >= 10) & (ours % 2 == 0)).any()


@pytest.mark.parametrize("rp,ngram,nb,lp", [(1.3, None, 3, 1.0), (None, 2, 4, -1.0), (1.15, 3, 3, 1.0)])
def test_beam_with_logits_processors_equals_transformers(tiny_opt, rp, ngram, nb, lp):
    torch.manual_seed(4)
    prompt = torch.randint(4, 40, (2, 6))
    step, first = _stepper(tiny_opt, prompt, 2 * nb)
    ours = beam_search(step, first, 2, nb, 9, lp, 3, 1, False, 1, processors=_procs(rp, ngram), prefix=prompt)
    hf = _hf(tiny_opt, prompt, max_new_tokens=9, do_sample=False, num_beams=nb, length_penalty=lp, eos_token_id=3, repetition_penalty=rp,
             no_repeat_ngram_size=ngram, early_stopping=False)
    _eq(ours, hf)


def test_greedy_with_stopping_criteria_equals_transformers(tiny_opt):
    from transformers import StoppingCriteriaList

    torch.manual_seed(5)
    prompt = torch.randint(4, 40, (2, 5))
    step, first = _stepper(tiny_opt, prompt, 2)
    crit = StoppingCriteriaList([_StopOnLength(5 + 4)])
    ours = sample_loop(step, first, 12, eos_id=-1, pad_id=1, greedy=True, stopping=crit, prefix=prompt)
    hf = _hf(tiny_opt, prompt, max_new_tokens=12, do_sample=False, num_beams=1, eos_token_id=None, stopping_criteria=StoppingCriteriaList([_StopOnLength(5 + 4)]))
    assert ours.shape[1] == 4
    _eq(ours, hf)


# ---- round 5: stopping criteria inside BEAM search: a flagged candidate finishes like an EOS one (hf `_beam_search` step d) ----
class _StopOnToken:
    """a user StoppingCriteria: a row is done once its last id is in `ids` (per-row, unlike _StopOnLength)"""

    def __init__(self, ids):
        self.ids = torch.tensor(ids)

    def __call__(self, input_ids, scores, **kw):
        return torch.isin(input_ids[:, -1], self.ids.to(input_ids.device))


@pytest.mark.parametrize("nb,lp,crit_kind", [(3, 1.0, "len"), (4, -1.0, "len"), (3, 1.0, "tok"), (5, -1.0, "tok")])
def test_beam_with_stopping_criteria_equals_transformers(tiny_opt, nb, lp, crit_kind):
    from transformers import StoppingCriteriaList

    torch.manual_seed(6)
    prompt = torch.randint(4, 40, (2, 6))
    mk = (lambda: _StopOnLength(6 + 4)) if crit_kind == "len" else (lambda: _StopOnToken([7, 12, 25, 30, 31]))
    step, first = _stepper(tiny_opt, prompt, 2 * nb)
    ours = beam_search(step, first, 2, nb, 9, lp, -1, 1, False, 1, stopping=StoppingCriteriaList([mk()]), prefix=prompt)
    hf = _hf(tiny_opt, prompt, max_new_tokens=9, do_sample=False, num_beams=nb, length_penalty=lp, eos_token_id=None, early_stopping=False,
             stopping_criteria=StoppingCriteriaList([mk()]))
    if crit_kind == "len":
        assert ours.shape[1] == 4
    _eq(ours, hf, pad=-1)  # hf fills the unused tail of a hypothesis with -1 when there is no EOS id; so does beam_search


def test_beam_with_max_time_returns_finished_hypotheses(tiny_opt):
    """generate(num_beams > 1, max_time=...): hf's MaxTimeCriteria flags every candidate once the budget is spent; the candidates of that
    step must come back as hypotheses (not an empty / stale finished set)."""
    from transformers import MaxTimeCriteria, StoppingCriteriaList

    torch.manual_seed(7)
    prompt = torch.randint(4, 40, (2, 6))
    step, first = _stepper(tiny_opt, prompt, 6)
    ours = beam_search(step, first, 2, 3, 9, 1.0, -1, 1, False, 1, stopping=StoppingCriteriaList([MaxTimeCriteria(max_time=0.0)]), prefix=prompt)
    hf = _hf(tiny_opt, prompt, max_new_tokens=9, do_sample=False, num_beams=3, eos_token_id=None, early_stopping=False, max_time=0.0)
    assert ours.shape[1] == 1
    _eq(ours, hf, pad=-1)


def test_greedy_custom_criterion_without_eos_keeps_real_tokens(tiny_opt):
    """hf `_sample` pads finished rows only when an EOS criterion exists: with eos_token_id=None and a per-row criterion the rows that
    stopped early keep receiving real tokens until every row is done."""
    from transformers import StoppingCriteriaList

    torch.manual_seed(8)
    prompt = torch.randint(4, 40, (3, 5))
    ids = [7, 12, 25, 30, 31, 9, 14]
    step, first = _stepper(tiny_opt, prompt, 3)
    ours = sample_loop(step, first, 12, eos_id=-1, pad_id=1, greedy=True, stopping=StoppingCriteriaList([_StopOnToken(ids)]), prefix=prompt)
    hf = _hf(tiny_opt, prompt, max_new_tokens=12, do_sample=False, num_beams=1, eos_token_id=None, stopping_criteria=StoppingCriteriaList([_StopOnToken(ids)]))
    _eq(ours, hf)


# ---- round 6 ----
class _StopOnScore:
    """a user StoppingCriteria that READS `scores` (hf `_beam_search` passes the candidates' running log-probabilities): done once a row's
    score falls below a threshold.  With `scores=None` it would raise."""

    def __init__(self, thr):
        self.thr = thr

    def __call__(self, input_ids, scores, **kw):
        assert scores is not None and scores.shape[0] == input_ids.shape[0]
        return scores.reshape(input_ids.shape[0], -1)[:, -1] < self.thr


def test_beam_stopping_criteria_receive_the_candidate_scores(tiny_opt):
    from transformers import StoppingCriteriaList

    torch.manual_seed(9)
    prompt = torch.randint(4, 40, (2, 6))
    step, first = _stepper(tiny_opt, prompt, 6)
    out = beam_search(step, first, 2, 3, 9, 1.0, -1, 1, False, 1, stopping=StoppingCriteriaList([_StopOnScore(-6.0)]), prefix=prompt)
    assert out.shape[0] == 2 and 1 <= out.shape[1] <= 9


def test_beam_fill_is_the_pad_id_when_the_model_has_an_eos_that_cannot_fire(tiny_opt):
    """generate(min_new_tokens >= max_new_tokens) empties the EOS list for the search (EOS can never fire) but the model still HAS an EOS id:
    hf fills the unused tail of a hypothesis shortened by a stopping criterion with the pad id, not -1 (`fill_id`)."""
    from transformers import StoppingCriteriaList

    torch.manual_seed(6)
    prompt = torch.randint(4, 40, (2, 6))
    step, first = _stepper(tiny_opt, prompt, 6)
    crit = lambda: StoppingCriteriaList([_StopOnToken([7, 12, 25, 30, 31])])
    a = beam_search(step, first, 2, 3, 9, 1.0, -1, 1, False, 1, stopping=crit(), prefix=prompt)
    step, first = _stepper(tiny_opt, prompt, 6)
    b = beam_search(step, first, 2, 3, 9, 1.0, -1, 1, False, 1, stopping=crit(), prefix=prompt, fill_id=1)
    assert a.shape == b.shape
    assert torch.equal(torch.where(a < 0, torch.ones_like(a), a), b)  # same hypotheses; only the fill differs
    assert (b >= 0).all()
