"""Multinomial sampling over the HIP decode step (host-side plumbing; the step itself — one token through the language model on the
KV cache — is the HIP kernel path).

What `transformers` does for `generate(do_sample=True, num_beams=1)` [hf generation/utils.py `_sample`; the reference inherits it:
ref:eilev/model/v2.py:312-322, exercised by ref:tests/model/test_model_v2.py:194]: per step the next-token logits go through the
warpers in this order — temperature (logits / T), top-k (keep the k largest, rest -inf), top-p (smallest set of tokens whose
probability mass reaches top_p, at least one kept) — then softmax and one `torch.multinomial` draw per row; rows that produced EOS
emit the pad id from then on; the loop ends when every row has finished.  HF's defaults when only `do_sample=True` is given:
temperature 1.0, top_k 50, top_p 1.0."""
from __future__ import annotations

import torch


def warp_logits(logits: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """(rows, vocab) fp32 -> warped scores (filtered entries are -inf)."""
    scores = logits.float()
    if temperature is not None and temperature != 1.0:
        if temperature <= 0:
            raise ValueError("temperature must be > 0")
        scores = scores / float(temperature)
    if top_k and top_k > 0:
        k = min(int(top_k), scores.shape[-1])
        kth = torch.topk(scores, k, dim=-1).values[..., -1:]
        scores = scores.masked_fill(scores < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        if not 0.0 < top_p:
            raise ValueError("top_p must be in (0, 1]")
        srt, idx = torch.sort(scores, dim=-1, descending=False)
        cum = srt.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1.0 - float(top_p))   # the low-probability tail whose mass stays below 1 - top_p
        remove[..., -1:] = False               # min_tokens_to_keep = 1
        scores = scores.masked_fill(remove.scatter(-1, idx, remove), float("-inf"))
    return scores


def eos_list(eos_id) -> list:
    """`eos_token_id` as HF accepts it (None / negative = disabled, an int, or several ids) -> list of ints."""
    if eos_id is None:
        return []
    if isinstance(eos_id, (list, tuple)):
        return [int(e) for e in eos_id if e is not None and int(e) >= 0]
    return [int(eos_id)] if int(eos_id) >= 0 else []


def sample_loop(step, first_logits: torch.Tensor, max_new_tokens: int, eos_id=-1, pad_id: int = 0, temperature: float = 1.0,
                top_k: int = 50, top_p: float = 1.0, generator: torch.Generator | None = None, min_new_tokens: int = 0,
                greedy: bool = False, processors=None, stopping=None, prefix: torch.Tensor | None = None) -> torch.Tensor:
    """`step(next_tokens (R,), row_src (R,)) -> logits (R, vocab)` is the decode step of engine.beam_decode / t5_beam (row_src is the
    identity here).  Returns (R, n) new tokens, n <= max_new_tokens (the loop stops once every row has produced EOS, as HF does).

    ``eos_id`` may be a list (any of the ids finishes a row); ``min_new_tokens`` is HF's MinNewTokensLengthLogitsProcessor (the EOS
    logits are -inf while fewer tokens than that have been generated; applied before the warpers); ``greedy=True`` takes the argmax
    instead of drawing (the host-side form of greedy search used for the stopping rules the captured device step does not cover).

    ``processors``: a `transformers.LogitsProcessorList` (or any callable ``(input_ids, scores) -> scores``) applied to every step's scores
    before the warpers, as hf `_sample` does — `repetition_penalty`, `no_repeat_ngram_size` and user `logits_processor`s arrive here;
    ``stopping``: a `StoppingCriteriaList` / callable ``(input_ids, scores) -> bool per row`` (`stopping_criteria`, `max_time`), a row it
    flags is finished like one that produced EOS.  ``prefix`` (R, P): the ids the processors see in front of the generated ones — none for
    the OPT path (the reference drives the LM with inputs_embeds: hf then starts from an empty id tensor), the start token for T5."""
    R = first_logits.shape[0]
    dev = first_logits.device
    ident = torch.arange(R, device=dev)
    unfinished = torch.ones(R, dtype=torch.bool, device=dev)
    eos = eos_list(eos_id)
    eos_t = torch.tensor(eos, dtype=torch.int64, device=dev) if eos else None
    out = []
    logits = first_logits
    prev = prefix.to(dev, torch.int64) if prefix is not None else torch.zeros((R, 0), dtype=torch.int64, device=dev)
    for t in range(max_new_tokens):
        if processors is not None:
            logits = processors(prev, logits.float())
        if eos_t is not None and t < int(min_new_tokens):
            logits = logits.float().index_fill(-1, eos_t, float("-inf"))
        if greedy:
            nxt = logits.argmax(dim=-1)
        else:
            probs = warp_logits(logits, temperature, top_k, top_p).softmax(dim=-1)
            nxt = torch.multinomial(probs, 1, generator=generator).squeeze(1)
        if eos_t is not None:  # hf `_sample` pads finished rows only when an EOS criterion exists (`has_eos_stopping_criteria`); rows finished
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, int(pad_id)))  # by a custom criterion alone keep their real tokens
        out.append(nxt)
        prev = torch.cat((prev, nxt.view(R, 1)), dim=1)
        if eos_t is not None:
            unfinished = unfinished & ~torch.isin(nxt, eos_t)
        if stopping is not None:
            done = stopping(prev, logits)
            unfinished = unfinished & ~(done.to(dev) if torch.is_tensor(done) else torch.full((R,), bool(done), device=dev))
        if (eos_t is not None or stopping is not None) and not bool(unfinished.any()):
            break
        if t + 1 < max_new_tokens:
            logits = step(nxt, ident)
    return torch.stack(out, dim=1)
