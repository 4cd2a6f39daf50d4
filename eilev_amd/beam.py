"""Beam search bookkeeping (host side) for ``generate(num_beams > 1)``.

The sample script's default is ``num_beams=5, length_penalty=-1, max_new_tokens=32, eos_token_id=50118``
(ref:samples/eilev_generate_action_narration.py:60-73), executed by HF ``GenerationMixin._beam_search``
(hf generation/utils.py:3208-3560, called with ``inputs_embeds`` only, so the prompt length seen by the scorer is 0).
This module re-implements that selection rule — vectorised top-2K continuation pick, running / finished beam
sets, length penalty ``score / len**penalty``, the ``early_stopping=False`` heuristic — independently of the
model: it only needs ``step(tokens, beam_src) -> logits`` which (a) reorders the KV cache rows by ``beam_src``,
(b) feeds ``tokens`` and (c) returns the fp32 next-token logits of every row.  The same code drives the HIP
engine (GPU tensors) and, in the tests, the CPU oracle.
"""
from __future__ import annotations

import torch

NEG = -1.0e9


def _take(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """Gather along the beam axis (dim 1) with per-batch indices ``idx`` (B, k)."""
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.take_along_dim(t, idx.expand(-1, -1, *t.shape[2:]), dim=1)


@torch.no_grad()
def beam_search(step, first_logits: torch.Tensor, batch: int, num_beams: int, max_new_tokens: int, length_penalty: float = 1.0,
                eos_id=-1, pad_id: int = 1, early_stopping=False, num_return_sequences: int = 1, sampler: dict | None = None,
                min_new_tokens: int = 0, processors=None, stopping=None, prefix: torch.Tensor | None = None) -> torch.Tensor:
    """``first_logits``: (batch, vocab) fp32 from the prefill.  Returns int64 (batch * num_return_sequences, n_generated).

    ``sampler`` (``generate(num_beams > 1, do_sample=True)``, hf `_get_top_k_continuations`): the 2K continuations of a step are DRAWN
    without replacement from softmax(warped log-probabilities + beam scores) over all beams x vocabulary instead of being the top 2K,
    and keep their draw order (HF lets only the first K drawn finish); everything else is the same bookkeeping.

    ``processors`` (`LogitsProcessorList` / callable): applied to every step's LOG-PROBABILITIES with each beam's own ids, as hf `_beam_search`
    does (`repetition_penalty`, `no_repeat_ngram_size`, user processors); ``stopping``: the search ends when it flags every row;
    ``prefix`` (batch, P): ids in front of the generated ones (see sampling.sample_loop)."""
    dev = first_logits.device
    B, nb, T = batch, num_beams, max_new_tokens
    V = first_logits.shape[-1]
    from .sampling import eos_list

    eos = eos_list(eos_id)  # one id, several, or none (hf: `eos_token_id` may be a list)
    eos_t = torch.tensor(eos, dtype=torch.int64, device=dev) if eos else None
    keep = max(2, 1 + len(eos)) * nb  # hf generation/utils.py:3285-3286 beams_to_keep
    top_mask = torch.arange(keep, device=dev) < nb

    run_seq = torch.full((B, nb, T), pad_id, dtype=torch.int64, device=dev)
    fin_seq = run_seq.clone()
    run_len = torch.zeros((B, nb), dtype=torch.int64, device=dev)  # generated length of every finished hypothesis
    fin_len = run_len.clone()
    run_score = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    run_score[:, 1:] = NEG  # only beam 0 is live at the first step (all beams hold the same prompt)
    fin_score = torch.full((B, nb), NEG, dtype=torch.float32, device=dev)
    finished = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    can_improve = torch.ones((B, 1), dtype=torch.bool, device=dev)
    offs = (torch.arange(B, device=dev) * nb).view(B, 1)

    logits = first_logits.float().repeat_interleave(nb, dim=0)  # (B*nb, V): identical rows, like HF's expanded prefill
    cur = 0
    while True:
        logp = torch.log_softmax(logits, dim=-1)
        if processors is not None or stopping is not None:
            seen = run_seq[:, :, :cur].reshape(B * nb, cur)
            if prefix is not None:
                seen = torch.cat((prefix.to(dev, torch.int64).repeat_interleave(nb, dim=0), seen), dim=1)
            if stopping is not None and cur > 0:
                done_all = stopping(seen, logp)
                if bool(done_all.all() if torch.is_tensor(done_all) else done_all):
                    break
            if processors is not None:
                logp = processors(seen, logp)
        if eos_t is not None and cur < int(min_new_tokens):  # MinNewTokensLengthLogitsProcessor acts on the log-probabilities here
            logp = logp.index_fill(-1, eos_t, float("-inf"))
        if sampler is not None:
            from .sampling import warp_logits

            logp = warp_logits(logp, sampler.get("temperature", 1.0), sampler.get("top_k", 50), sampler.get("top_p", 1.0))
        logp = logp.view(B, nb, V) + run_score[:, :, None]
        if sampler is not None:
            top_ix = torch.multinomial(torch.softmax(logp.view(B, nb * V), dim=-1), keep, generator=sampler.get("generator"))
            top_lp = torch.gather(logp.view(B, nb * V), 1, top_ix)
        else:
            top_lp, top_ix = torch.topk(logp.view(B, nb * V), keep, dim=1)
        src = top_ix // V
        tok = top_ix % V
        cand = _take(run_seq, src)
        cand[:, :, cur] = tok
        hit = torch.isin(tok, eos_t) if eos_t is not None else torch.zeros_like(tok, dtype=torch.bool)
        if cur + 1 >= T:
            hit = torch.ones_like(hit)

        # beams that keep running: best `nb` candidates that did not stop
        live_lp = top_lp + hit.float() * NEG
        nxt = torch.topk(live_lp, nb, dim=1).indices
        run_seq = _take(cand, nxt)
        run_score = _take(live_lp, nxt)
        beam_src = _take(src, nxt)

        # finished set: only the top `nb` candidates may finish; score with the length penalty
        just_done = hit & top_mask[None, :]
        done_lp = top_lp / float(cur + 1) ** length_penalty
        if early_stopping is True:
            done_lp = done_lp + torch.all(finished, dim=1, keepdim=True).float() * NEG
        done_lp = done_lp + (~can_improve).float() * NEG + (~just_done).float() * NEG
        all_seq = torch.cat((fin_seq, cand), dim=1)
        all_score = torch.cat((fin_score, done_lp), dim=1)
        all_len = torch.cat((fin_len, torch.full_like(tok, cur + 1)), dim=1)
        all_done = torch.cat((finished, just_done), dim=1)
        best = torch.topk(all_score, nb, dim=1).indices
        fin_seq, fin_score, fin_len, finished = _take(all_seq, best), _take(all_score, best), _take(all_len, best), _take(all_done, best)

        cur += 1
        # early_stopping=False heuristic: can the best running beam still beat the worst finished one?
        ref_len = T if (early_stopping == "never" and length_penalty > 0.0) else cur
        best_running = run_score[:, :1] / float(ref_len) ** length_penalty
        worst_done = torch.where(finished, fin_score.min(dim=1, keepdim=True).values, torch.full_like(fin_score, NEG))
        can_improve = can_improve & torch.any(best_running > worst_done, dim=1, keepdim=True)
        if cur >= T:  # the budget: every candidate of this step was forced to finish (hit is all ones) — known on the host, no read-back
            break
        # The early exits (nothing can improve any more; early_stopping=True and every beam finished) read device values back: a host
        # synchronisation per step, during which the GPU idles behind the ~20 small selection kernels above.  They are looked at every 4th
        # step only: once an exit condition holds it keeps holding and the finished set is frozen (done_lp is masked by ~can_improve /
        # all-finished), so up to three extra steps change nothing that is returned.
        if cur % 4 == 0:
            open_beam = not (bool(torch.all(finished)) and early_stopping is True)
            if not (bool(torch.any(can_improve)) and open_beam):
                break
        logits = step(run_seq[:, :, cur - 1].reshape(-1), (beam_src + offs).reshape(-1)).float()

    out = fin_seq[:, :num_return_sequences].reshape(B * num_return_sequences, T)
    n = int(fin_len[:, :num_return_sequences].max().item())
    return out[:, : max(n, 0)]
