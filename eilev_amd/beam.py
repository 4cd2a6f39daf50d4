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
def beam_search_device(step_dev, logits_buf: torch.Tensor, first_logits: torch.Tensor, batch: int, num_beams: int, max_new_tokens: int,
                       length_penalty: float = 1.0, eos_id=-1, pad_id: int = 1, early_stopping=False, num_return_sequences: int = 1,
                       use_graph: bool = True, check_every: int = 4, topk_fn=None, advance_fn=None, anc_state=None) -> torch.Tensor:
    """The plain beam search of `beam_search` (no sampler, processors, stopping criteria or minimum length: the sample script's call,
    ref:samples/eilev_generate_action_narration.py:60-73) with NOTHING per step on the host: the step index lives on the device, every
    slice by it is an index tensor, the hypotheses are updated in place — so selection + ancestor-table update + the HIP decode step
    are ONE captured graph replayed per generated token (the loop of `beam_search` launches ~25 small torch kernels per step from Python).

    ``step_dev(next_tokens (R,), beam_src (R,) int64)``: capture-safe decode step that leaves the fp32 logits of every row in
    ``logits_buf`` (R, V).  Same arithmetic, same torch ops and tie behaviour as `beam_search` (tests/test_beam_device_loop.py compares
    the two loops step for step on the CPU); the early exits are looked at every ``check_every`` steps as there.

    ``topk_fn(logits_buf, run_score (B, K) f32) -> (values (R, keep) f32, token ids (R, keep) int32)``: the per-row best `keep` of
    log_softmax + running score (`eilev_topk_logprob`: one kernel instead of torch's log_softmax + top-k over K x vocabulary, ~0.25 ms per
    step at 5 x 50 272); the global top `keep` of a sample lie among its rows' top `keep`, merged here.

    ``advance_fn(row_lp, row_tok, st)`` (with ``topk_fn``): the WHOLE bookkeeping of a step as one kernel (`eilev_beam_advance`; st = the
    state tensors below, updated in place, plus the tokens to feed) — the ~35 small torch kernels of `select` cost ~4 us each even inside a
    graph.  ``anc_state`` = (state, tokens): the decode step's device counter (cur = state[0] - 1) and token buffer; `step_dev` is then
    called with (None, None): the kernel already wrote the tokens and the ancestor table."""
    dev = first_logits.device
    B, nb, T = batch, num_beams, max_new_tokens
    V = first_logits.shape[-1]
    from .sampling import eos_list

    eos = eos_list(eos_id)
    eos_t = torch.tensor(eos, dtype=torch.int64, device=dev) if eos else None
    keep = max(2, 1 + len(eos)) * nb
    top_mask = (torch.arange(keep, device=dev) < nb)[None, :]
    lp = float(length_penalty)

    run_seq = torch.full((B, nb, T), pad_id, dtype=torch.int64, device=dev)
    fin_seq = run_seq.clone()
    run_len = torch.zeros((B, nb), dtype=torch.int64, device=dev)
    fin_len = run_len.clone()
    run_score = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    run_score[:, 1:] = NEG
    fin_score = torch.full((B, nb), NEG, dtype=torch.float32, device=dev)
    finished = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    can_improve = torch.ones((B, 1), dtype=torch.bool, device=dev)
    offs = (torch.arange(B, device=dev) * nb).view(B, 1)
    cur_t = torch.zeros((), dtype=torch.int64, device=dev)  # tokens selected so far
    next_tok = torch.zeros(B * nb, dtype=torch.int64, device=dev)
    next_src = torch.zeros(B * nb, dtype=torch.int64, device=dev)
    flags = torch.zeros(2, dtype=torch.bool, device=dev)  # [any(can_improve), all(finished)]
    # x / float(n) ** length_penalty of `beam_search` (a tensor divided by a Python scalar) as a table over n = 1..T: ATen divides by the
    # scalar cast to fp32 on the CPU and multiplies by its fp32 reciprocal on the GPU (BinaryDivTrueKernel) — the same here, bit for bit
    pow_tab = torch.tensor([float(n) ** lp for n in range(1, T + 1)], dtype=torch.float64).to(torch.float32).to(dev)
    on_gpu = dev.type == "cuda"
    if on_gpu:
        pow_tab = torch.ones((), dtype=torch.float32, device=dev) / pow_tab

    def over_len_pow(x, idx):  # x / float(idx + 1) ** lp; idx: 0-d int64 tensor on the device
        f = pow_tab.index_select(0, idx.view(1))  # (not pow_tab[idx]: indexing by a 0-d tensor reads it back on the host)
        return x * f if on_gpu else x / f

    fused = advance_fn is not None and topk_fn is not None
    if fused:
        finished = finished.to(torch.uint8)
        can_improve = can_improve.view(B).to(torch.uint8)
        st = dict(run_seq=run_seq, run_score=run_score, fin_seq=fin_seq, fin_score=fin_score, fin_len=fin_len, finished=finished,
                  can_improve=can_improve, pow_tab=pow_tab, reciprocal=on_gpu, eos=eos, keep=keep,
                  early=1 if early_stopping is True else (2 if (early_stopping == "never" and lp > 0.0) else 0))

    def select_fused():
        row_lp, row_tok = topk_fn(logits_buf, run_score)
        advance_fn(row_lp, row_tok, st)
        flags[0] = can_improve.any()
        flags[1] = finished.all()

    def select():
        if fused:
            return select_fused()
        if topk_fn is not None:
            row_lp, row_tok = topk_fn(logits_buf, run_score)
            top_lp, pos = torch.topk(row_lp.view(B, nb * keep), keep, dim=1)
            src = pos // keep
            tok = row_tok.view(B, nb * keep).gather(1, pos).to(torch.int64)
        else:
            logp = torch.log_softmax(logits_buf.view(B * nb, V).float(), dim=-1).view(B, nb, V) + run_score[:, :, None]
            top_lp, top_ix = torch.topk(logp.view(B, nb * V), keep, dim=1)
            src = top_ix // V
            tok = top_ix % V
        cand = _take(run_seq, src)
        cand.scatter_(2, cur_t.view(1, 1, 1).expand(B, keep, 1), tok.unsqueeze(-1))  # cand[:, :, cur] = tok
        hit = (tok.unsqueeze(-1) == eos_t).any(-1) if eos_t is not None else torch.zeros_like(tok, dtype=torch.bool)
        hit = hit | (cur_t + 1 >= T)  # the budget: every candidate of the last step finishes
        live_lp = top_lp + hit.float() * NEG
        nxt = torch.topk(live_lp, nb, dim=1).indices
        new_run_seq, new_run_score, beam_src = _take(cand, nxt), _take(live_lp, nxt), _take(src, nxt)
        just_done = hit & top_mask
        done_lp = over_len_pow(top_lp, cur_t)
        if early_stopping is True:
            done_lp = done_lp + torch.all(finished, dim=1, keepdim=True).float() * NEG
        done_lp = done_lp + (~can_improve).float() * NEG + (~just_done).float() * NEG
        all_seq = torch.cat((fin_seq, cand), dim=1)
        all_score = torch.cat((fin_score, done_lp), dim=1)
        all_len = torch.cat((fin_len, (cur_t + 1).expand_as(tok)), dim=1)
        all_done = torch.cat((finished, just_done), dim=1)
        best = torch.topk(all_score, nb, dim=1).indices
        n_fs, n_fsc, n_fl, n_fd = _take(all_seq, best), _take(all_score, best), _take(all_len, best), _take(all_done, best)
        ref_idx = torch.full_like(cur_t, T - 1) if (early_stopping == "never" and lp > 0.0) else cur_t
        best_running = over_len_pow(new_run_score[:, :1], ref_idx)
        worst_done = torch.where(n_fd, n_fsc.min(dim=1, keepdim=True).values, torch.full_like(n_fsc, NEG))
        n_ci = can_improve & torch.any(best_running > worst_done, dim=1, keepdim=True)
        # the hypotheses are updated IN PLACE (a replayed graph works on the same storage)
        next_tok.copy_(_take(tok, nxt).reshape(-1))
        next_src.copy_((beam_src + offs).reshape(-1))
        run_seq.copy_(new_run_seq); run_score.copy_(new_run_score)
        fin_seq.copy_(n_fs); fin_score.copy_(n_fsc); fin_len.copy_(n_fl); finished.copy_(n_fd); can_improve.copy_(n_ci)
        flags[0] = n_ci.any()
        flags[1] = n_fd.all()
        cur_t.add_(1)

    def iteration():
        select()
        if fused:
            step_dev(None, None)
        else:
            step_dev(next_tok, next_src)

    logits_buf.view(B * nb, V).copy_(first_logits.float().repeat_interleave(nb, dim=0))
    graph = None
    cur = 0
    while True:
        if cur + 1 >= T:  # the last selection: no decode step follows
            select()
            cur += 1
            break
        if graph is None or not use_graph:
            iteration()  # (the first one runs eagerly: it is also the warm-up the capture needs)
            if use_graph and dev.type == "cuda" and T > 3:
                graph = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                snap = [t_.clone() for t_ in (run_seq, run_score, fin_seq, fin_score, fin_len, finished, can_improve, cur_t, next_tok, next_src, flags)]
                with torch.cuda.stream(side):
                    with torch.cuda.graph(graph, stream=side):
                        iteration()
                torch.cuda.current_stream(dev).wait_stream(side)
                # (capture records, it does not run — but keep the state exactly as the eager iteration left it whatever the runtime did)
                for t_, v in zip((run_seq, run_score, fin_seq, fin_score, fin_len, finished, can_improve, cur_t, next_tok, next_src, flags), snap):
                    t_.copy_(v)
        else:
            graph.replay()
        cur += 1
        if cur % check_every == 0:
            f = flags.tolist()
            if not (f[0] and not (f[1] and early_stopping is True)):
                break
    out = fin_seq[:, :num_return_sequences].reshape(B * num_return_sequences, T)
    n = int(fin_len[:, :num_return_sequences].max().item())
    return out[:, : max(n, 0)]


@torch.no_grad()
def beam_search(step, first_logits: torch.Tensor, batch: int, num_beams: int, max_new_tokens: int, length_penalty: float = 1.0,
                eos_id=-1, pad_id: int = 1, early_stopping=False, num_return_sequences: int = 1, sampler: dict | None = None,
                min_new_tokens: int = 0, processors=None, stopping=None, prefix: torch.Tensor | None = None, fill_id: int | None = None) -> torch.Tensor:
    """``first_logits``: (batch, vocab) fp32 from the prefill.  Returns int64 (batch * num_return_sequences, n_generated).

    ``sampler`` (``generate(num_beams > 1, do_sample=True)``, hf `_get_top_k_continuations`): the 2K continuations of a step are DRAWN
    without replacement from softmax(warped log-probabilities + beam scores) over all beams x vocabulary instead of being the top 2K,
    and keep their draw order (HF lets only the first K drawn finish); everything else is the same bookkeeping.

    ``processors`` (`LogitsProcessorList` / callable): applied to every step's LOG-PROBABILITIES with each beam's own ids, as hf `_beam_search`
    does (`repetition_penalty`, `no_repeat_ngram_size`, user processors); ``stopping``: a candidate it flags finishes like one that produced EOS (hf step d); the search ends when every candidate is flagged;
    ``prefix`` (batch, P): ids in front of the generated ones (see sampling.sample_loop)."""
    dev = first_logits.device
    B, nb, T = batch, num_beams, max_new_tokens
    V = first_logits.shape[-1]
    from .sampling import eos_list

    eos = eos_list(eos_id)  # one id, several, or none (hf: `eos_token_id` may be a list)
    eos_t = torch.tensor(eos, dtype=torch.int64, device=dev) if eos else None
    keep = max(2, 1 + len(eos)) * nb  # hf generation/utils.py:3285-3286 beams_to_keep
    top_mask = torch.arange(keep, device=dev) < nb

    # hf `_beam_search`: `output_fill_value = pad_token_id or eos_token_id[0] if eos_token_id is not None else -1` — without an EOS id the unused
    # tail of a hypothesis is -1, whatever the pad id (reachable only through stopping criteria: every hypothesis has full length otherwise)
    # ``fill_id``: the caller knows better (generate() empties the EOS list when min_new_tokens >= max_new_tokens; hf still fills with the pad id there)
    fill = int(fill_id) if fill_id is not None else (int(pad_id) if eos else -1)
    run_seq = torch.full((B, nb, T), fill, dtype=torch.int64, device=dev)
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
    check_every = 4 if (sampler is None and processors is None and stopping is None) else 1
    cur = 0
    while True:
        logp = torch.log_softmax(logits, dim=-1)
        if processors is not None:
            seen = run_seq[:, :, :cur].reshape(B * nb, cur)
            if prefix is not None:
                seen = torch.cat((prefix.to(dev, torch.int64).repeat_interleave(nb, dim=0), seen), dim=1)
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
        if stopping is not None:
            # hf `_beam_search` step d (generation/utils.py:3438-3444): the criteria see every CANDIDATE (prompt ids + the ids generated so
            # far + the candidate token) and a flagged candidate finishes exactly like one that produced EOS — it enters the finished set with
            # its length-penalised score and cannot keep running.  (breaking out of the loop instead lost those hypotheses.)
            cs = cand[:, :, : cur + 1].reshape(B * keep, cur + 1)
            if prefix is not None:
                cs = torch.cat((prefix.to(dev, torch.int64).repeat_interleave(keep, dim=0), cs), dim=1)
            flagged = stopping(cs, top_lp.reshape(B * keep))  # (hf passes the step's scores)
            if not torch.is_tensor(flagged):
                flagged = torch.full((B * keep,), bool(flagged), dtype=torch.bool, device=dev)
            hit = hit | flagged.to(dev).view(B, keep)
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
        if stopping is not None and bool(hit.all()):  # hf: `valid_continuations` — every candidate was stopped, nothing can continue
            break
        # The early exits (nothing can improve any more; early_stopping=True and every beam finished) read device values back: a host
        # synchronisation per step, during which the GPU idles behind the ~20 small selection kernels above.  They are looked at every 4th
        # step only: once an exit condition holds it keeps holding and the finished set is frozen (done_lp is masked by ~can_improve /
        # all-finished), so up to three extra steps change nothing that is returned.  With a sampler, logits processors or stopping criteria
        # the test runs EVERY step: extra iterations would draw from the torch generator and call the user's (possibly stateful)
        # callbacks more often than hf does.
        if cur % check_every == 0:
            open_beam = not (bool(torch.all(finished)) and early_stopping is True)
            if not (bool(torch.any(can_improve)) and open_beam):
                break
        logits = step(run_seq[:, :, cur - 1].reshape(-1), (beam_src + offs).reshape(-1)).float()

    out = fin_seq[:, :num_return_sequences].reshape(B * num_return_sequences, T)
    n = int(fin_len[:, :num_return_sequences].max().item())
    return out[:, : max(n, 0)]
