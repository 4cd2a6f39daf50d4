"""Clip sharding across ranks (one process per GPU) and the single exchange step of the path.

The reference never shards inside a sample; its closest ancestor is the accelerate
all-gather of ref:scripts/general/generate_narration_texts.py:124-127.  Here the unit is the CLIP:
all clips of a global step are dealt round-robin to the ranks (clip c -> rank c % G), every rank runs
ViT + Q-Former + projection on its clips, ONE all-gather (RCCL over xGMI with backend "nccl") exchanges
the projected query tokens (num_query x Dt bf16 per clip, 164 KB at OPT-2.7B), and the language model
runs data-parallel over samples with replicated weights.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def deal_clips(num_clips: int, world: int, rank: int):
    """Global clip indices owned by `rank` (round-robin)."""
    return list(range(rank, num_clips, world))


def max_local_clips(num_clips: int, world: int) -> int:
    return (num_clips + world - 1) // world


def gather_clip_tokens(local_feats: torch.Tensor, num_clips: int, rows_per_clip: int, group=None) -> torch.Tensor:
    """All-gather the per-clip projected tokens and return them in GLOBAL clip order.

    local_feats: (n_local * rows_per_clip, D) for the clips of deal_clips(num_clips, world, rank), in that order.
    Returns (num_clips * rows_per_clip, D) on every rank.  world == 1 (or no process group): identity.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_feats
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    D = local_feats.shape[1]
    slots = max_local_clips(num_clips, world)
    send = local_feats.new_zeros((slots * rows_per_clip, D))  # pad so every rank sends the same size
    send[: local_feats.shape[0]] = local_feats
    recv = local_feats.new_empty((world, slots, rows_per_clip, D))
    dist.all_gather_into_tensor(recv.view(world * slots * rows_per_clip, D), send, group=group)
    # rank r, slot s holds global clip s * world + r  ->  transpose to clip-major and cut the padding
    out = recv.permute(1, 0, 2, 3).reshape(slots * world, rows_per_clip, D)[:num_clips]
    return out.reshape(num_clips * rows_per_clip, D)


def my_samples(num_samples: int, world: int, rank: int):
    """Contiguous block of samples whose language-model pass runs on `rank`."""
    per = (num_samples + world - 1) // world
    return list(range(rank * per, min(num_samples, (rank + 1) * per)))


def gather_token_ids(local_ids: torch.Tensor, num_samples: int, pad_id: int = 1, group=None) -> torch.Tensor:
    """Generated token ids of every sample of the global step, in sample order, on every rank — the analogue of the
    reference's `accelerator.gather_for_metrics(generated_ids)` (ref:scripts/general/generate_narration_texts.py:124-127).

    local_ids: (n_local, T_rank) int ids of the samples of my_samples(num_samples, world, rank), in that order; T_rank may differ
    between ranks (rows are right-padded with pad_id to the longest).  One all-gather of (per, T) int64 per rank,
    per = ceil(num_samples / world) (ranks with fewer samples send pad rows that are cut off).
    world == 1 (or no process group): identity."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_ids
    world = dist.get_world_size(group)
    per = (num_samples + world - 1) // world
    # ranks stop at different lengths once EOS is on (and a rank without samples has no length at all): agree on the longest
    # first and right-pad with pad_id — accelerate's pad_across_processes(dim=1) in the reference (:124)
    t = torch.tensor([local_ids.shape[1] if local_ids.shape[0] else 0], dtype=torch.int64, device=local_ids.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    T = int(t.item())
    send = local_ids.new_full((per, T), pad_id)
    send[: local_ids.shape[0], : local_ids.shape[1]] = local_ids
    recv = local_ids.new_empty((world * per, T))
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv[:num_samples] if num_samples == world * per else torch.cat(
        [recv[r * per: r * per + len(my_samples(num_samples, world, r))] for r in range(world)])


# ---- the exchange that sends each rank only the clips of ITS samples (all-to-all-v), in encode chunks ---------------------------
class ExchangePlan:
    """Who encodes which clip, who consumes it, and the per-round send / receive blocks — pure host arithmetic.

    A global step has ``num_samples`` samples of ``clips_per_sample`` clips (global clip c belongs to sample c // clips_per_sample).
    Clip c is ENCODED on rank c % world (deal_clips) and CONSUMED by the rank that runs the language model of its sample
    (my_samples: contiguous blocks).  Every rank encodes its clips in chunks of ``chunk_clips`` (one ViT launch group) and ships
    a chunk as soon as it is projected, so that the exchange of round j runs under the ViT of round j + 1.  Because both the
    deal and the sample split are monotone in c, the rows a rank sends to one peer within a round are CONTIGUOUS in its chunk
    (no packing), and a receiver stages rounds back to back, source-major inside a round; ``order`` maps the staging slots to
    global clip order.  All quantities are in CLIPS; multiply by rows_per_clip for rows."""

    def __init__(self, num_samples: int, clips_per_sample: int, world: int, rank: int, chunk_clips: int):
        if min(num_samples, clips_per_sample, world, chunk_clips) < 1 or not (0 <= rank < world):
            raise ValueError("bad exchange plan arguments")
        self.num_samples, self.clips_per_sample, self.world, self.rank, self.chunk_clips = num_samples, clips_per_sample, world, rank, chunk_clips
        C = num_samples * clips_per_sample
        self.num_clips = C
        per = (num_samples + world - 1) // world
        self.n_local = len(range(rank, C, world))
        self.rounds = max(1, -(-max_local_clips(C, world) // chunk_clips))
        R, G = self.rounds, world
        self.send_rows = [[0] * G for _ in range(R)]
        self.send_off = [[0] * G for _ in range(R)]
        self.recv_rows = [[0] * G for _ in range(R)]
        self.recv_off = [[0] * G for _ in range(R)]
        incoming = []  # (round, source, clip) of the clips this rank consumes
        for c in range(C):
            q, i = c % G, c // G
            j, r = i // chunk_clips, (c // clips_per_sample) // per
            if q == rank:
                if self.send_rows[j][r] == 0:
                    self.send_off[j][r] = i - j * chunk_clips
                assert self.send_off[j][r] + self.send_rows[j][r] == i - j * chunk_clips, "blocks per peer must be contiguous"
                self.send_rows[j][r] += 1
            if r == rank:
                incoming.append((j, q, c))
                self.recv_rows[j][q] += 1
        incoming.sort()
        slot = 0
        for j in range(R):
            for q in range(G):
                self.recv_off[j][q] = slot
                slot += self.recv_rows[j][q]
        self.n_consumed = len(incoming)
        # staging slot of my k-th consumed clip in GLOBAL clip order
        by_clip = sorted(range(len(incoming)), key=lambda s: incoming[s][2])
        self.order = by_clip
        self.consumed_clips = [incoming[s][2] for s in by_clip]
        self.identity = by_clip == list(range(len(by_clip)))

    def chunk_range(self, j: int):
        """Local clip indices [a, b) that this rank encodes in round j (empty when it has fewer chunks than the longest rank)."""
        a = min(self.n_local, j * self.chunk_clips)
        return a, min(self.n_local, a + self.chunk_clips)
