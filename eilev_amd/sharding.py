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

    local_ids: (n_local, T) int ids of the samples of my_samples(num_samples, world, rank), in that order.  One all-gather of
    (per, T) int64 per rank, per = ceil(num_samples / world) (ranks with fewer samples send pad rows that are cut off).
    world == 1 (or no process group): identity."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_ids
    world = dist.get_world_size(group)
    per = (num_samples + world - 1) // world
    T = local_ids.shape[1]
    send = local_ids.new_full((per, T), pad_id)
    send[: local_ids.shape[0]] = local_ids
    recv = local_ids.new_empty((world * per, T))
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv[:num_samples] if num_samples == world * per else torch.cat(
        [recv[r * per: r * per + len(my_samples(num_samples, world, r))] for r in range(world)])
