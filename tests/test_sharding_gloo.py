"""N>1 path on CPU: world_size-2 gloo processes run the clip deal + all-gather + sample split and must
reproduce the single-process ordering exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eilev_amd.sharding import deal_clips, gather_clip_tokens, max_local_clips, my_samples


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_clips, rows, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = 6
        # "encode": clip c -> rows filled with c*100 + row index (stands in for ViT+Q-Former+projection)
        mine = deal_clips(num_clips, world, rank)
        local = torch.stack([torch.full((rows, D), float(c * 100)) + torch.arange(rows).float()[:, None] for c in mine]) if mine else torch.zeros(0, rows, D)
        local = local.reshape(-1, D).to(torch.bfloat16)
        full = gather_clip_tokens(local, num_clips, rows)
        ret[rank] = full.float()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_clips", [34, 17, 5])
def test_gather_restores_global_clip_order(num_clips):
    world, rows = 2, 4
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, num_clips, rows, ret), nprocs=world, join=True)
    expect = torch.stack([torch.full((rows, 6), float(c * 100)) + torch.arange(rows).float()[:, None] for c in range(num_clips)]).reshape(-1, 6)
    for r in range(world):
        assert torch.equal(ret[r], expect.to(torch.bfloat16).float())


def test_deal_and_sample_split_cover_everything():
    for world in (1, 2, 4, 8):
        for n in (1, 17, 136, 137):
            got = sorted(c for r in range(world) for c in deal_clips(n, world, r))
            assert got == list(range(n))
            assert max(len(deal_clips(n, world, r)) for r in range(world)) == max_local_clips(n, world)
        for s in (1, 8, 64):
            got = [x for r in range(world) for x in my_samples(s, world, r)]
            assert got == list(range(s))


def test_single_process_gather_is_identity():
    x = torch.randn(8, 3)
    assert gather_clip_tokens(x, 2, 4) is x
