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


def _grad_worker(rank, world, port, ret):
    from eilev_amd.train import allreduce_gradients

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        params = {f"p{i}": torch.zeros(shape, requires_grad=True) for i, shape in enumerate([(5, 7), (300,), (64, 33), (1, 4, 8)])}
        params["frozen"] = torch.zeros(3)  # not trainable: skipped
        for i, (k, p) in enumerate(params.items()):
            if p.requires_grad and not (rank == 1 and k == "p1"):  # rank 1 has no gradient for p1 (e.g. an unused branch)
                p.grad = torch.full_like(p, float(rank + 1)) * (i + 1)
        allreduce_gradients(params, bucket_bytes=1024)  # several buckets
        ret[rank] = {k: p.grad.clone() for k, p in params.items() if p.requires_grad}
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks():
    """train_v2 under torchrun: every rank ends with the mean gradient (ref:scripts/general/train_v2.py via accelerate DDP)."""
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_grad_worker, args=(world, port, ret), nprocs=world, join=True)
    for i, k in enumerate(["p0", "p1", "p2", "p3"]):
        expect = (1.0 * (i + 1) + (0.0 if k == "p1" else 2.0 * (i + 1))) / 2
        for r in range(world):
            assert torch.allclose(ret[r][k], torch.full_like(ret[r][k], expect)), (k, r)


def _ids_worker(rank, world, port, num_samples, ret):
    from eilev_amd.sharding import gather_token_ids

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = my_samples(num_samples, world, rank)
        local = torch.tensor([[s * 10 + t for t in range(5)] for s in mine], dtype=torch.int64).reshape(len(mine), 5)
        ret[rank] = gather_token_ids(local, num_samples)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_samples", [8, 5, 1])
def test_gather_generated_ids_in_sample_order(num_samples):
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_ids_worker, args=(world, port, num_samples, ret), nprocs=world, join=True)
    expect = torch.tensor([[s * 10 + t for t in range(5)] for s in range(num_samples)], dtype=torch.int64)
    for r in range(world):
        assert torch.equal(ret[r], expect), (r, ret[r])
