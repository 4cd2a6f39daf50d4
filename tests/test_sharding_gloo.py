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


def _ragged_ids_worker(rank, world, port, ret):
    from eilev_amd.sharding import gather_token_ids

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 generated 5 tokens for its 2 samples, rank 1 stopped after 3 (EOS): lengths differ between ranks
        T = 5 if rank == 0 else 3
        mine = my_samples(3, world, rank)
        local = torch.tensor([[s * 10 + t for t in range(T)] for s in mine], dtype=torch.int64).reshape(len(mine), T)
        ret[rank] = gather_token_ids(local, 3, pad_id=1)
    finally:
        dist.destroy_process_group()


def test_gather_generated_ids_pads_unequal_lengths():
    """each rank's generated length differs once EOS is on (the reference pads across processes first,
    ref:scripts/general/generate_narration_texts.py:124)."""
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_ragged_ids_worker, args=(world, port, ret), nprocs=world, join=True)
    expect = torch.tensor([[0, 1, 2, 3, 4], [10, 11, 12, 13, 14], [20, 21, 22, 1, 1]])
    for r in range(world):
        assert torch.equal(ret[r], expect), (r, ret[r])


def _exchange_worker(rank, world, port, num_samples, cps, chunk, ret):
    from eilev_amd.comm import ClipExchange
    from eilev_amd.sharding import ExchangePlan

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, D = 4, 6
        plan = ExchangePlan(num_samples, cps, world, rank, chunk)
        ex = ClipExchange(plan, rows, D, torch.bfloat16, "cpu", transport="torch")
        mine = deal_clips(num_samples * cps, world, rank)
        outs = []
        for step in range(2):  # two steps through the same exchange object (staging is renewed per step)
            for j in range(plan.rounds):
                a, b = plan.chunk_range(j)
                buf = ex.chunk_buffer(j)
                for i, c in enumerate(mine[a:b]):  # "encode": rows of clip c = c * 100 + row index (+ step)
                    buf[i * rows:(i + 1) * rows] = (torch.full((rows, D), float(c * 100 + step)) + torch.arange(rows).float()[:, None]).to(torch.bfloat16)
                ex.send_round(j, buf)
            outs.append(ex.finish().float())
        ret[rank] = (outs, plan.consumed_clips)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_samples,cps,chunk", [(4, 17, 8), (5, 3, 2), (2, 17, 136), (3, 5, 1)])
def test_exchange_delivers_my_samples_clips_in_global_order(num_samples, cps, chunk):
    """The all-to-all-v form of the exchange (each rank receives only the clips of ITS samples), in encode chunks."""
    world, rows, D = 2, 4, 6
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_exchange_worker, args=(world, port, num_samples, cps, chunk, ret), nprocs=world, join=True)
    seen = []
    for r in range(world):
        outs, clips = ret[r]
        assert clips == [c for s in my_samples(num_samples, world, r) for c in range(s * cps, (s + 1) * cps)]
        seen += clips
        for step, got in enumerate(outs):
            expect = torch.cat([torch.full((rows, D), float(c * 100 + step)) + torch.arange(rows).float()[:, None] for c in clips]) \
                if clips else torch.zeros(0, D)
            assert torch.equal(got, expect.to(torch.bfloat16).float()), (r, step)
    assert sorted(seen) == list(range(num_samples * cps))


def test_exchange_plan_simulated_for_8_ranks():
    """Every rank's plan, executed by hand for world 8 (the driver's scaling run): blocks match pairwise, nothing is lost."""
    from eilev_amd.sharding import ExchangePlan

    for S, cps, G, chunk in [(256, 17, 8, 136), (8, 17, 8, 17), (1, 17, 8, 3), (7, 33, 8, 16), (64, 17, 4, 136)]:
        plans = [ExchangePlan(S, cps, G, r, chunk) for r in range(G)]
        C = S * cps
        local = [list(range(r, C, G)) for r in range(G)]
        staging = [[None] * p.n_consumed for p in plans]
        for j in range(plans[0].rounds):
            for q in range(G):
                a, b = plans[q].chunk_range(j)
                chunk_clips = local[q][a:b]
                assert sum(plans[q].send_rows[j]) == len(chunk_clips)
                for r in range(G):
                    n, o = plans[q].send_rows[j][r], plans[q].send_off[j][r]
                    assert plans[r].recv_rows[j][q] == n
                    staging[r][plans[r].recv_off[j][q]: plans[r].recv_off[j][q] + n] = chunk_clips[o:o + n]
        for r in range(G):
            assert [staging[r][s] for s in plans[r].order] == plans[r].consumed_clips
            assert plans[r].consumed_clips == [c for s in my_samples(S, G, r) for c in range(s * cps, (s + 1) * cps)]


def test_exchange_c_abi_single_rank_is_a_copy():
    """world == 1 through the C ABI (oracle build: host pointers): the block a rank keeps is copied to its receive slot."""
    import ctypes as C

    import numpy as np

    from oracle.runner import lib as oracle_lib

    lib = oracle_lib()
    send = np.arange(40, dtype=np.uint8).reshape(10, 4)
    recv = np.zeros((12, 4), np.uint8)
    i64 = lambda *v: (C.c_int64 * len(v))(*v)
    lib.eilev_exchange_clip_tokens.restype = C.c_int
    rc = lib.eilev_exchange_clip_tokens(None, send.ctypes.data_as(C.c_void_p), i64(6), i64(2), recv.ctypes.data_as(C.c_void_p), i64(6), i64(3),
                                        C.c_int(1), C.c_int(0), C.c_int64(4), None)
    assert rc == 0 and np.array_equal(recv[3:9], send[2:8]) and not recv[:3].any() and not recv[9:].any()
    allb = np.zeros((10, 4), np.uint8)
    lib.eilev_gather_clip_tokens.restype = C.c_int
    rc = lib.eilev_gather_clip_tokens(None, send.ctypes.data_as(C.c_void_p), allb.ctypes.data_as(C.c_void_p), i64(10), C.c_int(1), C.c_int(0),
                                      C.c_int64(4), None)
    assert rc == 0 and np.array_equal(allb, send)
    assert lib.eilev_exchange_clip_tokens(None, None, i64(0, 0), i64(0, 0), None, i64(0, 0), i64(0, 0), C.c_int(2), C.c_int(0), C.c_int64(4), None) == -2
