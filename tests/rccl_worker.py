"""One rank of tests/test_hip_sharded.py::test_two_real_rccl_ranks (spawned, one process per GPU; RANK / WORLD_SIZE / MASTER_* from the env).

Every rank builds the ExchangePlan of a small global step, fills the rows it "encoded" with a pattern that names the global clip
(row value = clip * 64 + row-in-clip, every column), runs the exchange through the DIRECT RCCL transport (eilev_exchange_clip_tokens:
grouped ncclSend / ncclRecv on a side stream) chunk by chunk, and checks that what arrives is exactly the clips of ITS samples in
global clip order; then the all-gather form (eilev_gather_clip_tokens) and the gradient all-reduce buckets.  Exit code 0 = all good."""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    from eilev_amd import abi
    from eilev_amd.comm import ClipExchange, RcclComm
    from eilev_amd.sharding import ExchangePlan, deal_clips

    rpc, width = 4, 64
    comm = RcclComm(dev)
    for num_samples, cps, chunk in ((4, 3, 2), (5, 17, 4), (3, 5, 100)):
        plan = ExchangePlan(num_samples, cps, world, rank, chunk)
        mine = deal_clips(num_samples * cps, world, rank)
        rows = torch.empty((len(mine) * rpc, width), dtype=torch.bfloat16, device=dev)
        for i, c in enumerate(mine):
            for r in range(rpc):
                rows[i * rpc + r] = float((c * 8 + r) % 251)
        ex = ClipExchange(plan, rpc, width, torch.bfloat16, dev, transport="rccl", comm=comm)
        ex.timing = []
        for j in range(plan.rounds):
            a, b = plan.chunk_range(j)
            buf = ex.chunk_buffer(j)
            buf.copy_(rows[a * rpc: b * rpc])
            ex.send_round(j, buf)
        got = ex.finish()
        torch.cuda.synchronize(dev)
        want = torch.empty_like(got)
        for i, c in enumerate(plan.consumed_clips):
            for r in range(rpc):
                want[i * rpc + r] = float((c * 8 + r) % 251)
        assert got.shape == (plan.n_consumed * rpc, width) and torch.equal(got, want), (rank, num_samples, cps, chunk)
        assert any(m[0] == "round" for m in ex.timing) and any(m[0] == "wait" for m in ex.timing)
    # latency-mode form: all-gather of equal-sized blocks through the same communicator
    lib = abi.load_hip()
    blk = torch.full((6, width), float(rank + 1), dtype=torch.bfloat16, device=dev)
    allb = torch.empty((world * 6, width), dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream(dev)
    counts = (C.c_int64 * world)(*([6] * world))
    abi.check(lib.eilev_gather_clip_tokens(comm.handle, C.c_void_p(blk.data_ptr()), C.c_void_p(allb.data_ptr()), counts, world, rank, width * 2,
                                           C.c_void_p(st.cuda_stream)), "eilev_gather_clip_tokens")
    torch.cuda.synchronize(dev)
    for q in range(world):
        assert bool((allb[q * 6:(q + 1) * 6] == float(q + 1)).all())
    # gradient all-reduce of the training step (torch.distributed's RCCL): mean over ranks
    from eilev_amd.train import allreduce_gradients

    p = torch.nn.Parameter(torch.zeros(1000, device=dev))
    p.grad = torch.full_like(p, float(rank))
    allreduce_gradients([p])
    assert torch.allclose(p.grad, torch.full_like(p, (world - 1) / 2.0))
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: ok", flush=True)


if __name__ == "__main__":
    main()
