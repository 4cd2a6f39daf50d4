"""The exchange step of the sharded path: projected clip tokens travel to the rank that runs their sample's language model.

Two transports behind one interface:

* ``"rccl"`` — the product path on MI355X: `eilev_exchange_clip_tokens` / `eilev_gather_clip_tokens` of libeilev_hip.so call
  RCCL directly (grouped ncclSend / ncclRecv, ncclAllGather) on a SIDE stream; the communicator is created from a
  ncclUniqueId that rank 0 draws and `torch.distributed` only carries to the other ranks (bootstrap, not data path).
* ``"torch"`` — `torch.distributed.all_to_all_single` on the default process group: what the world-size-2 gloo tests run on
  the CPU and what `bench.py --exchange torch` / `--share-gpu` time.  (`bench.py --exchange rccl` exits non-zero when the direct
  communicator cannot be created: no silent fallback.)

Both follow :class:`eilev_amd.sharding.ExchangePlan`: per encode round every rank sends contiguous blocks of its chunk and
receives into a staging buffer; `finish()` returns the consumed clips in global clip order.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import abi
from .sharding import ExchangePlan


def _loaded_librccl() -> str | None:
    """Path of the librccl the process has mapped (PyTorch ships its own copy; a second copy must not be loaded next to it)."""
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "librccl" in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None


class RcclComm:
    """ncclComm_t created through the C ABI (include/eilev.h stage 3b).  One per process, bound to the current device."""

    def __init__(self, device, group=None):
        self.lib = abi.load_hip()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        path = _loaded_librccl()
        abi.check(self.lib.eilev_comm_bind(path.encode() if path else None), "eilev_comm_bind")
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            abi.check(self.lib.eilev_comm_unique_id(ident), "eilev_comm_unique_id")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0, group=group)  # bootstrap only: 128 bytes through the existing process group
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        with torch.cuda.device(device):
            abi.check(self.lib.eilev_comm_init(C.byref(self.handle), self.world, self.rank, ident), "eilev_comm_init")

    def close(self):
        if self.handle:
            self.lib.eilev_comm_destroy(self.handle)
            self.handle = C.c_void_p()


def _i64(values):
    return (C.c_int64 * len(values))(*values)


class ClipExchange:
    """Runs one ExchangePlan: `send_round(j, chunk_rows)` per encode chunk, then `finish()`.

    chunk_rows: (n_chunk_clips * rows_per_clip, width) tensor produced on the CURRENT stream.  On a GPU every round is
    launched on a side stream behind an event, so the main stream goes straight on to the next chunk's ViT."""

    def __init__(self, plan: ExchangePlan, rows_per_clip: int, width: int, dtype, device, transport: str = "auto", group=None, comm=None):
        self.plan, self.rpc, self.width, self.dtype = plan, rows_per_clip, width, dtype
        self.device = torch.device(device)
        self.group = group
        self.gpu = self.device.type == "cuda"
        if transport == "auto":
            transport = "rccl" if self.gpu else "torch"
        if plan.world == 1:
            transport = "local"
        if transport not in ("rccl", "torch", "local"):
            raise ValueError(transport)
        self.transport = transport
        self.comm = comm
        if transport == "rccl" and comm is None:
            self.comm = RcclComm(self.device, group)
        self.lib = abi.load_hip() if (self.gpu and transport != "torch") else None
        self.side = torch.cuda.Stream(self.device) if (self.gpu and plan.world > 1) else None
        self.row_bytes = width * torch.empty((), dtype=dtype).element_size()
        # optional timing (bench.py): per round an event pair on the SIDE stream around the transfer, per step an event pair on the
        # main stream around the final wait for the side stream (= the part of the exchange the ViT did not cover)
        self.timing = None
        self._new_staging()

    def _new_staging(self):
        self.staging = torch.empty((self.plan.n_consumed * self.rpc, self.width), dtype=self.dtype, device=self.device)
        self._held = []

    def chunk_buffer(self, j: int):
        """Where the projection of round j should write its rows.  At world == 1 that is the final buffer itself (the plan is
        the identity), so the single-GPU path has no copy at all."""
        a, b = self.plan.chunk_range(j)
        if self.plan.world == 1:
            return self.staging[a * self.rpc: b * self.rpc]
        return torch.empty(((b - a) * self.rpc, self.width), dtype=self.dtype, device=self.device)

    def send_round(self, j: int, chunk_rows: torch.Tensor):
        p = self.plan
        if p.world == 1:
            a, _ = p.chunk_range(j)
            if chunk_rows.data_ptr() != self.staging[a * self.rpc:].data_ptr():  # caller did not use chunk_buffer()
                self.staging[a * self.rpc: a * self.rpc + chunk_rows.shape[0]].copy_(chunk_rows)
            return
        k = self.rpc
        srows, soff = [n * k for n in p.send_rows[j]], [o * k for o in p.send_off[j]]
        rrows, roff = [n * k for n in p.recv_rows[j]], [o * k for o in p.recv_off[j]]
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            chunk_rows.record_stream(self.side)
        self._held.append(chunk_rows)
        t_on = self.timing is not None and self.side is not None
        if t_on:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(self.side)
        if self.transport == "rccl":
            with torch.cuda.stream(self.side):
                abi.check(self.lib.eilev_exchange_clip_tokens(
                    self.comm.handle, C.c_void_p(chunk_rows.data_ptr()) if chunk_rows.numel() else None, _i64(srows), _i64(soff),
                    C.c_void_p(self.staging.data_ptr()) if self.staging.numel() else None, _i64(rrows), _i64(roff), p.world, p.rank,
                    self.row_bytes, C.c_void_p(self.side.cuda_stream)), "eilev_exchange_clip_tokens")
            if t_on:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(self.side)
                self.timing.append(("round", e0, e1, sum(srows) * self.row_bytes, sum(rrows) * self.row_bytes))
            return
        # torch.distributed transport: blocks per peer are contiguous and in rank order on both sides
        first = min((o for o, n in zip(soff, srows) if n), default=0)
        send = chunk_rows[first: first + sum(srows)]
        rfirst = roff[0]
        recv = self.staging[rfirst: rfirst + sum(rrows)]
        ctx = torch.cuda.stream(self.side) if self.side is not None else _null()
        with ctx:
            if self.gpu and dist.get_backend(self.group) == "gloo":
                # device tensors over a gloo group (bench.py --share-gpu: N ranks rehearsing on ONE GPU): staged through the host explicitly —
                # `.cpu()` waits for the side stream (which waited for the producer), the collective runs on host tensors, the copy back is
                # ordered on the side stream.  (Round 6: gloo's own device staging gave one corrupted clip in one of ~7 eight-rank runs.)
                recv_h = torch.empty(recv.shape, dtype=recv.dtype)
                dist.all_to_all_single(recv_h, send.contiguous().cpu(), output_split_sizes=rrows, input_split_sizes=srows, group=self.group)
                recv.copy_(recv_h.pin_memory() if recv_h.numel() else recv_h, non_blocking=False)
            else:
                dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=rrows, input_split_sizes=srows, group=self.group)
        if t_on:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(self.side)
            self.timing.append(("round", e0, e1, sum(srows) * self.row_bytes, sum(rrows) * self.row_bytes))

    def finish(self) -> torch.Tensor:
        """Rows of the clips this rank consumes, in global clip order: (n_consumed * rows_per_clip, width)."""
        if self.side is not None:
            if self.timing is not None:
                w0 = torch.cuda.Event(enable_timing=True)
                w0.record()
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            if self.timing is not None:
                w1 = torch.cuda.Event(enable_timing=True)
                w1.record()
                self.timing.append(("wait", w0, w1, 0, 0))
        out = self.staging
        if not self.plan.identity:
            idx = torch.tensor(self.plan.order, device=self.device)
            out = out.view(self.plan.n_consumed, self.rpc, self.width).index_select(0, idx).view(-1, self.width)
        self._new_staging()  # the next step must not overwrite rows the language model of this step still reads
        return out


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
