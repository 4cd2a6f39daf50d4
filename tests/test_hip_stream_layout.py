"""-m gpu: the stream layout of the OPT decode matrices (round 5; include/eilev.h eilev_stream_layout_pack, EilevOptWeights.layers_stream).

A decode step of 17..32 rows streams every matrix once through gemm_rows32_kernel.  The packed copy holds the same values in the order the
kernel's load instructions consume them, so (a) the packing is a permutation that a numpy restatement of the kernel's row / fragment
assignment reproduces, and (b) a decode step with the copies attached gives bit-identical logits to the step on the checkpoint layout —
whose parity with the oracle tests/test_hip_real_shapes.py::test_batch_decode_step_at_real_widths_vs_oracle pins.
Arithmetic: nn.Linear of hf OPTDecoderLayer (modeling_opt.py:226-247) via ref:eilev/model/v2.py:318-322; unchanged by the layout."""
import ctypes as C

import numpy as np
import pytest
import torch

from eilev_amd import abi

from hip_utils import P, models, stream_ptr

pytestmark = pytest.mark.gpu


def rows32_shape(n, k, n_cu):
    """gemm_skinny.h rows32_shape: (K splits, workgroups along N) of a launch, or None."""
    if k % 256:
        return None
    nb, pw, k5 = (n + 15) // 16, k // 256, 0
    for c in (1, 2, 4, 8):
        if pw % c == 0 and (pw // c in (10, 5) or (pw // c == 8 and c == 1)):
            k5 = c
            if nb * c >= n_cu or pw // c == 5:
                break
    if not k5:
        return None
    cus = max(1, n_cu // k5) if k5 > 1 else n_cu
    return k5, min(nb, cus)


def stream_layout(w, grid_x):
    """Workgroup x owns rows [r0, r0 + cnt); per 16-row block (nv rows) and k-step of 32: the nv x 32 fragment as one piece, element
    (row l15, 8-element chunk lg) at (l15 * 4 + lg) * 8."""
    n, k = w.shape
    out = np.empty(n * k, dtype=w.dtype)
    per, rem = divmod(n, grid_x)
    for x in range(grid_x):
        r0, cnt = x * per + min(x, rem), per + (1 if x < rem else 0)
        for j in range((cnt + 15) // 16):
            nv = min(16, cnt - 16 * j)
            blk = w[r0 + 16 * j: r0 + 16 * j + nv].reshape(nv, k // 32, 4, 8).transpose(1, 0, 2, 3)
            out[(r0 + 16 * j) * k: (r0 + 16 * j + nv) * k] = blk.reshape(-1)
    return out.reshape(n, k)


@pytest.mark.parametrize("n,k", [(7680, 2560), (2560, 2560), (10240, 2560), (2560, 10240), (1000, 2560), (50272, 2560), (4096, 2048)])
def test_pack_is_the_kernels_permutation(n, k):
    lib = abi.load_hip()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    w = torch.arange(n * k, dtype=torch.int32, device="cuda").remainder(65521).to(torch.int16).view(n, k)  # distinct-ish 16-bit patterns
    src = w.view(torch.bfloat16)
    out = torch.empty_like(src)
    rc = lib.eilev_stream_layout_pack(P(src), n, k, P(out), stream_ptr())
    shape = rows32_shape(n, k, n_cu)
    assert shape is not None and rc == 0, (rc, shape)
    torch.cuda.synchronize()
    want = stream_layout(w.cpu().numpy(), shape[1])
    assert np.array_equal(out.view(torch.int16).cpu().numpy(), want)
    assert np.array_equal(np.sort(want.reshape(-1)), np.sort(w.cpu().numpy().reshape(-1)))  # a permutation


def test_pack_rejects_what_the_decode_kernel_does_not_take():
    lib = abi.load_hip()
    a = torch.zeros(64 * 384, dtype=torch.bfloat16, device="cuda")
    b = torch.empty_like(a)
    assert lib.eilev_stream_layout_pack(P(a), 64, 384, P(b), stream_ptr()) == -2  # k % 256
    assert lib.eilev_stream_layout_pack(P(a), 32, 768, P(b), stream_ptr()) == -2  # k / 256 = 3: no 5 / 8 / 10-step slice
    assert lib.eilev_stream_layout_pack(P(a), 384, 64, P(b), stream_ptr()) == -2  # a test-sized model: the engine keeps the checkpoint layout
    assert lib.eilev_stream_layout_pack(P(a), 64, 256, P(a), stream_ptr()) == -1  # in place
    assert lib.eilev_stream_layout_pack(None, 64, 256, P(b), stream_ptr()) == -1


@pytest.mark.parametrize("B", [17, 20, 32])
def test_decode_step_bit_identical_with_stream_layout(B):
    """One eilev_opt_decode_step at OPT-2.7B widths (one block) on the checkpoint layout and with the stream-layout copies attached:
    identical fp32 logits and ids (q|k|v, fc1, lm_head unsplit; out_proj with 2 and fc2 with 4 K splits)."""
    cfg, _, eng = models("real_1l")
    d = eng.dims
    rng = np.random.default_rng(11)
    L, cap = 40, 44
    ids = torch.from_numpy(rng.integers(4, 50000, size=(B, L + 1)).astype(np.int64)).cuda()
    am = torch.ones((B, L), dtype=torch.int32, device="cuda")
    am[1, :7] = 0
    emb = eng.embed_scatter(ids, None, None)
    kv0 = eng.new_kv_cache(B, cap)
    eng.prefill(emb[:, :L].contiguous(), am, kv_cache=kv0, kv_capacity=cap)
    n_valid = am.sum(dim=1).to(torch.int32).contiguous()
    ws = torch.empty(int(eng.lib.eilev_opt_workspace_bytes(C.byref(d), B, 1)), dtype=torch.uint8, device="cuda")

    def step():
        kv = kv0.clone()
        state = torch.tensor([1, B], dtype=torch.int32, device="cuda")
        tokens = ids[:, L].contiguous()
        finished = torch.zeros(B, dtype=torch.uint8, device="cuda")
        out = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
        logits = torch.empty((B, d.vocab), dtype=torch.float32, device="cuda")
        rc = eng.lib.eilev_opt_decode_step(C.byref(d), C.byref(eng.pack.opt), P(tokens), P(state), P(am), P(n_valid), B, L, P(kv), cap, P(logits),
                                           P(finished), -1, 1, P(out), 4, P(ws), ws.numel(), stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        return logits.cpu().numpy(), out[:, 1].cpu().numpy(), kv

    saved, eng._stream_keep = eng._stream_keep, None
    try:
        abi.attach_opt_stream(eng.pack, None, None)
        plain, plain_ids, kv_a = step()
        assert eng.ensure_stream_layout(B)
        assert eng.pack.opt.layers_stream and eng.pack.opt.lm_head_stream
        packed, packed_ids, kv_b = step()
    finally:
        abi.attach_opt_stream(eng.pack, None, None)
        eng._stream_keep = saved
        eng._dec_cache = None
    assert np.array_equal(plain, packed)
    assert np.array_equal(plain_ids, packed_ids)
    assert torch.equal(kv_a, kv_b)


@pytest.mark.parametrize("B", [17, 20, 32])
def test_decode_step_bit_identical_with_row_block_activations(probes, B):
    """Inside a decode step of 17..32 rows the attention rows, the LayerNorm rows and the fc1 rows travel between the kernels in the
    row-block layout (csrc/common.h frag32_index: what an MFMA operand load of the next GEMV reads is one contiguous kilobyte).  Same
    values, same arithmetic: the step gives the logits, the ids and the cache of the row-major step (probe build switch)."""
    cfg, _, eng = models("real_1l")
    d = eng.dims
    rng = np.random.default_rng(13)
    L, cap = 300, 304  # two key ranges in the attention loop
    ids = torch.from_numpy(rng.integers(4, 50000, size=(B, L + 1)).astype(np.int64)).cuda()
    am = torch.ones((B, L), dtype=torch.int32, device="cuda")
    am[2, :270] = 0
    emb = eng.embed_scatter(ids, None, None)
    kv0 = eng.new_kv_cache(B, cap)
    eng.prefill(emb[:, :L].contiguous(), am, kv_cache=kv0, kv_capacity=cap)
    n_valid = am.sum(dim=1).to(torch.int32).contiguous()
    ws = torch.empty(int(probes.eilev_opt_workspace_bytes(C.byref(d), B, 1)), dtype=torch.uint8, device="cuda")
    res = {}
    try:
        for frag in (0, 1):
            probes.eilev_debug_decode_frag(frag)
            kv = kv0.clone()
            state = torch.tensor([1, B], dtype=torch.int32, device="cuda")
            tokens = ids[:, L].contiguous()
            finished = torch.zeros(B, dtype=torch.uint8, device="cuda")
            out = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
            logits = torch.empty((B, d.vocab), dtype=torch.float32, device="cuda")
            ws.fill_(0x7f)  # stale rows of the 32-row buffers must not matter
            rc = probes.eilev_opt_decode_step(C.byref(d), C.byref(eng.pack.opt), P(tokens), P(state), P(am), P(n_valid), B, L, P(kv), cap, P(logits),
                                              P(finished), -1, 1, P(out), 4, P(ws), ws.numel(), stream_ptr())
            assert rc == 0
            torch.cuda.synchronize()
            res[frag] = (logits.cpu().numpy(), out[:, 1].cpu().numpy(), kv)
    finally:
        probes.eilev_debug_decode_frag(1)
    assert np.isfinite(res[1][0]).all()
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])
    assert torch.equal(res[0][2], res[1][2])


def test_greedy_decode_ids_with_and_without_stream_layout():
    """engine.greedy_decode at 32 rows (hipGraph replay): the lazily packed copies change no id."""
    cfg, _, eng = models("real_1l")
    rng = np.random.default_rng(3)
    B, L = 32, 24
    emb = torch.from_numpy((rng.standard_normal((B, L, eng.dims.t_hidden)) * 0.05).astype(np.float32)).cuda().to(torch.bfloat16)
    am = torch.ones((B, L), dtype=torch.int32, device="cuda")
    am[3, :6] = 0
    saved_flag, saved_keep = eng.decode_stream_layout, eng._stream_keep
    try:
        eng.decode_stream_layout, eng._stream_keep, eng._dec_cache = False, None, None
        abi.attach_opt_stream(eng.pack, None, None)
        a = eng.greedy_decode(emb, am, 6, eos_id=-1).cpu().numpy()
        assert not eng.pack.opt.layers_stream
        eng.decode_stream_layout, eng._stream_keep, eng._dec_cache = True, None, None
        b = eng.greedy_decode(emb, am, 6, eos_id=-1).cpu().numpy()
        assert eng.pack.opt.layers_stream
    finally:
        abi.attach_opt_stream(eng.pack, None, None)
        eng.decode_stream_layout, eng._stream_keep, eng._dec_cache = saved_flag, saved_keep, None
    assert np.array_equal(a, b)
