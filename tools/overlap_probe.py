#!/usr/bin/env python
"""GPU probe: can the HBM-bound greedy decode of step i run UNDER the MFMA-bound ViT encode of step i + 1?

The persistent GEMMs own every CU's LDS and registers, so two plain streams serialise (DESIGN history, round 1).  Here the decode gets
its own CUs: `hipExtStreamCreateWithCUMask` gives the language-model stream k CUs (k / 8 per XCD) and the encode stream the other 256 - k,
and the persistent kernels size their grids for 256 - k (probe switch eilev_debug_grid_cus).  Measured, one process, one box:
  A  the four ViT GEMMs of a block (folded-LayerNorm forms, M = 279 616) in a loop, whole chip
  B  the batch-32 OPT-2.7B decode (hipGraph, 31 tokens after a 32 x 960 prefill), whole chip
  C  A on its 256 - k CUs alone,  D  B on its k CUs alone,  E  C and D at the same time
and from them what a software-pipelined step (decode of step i under the encode of step i + 1) would take.

    python tools/overlap_probe.py [k ...]        (default 8 16)
"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from eilev_amd import abi

abi.use_probes()
lib = abi.load_hip()
raw = C.CDLL(abi.HIP_LIB_PATH)
HERE = os.path.join(ROOT, "tools", "probes")
so = os.path.join(HERE, "libcu_census.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "cu_census.hip")])
cen = C.CDLL(so)
dev = torch.device("cuda", 0)
torch.cuda.init()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (NCU + 31) // 32
    m = (C.c_uint32 * words)()
    for b in bits:
        m[b // 32] |= 1 << (b % 32)
    h = C.c_void_p()
    rc = cen.masked_stream_create(C.byref(h), m, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask: {rc}"
    return torch.cuda.ExternalStream(h.value, device=dev)


def census(stream, blocks=512):
    out = torch.zeros(blocks * 2, dtype=torch.int32, device=dev)
    # 1024-thread workgroups with 64 KiB of LDS: at most two per CU, so 512 of them cover every CU the stream may use
    assert cen.census_launch(P(out), blocks, 1024, 65536, 3000, C.c_void_p(stream.cuda_stream)) == 0
    stream.synchronize()
    o = out.cpu().numpy().reshape(blocks, 2).astype(np.int64)
    xcc, hw = o[:, 0] & 15, o[:, 1]
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    ids = sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
    per = [sum(1 for i in ids if i[0] == x) for x in range(8)]
    return len(ids), per


ks = [int(x) for x in sys.argv[1:]] or [8, 16]
full = torch.cuda.Stream(dev)
n, per = census(full)
print(f"census, unmasked stream: {n} distinct CUs, per XCC {per}", flush=True)
# which mask bits are which CUs?  bits 0..7 / bits 0, 32, 64 .. / bits 0..31
for name, bits in (("bits 0-7", range(8)), ("bits 0,32,..,224", range(0, 256, 32)), ("bits 0-31", range(32)), ("bits 0,8,16,..,248", range(0, 256, 8))):
    n, per = census(masked_stream(list(bits)))
    print(f"census, mask {name}: {n} distinct CUs, per XCC {per}", flush=True)

# ---- workloads ----------------------------------------------------------------------------------------------------------------------
M = 279616
SH = {"fc1_ln": (6144, 1408, 1), "qkv_ln": (4224, 1408, 0), "fc2_st": (1408, 6144, 0), "proj_st": (1408, 1408, 0)}
bufs = {}
for nm, (n_, k_, epi) in SH.items():
    a = torch.randn(M, k_, device=dev).to(torch.bfloat16)
    w = (torch.randn(n_, k_, device=dev) / k_ ** 0.5).to(torch.bfloat16)
    b = torch.randn(n_, device=dev).to(torch.bfloat16)
    o = torch.empty(M, n_, device=dev, dtype=torch.bfloat16)
    if nm.endswith("_ln"):
        cs = torch.randn(n_, device=dev)
        rows = torch.stack([torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.1], 1).contiguous()
        bufs[nm] = (a, w, b, cs, rows, o, n_, k_, epi)
    else:
        r = torch.randn(M, n_, device=dev).to(torch.bfloat16)
        stats = torch.empty(((n_ + 63) // 64, M, 2), device=dev)
        bufs[nm] = (a, w, b, r, stats, o, n_, k_, epi)
FLOP_BLOCK = sum(2.0 * M * n_ * k_ for n_, k_, _ in SH.values())


def gemm_block(sp):
    for nm, t in bufs.items():
        if nm.endswith("_ln"):
            a, w, b, cs, rows, o, n_, k_, epi = t
            assert lib.eilev_linear_lnfold(P(a), P(w), P(b), P(cs), P(rows), P(o), M, n_, k_, epi, sp) == 0
        else:
            a, w, b, r, stats, o, n_, k_, epi = t
            assert lib.eilev_linear_stats(P(a), P(w), P(b), P(r), P(o), M, n_, k_, P(stats), sp) == 0


def gemm_loop(stream, blocks):
    sp = C.c_void_p(stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record()
        for _ in range(blocks):
            gemm_block(sp)
        e1.record()
    return e0, e1


from bench import build_inputs, random_weights  # noqa: E402  (the bench's own synthetic model and inputs)
from eilev_amd.configs import blip2_config  # noqa: E402
from eilev_amd.engine import HipEngine  # noqa: E402

cfg = blip2_config("opt27")
eng = HipEngine(cfg, random_weights(cfg, dev), device=dev, parts=("opt",))
S, L, D = 32, 960, cfg.text_config.hidden_size
emb = (0.5 * torch.randn(S, L, D, device=dev)).to(torch.bfloat16)
am = torch.ones(S, L, dtype=torch.int64, device=dev)


def decode(stream):
    """greedy_decode on `stream`; returns (prefill_done event, end event)."""
    eng.timing = []
    with torch.cuda.stream(stream):
        eng.greedy_decode(emb, am, 32, eos_id=-1, pad_id=1, use_graph=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
    ev = [e for nme, e in eng.timing if nme == "prefill_done"][-1]
    eng.timing = None
    return ev, e1


def ms(e0, e1):
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


raw.eilev_debug_grid_cus(0)
for _ in range(2):
    ms(*gemm_loop(full, 2))
    ms(*decode(full))
A = ms(*gemm_loop(full, 20)) / 20
B = ms(*decode(full)) / 31
print(f"A  GEMMs of one ViT block, whole chip: {A:.3f} ms ({FLOP_BLOCK / A / 1e9:.0f} TFLOP/s)   B  decode, whole chip: {B:.3f} ms/token", flush=True)
for k in ks:
    per_x = k // 8
    # CU mask bit b -> (XCC b % 8, CU b / 8) on this part (the census above shows it): the decode gets the LAST per_x CUs of every XCC
    dec_bits = [b for b in range(NCU) if (b // 8) >= NCU // 8 - per_x]
    enc_bits = [b for b in range(NCU) if b not in set(dec_bits)]
    sd, se_ = masked_stream(dec_bits), masked_stream(enc_bits)
    nd, pd = census(sd)
    ne, pe = census(se_)
    print(f"k = {k}: decode stream {nd} CUs {pd}, encode stream {ne} CUs {pe}", flush=True)
    raw.eilev_debug_grid_cus(NCU - k)
    ms(*gemm_loop(se_, 2))
    Cc = ms(*gemm_loop(se_, 20)) / 20
    raw.eilev_debug_grid_cus(0)
    ms(*decode(sd))
    Dd = ms(*decode(sd)) / 31
    blocks = max(20, int(1.3 * Dd * 31 / Cc) + 4)  # the GEMM loop outlasts the decode
    raw.eilev_debug_grid_cus(NCU - k)
    g0, g1 = gemm_loop(se_, blocks)
    raw_dec = decode(sd)  # (prefill on the decode stream first: it also runs on the k CUs — only the decode part is read)
    torch.cuda.synchronize()
    raw.eilev_debug_grid_cus(0)
    Ec, Ed = g0.elapsed_time(g1) / blocks, raw_dec[0].elapsed_time(raw_dec[1]) / 31
    enc_ms = 2133.0  # encode of a bench step (ViT + Q-Former), whole chip, ms (profiles/r05_bench_mid.json)
    step = 2437.0
    dec_ms, pre_ms = 31 * B, step - enc_ms - 31 * B
    piped = max(enc_ms * Ec / A, 31 * Ed) + pre_ms
    print(f"k = {k}: C  GEMMs on {NCU - k} CUs alone {Cc:.3f} ms ({Cc / A:.3f} x A)   D  decode on {k} CUs alone {Dd:.3f} ms/token ({Dd / B:.2f} x B)   "
          f"E  together: GEMMs {Ec:.3f} ms ({Ec / A:.3f} x A), decode {Ed:.3f} ms/token ({Ed / B:.2f} x B; {31 * Ed:.0f} ms for 31 tokens over {blocks} GEMM blocks)", flush=True)
    print(f"        pipelined step estimate: max(encode {enc_ms:.0f} x {Ec / A:.3f}, decode {31 * Ed:.0f}) + prefill {pre_ms:.0f} = {piped:.0f} ms vs {step:.0f} ms serial "
          f"-> {step / piped:.3f} x", flush=True)
