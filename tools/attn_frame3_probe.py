#!/usr/bin/env python
"""GPU probe: attn_frame3_kernel (probe flag 16 forces it) against attn_frame_kernel (flag 32 forces that one) on the ViT shape."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from eilev_amd import abi

abi.use_probes()  # the eilev_debug_* switches live in the probe build only (libeilev_hip_probes.so)

lib = abi.load_hip()
raw = C.CDLL(abi.HIP_LIB_PATH)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
h, sq, hd = 16, 257, 88
D = h * hd
batches = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2", "17", "100", "544", "1088"])]
reps = int(os.environ.get("REPS", "10"))
for b in batches:
    torch.manual_seed(b)
    qkv = (torch.randn(b, sq, 3 * D, device="cuda") * 1.5).to(torch.bfloat16)
    outs, times = {}, {}
    modes = [64] + [int(x) for x in os.environ.get("MODES", "32").split(",")]  # mode = probe flags << 1: 64 = old kernel, 32 = new
    for mode in modes:
        raw.eilev_debug_attn_v1(mode)
        o = torch.full((b, sq, D), float("nan"), device="cuda", dtype=torch.bfloat16)
        call = lambda: lib.eilev_attention(C.c_void_p(qkv.data_ptr()), C.c_void_p(qkv.data_ptr() + 2 * D), C.c_void_p(qkv.data_ptr() + 4 * D),
                                           C.c_void_p(o.data_ptr()), b, h, sq, sq, hd, 3 * D, 3 * D, 3 * D, hd ** -0.5, 0, None, st())
        for _ in range(2):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        times[mode] = e0.elapsed_time(e1) / reps * 1e3
        outs[mode] = o.float().clone()
    raw.eilev_debug_attn_v1(0)
    # fp32 reference on a few frames
    nb = min(b, 4)
    q, k, v = [t.float().view(nb, sq, h, hd).transpose(1, 2) for t in qkv[:nb].split(D, dim=-1)]
    ref = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v
    ref = ref.transpose(1, 2).reshape(nb, sq, D)
    e_old = (outs[64][:nb] - ref).abs().max().item()
    for m in modes[1:]:
        d_new = (outs[m] - outs[64]).abs().max().item()
        e_new = (outs[m][:nb] - ref).abs().max().item()
        e_cls = (outs[m][:nb, sq - 1] - ref[:, sq - 1]).abs().max().item()
        nan = int(torch.isnan(outs[m]).sum().item())
        print(f"frames {b:5d}: frame {times[64]:8.1f} us  frame3[{m}] {times[m]:8.1f} us  ({times[64] / times[m]:.3f}x)   max|new-old| {d_new:.4g}  "
              f"err vs fp32: old {e_old:.4g} new {e_new:.4g} (row 256: {e_cls:.4g})  nan {nan}", flush=True)
