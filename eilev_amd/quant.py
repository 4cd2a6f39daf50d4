"""fp8 (OCP e4m3, torch.float8_e4m3fn) weight-only quantisation with one fp32 scale per output channel — the weight format of
``eilev_linear_w8`` (include/eilev.h; BASELINE configs[4] "fp8 MFMA weights").

    scale[n] = max_k |W[n, k]| / 448          (448 = largest finite e4m3 value; an all-zero row gets scale 1)
    Wq[n, k] = e4m3(W[n, k] / scale[n])       (round to nearest even — torch's cast)
    W ≈ Wq * scale[n]

The linear layer then computes ``(x @ Wq.T) * scale + bias``: the scale is factored out of the sum, so the kernel
multiplies bf16 activations with exactly-represented weights and scales the fp32 sum once per output.
"""
from __future__ import annotations

E4M3_MAX = 448.0


def quantize_e4m3_per_channel(weight):
    """weight: floating tensor (N, K) -> (uint8 tensor (N, K) holding the e4m3 bytes, float32 scales (N,))."""
    import torch

    w = weight.detach().to(torch.float32)
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
    q = (w / scale[:, None]).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale.contiguous()


def dequantize(q_bytes, scale):
    """float32 (N, K) = e4m3(q_bytes) * scale[:, None] (the matrix the quantised layer effectively multiplies by)."""
    import torch

    return q_bytes.view(torch.float8_e4m3fn).to(torch.float32) * scale[:, None]


def linear_w8(x, q_bytes, scale, bias=None, residual=None, epilogue: int = 0, out_dtype=None, lib=None):
    """``epilogue((x @ dq(Wq).T) * scale + bias) (+ residual)`` on the GPU through ``eilev_linear_w8``.  x: bf16 (M, K) CUDA."""
    import ctypes as C

    import torch

    from . import abi

    if not x.is_cuda or x.dtype != torch.bfloat16:
        raise RuntimeError("linear_w8 runs on the GPU (HIP library) with bf16 activations")
    lib = lib or abi.load_hip()
    m, k = x.shape
    n = q_bytes.shape[0]
    out_f32 = out_dtype == torch.float32
    out = torch.empty((m, n), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    nb = lib.eilev_linear_w8_scratch_bytes(m, n, k)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=x.device)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    abi.check(lib.eilev_linear_w8(p(x.contiguous()), p(q_bytes), p(scale), p(bias), p(residual), p(out), m, n, k, epilogue, int(out_f32),
                                  p(ws), nb, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "eilev_linear_w8")
    return out
