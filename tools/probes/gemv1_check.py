"""GPU probe: eilev_linear_rows at m = 1 (gemv1_kernel) against torch on the decode shapes: which outputs are wrong / unwritten."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from eilev_amd import abi
lib = abi.load_hip()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
torch.manual_seed(0)
for (n, k, ln, epi, resid, f32) in [(7680, 2560, True, 0, False, False), (2560, 2560, False, 0, True, False), (10240, 2560, True, 2, False, False),
                                    (2560, 10240, False, 0, True, False), (50272, 2560, True, 0, False, True), (4096, 16384, False, 0, True, False)]:
    x = (torch.randn(1, k, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)
    g = (torch.randn(k, device="cuda") * 0.3 + 1).to(torch.bfloat16) if ln else None
    b = (torch.randn(k, device="cuda") * 0.2).to(torch.bfloat16) if ln else None
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    bias = (torch.randn(n, device="cuda") * 0.5).to(torch.bfloat16)
    r = (torch.randn(1, n, device="cuda") * 2).to(torch.bfloat16) if resid else None
    out = torch.full((1, n), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    for rep in range(3):
        out.fill_(float("nan"))
        rc = lib.eilev_linear_rows(P(x), P(g), P(b), C.c_float(1e-5), P(w), P(bias), P(r), P(out), 1, n, k, epi, int(f32), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        xn = torch.nn.functional.layer_norm(x.float(), (k,), g.float(), b.float(), 1e-5).to(torch.bfloat16).float() if ln else x.float()
        ref = xn @ w.float().T + bias.float()
        if epi == 2: ref = torch.relu(ref)
        if resid: ref = ref + r.float()
        got = out.float()
        bad = (~torch.isfinite(got)) | ((got - ref).abs() > 0.05 * ref.abs().max())
        idx = torch.nonzero(bad[0]).flatten().tolist()
        print(f"n={n} k={k} ln={ln} rc={rc} rep={rep}: bad {len(idx)} of {n}; first {idx[:24]}", flush=True)
