import ctypes as C, sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/eilev_amd") else os.getcwd())
import torch
from eilev_amd import abi
lib = abi.load_hip()
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for (m, n, k, resid) in [(279616 - 192, 1408, 1408, True), (30720 + 64, 2560, 2560, False), (279616 + 64, 4224, 1408, False), (65536 + 16, 6144, 1408, False)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    extra = 512
    buf = torch.full((m + extra, n), 7.0, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(m, n, device="cuda", dtype=torch.bfloat16) if resid else None
    rc = lib.eilev_linear(P(a), P(w), P(b), P(r) if resid else None, P(buf), m, n, k, 0, 0, st)
    torch.cuda.synchronize()
    tail = buf[m:]
    bad = int((tail != 7.0).sum().item())
    print(f"M={m} N={n} K={k} resid={resid}: rc={rc}, canary elements overwritten: {bad}", flush=True)
