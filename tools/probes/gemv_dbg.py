import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from eilev_amd import abi
lib = abi.load_hip()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
m, n, k = 1, 16, 512
for name, x, w in (("ones", torch.ones(m, k), torch.ones(n, k)),
                   ("x=arange/512,w=1", (torch.arange(k).float() / 512).repeat(m, 1), torch.ones(n, k)),
                   ("x=1,w=row index", torch.ones(m, k), torch.arange(n).float()[:, None].repeat(1, k)),
                   ("x=e0", torch.nn.functional.one_hot(torch.tensor([0]), k).float(), torch.arange(k).float()[None].repeat(n, 1) / 64),
                   ("x=e9", torch.nn.functional.one_hot(torch.tensor([9]), k).float(), torch.arange(k).float()[None].repeat(n, 1) / 64)):
    xd, wd = x.cuda().bfloat16().contiguous(), w.cuda().bfloat16().contiguous()
    out = torch.full((m, n), float("nan"), device="cuda")
    rc = lib.eilev_linear_rows(P(xd), None, None, C.c_float(1e-5), P(wd), None, None, P(out), m, n, k, 0, 1, st())
    torch.cuda.synchronize()
    print(name, rc, out[0].tolist(), "expected", (xd.float() @ wd.float().t())[0].tolist())
