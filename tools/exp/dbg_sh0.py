import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes as C
import torch
from rasterizer.cuda import _call, _ptr, _stream
from gs_fused import spherical_harmonics_split
n = 1000
g = torch.Generator(device="cuda").manual_seed(0)
dc = (torch.rand(n, 3, device="cuda", generator=g) - 0.5) / 0.28209479177387814
rest = torch.zeros(n, 0, 3, device="cuda")
dirs = torch.randn(n, 3, device="cuda", generator=g)
a = spherical_harmonics_split(0, dirs, dc, rest, shift=0.5, clamp_zero=True)
b = torch.empty(n, 3, device="cuda")
_call("gsr_sh_forward_split", C.c_uint(n), C.c_uint(0), C.c_uint(0), _ptr(dirs), _ptr(dc), _ptr(rest), _ptr(b), C.c_float(0.5), C.c_int(1), _stream(dc.device))
torch.cuda.synchronize()
print("equal", torch.equal(a, b), "max diff", (a - b).abs().max().item(), "nan", torch.isnan(b).any().item())
d = (a != b).nonzero()[:5]
print(d, a[a != b][:5], b[a != b][:5], dc[a != b][:5])
