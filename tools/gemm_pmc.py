import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
M, N, K = 32768, 1024, 2048
a = torch.rand(M, K, device="cuda") - 0.5
w = torch.rand(N, K, device="cuda") - 0.5
out = torch.empty(M, N, device="cuda")
for _ in range(6):
    lib.call("dc_linear_forward", a, K, w, K, M, N, K, out, N, 1)
    torch.mm(a, w.t(), out=out)
torch.cuda.synchronize()
