"""Debug: the tail of the segmentation head (Linear(256,128)+bias -> LeakyReLU -> Linear(128,50)+bias) fwd/bwd against fp64."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deltaconv_amd as dc
from deltaconv_amd.nn import fused
from deltaconv_amd.nn.mlp import Linear
dev = "cuda"
def rel(a, b):
    b = b.to(a.device).double(); a = a.double()
    return float((a - b).abs().max() / b.abs().max())
torch.manual_seed(0)
for M in (4096, 8192):
    x = torch.randn(M, 256, device=dev)
    l4, l6 = Linear(256, 128).to(dev), Linear(128, 50).to(dev)
    act = torch.nn.LeakyReLU(0.2)
    xr = x.clone().requires_grad_(True)
    y = l6(act(l4(xr)))
    g = torch.randn_like(y)
    y.backward(g)
    x64 = x.double().requires_grad_(True)
    w4, b4, w6, b6 = (t.detach().double().requires_grad_(True) for t in (l4.weight, l4.bias, l6.weight, l6.bias))
    y64 = torch.nn.functional.linear(torch.nn.functional.leaky_relu(torch.nn.functional.linear(x64, w4, b4), 0.2), w6, b6)
    y64.backward(g.double())
    print(M, "fwd", rel(y, y64), "dx", rel(xr.grad, x64.grad), "dW4", rel(l4.weight.grad, w4.grad), "db4", rel(l4.bias.grad, b4.grad),
          "dW6", rel(l6.weight.grad, w6.grad), "db6", rel(l6.bias.grad, b6.grad))
    dy = torch.randn(M, 50, device=dev)
    w = torch.randn(50, 128, device=dev) / math.sqrt(128)
    print("   mm_nn [M,50]x[50,128]", rel(fused.mm_nn(dy, w), dy.double() @ w.double()),
          "gemm_tn [M,50]^T[M,128]", rel(fused.gemm_tn(dy, x[:, :128].contiguous()), dy.double().t() @ x[:, :128].double()),
          "gemm_tn [M,128]^T[M,256]", rel(fused.gemm_tn(x[:, :128].contiguous(), x), x[:, :128].double().t() @ x.double()))
