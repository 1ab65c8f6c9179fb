"""Debug: one [Linear -> BatchNorm -> LeakyReLU] block (fused node) fwd/bwd vs fp64, at the shapes of the C4 2-cloud head."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deltaconv_amd as dc
from deltaconv_amd.nn import MLP
def rel(a, c): return float((a.double().cpu() - c.double().cpu()).abs().max() / c.double().abs().max())
for R in (4096, 8192, 2048, 4097):
    for scale in (1.0, 1e-4):
        torch.manual_seed(R)
        mlp = MLP([256, 256]).cuda().train()
        blk = mlp[0]
        with torch.no_grad():
            blk[1].bn.weight.copy_(torch.randn(256)); blk[1].bn.bias.copy_(torch.randn(256))
        x = torch.randn(R, 256, device="cuda", requires_grad=True)
        g = torch.randn(R, 256, device="cuda") * scale
        y = mlp(x); y.backward(g)
        x64 = x.detach().double().cpu().requires_grad_(True)
        W = blk[0].weight.detach().double().cpu().requires_grad_(True)
        ga = blk[1].bn.weight.detach().double().cpu().requires_grad_(True); be = blk[1].bn.bias.detach().double().cpu().requires_grad_(True)
        h = x64 @ W.t()
        z = torch.nn.functional.batch_norm(h, None, None, ga, be, True, 0.1, 1e-5)
        y64 = torch.nn.functional.leaky_relu(z, 0.2); y64.backward(g.double().cpu())
        print(R, scale, "y", f"{rel(y, y64):.1e}", "dx", f"{rel(x.grad, x64.grad):.1e}", "dW", f"{rel(blk[0].weight.grad, W.grad):.1e}",
              "dgamma", f"{rel(blk[1].bn.weight.grad, ga.grad):.1e}", "dbeta", f"{rel(blk[1].bn.bias.grad, be.grad):.1e}")
