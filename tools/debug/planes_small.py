import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deltaconv_amd._lib import lib
from deltaconv_amd.nn import fused
opt = lib.raw("dc_set_option")
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
for (M, N, K) in [(1024, 64, 256), (1024, 128, 256), (2048, 128, 384), (1024, 256, 64), (2048, 128, 192), (1024, 64, 64), (512, 64, 64), (1024, 256, 512)]:
    fused._planes_reset()
    g = torch.Generator().manual_seed(M + N + K)
    x, dy, h = torch.randn(M, K, generator=g).cuda(), torch.randn(M, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    coefs = torch.randn(5 * N, generator=g).cuda()
    w = torch.nn.Parameter((torch.randn(N, K, generator=g) / math.sqrt(K)).cuda())
    def run():
        with torch.no_grad():
            y, dx = fused.mm_nt(x, w), fused.mm_nn(dy, w)
            dxp = torch.empty(M, K, device="cuda")
            fused._hint_planes(w, True)
            lib.call("dc_linear_bn_backward_input", dy, N, h, N, coefs, 0.2, w, K, M, N, K, dxp, K, 0, 0)
            acc = torch.ones(M, K + 8, device="cuda")
            fused.mm_nn(dy, w, out=acc[:, 4:4 + K], accumulate=True)
        return y, dx, dxp, acc
    opt(9, 1); a = run(); opt(9, 0); b = run()
    ry, rdx = x.double() @ w.double().t(), dy.double() @ w.double()
    print((M, N, K), "fwd", f"{rel(a[0], ry):.1e} {rel(b[0], ry):.1e}", "dX", f"{rel(a[1], rdx):.1e} {rel(b[1], rdx):.1e}",
          "dX-pro a-vs-b", f"{rel(b[2], a[2]):.1e}", "acc a-vs-b", f"{rel(b[3], a[3]):.1e}")
