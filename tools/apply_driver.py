"""Launch the ELL apply kernels a fixed number of times (for rocprofv3 --pmc / --kernel-trace runs)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib                                   # noqa: E402
from deltaconv_amd.data import synthetic_batch                       # noqa: E402
from deltaconv_amd.geometry import Graph, build_grad_div, build_tangent_basis  # noqa: E402
import deltaconv_amd as dc                                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--points", type=int, default=1024)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--C", type=int, default=64)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--opt", type=int, nargs=2, action="append", default=[])
a = ap.parse_args()
for key, val in a.opt:
    lib.raw("dc_set_option")(key, val)
b = synthetic_batch(a.batch, a.points, seed=7).to("cuda")
info = dc.models.deltanet_base._ptr_info(b)
g = Graph.knn(b.pos, a.k, ptr_info=info)
xb, yb = build_tangent_basis(b.norm)
grad, div = build_grad_div(b.pos, b.norm, xb, yb, g, b.batch)
n, k, C = g.n, g.k, a.C
x = torch.randn(n, C, device="cuda")
v = torch.randn(2 * n, C, device="cuda")
y2 = torch.empty(2 * n, C, device="cuda")
y1 = torch.empty(n, C, device="cuda")
y3 = torch.empty(n, 3 * C, device="cuda")
torch.cuda.synchronize()
for _ in range(a.iters):
    lib.call("dc_apply_grad", grad.coef, g.nbr, n, k, x, C, C, y2, C)
    lib.call("dc_apply_div", div.coef, g.nbr, n, k, v, C, C, y1, C)
    lib.call("dc_apply_div_curl_norm", div.coef, g.nbr, n, k, v, C, C, y3, 3 * C)
torch.cuda.synchronize()
print("done", n, k, C)
