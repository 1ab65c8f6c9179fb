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
ap.add_argument("--gather", action="store_true", help="the gather-path kernels instead of the product's dispatch (tile plan)")
ap.add_argument("--transposed", action="store_true", help="the backward family (transposed applies + max-aggregation backward)")
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
from deltaconv_amd import _ops                                      # noqa: E402
if a.gather:
    g._tile_plan = False
else:
    g.tile_plan(force_P=64 if k <= 24 else 32)          # the plan the training step uses at this size
    torch.cuda.synchronize()
if a.transposed:
    dcn, dvv, o2c = torch.randn(n, 3 * C, device="cuda"), torch.zeros(2 * n, C, device="cuda"), torch.zeros(n, 2 * C, device="cuda")
    arg = torch.randint(0, k, (n, C), device="cuda").to(torch.uint8)
    if a.gather:
        g._tile_plan_T = False
    g.csc(); grad.coefT(); div.coefT()
    if not a.gather:
        grad.coefTt(); div.coefTt()
    torch.cuda.synchronize()
    for _ in range(a.iters):
        _ops.bwd_div_curl_norm(div, dcn, C, 3 * C, v, C, dvv, C, 1)
        _ops.bwd_apply("hodge", grad, v, C, C, o2c, 2 * C, 1)
        _ops.bwd_grad_sum(grad, v, C, C, x, C, None, 0, y1, C)
        _ops.bwd_knn_max(g, arg, x, C, C, y1, C, 0)
    torch.cuda.synchronize()
    print("done (transposed)", n, k, C)
    sys.exit(0)
for _ in range(a.iters):
    _ops.fwd_apply("grad", grad, x, C, C, y2, C)
    _ops.fwd_apply("div", div, v, C, C, y1, C)
    _ops.fwd_apply("div_curl_norm", div, v, C, C, y3, 3 * C)
torch.cuda.synchronize()
print("done", n, k, C)
