"""A/B of the max-aggregation backward: edge-at-a-time generic transposed skeleton (option 1 = 1) vs the batched
kernel, on the bench's graph.  us per launch from a captured HIP graph; bit-equality of the two results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.geometry import Graph
from tools.gemm_lab import timeit

b = synthetic_batch(32, 1024, seed=7).to("cuda")
g = Graph.knn(b.pos, 20, b.batch)
tptr, tedge = g.csc()
n, k = g.n, g.k
for C in (64, 128, 256):
    h = torch.randn(n, C, device="cuda")
    out = torch.empty(n, C, device="cuda")
    arg = torch.empty(n, C, dtype=torch.uint8, device="cuda")
    lib.call("dc_knn_max", g.nbr, n, k, h, C, C, out, C, arg)
    dout = torch.randn(n, C, device="cuda")
    res, ts = {}, {}
    for opt in (1, 0):
        lib.raw("dc_set_option")(1, opt)
        dh = torch.empty(n, C, device="cuda")
        ts[opt] = timeit(lambda: lib.call("dc_knn_max_backward", tptr, tedge, n, k, arg, dout, C, C, dh, C, 0))
        res[opt] = dh.clone()
    nbytes = 9 * C * n + 4 * n * k
    print(f"knn_max_backward C={C}: generic {ts[1]:7.1f} us ({nbytes / ts[1] / 1e3:7.1f} GB/s)  batched {ts[0]:7.1f} us "
          f"({nbytes / ts[0] / 1e3:7.1f} GB/s)  bit-equal {torch.equal(res[0], res[1])}")
