"""dc_knn_max_affine_tiled (BatchNorm + activation folded into the max aggregation) at the C2 shape: us per launch under graph
replay for C = 64 / 128 / 256, and bit-identity (values + slots) against the gather kernel dc_knn_max_affine, with negative
scales and overflowing tiles included.   python tools/knnmax_affine_ab.py"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc  # noqa: F401
from deltaconv_amd._lib import lib
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.geometry.graph import Graph
from deltaconv_amd.models.deltanet_base import _ptr_info

b = synthetic_batch(32, 1024, seed=7).to("cuda")
g = Graph.knn(b.pos, 20, ptr_info=_ptr_info(b))
plan = g.tile_plan()
n, k = g.n, g.k
for C in (64, 128, 256):
    torch.manual_seed(C)
    h = torch.randn(n, C, device="cuda")
    scale = torch.randn(C, device="cuda")          # both signs
    shift = torch.randn(C, device="cuda")
    outs = []
    for tiled in (False, True):
        o = torch.empty(n, C, device="cuda"); a = torch.empty(n, C, dtype=torch.uint8, device="cuda")
        if tiled:
            lib.call("dc_knn_max_affine_tiled", plan.blob, g.nbr, *plan.args, h, C, C, scale, shift, 0.2, o, C, a)
        else:
            lib.call("dc_knn_max_affine", g.nbr, n, k, h, C, C, scale, shift, 0.2, o, C, a)
        outs.append((o, a))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    o, a = outs[1]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            lib.call("dc_knn_max_affine_tiled", plan.blob, g.nbr, *plan.args, h, C, C, scale, shift, 0.2, o, C, a)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(50):
            lib.call("dc_knn_max_affine_tiled", plan.blob, g.nbr, *plan.args, h, C, C, scale, shift, 0.2, o, C, a)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"C={C}: tiled == gather: {same}; {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch")
