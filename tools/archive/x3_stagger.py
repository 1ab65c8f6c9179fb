"""Sweep of the first-round phase shift (option 4, percent of a K loop; -1 = off) on the shapes with >= 2 rounds of 128 x 128 workgroups:
time per launch under graph replay, forward and input-gradient products, split and exact."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
from x3_ab import graph_us  # noqa (runs nothing at import besides definitions?)

opt = lib.raw("dc_set_option")
dev = "cuda"
vals = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "-1,25,50,75,100,150").split(",")]
print("# us per launch; columns = phase shift in percent of a K loop")
print(f"{'M x N x K':20s} {'product':6s} {'path':6s} " + " ".join(f"{v:7d}" for v in vals))
torch.manual_seed(0)
for (M, N, K) in [(32768, 1024, 448), (32768, 1024, 512), (32768, 512, 256), (65536, 256, 256), (65536, 256, 384)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    dy = torch.randn(M, N, device=dev)
    y, dx = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
    fns = {"fwd": lambda: lib.call("dc_linear_forward", x, K, w, K, M, N, K, y, N, 0),
           "dX": lambda: lib.call("dc_linear_backward_input", dy, N, w, K, M, N, K, dx, K, 0, 0)}
    for name, fn in fns.items():
        for exact in (0, 1):
            opt(3, exact)
            row = []
            for v in vals:
                opt(4, v)
                row.append(graph_us(fn))
            print(f"{M:6d}x{N:5d}x{K:4d}   {name:6s} {'exact' if exact else 'split':6s} " + " ".join(f"{t:7.1f}" for t in row), flush=True)
opt(3, 0); opt(4, 0)
