"""Lab: forward / input-gradient products per forced tile (tile argument 0 = policy, 1..4 = 128x128, 128x64, 64x64, 64x128), exact chain
(option 3 = 1) vs split products on every tile (option 3 = 2), us per call under graph replay."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from deltaconv_amd._lib import lib
from x3_ab import graph_us

opt = lib.raw("dc_set_option")
dev = "cuda"
print(f"{'M x N x K':20s} {'prod':4s} " + " ".join(f"{('ex' if e == 1 else 'sp') + str(t):>7s}" for e in (1, 2) for t in (0, 1, 2, 3, 4)))
for (M, N, K) in [(32768, 128, 256), (32768, 128, 128), (32768, 256, 128), (65536, 128, 384), (32768, 64, 256), (32768, 256, 64), (65536, 128, 64)]:
    x, w, dy = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
    y, dx = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
    for name in ("fwd", "dX"):
        row = []
        for e in (1, 2):
            for t in (0, 1, 2, 3, 4):
                opt(3, e)
                if name == "fwd":
                    row.append(graph_us(lambda: lib.call("dc_linear_forward", x, K, w, K, M, N, K, y, N, t)))
                else:
                    row.append(graph_us(lambda: lib.call("dc_linear_backward_input", dy, N, w, K, M, N, K, dx, K, 0, t)))
        opt(3, 0)
        print(f"{M:6d}x{N:5d}x{K:4d}   {name:4s} " + " ".join(f"{v:7.1f}" for v in row), flush=True)
