"""Lab: what the statistics epilogue + finaliser cost on top of the plain forward product (graph replay, us per call):
dc_linear_forward vs dc_linear_bn_stats_forward on model shapes.   DELTACONV_HIP_LIB selects a library build."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from deltaconv_amd._lib import lib
from x3_ab import graph_us

dev = "cuda"
print(f"{'M x N x K':20s} {'plain':>8s} {'+stats':>8s} {'delta':>7s}")
for (M, N, K) in [(32768, 1024, 448), (32768, 256, 256), (32768, 128, 256), (65536, 128, 128), (32768, 64, 64)]:
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    y, coef = torch.empty(M, N, device=dev), torch.empty(4, N, device=dev)
    g = torch.ones(N, device=dev)
    nb = lib.raw("dc_linear_stats_workspace_bytes")(M, N, K, 0)
    ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
    t0 = graph_us(lambda: lib.call("dc_linear_forward", x, K, w, K, M, N, K, y, N, 0))
    t1 = graph_us(lambda: lib.call("dc_linear_bn_stats_forward", x, K, w, K, M, N, K, y, N, g, g, 1e-5, 0.1, None, None,
                                   coef[0], coef[1], coef[2], coef[3], 0, ws, nb))
    print(f"{M:6d}x{N:5d}x{K:4d}   {t0:8.1f} {t1:8.1f} {t1 - t0:7.1f}", flush=True)
