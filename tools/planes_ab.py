"""A/B per product: weight operand from pre-split bf16 planes (dc_presplit_weights + BP kernels) vs the in-loop split / exact
chain of round 3 (option 9 = 1); us per launch under HIP-graph replay (20 launches per graph)."""
import os, sys, math
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
from deltaconv_amd.nn import fused
from tools.x3_ab import graph_us
opt = lib.raw("dc_set_option")
shapes = [(32768, 1024, 448), (32768, 1024, 512), (32768, 256, 512), (65536, 256, 256), (32768, 128, 256), (65536, 128, 384),
          (32768, 128, 128), (32768, 64, 256), (32768, 64, 64), (65536, 64, 128), (32768, 448, 1024)]
print(f"{'M x N x K':22s} {'product':6s} {'no planes':>10s} {'planes':>10s} {'ratio':>6s}")
for (M, N, K) in shapes:
    fused._planes_reset()
    x, dy = torch.randn(M, K, device="cuda"), torch.randn(M, N, device="cuda")
    w = torch.nn.Parameter(torch.randn(N, K, device="cuda") / math.sqrt(K))
    with torch.no_grad():
        fused.mm_nt(x, w); fused.mm_nn(dy, w)
        fused.presplit_begin()
        for name, fn in (("fwd", lambda: fused.mm_nt(x, w)), ("dX", lambda: fused.mm_nn(dy, w))):
            t = {}
            for v in (1, 0):
                opt(9, v)
                t[v] = graph_us(fn)
            opt(9, 0)
            print(f"{M:6d}x{N:5d}x{K:5d}   {name:6s} {t[1]:10.1f} {t[0]:10.1f} {t[1] / t[0]:6.2f}", flush=True)
