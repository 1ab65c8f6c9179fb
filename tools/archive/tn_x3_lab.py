"""Lab: weight-gradient product dW = dY^T X per forced tile (option 5: 1..4 = 128x128, 128x64, 64x64, 64x128; 0 = plan) with the exact
chain (option 3 = 1) and the split products on every tile (option 3 = 2), us per call incl. the slab reduction (graph replay)."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from deltaconv_amd._lib import lib
from x3_ab import graph_us

opt = lib.raw("dc_set_option")
dev = "cuda"
print(f"{'R x M x N':20s} " + " ".join(f"{('ex' if e == 1 else 'sp') + str(t):>7s}" for e in (1, 2) for t in (0, 1, 2, 3, 4)))
for (R, M, N) in [(32768, 128, 256), (32768, 128, 128), (65536, 128, 384), (32768, 64, 256), (32768, 256, 128), (65536, 64, 128), (32768, 256, 512)]:
    a, b = torch.randn(R, M, device=dev), torch.randn(R, N, device=dev)
    c = torch.empty(M, N, device=dev)
    row = []
    for e in (1, 2):
        for t in (0, 1, 2, 3, 4):
            opt(3, e); opt(5, t)
            nb = lib.raw("dc_gemm_tn_workspace_bytes")(R, M, N)
            ws = torch.empty((nb + 3) // 4, device=dev)
            row.append(graph_us(lambda: lib.call("dc_gemm_tn", a, M, b, N, R, M, N, c, N, 0, ws, ws.numel() * 4)))
    opt(3, 0); opt(5, 0)
    print(f"{R:6d}x{M:4d}x{N:4d}     " + " ".join(f"{v:7.1f}" for v in row), flush=True)
