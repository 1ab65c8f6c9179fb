"""Launch the forward / input-gradient / weight-gradient products on three model shapes (for rocprofv3 --pmc, tools/gpu_x3_pmc.sh).
DC_GEMM_EXACT=1 selects the exact fp32 chain."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
for (M, N, K) in [(32768, 1024, 448), (65536, 256, 384), (32768, 128, 256)]:
    x = torch.rand(M, K, device="cuda") - 0.5
    w = torch.rand(N, K, device="cuda") - 0.5
    dy = torch.rand(M, N, device="cuda") - 0.5
    y, dx, dw = torch.empty(M, N, device="cuda"), torch.empty(M, K, device="cuda"), torch.empty(N, K, device="cuda")
    wsb = lib.raw("dc_gemm_tn_workspace_bytes")(M, N, K)
    ws = torch.empty(wsb // 4 + 16, device="cuda")
    for _ in range(4):
        lib.call("dc_linear_forward", x, K, w, K, M, N, K, y, N, 0)
        lib.call("dc_linear_backward_input", dy, N, w, K, M, N, K, dx, K, 0, 0)
        lib.call("dc_gemm_tn", dy, N, x, K, M, N, K, dw, K, 0, ws, wsb)
    torch.cuda.synchronize()
