"""Launch the hand-written forward GEMM (and the vendor library for the same product) on the MODEL's shapes a fixed number of
times, for rocprofv3 --pmc runs (tools/gpu_pmc_gemm.sh).  Shapes: embedding 32768 x 1024 x 512, C5's 65536 x 256 x 384 family,
the single-wave 32768 x 128 x 256, and the K = 2048 lab shape of round 2 for continuity."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
for (M, N, K) in [(32768, 1024, 512), (65536, 256, 384), (32768, 128, 256), (32768, 1024, 2048)]:
    a = torch.rand(M, K, device="cuda") - 0.5
    w = torch.rand(N, K, device="cuda") - 0.5
    out = torch.empty(M, N, device="cuda")
    for _ in range(6):
        lib.call("dc_linear_forward", a, K, w, K, M, N, K, out, N, 0)
        torch.mm(a, w.t(), out=out)
    torch.cuda.synchronize()
