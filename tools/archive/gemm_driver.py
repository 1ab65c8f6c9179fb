"""Launch the own MFMA weight-gradient GEMM and the library GEMM a fixed number of times (rocprofv3 --pmc)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
from deltaconv_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
for (R, M, N) in ((65536, 256, 256), (32768, 256, 128)):
    A, B = torch.randn(R, M, device="cuda"), torch.randn(R, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    nb = lib.raw("dc_gemm_tn_workspace_bytes")(R, M, N)
    ws = torch.empty((nb + 3) // 4, device="cuda")
    for _ in range(20):
        lib.call("dc_gemm_tn", A, M, B, N, R, M, N, C, N, 0, ws, ws.numel() * 4)
X, W = torch.randn(32768, 512, device="cuda"), torch.randn(1024, 512, device="cuda")
for _ in range(20):
    Y = X @ W.t()
torch.cuda.synchronize()
print("done")
