"""Forward and input-gradient products of three model shapes, the weight operand (a) from its pre-split bf16 planes as the model runs
them, (b) split inside the K loop (DC_NO_PLANES=1 -> option 9) -- for the rocprofv3 --pmc passes of tools/pmc_gemm_planes.sh."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd  # noqa: F401
from deltaconv_amd._lib import lib
from deltaconv_amd.nn import fused
if os.environ.get("DC_NO_PLANES", "0") == "1":
    lib.raw("dc_set_option")(9, 1)
for (M, N, K) in [(32768, 1024, 512), (65536, 256, 256), (32768, 128, 128)]:
    x = torch.rand(M, K, device="cuda") - 0.5
    w = torch.nn.Parameter(torch.rand(N, K, device="cuda") - 0.5)
    dy = torch.rand(M, N, device="cuda") - 0.5
    y, dx = torch.empty(M, N, device="cuda"), torch.empty(M, K, device="cuda")
    with torch.no_grad():
        for _ in range(4):
            fused._hint_planes(w, False)
            lib.call("dc_linear_forward", x, K, w, K, M, N, K, y, N, 0)
            fused._hint_planes(w, True)
            lib.call("dc_linear_backward_input", dy, N, w, K, M, N, K, dx, K, 0, 0)
    torch.cuda.synchronize()
