"""How the forward GEMM's rate depends on K (per-tile prologue / epilogue share) and on the tile count."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
from tools.gemm_lab import timeit
DEV = "cuda"
for (M, N) in [(32768, 1024), (32768, 256), (65536, 256)]:
    for K in (128, 256, 512, 1024, 2048, 4096):
        a = torch.rand(M, K, device=DEV) - 0.5
        w = torch.rand(N, K, device=DEV) - 0.5
        out = torch.empty(M, N, device=DEV)
        t = timeit(lambda: lib.call("dc_linear_forward", a, K, w, K, M, N, K, out, N, 1), iters=10)
        tl = timeit(lambda: torch.mm(a, w.t(), out=out), iters=10)
        fl = 2.0 * M * N * K
        print(f"{M}x{N}x{K}: own {t:8.1f} us {fl / t / 1e6:6.1f} TF | library(default) {tl:8.1f} us {fl / tl / 1e6:6.1f} TF", flush=True)
