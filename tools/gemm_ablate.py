"""Ablation of the GEMM main loop (timing only; results are wrong under ablation): which part costs what."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
from tools.gemm_lab import timeit

DEV = "cuda"
for (M, N, K) in [(32768, 1024, 512), (32768, 256, 512), (65536, 256, 256), (32768, 64, 256)]:
    a = torch.rand(M, K, device=DEV) - 0.5
    w = torch.rand(N, K, device=DEV) - 0.5
    out = torch.empty(M, N, device=DEV)
    fl = 2.0 * M * N * K
    for tile in (1, 3):
        row = []
        for abl in (0, 1, 2, 3, 4, 7):
            lib.raw("dc_set_option")(4, abl)
            t = timeit(lambda: lib.call("dc_linear_forward", a, K, w, K, M, N, K, out, N, tile))
            row.append(f"abl{abl}: {t:7.1f} us {fl / t / 1e6:6.1f} TF")
        lib.raw("dc_set_option")(4, 0)
        print(f"{M}x{N}x{K} tile {tile}: " + " | ".join(row), flush=True)
