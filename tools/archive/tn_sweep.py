"""Lab: tile / slab-count sweep of the LDS-staged weight-gradient GEMM (csrc/gemm.hip A_KM x B_KN, options 5 / 6)
on the dW shapes of C2 and C5.  Run on the GPU box:  python tools/tn_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib
from tools.gemm_lab import timeit, TILES

SHAPES = [("embed dW", 32768, 1024, 512), ("C2 L3 dWs", 32768, 256, 512), ("C2 L3 dWm", 32768, 256, 128), ("C2 L2 dWv", 65536, 256, 256),
          ("C2 L2 dWs", 32768, 128, 256), ("C2 L2 dWm", 32768, 128, 64), ("C2 L1 dWv", 65536, 128, 192),
          ("C2 L1 dWs", 32768, 64, 256), ("C2 L1 dWm", 32768, 64, 64),
          ("C5 dWs", 32768, 128, 512), ("C5 dWm", 32768, 128, 128), ("C5 dWv", 65536, 256, 384)]


def main():
    opt = lib.raw("dc_set_option")
    for label, R, M, N in SHAPES:
        a = torch.rand(R, M, device="cuda") - 0.5
        b = torch.rand(R, N, device="cuda") - 0.5
        out = torch.empty(M, N, device="cuda")
        ws = torch.empty(min(512 * M * N, 1 << 27) + 1024, device="cuda")        # room for any slab count of the sweep
        opt(2, 2); opt(5, 0); opt(6, 0)
        t_direct = timeit(lambda: lib.call("dc_gemm_tn", a, M, b, N, R, M, N, out, N, 0, ws, ws.numel() * 4))
        opt(2, 0)
        t_auto = timeit(lambda: lib.call("dc_gemm_tn", a, M, b, N, R, M, N, out, N, 0, ws, ws.numel() * 4))
        fl = 2.0 * R * M * N
        print(f"{label} {R}x{M}x{N}: direct {t_direct:6.1f} us  lds-auto {t_auto:6.1f} us  ({fl / min(t_direct, t_auto) / 1e6:5.1f} TF/s)")
        best = (1e9, None)
        for t in (1, 2, 3, 4):
            bm = 128 if t in (1, 2) else 64
            bn = 128 if t in (1, 4) else 64
            if bm > max(M, 64) or bn > max(N, 64):
                continue
            row = []
            for slabs in (16, 32, 64, 128, 256, 512):
                tiles = -(-M // bm) * -(-N // bn)
                if tiles * slabs > 4096 or R // slabs < 64 or slabs * M * N > (1 << 27):
                    row.append("     -")
                    continue
                opt(5, t); opt(6, slabs)
                us = timeit(lambda: lib.call("dc_gemm_tn", a, M, b, N, R, M, N, out, N, 0, ws, ws.numel() * 4))
                row.append(f"{us:6.1f}")
                if us < best[0]:
                    best = (us, (TILES[t], slabs))
            print(f"    tile {TILES[t]:>8}: slabs 16..512  " + " ".join(row))
        print(f"    best {best[0]:.1f} us {best[1]}  -> {fl / best[0] / 1e6:5.1f} TF/s", flush=True)
    opt(2, 0); opt(5, 0); opt(6, 0)


if __name__ == "__main__":
    main()
