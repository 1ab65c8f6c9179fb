"""Timing sweep of the hand-written forward / input-gradient GEMMs (csrc/gemm.hip) against the vendor library on
the shapes of the C2 step.  Run on the GPU box:  python tools/gemm_lab.py [> gpurun_out/gemm_lab.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd._lib as _L
if os.environ.get("DC_AB_LIB"):            # A/B runs: another build of the library (tools/ab/, not tracked)
    _L.LIB_PATH = os.path.abspath(os.environ["DC_AB_LIB"])
from deltaconv_amd._lib import lib

DEV = "cuda"
TILES = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x64", 4: "64x128"}


def timeit(fn, iters=20, warm=3):
    """us per call, measured on a HIP graph holding `iters` back-to-back calls (no host launch overhead, like the
    captured training step)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30                      # min over replays: filters clock ramps and neighbours on the box
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


FWD = [  # (label, M, N, K)
    ("L1 hm", 32768, 64, 64), ("L1 hs", 32768, 64, 256), ("L1 PQ", 65536, 128, 192),
    ("L2 hm", 32768, 128, 64), ("L2 hs", 32768, 128, 256), ("L2 PQ", 65536, 256, 256),
    ("L3 hm", 32768, 256, 128), ("L3 hs", 32768, 256, 512), ("embed", 32768, 1024, 512),
    ("L0 PQ", 65536, 128, 70), ("L0 hs", 32768, 64, 12),
]
BWD = [  # dX[M,K] = dY[M,N] W[N,K]
    ("embed dX", 32768, 1024, 512), ("L3 d_xcat", 32768, 256, 512), ("L3 dx", 32768, 256, 128),
    ("L2 dv_cat", 65536, 256, 256), ("L2 d_xcat", 32768, 128, 256), ("L2 dx", 32768, 128, 64),
    ("L1 dv_cat", 65536, 128, 192), ("L1 d_xcat", 32768, 64, 256), ("L1 dx", 32768, 64, 64),
    ("L0 dv_cat", 65536, 128, 64),
]


TN = [  # dW[M,N] = dY[R,M]^T X[R,N]
    ("embed dW", 32768, 1024, 512), ("L3 dWs", 32768, 256, 512), ("L3 dWm", 32768, 256, 128),
    ("L2 dWv", 65536, 256, 256), ("L2 dWs", 32768, 128, 256), ("L2 dWm", 32768, 128, 64),
    ("L1 dWv", 65536, 128, 192), ("L1 dWs", 32768, 64, 256), ("L1 dWm", 32768, 64, 64),
]


def main():
    torch.manual_seed(0)
    from deltaconv_amd.tuning import enable_tuned_gemms
    enable_tuned_gemms()           # the library column = the shipped per-shape tuned solutions (what r01 ran)
    print(f"{'shape':<34}{'library':>10}" + "".join(f"{TILES[t]:>10}" for t in TILES) + "   best TF/s (lib TF/s)")
    for kind, table in (("fwd", FWD), ("dX", BWD)):
        for label, M, N, K in table:
            if kind == "fwd":
                a = torch.rand(M, K, device=DEV) - 0.5
                w = torch.rand(N, K, device=DEV) - 0.5
                out = torch.empty(M, N, device=DEV)
                t_lib = timeit(lambda: torch.mm(a, w.t(), out=out))
                ts = {t: timeit(lambda t=t: lib.call("dc_linear_forward", a, K, w, K, M, N, K, out, N, t)) for t in TILES}
                # with the statistics epilogue
                coef = torch.empty(4, N, device=DEV)
                g = torch.ones(N, device=DEV)
                nb = lib.raw("dc_linear_stats_workspace_bytes")(M, N, K, 0)
                ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=DEV)
                t_st = timeit(lambda: lib.call("dc_linear_bn_stats_forward", a, K, w, K, M, N, K, out, N, g, g, 1e-5, 0.1,
                                               None, None, coef[0], coef[1], coef[2], coef[3], 0, ws, nb))
                nb2 = lib.raw("dc_bn_workspace_bytes")(M, N)
                ws2 = torch.empty((nb2 + 7) // 8, dtype=torch.float64, device=DEV)
                t_sep = timeit(lambda: lib.call("dc_bn_stats", out, M, N, N, g, g, 1e-5, 0.1, None, None, coef[0], coef[1],
                                                coef[2], coef[3], ws2, nb2))
                extra = f"   +stats fused {t_st:7.1f} us vs separate stats pass {t_sep:6.1f} us"
            else:
                a = torch.rand(M, N, device=DEV) - 0.5
                w = torch.rand(N, K, device=DEV) - 0.5
                out = torch.empty(M, K, device=DEV)
                t_lib = timeit(lambda: torch.mm(a, w, out=out))
                ts = {t: timeit(lambda t=t: lib.call("dc_linear_backward_input", a, N, w, K, M, N, K, out, K, 0, t))
                      for t in TILES}
                extra = ""
            fl = 2.0 * M * N * K
            best = min(ts.values())
            print(f"{kind + ' ' + label + f' {M}x{N}x{K}':<34}{t_lib:10.1f}" + "".join(f"{ts[t]:10.1f}" for t in TILES)
                  + f"   {fl / best / 1e6:6.1f} ({fl / t_lib / 1e6:6.1f})" + extra, flush=True)


    print("weight gradient dW = dY^T X: library | direct-load kernel (r01) | LDS-staged kernel")
    for label, R, M, N in TN:
        a = torch.rand(R, M, device=DEV) - 0.5
        b = torch.rand(R, N, device=DEV) - 0.5
        out = torch.empty(M, N, device=DEV)
        nb = lib.raw("dc_gemm_tn_workspace_bytes")(R, M, N)
        ws = torch.empty((nb + 3) // 4, device=DEV)
        t_lib = timeit(lambda: torch.mm(a.t(), b, out=out))
        res = {}
        for impl in (2, 0):
            lib.raw("dc_set_option")(2, impl)
            res[impl] = timeit(lambda: lib.call("dc_gemm_tn", a, M, b, N, R, M, N, out, N, 0, ws, ws.numel() * 4))
            ref = a.double().t() @ b.double()
            err = float((out.double() - ref).abs().max() / ref.abs().max())
            assert err < 1e-4, (label, impl, err)
        lib.raw("dc_set_option")(2, 0)
        fl = 2.0 * R * M * N
        print(f"tn {label + f' {R}x{M}x{N}':<30}{t_lib:10.1f}{res[2]:10.1f}{res[0]:10.1f}   TF/s {fl / t_lib / 1e6:6.1f} "
              f"{fl / res[2] / 1e6:6.1f} {fl / res[0] / 1e6:6.1f}", flush=True)


if __name__ == "__main__":
    main()
