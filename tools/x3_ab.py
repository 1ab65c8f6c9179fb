"""A/B of the dense products: exact fp32 MFMA chain (option 3 = 1) against the bf16 split products (default), per model
GEMM shape and per product (forward Y = X W^T, input gradient dX = dY W, weight gradient dW = dY^T X): error against an
fp64 product and time per launch under HIP-graph replay (20 launches per graph)."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib

dev = "cuda"
opt = lib.raw("dc_set_option")


def graph_us(fn, n=20, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


def err(y, ref):
    return float((y.double() - ref).abs().max() / ref.abs().max())


def main():
    shapes = [(32768, 1024, 448), (32768, 1024, 512), (32768, 256, 512), (65536, 256, 256), (32768, 128, 256),
              (65536, 128, 384), (32768, 128, 128), (32768, 64, 256), (32768, 64, 64), (65536, 64, 128)]
    if len(sys.argv) > 1:
        shapes = shapes[:int(sys.argv[1])]
    print(f"# {torch.cuda.get_device_name(0)}; error = max |y - y64| / max |y64|; time per launch (graph replay)")
    print(f"{'M x N x K':20s} {'product':8s} {'err exact':>10s} {'err split':>10s} {'exact us':>9s} {'split us':>9s} {'ratio':>6s}")
    torch.manual_seed(0)
    for (M, N, K) in shapes:
        x = torch.randn(M, K, device=dev) * torch.exp(2 * torch.randn(M, K, device=dev))
        x[x.abs() < 0.3] = 0                                    # post-ReLU-like operand: zeros and several binades
        w = torch.randn(N, K, device=dev) / K ** 0.5
        dy = torch.randn(M, N, device=dev) * torch.exp(torch.randn(M, N, device=dev))
        y, dx, dw = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
        ws_bytes = lib.raw("dc_gemm_tn_workspace_bytes")(M, N, K)
        ws = torch.empty(ws_bytes // 4 + 16, device=dev)
        sub = slice(0, 2048)
        refs = {"fwd": x[sub].double() @ w.double().t(), "dX": dy[sub].double() @ w.double(),
                "dW": dy.double().t() @ x.double()}
        fns = {"fwd": lambda: lib.call("dc_linear_forward", x, K, w, K, M, N, K, y, N, 0),
               "dX": lambda: lib.call("dc_linear_backward_input", dy, N, w, K, M, N, K, dx, K, 0, 0),
               "dW": lambda: lib.call("dc_gemm_tn", dy, N, x, K, M, N, K, dw, K, 0, ws, ws_bytes)}
        outs = {"fwd": lambda: y[sub], "dX": lambda: dx[sub], "dW": lambda: dw}
        for name in ("fwd", "dX", "dW"):
            res = {}
            for exact in (1, 0):
                opt(3, exact)
                fns[name]()
                torch.cuda.synchronize()
                e = err(outs[name](), refs[name])
                res[exact] = (e, graph_us(fns[name]))
            opt(3, 0)
            print(f"{M:6d}x{N:5d}x{K:4d}   {name:8s} {res[1][0]:10.2e} {res[0][0]:10.2e} {res[1][1]:9.1f} {res[0][1]:9.1f} "
                  f"{res[1][1] / res[0][1]:6.2f}", flush=True)


if __name__ == "__main__":
    main()
