"""Per-kernel micro-benchmark at the BASELINE configs' shapes (runs on the GPU box).
Each kernel family is timed back-to-back with HIP events on the launch stream; reports us/launch and
GB/s against its ALGORITHMIC bytes (inputs once + outputs once + operator once)."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc                      # noqa: E402
from deltaconv_amd import _ops                  # noqa: E402
from deltaconv_amd._lib import lib              # noqa: E402
from deltaconv_amd.data import synthetic_batch  # noqa: E402
from deltaconv_amd.geometry import Graph, build_grad_div, build_tangent_basis  # noqa: E402


def timeit(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--channels", type=int, nargs="+", default=[64, 128, 256])
    ap.add_argument("--json", default=None)
    ap.add_argument("--sorted", action="store_true", help="Morton-sort the points of every cloud first")
    a = ap.parse_args()
    dev = "cuda"
    b = synthetic_batch(a.batch, a.points, seed=7)
    if a.sorted:
        q = ((b.pos.clamp(-1, 1) * 0.5 + 0.5) * 1023).long()

        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        perm = torch.sort((b.batch << 32) | code, stable=True).indices
        b.pos, b.norm = b.pos[perm].contiguous(), b.norm[perm].contiguous()
    b = b.to(dev)
    rows = []

    def rec(name, us, nbytes, **kw):
        rows.append(dict(kernel=name, us=round(us, 2), GBs=round(nbytes / us / 1e3, 1), MB=round(nbytes / 1e6, 2), **kw))
        print(f"{name:38s} {us:9.2f} us  {nbytes / 1e6:8.2f} MB  {nbytes / us / 1e3:9.1f} GB/s  {kw}")

    info = dc.models.deltanet_base._ptr_info(b)
    n, k = b.pos.shape[0], a.k
    E = n * k
    for lanes in (1, 8, 64):
        rec(f"knn lanes={lanes}", timeit(lambda: Graph.knn(b.pos, k, ptr_info=info, lanes_per_query=lanes), 20, 3),
            12 * n + 4 * E)
    graph = Graph.knn(b.pos, k, ptr_info=info)
    xb, yb = build_tangent_basis(b.norm)
    rec("tangent_basis", timeit(lambda: build_tangent_basis(b.norm)), 36 * n)
    rec("mls_assemble", timeit(lambda: build_grad_div(b.pos, b.norm, xb, yb, graph, b.batch), 20, 3), 48 * n + 20 * E)
    grad, div = build_grad_div(b.pos, b.norm, xb, yb, graph, b.batch)

    def csc():
        graph._csc = None
        graph.csc()
    rec("csc_build", timeit(csc, 20, 3), 12 * E)
    tptr, tedge = graph.csc()
    GT, DT = grad.coefT(), div.coefT()
    for remap in (1, 0):
        lib.raw("dc_set_option")(0, remap)
        for C in (a.channels if remap else a.channels[:1]):
            x = torch.randn(n, C, device=dev)
            v = torch.randn(2 * n, C, device=dev)
            dcn = torch.randn(n, 3 * C, device=dev)
            y2 = torch.empty(2 * n, C, device=dev)
            y1 = torch.empty(n, C, device=dev)
            y3 = torch.empty(n, 3 * C, device=dev)
            arg = torch.empty(n, C, dtype=torch.uint8, device=dev)
            ab = 12 * C * n + 12 * E
            kw = dict(C=C, remap=remap)
            call = lib.call
            rec("apply_grad", timeit(lambda: call("dc_apply_grad", grad.coef, graph.nbr, n, k, x, C, C, y2, C)), ab, **kw)
            rec("apply_div", timeit(lambda: call("dc_apply_div", div.coef, graph.nbr, n, k, v, C, C, y1, C)), ab, **kw)
            rec("apply_div_curl_norm", timeit(lambda: call("dc_apply_div_curl_norm", div.coef, graph.nbr, n, k, v, C, C, y3, 3 * C)),
                20 * C * n + 12 * E, **kw)
            rec("apply_hodge", timeit(lambda: call("dc_apply_hodge", grad.coef, graph.nbr, n, k, dcn, C, 3 * C, y2, C)),
                16 * C * n + 12 * E, **kw)
            rec("apply_grad_T", timeit(lambda: call("dc_apply_grad_T", GT, tptr, tedge, n, k, y2, C, C, y1, C, 0)),
                12 * C * n + 16 * E, **kw)
            rec("apply_div_T", timeit(lambda: call("dc_apply_div_T", DT, tptr, tedge, n, k, y1, C, C, y2, C, 0)),
                12 * C * n + 16 * E, **kw)
            rec("apply_div_curl_norm_T", timeit(lambda: call("dc_apply_div_curl_norm_T", DT, tptr, tedge, n, k, y3, C, 3 * C, v, C, y2, C, 0)),
                28 * C * n + 16 * E, **kw)
            rec("apply_hodge_T", timeit(lambda: call("dc_apply_hodge_T", GT, tptr, tedge, n, k, y2, C, C, dcn, 3 * C, 0)),
                16 * C * n + 16 * E, **kw)
            rec("knn_max", timeit(lambda: call("dc_knn_max", graph.nbr, n, k, x, C, C, y1, C, arg)), 9 * C * n + 4 * E, **kw)
            rec("knn_max_backward", timeit(lambda: call("dc_knn_max_backward", tptr, tedge, n, k, arg, y1, C, C, x, C, 0)),
                9 * C * n + 8 * E, **kw)
            # torch reference points for the dense stream at the same shape
            w = torch.randn(C, 4 * C, device=dev)
            xc = torch.randn(n, 4 * C, device=dev)
            t = timeit(lambda: torch.nn.functional.linear(xc, w))
            rec("torch linear [Nt,4C]x[4C,C]", t, 4 * (5 * C * n + 4 * C * C), TF=round(2 * n * 4 * C * C / t / 1e6, 1), **kw)
            bn = torch.nn.BatchNorm1d(C).to(dev).train()
            rec("torch batch_norm+leaky [Nt,C]", timeit(lambda: torch.nn.functional.leaky_relu(bn(x), 0.2)), 8 * C * n, **kw)
            from deltaconv_amd.nn import fused
            rec("fused bn_stats+act fwd [Nt,C]", timeit(lambda: fused.bn_act(x, bn, 0.2)), 12 * C * n, **kw)
            xg = x.clone().requires_grad_(True)
            yg = fused.bn_act(xg, bn, 0.2)
            gy = torch.randn_like(yg)
            rec("fused bn_act bwd [Nt,C]", timeit(lambda: torch.autograd.grad(yg, xg, gy, retain_graph=True)), 20 * C * n, **kw)
    lib.raw("dc_set_option")(0, 1)
    from deltaconv_amd.nn import fused
    from deltaconv_amd.tuning import enable_tuned_gemms
    enable_tuned_gemms()                      # the library side of the comparison uses its tuned solutions
    for (R, M, N) in ((32768, 64, 64), (32768, 64, 256), (32768, 128, 64), (32768, 128, 256), (32768, 256, 128),
                      (32768, 256, 512), (65536, 128, 192), (65536, 256, 256)):
        A, Bm = torch.randn(R, M, device=dev), torch.randn(R, N, device=dev)
        t1 = timeit(lambda: fused.gemm_tn(A, Bm), 50, 5)
        t2 = timeit(lambda: A.t() @ Bm, 50, 5)
        rec(f"gemm_tn MFMA {R}x{M}x{N}", t1, 4 * R * (M + N), TF=round(2 * R * M * N / t1 / 1e6, 1))
        rec(f"gemm_tn library {R}x{M}x{N}", t2, 4 * R * (M + N), TF=round(2 * R * M * N / t2 / 1e6, 1))
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
