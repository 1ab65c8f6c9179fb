"""Lab (round 6): does the ROW STRIDE of the operands explain why the graded apply is slower inside the step than alone?

Inside the layer the fused div|curl|norm apply reads v out of the next layer's concat buffer (row stride 2 ci + co floats)
and writes into x_cat (row stride 4 ci floats): for ci = 64 these are 768 / 1024-byte strides, for the isolated benchmark
256 / 768 bytes.  A stride that is a multiple of 1 KiB maps the 256-byte row pieces onto a quarter of the L2 channels.

    python tools/stride_lab.py [C=64] > profiles/r06_stride_lab.txt

HIP events around graph replays of 50 launches on 8 rotating operand sets; the same launch at every (ldv, ldo) pair."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc                               # noqa: E402
from deltaconv_amd import _ops                           # noqa: E402
from deltaconv_amd.data import synthetic_batch           # noqa: E402

DEV = "cuda"


def timed(calls, rounds=4, reps=5):
    for c in calls:
        c()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rounds):
            for c in calls:
                c()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * rounds * len(calls)) * 1e3)
    return best


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    b = synthetic_batch(32, 1024, seed=100).to(DEV)
    model = dc.models.DeltaNetClassification(3, 40).to(DEV)
    graph, grad, div = model.deltanet_base.build_operators(b)
    n, k = graph.n, graph.k
    sets = 8
    print(f"# {torch.cuda.get_device_name(0)}; n = {n}, k = {k}, C = {C}; us per launch, {sets} rotating operand sets")

    def bufs(ld, rows):
        return [torch.randn(rows, ld, device=DEV) for _ in range(sets)]

    print("## tiled div|curl|norm forward: v [2n, C] at row stride ldv -> out [n, 3C] at row stride ldo (floats)")
    for ldv in (C, 3 * C, 4 * C, 4 * C + 32, 8 * C, 8 * C + 32):
        for ldo in (3 * C, 4 * C, 4 * C + 32):
            vs, os_ = bufs(ldv, 2 * n), bufs(ldo, n)
            calls = [(lambda v=v, o=o: _ops.fwd_apply("div_curl_norm", div, v[:, :C], C, ldv, o[:, :3 * C], ldo)) for v, o in zip(vs, os_)]
            print(f"ldv {ldv:4d} ({ldv * 4:5d} B)  ldo {ldo:4d} ({ldo * 4:5d} B): {timed(calls):7.2f} us", flush=True)
            del vs, os_, calls
    print("## tiled hodge forward: [div | curl] block of x_cat (row stride ldx) -> v_cat block (row stride ldvc)")
    for ldx in (3 * C, 4 * C, 4 * C + 32):
        for ldvc in (C, 3 * C, 4 * C, 4 * C + 32):
            xs, vs = bufs(ldx, n), bufs(ldvc, 2 * n)
            calls = [(lambda x=x, v=v: _ops.fwd_apply("hodge", grad, x[:, :2 * C], C, ldx, v[:, :C], ldvc)) for x, v in zip(xs, vs)]
            print(f"ldx {ldx:4d}  ldvc {ldvc:4d}: {timed(calls):7.2f} us", flush=True)
            del xs, vs, calls
    print("## tiled grad forward: x [n, C] (row stride ldx) -> v_cat block (row stride ldvc)")
    for ldx in (C, 2 * C, 4 * C, 4 * C + 32):
        for ldvc in (C, 3 * C, 4 * C, 4 * C + 32):
            xs, vs = bufs(ldx, n), bufs(ldvc, 2 * n)
            calls = [(lambda x=x, v=v: _ops.fwd_apply("grad", grad, x[:, :C], C, ldx, v[:, :C], ldvc)) for x, v in zip(xs, vs)]
            print(f"ldx {ldx:4d}  ldvc {ldvc:4d}: {timed(calls):7.2f} us", flush=True)
            del xs, vs, calls
    print("## tiled max aggregation: h [n, C] (row stride ldh) -> out block (row stride ldo)")
    for ldh in (C, 4 * C, 4 * C + 32):
        for ldo in (C, 4 * C, 4 * C + 32):
            hs, os_ = bufs(ldh, n), bufs(ldo, n)
            arg = torch.empty(n, C, dtype=torch.uint8, device=DEV)
            calls = [(lambda h=h, o=o: _ops.fwd_knn_max(graph, h[:, :C], C, ldh, o[:, :C], ldo, arg)) for h, o in zip(hs, os_)]
            print(f"ldh {ldh:4d}  ldo {ldo:4d}: {timed(calls):7.2f} us", flush=True)
            del hs, os_, calls
    grad.coefTt(), div.coefTt()
    print("## tiled div|curl|norm TRANSPOSED: dout block of d_xcat (ldo), v (ldv), dv (lddv, accumulate)")
    for ldo, ldv in ((3 * C, C), (4 * C, 3 * C), (4 * C, 4 * C), (4 * C + 32, 4 * C + 32), (4 * C + 32, 3 * C)):
        ds, vs, dvs = bufs(ldo, n), bufs(ldv, 2 * n), bufs(ldv, 2 * n)
        calls = [(lambda d=d, v=v, dv=dv: _ops.bwd_div_curl_norm(div, d[:, :3 * C], C, ldo, v[:, :C], ldv, dv[:, :C], ldv, 1))
                 for d, v, dv in zip(ds, vs, dvs)]
        print(f"ldo {ldo:4d}  ldv {ldv:4d}: {timed(calls):7.2f} us", flush=True)
        del ds, vs, dvs, calls


main()
