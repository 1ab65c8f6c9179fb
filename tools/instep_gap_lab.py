"""Lab (round 6): why does the graded apply EXECUTE slower inside the training step (14.2 - 16.1 us, device-clock stamps) than in
back-to-back launches of itself (12.0 us)?  Cold operands are not it (rotating operand sets execute in 11.9 us).

Hypothesis: the clock.  Inside the step the kernel runs between matrix-pipe-heavy dense products that pull the power-limited
shader clock down (~1.64 GHz under the split products); alone it runs at the idle-boost clock.  The lab
replays graphs that alternate the apply with (B) a large dense product, (C) a streaming BatchNorm/activation pass of the same
duration class, (D) five tiny kernels, and reads the apply's execution time from the stamps (csrc/common.h: dc_stamp_*).

    python tools/instep_gap_lab.py > profiles/r06_instep_gap_lab.txt
"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc                               # noqa: E402
from deltaconv_amd import _ops                           # noqa: E402
from deltaconv_amd._lib import lib                       # noqa: E402
from deltaconv_amd.data import synthetic_batch           # noqa: E402
from deltaconv_amd.nn import fused                       # noqa: E402

DEV = "cuda"


def main():
    b = synthetic_batch(32, 1024, seed=100).to(DEV)
    model = dc.models.DeltaNetClassification(3, 40).to(DEV)
    graph, grad, div = model.deltanet_base.build_operators(b)
    n, C = graph.n, 64
    sets = 6
    vs = [torch.randn(2 * n, C, device=DEV) for _ in range(sets)]
    outs = [torch.empty(n, 3 * C, device=DEV) for _ in range(sets)]
    X, W = torch.randn(n, 512, device=DEV), torch.randn(1024, 512, device=DEV)
    Y = torch.empty(n, 1024, device=DEV)
    H = torch.randn(n, 1024, device=DEV)
    Hy = torch.empty_like(H)
    sc, sh = torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)
    tiny = torch.zeros(64, device=DEV)

    def apply(i):
        _ops.fwd_apply("div_curl_norm", div, vs[i % sets], C, C, outs[i % sets], 3 * C)

    def gemm():
        lib.call("dc_linear_forward", X, 512, W, 512, n, 1024, 512, Y, 1024, 0)

    def stream():
        lib.call("dc_bn_act", H, n, 1024, 1024, sc, sh, 0.2, None, 0, Hy, 1024)

    wsb = lib.raw("dc_bn_workspace_bytes")(n, 1024)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=DEV)
    coef = torch.empty(4, 1024, device=DEV)

    def reduce_only():      # reads 134 MB, writes a few KB: a heavy predecessor that leaves nothing dirty
        lib.call("dc_bn_stats", H, n, 1024, 1024, sc, sh, 1e-5, 0.1, None, None, coef[0], coef[1], coef[2], coef[3], ws, wsb)

    def tinies():
        for _ in range(5):
            lib.call("dc_bn_act", tiny, 1, 64, 64, sc, sh, 0.2, None, 0, tiny, 64)

    cases = {"A  apply only, back to back": None, "B  dense product (32768 x 1024 x 512) before every apply": gemm,
             "C  streaming BatchNorm/activation pass (268 MB) before every apply": stream, "D  five tiny kernels before every apply": tinies,
             "E  two dense products before every apply": lambda: (gemm(), gemm()),
             "F  column reduction (reads 134 MB, writes KBs) before every apply": reduce_only,
             "G  streaming pass, then five tiny kernels, then the apply": lambda: (stream(), tinies())}
    print(f"# {torch.cuda.get_device_name(0)}; graded apply (tiled div|curl|norm, C = 64, 32 x 1024 points, k = 20), {sets} rotating operand sets")
    print("# execution = device-clock stamps (first workgroup entry -> last workgroup exit), median over 25 launches x 8 replays")
    per = 25
    for name, pre in cases.items():
        for fn in ([pre] if pre else []) + [lambda: apply(0)]:
            fn()
        torch.cuda.synchronize()
        stamps = torch.zeros(per, 4, dtype=torch.int64, device=DEV)
        lib.raw("dc_stamp_buffer")(stamps.data_ptr(), per)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(per):
                    if pre:
                        pre()
                    apply(i)
            used = lib.raw("dc_stamp_count")()
        finally:
            lib.raw("dc_stamp_buffer")(None, 0)
        assert used == per, used
        meds, walls = [], []
        for _ in range(8):
            stamps[:, 0] = 2 ** 62
            stamps[:, 1] = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            rec = stamps.cpu()
            meds.append(float(((rec[:, 1] - rec[:, 0]).double() * 1e-2).median()))
            walls.append(e0.elapsed_time(e1) * 1e3 / per)
        print(f"{name:70s} apply execution {sorted(meds)[4]:6.2f} us   (graph: {sorted(walls)[4]:7.2f} us per iteration)", flush=True)


main()
