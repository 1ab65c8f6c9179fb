"""us per launch of the depth-2 centralised edge MLP (csrc/edge2.hip) at the ShapeNet shape (16 x 2048 points, k = 20, 3 -> 64 -> 64),
forward and backward entry points replayed from a HIP graph; the fp32 matrix-pipe floor beside them.
    python tools/edge2_bench.py [clouds=16] [points=2048] [k=20]"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.nn import fused

B, N, k = (int(a) for a in (sys.argv[1:4] + ["16", "2048", "20"][len(sys.argv) - 1:]))
b = synthetic_batch(B, N, seed=3).to("cuda")
graph = dc.geometry.Graph.knn(b.pos, k, b.batch)
graph.csc()
torch.manual_seed(0)
mlp = dc.nn.MLP([3, 64, 64]).cuda().train()
x = b.pos.clone().requires_grad_(True)


def fwd():
    return fused.edge_mlp2(x, graph, mlp[0][0], mlp[0][1].bn, 0.2, mlp[1][0], mlp[1][1].bn, 0.2)[0]


out = fwd()
dout = torch.randn_like(out)
out.backward(dout)
torch.cuda.synchronize()


def eager_us(fn, iters=100):
    """GPU time per call from HIP events around back-to-back eager calls (the GPU queue stays full: every call is 6-8 launches
    of 5-180 us, longer than their enqueue)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


with torch.no_grad():
    t_f = eager_us(lambda: fwd())
params = [x] + list(mlp.parameters())
t_fb = eager_us(lambda: torch.autograd.grad(fwd(), params, dout, allow_unused=True))
E = graph.n * k
flop = 2.0 * E * 64 * 64
print(f"{B} x {N} points, k = {k}: forward (z GEMM + edge statistics + edge2_fwd + finaliser + activation) {t_f:.1f} us; "
      f"forward + backward {t_fb:.1f} us; one E x 64 x 64 product at 157.3 TFLOP/s = {flop / 157.3e12 * 1e6:.1f} us "
      f"(forward pass 1 product, backward pass 3)")
