"""Repro of the ROCm 7.2 HIP-graph replay problem that deltaconv_amd/graph_step.py works around.

    python tools/graph_replay_repro.py l1 other small                                   # wrong loss from replay 1 on
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/graph_replay_repro.py l1 other small   # correct

One DeltaConv layer (forward+backward) is captured in a HIP graph and replayed three times; between
replays a few tiny eager kernels run on unrelated tensors ("other small"), on the layer's parameters
(default) or nothing ("none").  With the runtime's AQL-packet capture on, the replay after the eager
launches computes a different loss (5.81 -> 44496.07); with memset nodes in the graph it faults
("Write access to a read-only page").  Modes: l0|l1 (centralized first layer or not), nofuse, novec,
geomin (kNN+MLS inside the capture), other [small], none, only=<param-name-substring>|x0."""
import os as _os
_os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")   # show the bug by default
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc
from deltaconv_amd.data import synthetic_batch
mode = set(sys.argv[1:])
torch.manual_seed(1)
b = synthetic_batch(4, 256, seed=1).to("cuda")
base = dc.models.DeltaNetBase(3, [16], 1, 20, 1e-3, 1).cuda()
cin = 3 if "l0" in mode else 64
model = dc.nn.DeltaConv(cin, 128, depth=2, centralized=("l0" in mode), vector=("novec" not in mode)).cuda()
if "nofuse" in mode: model.fuse_layer = False
x0 = b.pos if "l0" in mode else torch.randn(b.pos.shape[0], cin, device="cuda")
ops = None if "geomin" in mode else base.build_operators(b)
def fb():
    g, G, D = ops if ops is not None else base.build_operators(b)
    v0 = G @ x0
    x, v = model(x0, v0, G, D, g)
    loss = x.pow(2).mean() + (v.pow(2).mean() if "novec" not in mode else 0)
    loss.backward(); return loss.detach()
for p in model.parameters(): p.grad = None
print("eager", float(fb()), float(fb()), flush=True)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        for p in model.parameters(): p.grad = None
        fb()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
for p in model.parameters(): p.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = fb()
others = [torch.randn(128, device="cuda") for _ in range(8)] if "small" in mode else [torch.randn(128, 64, device="cuda") for _ in range(50)]
snap = {n: t.detach().clone() for n, t in list(model.named_parameters()) + list(model.named_buffers())}
for it in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay", it, float(loss), flush=True)
    with torch.no_grad():
        sel = [m for m in mode if m.startswith("only=")]
        if sel:
            key = sel[0][5:]
            if key == "x0": x0.mul_(1.0)
            for n, p in model.named_parameters():
                if key in n: p.mul_(1.0)
        elif "none" in mode: pass
        elif "other" in mode:
            for t in others: t.mul_(1.0)
        else:
            for p in model.parameters(): p.mul_(1.0)
    torch.cuda.synchronize()
for n, t in list(model.named_parameters()) + list(model.named_buffers()):
    if "running" in n or "tracked" in n: continue
    if not torch.equal(snap[n], t.detach()): print("MEMCHANGED", n)
print("OK", sorted(mode), float(loss))
