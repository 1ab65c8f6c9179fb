"""Does the MEMORY order of the points matter?  Same clouds, (a) in the generator's order, (b) each cloud's points sorted along a
Morton curve (a per-cloud permutation of pos / norm: classification is invariant under it).  Prints ms per graph-replayed train
step for both, alternating.   python tools/morton_ab.py [B=32] [N=1024] [k=20] [reps=3] [steps=40]"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc
from deltaconv_amd.data import synthetic_batch, Batch
from deltaconv_amd.utils import calc_loss
from deltaconv_amd.graph_step import GraphedTrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 40


def morton(batch):
    pos = batch.pos.view(B, N, 3)
    lo = pos.min(dim=1, keepdim=True).values
    hi = pos.max(dim=1, keepdim=True).values
    q = ((pos - lo) / (hi - lo).clamp_min(1e-12) * 1023).long().clamp(0, 1023)
    key = torch.zeros(B, N, dtype=torch.long, device=pos.device)
    for bit in range(10):
        for ax in range(3):
            key |= ((q[..., ax] >> bit) & 1) << (3 * bit + ax)
    order = key.argsort(dim=1) + (torch.arange(B, device=pos.device) * N).view(B, 1)
    order = order.reshape(-1)
    return Batch(batch.pos[order].contiguous(), batch.batch, batch.norm[order].contiguous(), None, batch.y, None, batch.num_graphs)


torch.manual_seed(1)
model = dc.models.DeltaNetClassification(3, 40, num_neighbors=k).cuda().train()
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
raw = [synthetic_batch(B, N, seed=100 + i).to("cuda") for i in range(4)]
srt = [morton(b) for b in raw]
res = {"generator order": [], "morton order": []}
for r in range(reps):
    for name, bs in (("generator order", raw), ("morton order", srt)):
        static = Batch(bs[0].pos.clone(), bs[0].batch, bs[0].norm.clone(), None, bs[0].y.clone(), None, bs[0].num_graphs)
        g = GraphedTrainStep(model, calc_loss, static, optimizer=opt)
        for i in range(5):
            g(bs[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            g(bs[i % 4])
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / steps * 1e3)
        del g
for name, v in res.items():
    print(f"{name}: " + " ".join(f"{t:.3f}" for t in v) + f"  ms per step (min {min(v):.3f})")
