"""ms per graph-replayed C2 train step of the library named by DELTACONV_HIP_LIB (default: the in-tree build): 3 x 40 replays.
For A/B runs of lab builds: alternate processes on one box.   python tools/step_ms.py [reps=3] [steps=40]"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.utils import calc_loss
from deltaconv_amd.graph_step import GraphedTrainStep
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
torch.manual_seed(1)
model = dc.models.DeltaNetClassification(3, 40, num_neighbors=20).cuda().train()
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
batches = [synthetic_batch(32, 1024, seed=100 + i).to("cuda") for i in range(4)]
static = synthetic_batch(32, 1024, seed=99).to("cuda")
g = GraphedTrainStep(model, calc_loss, static, optimizer=opt)
out = []
for r in range(reps):
    for i in range(5):
        g(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        g(batches[i % 4])
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / steps * 1e3)
print(os.environ.get("DELTACONV_HIP_LIB", "in-tree"), " ".join(f"{t:.3f}" for t in out), "ms per step")
