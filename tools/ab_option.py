"""Same-box A/B of the bench step under a runtime option of the library: alternates `dc_set_option(key, a)` / `(key, b)`,
re-captures the HIP graph each time, prints ms per step.   python tools/ab_option.py <key> <a> <b> [reps=3] [steps=40]"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc
from deltaconv_amd._lib import lib
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.utils import calc_loss
from deltaconv_amd.graph_step import GraphedTrainStep
key, a, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
torch.manual_seed(1)
model = dc.models.DeltaNetClassification(3, 40, num_neighbors=20).cuda().train()
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
batches = [synthetic_batch(32, 1024, seed=100 + i).to("cuda") for i in range(4)]
static = synthetic_batch(32, 1024, seed=99).to("cuda")
res = {a: [], b: []}
for r in range(reps):
    for v in (a, b):
        lib.raw("dc_set_option")(key, v)
        g = GraphedTrainStep(model, calc_loss, static, optimizer=opt)
        for i in range(5):
            g(batches[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            g(batches[i % 4])
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e3)
        del g
lib.raw("dc_set_option")(key, 0)
for v in (a, b):
    print(f"option {key} = {v}: " + " ".join(f"{t:.3f}" for t in res[v]) + f"  ms per step (min {min(res[v]):.3f})")
