"""How close is the step to being host-bound?  Time to ENQUEUE 20 steps vs time until the GPU finishes them."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc
from deltaconv_amd.utils import calc_loss
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.dp import FlatGradDataParallel
dev = "cuda"
torch.manual_seed(1)
model = dc.models.DeltaNetClassification(3, 40).to(dev).train()
ddp = FlatGradDataParallel(model)
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
data = synthetic_batch(32, 1024, seed=100).to(dev)
def step():
    ddp.zero_grad(); loss = calc_loss(ddp(data), data.y); loss.backward(); ddp.reduce_gradients(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.3f} ms/step   total {1e3*(t2-t0)/20:.3f} ms/step   (host-bound if equal)")

from deltaconv_amd.graph_step import GraphedTrainStep
g = GraphedTrainStep(model, calc_loss, data, optimizer=opt)
def gstep():
    g()
for _ in range(5): gstep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): gstep()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"HIP graph: enqueue {1e3*(t1-t0)/20:.3f} ms/step   total {1e3*(t2-t0)/20:.3f} ms/step")
