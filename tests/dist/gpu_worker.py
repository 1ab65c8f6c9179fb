"""Runs under `python -m torch.distributed.run --nproc-per-node 1` on the GPU box (tests/test_gpu_dist.py):
RCCL process group with one rank, so every collective code path executes.
  1. GraphedTrainStep (HIP-graph replay of forward + loss + backward) + FlatGradDataParallel.reduce_gradients()
     (which re-points .grad at views of its flat buffer) + optimizer step, 3 steps on changing batches
     == the same 3 steps run eagerly: parameters and BatchNorm buffers bit-identical.
  1b. the two-graph step (GraphedTrainStep(..., optimizer, reducer)): gradient pack captured behind the backward pass,
     scale + optimizer update captured as a second graph, the all-reduce between the replays == eager, bit-identical.
  2. sync_bn=True (split statistics kernels + all-reduce of the fp64 sums, composed layer path) on one rank
     == per-rank statistics (fused layer path): logits / gradients agree to fp32 rounding.
  3. the sync_bn step captured in ONE graph (statistics + gradient collectives inside) == the same steps eagerly, bit-identical.
Prints one JSON line."""
import json
import os
import sys

os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import deltaconv_amd as dc
from deltaconv_amd import dp
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.graph_step import GraphedTrainStep
from deltaconv_amd.utils import calc_loss


def make(seed=5):
    torch.manual_seed(seed)
    m = dc.models.DeltaNetClassification(3, 40, num_neighbors=20).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.eval()
    return m, torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    batches = [synthetic_batch(4, 256, seed=60 + i).to("cuda") for i in range(3)]
    res = {}

    # ---- 1. graph replay + reduce_gradients + optimizer vs eager
    m1, o1 = make()
    d1 = dp.FlatGradDataParallel(m1, always_reduce=True)
    for b in [batches[0]] * 2 + batches:                 # 2 warm-up steps of the graphed twin, then the 3 compared ones
        d1.zero_grad()
        calc_loss(d1(b), b.y).backward()
        d1.reduce_gradients()
        o1.step()
    m2, o2 = make()
    d2 = dp.FlatGradDataParallel(m2, always_reduce=True)
    static = synthetic_batch(4, 256, seed=60).to("cuda")
    for _ in range(2):
        d2.zero_grad()
        calc_loss(d2(static), static.y).backward()
        d2.reduce_gradients()
        o2.step()
    step = GraphedTrainStep(m2, calc_loss, static, optimizer=None, warmup=0)
    for b in batches:
        step(b)
        d2.reduce_gradients()
        o2.step()
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    res["graph_dp_mismatches"] = [k for k in sd1 if not torch.equal(sd1[k], sd2[k])]

    # ---- 1b. the two-graph data-parallel step: replay (fwd + loss + bwd + gradient pack) -> all-reduce -> replay
    # (scale + optimizer update) == the same steps run eagerly
    m5, o5 = make()
    d5 = dp.FlatGradDataParallel(m5, always_reduce=True)
    static5 = synthetic_batch(4, 256, seed=60).to("cuda")
    step5 = GraphedTrainStep(m5, calc_loss, static5, optimizer=o5, warmup=2, reducer=d5)
    assert step5.graph_update is not None
    for b in batches:
        step5(b)
    sd5 = m5.state_dict()
    res["two_graph_dp_mismatches"] = [k for k in sd1 if not torch.equal(sd1[k], sd5[k])]

    # ---- 2. synchronised BatchNorm on one rank vs per-rank statistics
    b = batches[1]
    m3, _ = make(seed=6)
    m4, _ = make(seed=6)
    out3 = m3(b)
    calc_loss(out3, b.y).backward()
    d4 = dp.FlatGradDataParallel(m4, always_reduce=True, sync_bn=True)
    out4 = d4(b)
    calc_loss(out4, b.y).backward()
    dp.set_sync_bn(False)
    res["sync_logits_err"] = float((out3 - out4).abs().max() / out3.abs().max())
    gmax = max(float(p.grad.abs().max()) for p in m3.parameters() if p.grad is not None)
    worst = 0.0
    for (n3, p3), (n4, p4) in zip(m3.named_parameters(), m4.named_parameters()):
        assert (p3.grad is None) == (p4.grad is None), n3
        if p3.grad is not None:
            worst = max(worst, float((p3.grad - p4.grad).abs().max()) / max(float(p3.grad.abs().max()), 1e-3 * gmax))
    res["sync_grad_err"] = worst
    rb3 = {n: t for n, t in m3.named_buffers() if "running" in n}
    rb4 = {n: t for n, t in m4.named_buffers() if "running" in n}
    res["sync_running_err"] = max(float((rb3[n] - rb4[n]).abs().max() / rb3[n].abs().max().clamp_min(1e-6)) for n in rb3)
    # ---- 3. synchronised BatchNorm INSIDE ONE GRAPH (round 4): forward + backward with their statistics all-reduces, the
    # gradient all-reduce, scale and optimizer update captured together == the same steps run eagerly, bit for bit
    m6, o6 = make(seed=7)
    d6 = dp.FlatGradDataParallel(m6, always_reduce=True, sync_bn=True)
    for b_ in [batches[0]] * 2 + batches:
        d6.zero_grad()
        calc_loss(d6(b_), b_.y).backward()
        d6.reduce_gradients()
        o6.step()
    m7, o7 = make(seed=7)
    d7 = dp.FlatGradDataParallel(m7, always_reduce=True, sync_bn=True)
    static7 = synthetic_batch(4, 256, seed=60).to("cuda")
    step7 = GraphedTrainStep(m7, calc_loss, static7, optimizer=o7, warmup=2, reducer=d7)
    assert step7.capture_collectives and step7.graph_update is None
    for b_ in batches:
        step7(b_)
    dp.set_sync_bn(False)
    sd6, sd7 = m6.state_dict(), m7.state_dict()
    res["sync_graph_mismatches"] = [k for k in sd6 if not torch.equal(sd6[k], sd7[k])]
    print(json.dumps(res), flush=True)
    # graphs that hold captured collectives go before the communicator does
    del step7, step5, step
    import gc
    gc.collect()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
