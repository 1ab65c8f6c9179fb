"""Runs under `python -m torch.distributed.run --nproc-per-node 2` on the one GPU of the test box, backend gloo (two
processes cannot form an RCCL group on one device; gloo reduces GPU tensors through the host): each rank holds half of the
batch, BatchNorm statistics are synchronised (deltaconv_amd/dp.py: split statistics kernels + all-reduced fp64 sums), the
gradients are averaged by the flat all-reduce, one SGD step.  Rank 0 then runs ONE process on the full batch and compares.
Prints one JSON line on rank 0."""
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
from deltaconv_amd.data import synthetic_batch, Batch
from deltaconv_amd.utils import calc_loss


def make(seed=5):
    torch.manual_seed(seed)
    m = dc.models.DeltaNetClassification(3, 40, num_neighbors=20).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.eval()
    return m, torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)


def shard(b, r, world):
    n = b.pos.shape[0] // b.num_graphs
    per = b.num_graphs // world
    sl = slice(r * per * n, (r + 1) * per * n)
    return Batch(b.pos[sl], b.batch[sl] - r * per, b.norm[sl], None, b.y[r * per:(r + 1) * per], None, per)


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    full = synthetic_batch(4, 256, seed=61).to("cuda")
    m, opt = make()
    d = dp.FlatGradDataParallel(m, sync_bn=True)
    mine = shard(full, rank, world)
    d.zero_grad()
    out = d(mine)
    # per-rank loss = mean over the local clouds; the flat all-reduce averages over ranks = the global-batch mean
    calc_loss(out, mine.y).backward()
    d.reduce_gradients()
    grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    opt.step()
    dp.set_sync_bn(False)
    # replicas: parameters identical on both ranks after the step
    flat = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    logits = [torch.empty(out.shape, dtype=out.dtype) for _ in range(world)]
    dist.all_gather(logits, out.detach().cpu())
    if rank == 0:
        m1, o1 = make()
        o = m1(full)
        calc_loss(o, full.y).backward()
        res = {"replicas_equal": bool(all(torch.equal(other[0], t) for t in other))}
        res["logits_err"] = float((torch.cat(logits).cuda() - o).abs().max() / o.abs().max())
        worst = 0.0
        for n, p in m1.named_parameters():
            if p.grad is None:
                assert n not in grads, n
                continue
            worst = max(worst, float((grads[n] - p.grad).abs().sum() / p.grad.abs().sum().clamp_min(1e-12)))
        res["grad_l1_err"] = worst
        rb, rb1 = ({n: t for n, t in mm.named_buffers() if "running" in n} for mm in (m, m1))
        res["running_err"] = max(float((rb[n] - rb1[n]).abs().max() / rb1[n].abs().max().clamp_min(1e-6)) for n in rb1)
        o1.step()
        res["params_after_step_err"] = max(float((p - q).abs().max() / q.abs().max().clamp_min(1e-6))
                                           for p, q in zip(m.parameters(), m1.parameters()))
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
