"""Data-parallel path on CPU: world_size-2 gloo processes.  Checks (1) FlatGradDataParallel's
single flat all-reduce gives every rank the average gradient = the gradient of the global-batch
mean loss, (2) Batch.shard partitions clouds with no overlap, (3) the oracle model trained on two
shards (BN-free comparison) matches a single-process run on the full batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deltaconv_amd.dp import FlatGradDataParallel
    from deltaconv_amd.data import synthetic_batch
    torch.manual_seed(100 + rank)                        # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    ddp = FlatGradDataParallel(net)
    full = synthetic_batch(4, 32, seed=5, num_classes=4)
    shard = full.shard(rank, world)
    ddp.zero_grad()
    out = ddp(shard.pos).view(shard.num_graphs, 32, 4).mean(1)
    torch.nn.functional.cross_entropy(out, shard.y).backward()
    ddp.reduce_gradients()
    q.put((rank, [p.detach().clone() for p in net.parameters()], [p.grad.clone() for p in net.parameters()],
           shard.pos.clone(), shard.y.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_matches_global_batch():
    world, port = 2, 29533
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w0, g0, pos0, y0), (_, w1, g1, pos1, y1) = res
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)                         # broadcast made the replicas identical
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, atol=1e-7)           # both ranks hold the same averaged gradient
    sys.path.insert(0, ROOT)
    from deltaconv_amd.data import synthetic_batch
    full = synthetic_batch(4, 32, seed=5, num_classes=4)
    assert torch.equal(torch.cat([pos0, pos1]), full.pos) and torch.equal(torch.cat([y0, y1]), full.y)
    net = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    for p, w in zip(net.parameters(), w0):
        p.data.copy_(w)
    out = net(full.pos).view(4, 32, 4).mean(1)
    torch.nn.functional.cross_entropy(out, full.y).backward()
    for p, g in zip(net.parameters(), g0):
        assert torch.allclose(p.grad, g, atol=1e-6)      # = gradient of the global-batch mean loss


def test_shard_per_point_labels_and_categories():
    sys.path.insert(0, ROOT)
    from deltaconv_amd.data import synthetic_batch
    full = synthetic_batch(4, 16, seed=6, per_point_labels=True, categories=16, num_classes=50)
    parts = [full.shard(r, 2) for r in range(2)]
    assert torch.equal(torch.cat([p.pos for p in parts]), full.pos)
    assert torch.equal(torch.cat([p.y for p in parts]), full.y)
    assert torch.equal(torch.cat([p.category for p in parts]), full.category)
    assert all(p.num_graphs == 2 and p.batch.max() == 1 and p.batch.min() == 0 for p in parts)
    assert parts[1].ptr.tolist() == [0, 16, 32]
