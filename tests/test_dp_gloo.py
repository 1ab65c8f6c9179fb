"""Data-parallel path on CPU: world_size-2 gloo processes.  Checks (1) FlatGradDataParallel's
single flat all-reduce gives every rank the average gradient = the gradient of the global-batch
mean loss, (2) Batch.shard partitions clouds with no overlap, (3) the WHOLE model (the oracle's
DeltaNetClassification, every BatchNorm converted to the synchronised form of deltaconv_amd/dp.py) stepped on two
shards with sync_bn=True reproduces a single process on the full batch: logits, running statistics, gradients."""
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
    # numpy payloads: pickled by value (torch tensors travel as file descriptors the exiting worker may close first)
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()], [p.grad.numpy().copy() for p in net.parameters()],
           shard.pos.numpy().copy(), shard.y.numpy().copy()))
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
    tt = lambda t: [torch.from_numpy(a) for a in t] if isinstance(t, list) else torch.from_numpy(t)
    (_, w0, g0, pos0, y0), (_, w1, g1, pos1, y1) = [(r[0],) + tuple(tt(a) for a in r[1:]) for r in res]
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


def _oracle_model():
    import oracle
    torch.manual_seed(1)
    m = oracle.models.DeltaNetClassification(3, 10, conv_channels=(16, 16, 32), num_neighbors=8).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def _sync_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from deltaconv_amd.dp import FlatGradDataParallel, convert_sync_batchnorm
    from deltaconv_amd.data import synthetic_batch
    model = convert_sync_batchnorm(_oracle_model())
    ddp = FlatGradDataParallel(model, sync_bn=True)
    full = synthetic_batch(4, 96, seed=9, num_classes=10)
    shard = full.shard(rank, world)
    ddp.zero_grad()
    out = ddp(shard)
    oracle.loss.calc_loss(out, shard.y).backward()
    ddp.reduce_gradients()
    # numpy payloads: pickled by value (torch tensors travel as file descriptors the exiting worker may close first)
    q.put((rank, out.detach().numpy().copy(),
           {n: p.grad.numpy().copy() for n, p in model.named_parameters() if p.grad is not None},
           {n: b.numpy().copy() for n, b in model.named_buffers() if "running" in n}))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_whole_model_matches_single_process():
    world, port = 2, 29547
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    t = torch.from_numpy
    res = [(r, t(o), {n: t(g) for n, g in gs.items()}, {n: t(b) for n, b in bs.items()}) for r, o, gs, bs in res]
    sys.path.insert(0, ROOT)
    import oracle
    from deltaconv_amd.data import synthetic_batch
    ref = _oracle_model()
    full = synthetic_batch(4, 96, seed=9, num_classes=10)
    out = ref(full)
    oracle.loss.calc_loss(out, full.y).backward()
    logits = torch.cat([res[0][1], res[1][1]])
    assert torch.allclose(logits, out, atol=2e-4, rtol=1e-4)           # global statistics: every cloud sees the same BN
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert n not in res[0][2]
            continue
        for r in range(world):                                          # both ranks hold the same reduced gradient
            err = float((res[r][2][n] - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-3 * gmax)
            assert err < 2e-3, (n, r, err)
    for n, b in ref.named_buffers():
        if "running" in n:
            assert torch.allclose(res[0][3][n], b, atol=1e-5, rtol=1e-4), n
            assert torch.equal(res[0][3][n], res[1][3][n]), n           # replicas stay identical


def _rows_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deltaconv_amd import dp
    from deltaconv_amd.nn import fused
    bn = torch.nn.BatchNorm1d(8).train()
    res = {}

    def refused(rows):
        try:
            fused.check_bn_rows(bn, rows)
            return False
        except ValueError:
            return True
    res["per_rank_one_row"] = refused(1)                    # per-rank statistics: a single row has no variance
    dp.set_sync_bn(True)
    res["sync_one_row"] = refused(1)                        # global statistics over 2 ranks: allowed
    res["sync_two_rows"] = refused(2)
    dp.set_sync_bn(False)
    res["off_again"] = refused(1)
    bn.eval()
    res["eval_one_row"] = refused(1)                        # running statistics: any row count
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_check_bn_rows_under_synchronised_statistics():
    """ADVICE (round 2, medium): one row per rank is legal when BatchNorm statistics span the ranks (the categorical head
    of the segmentation net with one cloud per rank), and still refused with per-rank statistics, as the reference's
    functional.batch_norm does (deltaconv/nn/nonlin.py:29-30)."""
    world, port = 2, 29561
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r] == {"per_rank_one_row": True, "sync_one_row": False, "sync_two_rows": False, "off_again": True,
                          "eval_one_row": False}, res[r]
