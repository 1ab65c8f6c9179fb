"""Training loop with the semantics of the reference's experiments/train_modelnet.py (:20-142): SGD(lr 0.1,
momentum 0.9, wd 1e-4) + cosine annealing to 1e-3, label-smoothed cross entropy, train / evaluate per
epoch, state_dict checkpoints with the reference's key names -- on the MI355X path, data-parallel over
the GPUs of one node.  No dataset ships with this repo (the reference downloads ModelNet40), so by default
the clouds are synthetic; with `--data <ModelNet40 root>` (raw/<category>/<train|test>/*.off) the reference's
pipeline runs instead: NormalizeScale -> SamplePoints -> GeodesicFPS once, RandomScale + RandomTranslateGlobal
per access (train_modelnet.py:29-49), through `deltaconv_amd.datasets`.

    python examples/train_modelnet_like.py --epochs 3
    python examples/train_modelnet_like.py --data /data/ModelNet40 --epochs 50
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_modelnet_like.py
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as deltaconv                       # the drop-in: was `import deltaconv`
from deltaconv_amd.models import DeltaNetClassification
from deltaconv_amd.utils import calc_loss
from deltaconv_amd.dp import FlatGradDataParallel
from deltaconv_amd.data import synthetic_batch


def make_split(num_batches, batch_size, points, seed, device, num_classes=30):
    """Synthetic stand-in for the ModelNet40 loaders: the label IS the shape family of the cloud (30 families of closed surfaces
    r = 1 + a sin(m theta) cos(l phi), randomly rotated: deltaconv_amd.data.shape_family) -- a learnable task."""
    out = []
    for b in range(num_batches):
        data = synthetic_batch(batch_size, points, seed=seed + b, num_classes=num_classes, learnable=True)
        out.append(data.to(device))
    return out


def train_epoch(ddp, opt, loader):
    ddp.module.train()
    total, correct, count = 0.0, 0, 0
    for data in loader:
        ddp.zero_grad()
        out = ddp(data)
        loss = calc_loss(out, data.y)
        loss.backward()
        ddp.reduce_gradients()
        opt.step()
        total += float(loss) * data.num_graphs
        correct += int((out.argmax(1) == data.y).sum())
        count += data.num_graphs
    return total / count, correct / count


@torch.no_grad()
def evaluate(model, loader):
    model.eval()
    model.deltanet_base.cache_operators = True          # static test set: keep the operators
    correct = count = 0
    for data in loader:
        correct += int((model(data).argmax(1) == data.y).sum())
        count += data.num_graphs
    model.deltanet_base.cache_operators = False
    return correct / count


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--num_points", type=int, default=1024)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--grad_regularizer", type=float, default=0.001)
    ap.add_argument("--train_batches", type=int, default=8)
    ap.add_argument("--logdir", default="runs/modelnet_like")
    ap.add_argument("--data", default=None, help="ModelNet40 root with raw/<category>/<train|test>/*.off")
    ap.add_argument("--sampling_margin", type=int, default=8)
    args = ap.parse_args()

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(1)
    model = DeltaNetClassification(3, 40, num_neighbors=args.k, grad_regularizer=args.grad_regularizer).to(dev)
    ddp = FlatGradDataParallel(model)
    opt = deltaconv.optim.SGD(model.parameters(), lr=args.lr, momentum=0.9, weight_decay=1e-4)   # torch.optim.SGD, step = one launch
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, args.epochs, eta_min=0.001)
    if args.data is None:
        train = make_split(args.train_batches, args.batch_size, args.num_points, 1000 * (rank + 1), dev)
        test = make_split(4, args.batch_size, args.num_points, 777000, dev)
    else:
        import deltaconv_amd.transforms as T
        from deltaconv_amd.datasets import Compose, DataLoader, ModelNet
        pre = Compose((T.NormalizeScale(), T.SamplePoints(args.num_points * args.sampling_margin, include_normals=True),
                       T.GeodesicFPS(args.num_points)))
        aug = Compose((T.RandomScale((4 / 5, 5 / 4)), T.RandomTranslateGlobal(0.1)))
        tr = ModelNet(args.data, None, "40", True, transform=aug, pre_transform=pre)
        te = ModelNet(args.data, None, "40", False, pre_transform=pre)
        sampler = torch.utils.data.distributed.DistributedSampler(tr) if world > 1 else None
        on_dev = lambda loader: (b.to(dev) for b in loader)      # each rank collates and uploads its own shard
        train_loader = DataLoader(tr, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler, drop_last=True)
        test_loader = DataLoader(te, batch_size=args.batch_size, shuffle=False, drop_last=False)
        args.train_batches = len(train_loader)

        class _OnDevice:
            def __init__(self, loader): self.loader = loader
            def __iter__(self): return on_dev(self.loader)
        train, test = _OnDevice(train_loader), _OnDevice(test_loader)
    os.makedirs(args.logdir, exist_ok=True)
    for epoch in range(args.epochs):
        t0 = time.perf_counter()
        loss, acc = train_epoch(ddp, opt, train)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        test_acc = evaluate(model, test)
        sched.step()
        if rank == 0:
            print(json.dumps(dict(epoch=epoch, loss=round(loss, 4), train_acc=round(acc, 4), test_acc=round(test_acc, 4),
                                  clouds_per_s=round(world * args.train_batches * args.batch_size / dt, 1))))
            torch.save(model.state_dict(), os.path.join(args.logdir, "last.pt"))   # reference key names
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
