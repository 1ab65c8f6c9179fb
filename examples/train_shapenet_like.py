"""Part-segmentation training loop with the semantics of the reference's experiments/train_shapenet.py (:20-189):
DeltaNetSegmentation(conv_channels [64,128,256], mlp_depth 2, embedding 1024, categorical vector),
SGD(100 * lr, momentum 0.9, wd 1e-4) + cosine annealing to lr, plain mean cross entropy over the points,
mean part-IoU per shape (experiments/utils.py:27-51) after every epoch, state_dict checkpoint with the
reference's key names -- on the MI355X path, data-parallel over the GPUs of one node.

    python examples/train_shapenet_like.py --epochs 2                           # synthetic clouds
    python examples/train_shapenet_like.py --data /data/ShapeNet --epochs 200   # raw/<synset>/*.txt + train_test_split/
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_shapenet_like.py
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as deltaconv                       # the drop-in: was `import deltaconv`
from deltaconv_amd.models import DeltaNetSegmentation
from deltaconv_amd.utils import calc_loss, calc_shape_IoU
from deltaconv_amd.dp import FlatGradDataParallel
from deltaconv_amd.data import synthetic_batch


def shapenet_model(args, num_classes):
    """train_shapenet.py:76-89."""
    return DeltaNetSegmentation(in_channels=3, num_classes=num_classes, conv_channels=[64, 128, 256], mlp_depth=2,
                                embedding_size=1024, num_neighbors=args.k, grad_regularizer=args.grad_regularizer,
                                grad_kernel_width=args.grad_kernel, categorical_vector=True)


def synthetic_split(num_batches, args, seed, device):
    """Stand-in for the ShapeNet loaders: 16 categories, per-point labels inside the category's part range."""
    starts = [0, 4, 6, 8, 12, 16, 19, 22, 24, 28, 30, 36, 38, 41, 44, 47]
    parts = [4, 2, 2, 4, 4, 3, 3, 2, 4, 2, 6, 2, 3, 3, 3, 3]
    out = []
    g = torch.Generator().manual_seed(seed)
    for b in range(num_batches):
        data = synthetic_batch(args.batch_size, args.num_points, seed=seed + b, per_point_labels=True, categories=16,
                               num_classes=50)
        cat = torch.randint(0, 16, (args.batch_size,), generator=g)
        data.category = torch.nn.functional.one_hot(cat, 16).float()
        height = data.pos[:, 2].view(args.batch_size, -1)                      # the part = a height band of the shape
        band = ((height - height.min(1, keepdim=True).values) / (height.max(1, keepdim=True).values
                - height.min(1, keepdim=True).values + 1e-9) * torch.tensor(parts)[cat].view(-1, 1)).long()
        band = torch.minimum(band, (torch.tensor(parts)[cat] - 1).view(-1, 1))
        data.y = (band + torch.tensor(starts)[cat].view(-1, 1)).reshape(-1)
        out.append(data.to(device))
    return out


def train_epoch(ddp, opt, loader):
    ddp.module.train()
    total, count = 0.0, 0
    for data in loader:
        ddp.zero_grad()
        loss = calc_loss(ddp(data), data.y, smoothing=False)
        loss.backward()
        ddp.reduce_gradients()
        opt.step()
        total += float(loss) * data.num_graphs
        count += data.num_graphs
    return total / count


@torch.no_grad()
def evaluate(model, loader):
    """Mean part IoU per shape (train_shapenet.py:137-160)."""
    model.eval()
    ious = []
    for data in loader:
        pred = model(data).argmax(1).view(data.num_graphs, -1).cpu().numpy()
        true = data.y.view(data.num_graphs, -1).cpu().numpy()
        label = data.category.argmax(1).cpu().numpy()
        ious += calc_shape_IoU(pred, true, label, None)
    return float(np.mean(ious))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--num_points", type=int, default=2048)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--grad_regularizer", type=float, default=0.001)
    ap.add_argument("--grad_kernel", type=float, default=1)
    ap.add_argument("--train_batches", type=int, default=4)
    ap.add_argument("--logdir", default="runs/shapenet_like")
    ap.add_argument("--data", default=None, help="ShapeNet part root (raw/<synset>/*.txt, raw/train_test_split/*.json)")
    args = ap.parse_args()

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(1)
    model = shapenet_model(args, 50).to(dev)
    ddp = FlatGradDataParallel(model)
    opt = deltaconv.optim.SGD(model.parameters(), lr=100 * args.lr, momentum=args.momentum, weight_decay=1e-4)   # torch.optim.SGD, step = one launch
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, args.epochs, eta_min=args.lr)
    if args.data is None:
        train = synthetic_split(args.train_batches, args, 1000 * (rank + 1), dev)
        test = synthetic_split(2, args, 777000, dev)
    else:
        import deltaconv_amd.transforms as T
        from deltaconv_amd.datasets import Compose, DataLoader, ShapeNet
        pre = Compose((T.NormalizeScale(), T.GeodesicFPS(args.num_points)))                    # train_shapenet.py:30-33
        aug = Compose((T.RandomScale((2 / 3, 3 / 2)), T.RandomTranslateGlobal(0.2)))            # train_shapenet.py:35-38
        tr = ShapeNet(args.data, split="trainval", transform=aug, pre_transform=pre)
        te = ShapeNet(args.data, split="test", pre_transform=pre)
        sampler = torch.utils.data.distributed.DistributedSampler(tr) if world > 1 else None

        class _OnDevice:
            def __init__(self, loader):
                self.loader = loader

            def __iter__(self):
                return (b.to(dev) for b in self.loader)
        train = _OnDevice(DataLoader(tr, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler, drop_last=True))
        test = _OnDevice(DataLoader(te, batch_size=args.batch_size, shuffle=False, drop_last=False))
        args.train_batches = len(train.loader)
    os.makedirs(args.logdir, exist_ok=True)
    for epoch in range(args.epochs):
        t0 = time.perf_counter()
        loss = train_epoch(ddp, opt, train)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        miou = evaluate(model, test)
        sched.step()
        if rank == 0:
            print(json.dumps(dict(epoch=epoch, loss=round(loss, 4), test_mean_iou=round(miou, 4),
                                  clouds_per_s=round(world * args.train_batches * args.batch_size / dt, 1))))
            torch.save(model.state_dict(), os.path.join(args.logdir, "last.pt"))   # reference key names
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
