"""Experiment (one GPU, RCCL group of one rank): can the synchronised-BatchNorm step -- collectives inside forward and
backward -- and the gradient all-reduce be captured into ONE HIP graph?  Segmentation nets at the per-rank shapes of the
8-GPU BASELINE configurations (C5: 1 cloud x 4096 points, k = 30; C4: 2 clouds x 2048 points).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29581 tools/dp_syncbn_graph.py"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.utils import calc_loss
from deltaconv_amd.dp import FlatGradDataParallel, set_sync_bn
from deltaconv_amd.graph_step import GraphedTrainStep

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
CFG = {
    "C5 per rank: 1 x 4096, k=30": (1, 4096, 30, dict(in_channels=3, num_classes=8, conv_channels=[128] * 8, mlp_depth=1, embedding_size=512),
                                    dict(per_point_labels=True, num_classes=8)),
    "C4 per rank: 2 x 2048, k=20": (2, 2048, 20, dict(in_channels=3, num_classes=50, categorical_vector=True),
                                    dict(dup_frac=0.03, per_point_labels=True, categories=16, num_classes=50)),
}


def timed(fn, steps=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for name, (B, N, k, kw, bkw) in CFG.items():
    res = {}
    for mode in ("single graph, per-rank BN", "sync_bn eager", "sync_bn one graph (collectives captured)"):
        torch.manual_seed(1)
        set_sync_bn(False)
        model = dc.models.DeltaNetSegmentation(num_neighbors=k, **kw).cuda().train()
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)
        b = synthetic_batch(B, N, seed=5, **bkw).to("cuda")
        loss_fn = lambda out, y: calc_loss(out, y, smoothing=False)
        try:
            if mode.startswith("single"):
                g = GraphedTrainStep(model, loss_fn, b, optimizer=opt)
                res[mode] = timed(lambda: g())
            else:
                ddp = FlatGradDataParallel(model, always_reduce=True, sync_bn=True)

                def eager():
                    ddp.zero_grad()
                    loss_fn(ddp(b), b.y).backward()
                    ddp.reduce_gradients()
                    opt.step()
                if mode == "sync_bn eager":
                    res[mode] = timed(eager)
                else:
                    for _ in range(3):
                        eager()
                    torch.cuda.synchronize()
                    ddp.zero_grad()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        loss = loss_fn(ddp(b), b.y)
                        loss.backward()
                        ddp.reduce_gradients()
                        opt.step()
                    torch.cuda.synchronize()
                    l0 = float(loss)
                    res[mode] = timed(graph.replay)
                    res[mode + " loss"] = (l0, float(loss))
        except Exception as e:
            import traceback
            res[mode] = "FAILED: " + repr(e)[:120]
            print("".join(traceback.format_exc().splitlines(True)[-14:]), flush=True)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        torch.cuda.synchronize()
    print(name, res, flush=True)
dist.destroy_process_group()
