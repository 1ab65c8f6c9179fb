"""Throughput of every BASELINE.json configuration (SURVEY.md section 8 sizes) on one MI355X: full train
step (forward + loss + backward + optimizer update, train-mode BN/Dropout, operators rebuilt every step)
and eval-mode forward, eager launches and HIP-graph replay.  Synthetic inputs of the configured shape.

    python tools/bench_configs.py [--steps 20] > profiles/<round>_configs.txt

bench.py stays the headline (C2); this table is the measured context for the other rows."""
import argparse, os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.utils import calc_loss
from deltaconv_amd.graph_step import GraphedTrainStep

from deltaconv_amd import configs as C

# name: (B, N, k, normals, kind, model kwargs, batch kwargs, optimizer) -- the table of deltaconv_amd/configs.py (+ C1)
CONFIGS = {"C1 modelnet40 B=2 (min. train batch)": (2, 1024, 20, True, "cls", dict(in_channels=3, num_classes=40), {}, "sgd")}
_NAMES = {"C2": "C2 modelnet40 B=32", "C3": "C3 scanobjectnn B=32 N=2048 (no normals)", "C4": "C4 shapenet B=16 N=2048",
          "C5": "C5 shapeseg B=8 N=4096 k=30"}
for _key, _c in C.CONFIGS.items():
    CONFIGS[_NAMES[_key]] = (_c["B"], _c["N"], _c["k"], _c["normals"], _c["kind"], _c["model"], _c["batch"], _c["optimizer"])


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--only", default="", help="substring filter on the config name")
    ap.add_argument("--eager-only", action="store_true", help="only the eager train step (for rocprofv3 runs)")
    args = ap.parse_args()
    dev = "cuda"
    print(f"# {torch.cuda.get_device_name(0)}, torch {torch.__version__}; ms per step / clouds per second")
    print(f"{'config':44s} {'train eager':>18s} {'train graph':>18s} {'eval fwd':>18s}")
    for name, (B, N, k, normals, kind, kw, bkw, optname) in CONFIGS.items():
        if args.only not in name:
            continue
        torch.manual_seed(1)
        cls = dc.models.DeltaNetSegmentation if kind == "seg" else dc.models.DeltaNetClassification
        model = cls(num_neighbors=k, **kw).to(dev).train()
        batches = [synthetic_batch(B, N, seed=200 + i, normals=normals, **bkw).to(dev) for i in range(3)]
        smooth = kind != "seg"
        loss_fn = lambda out, y: calc_loss(out, y, smoothing=smooth)
        if optname == "sgd":     # train_modelnet.py:67 / train_shapeseg.py:82
            opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)
        else:
            opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True, capturable=True)
        it = [0]

        def eager():
            b = batches[it[0] % 3]; it[0] += 1
            for p in model.parameters():
                p.grad = None
            loss_fn(model(b), b.y).backward()
            opt.step()
        t_eager = timed(eager, args.steps)
        if args.eager_only:
            print(f"{name:44s} {t_eager:8.3f} ms", flush=True)
            continue
        static = synthetic_batch(B, N, seed=199, normals=normals, **bkw).to(dev)
        try:
            g = GraphedTrainStep(model, loss_fn, static, optimizer=opt)
            def graphed():
                g(batches[it[0] % 3]); it[0] += 1
            t_graph = timed(graphed, args.steps)
        except Exception as e:          # report, do not hide
            t_graph = float("nan"); print("# graph capture failed:", repr(e)[:200])
        model.eval()
        with torch.no_grad():
            t_eval = timed(lambda: model(batches[0]), args.steps)
        f = lambda t: f"{t:8.3f} / {B / t * 1e3:7.0f}"
        print(f"{name:44s} {f(t_eager):>18s} {f(t_graph):>18s} {f(t_eval):>18s}", flush=True)
        del model, opt, batches
        torch.cuda.empty_cache()


main()
