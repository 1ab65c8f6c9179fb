"""Same-box A/B of a PYTHON switch (module attribute, e.g. deltaconv_amd.nn.fused.USE_BIAS_ACT) on the BASELINE configurations
(graph-replayed train step).   python tools/ab_flag_cfg.py <module> <attr> [config substring ...]"""
import os, sys, time, importlib.util
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltaconv_amd as dc
from deltaconv_amd._lib import lib
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.utils import calc_loss
from deltaconv_amd.graph_step import GraphedTrainStep
src = open(os.path.join(ROOT, "tools", "bench_configs.py")).read()
from deltaconv_amd import configs as _C
ns = {"C": _C}
exec(src[src.index("CONFIGS = {"):src.index("def timed(")], ns)
import importlib
mod, attr = importlib.import_module(sys.argv[1]), sys.argv[2]
a, b = False, True
subs = sys.argv[3:] or ["C3", "C4", "C5"]
for name, (B, N, k, normals, kind, kw, bkw, optname) in ns["CONFIGS"].items():
    if not any(s in name for s in subs):
        continue
    torch.manual_seed(1)
    cls = dc.models.DeltaNetSegmentation if kind == "seg" else dc.models.DeltaNetClassification
    model = cls(num_neighbors=k, **kw).cuda().train()
    batches = [synthetic_batch(B, N, seed=200 + i, normals=normals, **bkw).to("cuda") for i in range(3)]
    static = synthetic_batch(B, N, seed=199, normals=normals, **bkw).to("cuda")
    smooth = kind != "seg"
    loss_fn = lambda out, y: calc_loss(out, y, smoothing=smooth)
    opt = (torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True) if optname == "sgd"
           else torch.optim.Adam(model.parameters(), lr=5e-4, fused=True, capturable=True))
    res = {a: [], b: []}
    for r in range(2):
        for v in (a, b):
            setattr(mod, attr, v)
            g = GraphedTrainStep(model, loss_fn, static, optimizer=opt)
            for i in range(4):
                g(batches[i % 3])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20):
                g(batches[i % 3])
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / 20 * 1e3)
            del g
    setattr(mod, attr, True)
    print(f"{name:44s} {sys.argv[1]}.{attr}: {a} -> " + " ".join(f"{t:.3f}" for t in res[a]) + f"   {b} -> " + " ".join(f"{t:.3f}" for t in res[b]), flush=True)
    del model, opt, batches, static
    torch.cuda.empty_cache()
