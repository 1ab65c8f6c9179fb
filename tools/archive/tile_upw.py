"""Lab: units (tile x 64-channel slab) per persistent tile workgroup (option 7; 0 = the launcher's choice: units / resident
workgroup slots, rounded up) on the forward apply family of a bench-like batch (B clouds x 1024 points, k = 20, C = 64), time per
launch under graph replay.   python tools/tile_upw.py 0,1,2,4 [B]"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltaconv_amd as dc
from deltaconv_amd._lib import lib
from deltaconv_amd.data import synthetic_batch
import bench

opt = lib.raw("dc_set_option")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
data = synthetic_batch(B, 1024, seed=3).to("cuda")
model = dc.models.DeltaNetClassification(3, 40, num_neighbors=20).cuda()
graph, grad, div = model.deltanet_base.build_operators(data)
vals = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,4").split(",")]
print("# us per launch: " + " ".join(f"{v:7d}" for v in vals))
rows = {}
for v in vals:
    if v == 99:                       # (lab) 99 = release the caching allocator's blocks first: fresh allocations again
        torch.cuda.empty_cache()
        v = 0
    opt(7, v)
    fam = bench.apply_roofline(graph, grad, div, 64, iters=400)["family"]
    for name, d in fam.items():
        rows.setdefault(name, []).append(d["us"])
opt(7, 0)
for name, r in rows.items():
    print(f"{name:16s} " + " ".join(f"{t:7.2f}" for t in r))
