"""Per-parameter gradient-summary errors of the seg golden fixture with csrc/edge2.hip on / off (round 5 debugging)."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
import deltaconv_amd as dc
from deltaconv_amd.data import Batch
from deltaconv_amd.nn import fused, layer as L
from tests.helpers import load_golden
from tests.test_gpu_model import MODELS, _model, _no_dropout, param_summaries

name = sys.argv[1] if len(sys.argv) > 1 else "model_seg_B2_N256_k20"
kind, kw, normals = MODELS[name]
g = load_golden(name)
res = {}
for use in (True, False):
    fused.USE_EDGE2 = use
    model = _no_dropout(_model(kind, kw, g["k"], g["lam"]).to("cuda").train())
    data = Batch(g["pos"], g["batch"], g["normal"] if normals else None, None, g["y"], g["category"] if "category" in g else None).to("cuda")
    L.SLOT_TAP[0] = []
    logits = model(data)
    slots = [s.cpu() for s in L.SLOT_TAP[0]]
    L.SLOT_TAP[0] = None
    oracle.loss.calc_loss(logits, data.y, smoothing=(kind != "seg")).backward()
    names, norms, dots = param_summaries(model)
    gn = g["gnorm_f64"].numpy(); den = gn + 1e-3 * gn.max()
    en = np.abs(np.array(norms) - gn) / den
    ed = np.abs(np.array(dots) - g["gdot_f64"].numpy()) / den
    res[use] = (names, en, ed, slots, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    print(f"edge2={use}: logits err {float((logits.cpu().double() - g['logits_f64']).abs().max() / g['logits_f64'].abs().max()):.2e}  max norm err {en.max():.2e}  max dot err {ed.max():.2e}")
names, en1, ed1, s1, g1 = res[True]
_, en0, ed0, s0, g0 = res[False]
for i, n in enumerate(names):
    if max(en1[i], ed1[i], en0[i], ed0[i]) > 5e-4:
        print(f"{n:60s} edge2 norm {en1[i]:.2e} dot {ed1[i]:.2e} | materialised norm {en0[i]:.2e} dot {ed0[i]:.2e}")
print("slot differences per layer:", [int((a != b).sum()) for a, b in zip(s1, s0)], "of", [a.numel() for a in s1])
for n in g1:
    d = float((g1[n] - g0[n]).abs().max() / g0[n].abs().max().clamp(min=1e-30))
    if d > 1e-3:
        print(f"grad {n:60s} edge2 vs materialised {d:.2e}")
