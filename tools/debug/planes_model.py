"""Debug: one train-mode step of the classification net (2 x 512 points) with and without pre-split weight planes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deltaconv_amd as dc
from deltaconv_amd.nn import fused
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.utils import calc_loss
def run(planes):
    fused.USE_WEIGHT_PLANES = planes
    fused._planes_reset()
    torch.manual_seed(3)
    m = dc.models.DeltaNetClassification(3, 40, num_neighbors=20).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.eval()
    b = synthetic_batch(2, 512, seed=9).to("cuda")
    out = m(b); loss = calc_loss(out, b.y); loss.backward()
    return out.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
o0, g0 = run(False)
for trial in range(2):
    o1, g1 = run(True)
    print("logits rel diff", float((o0 - o1).abs().max() / o0.abs().max()))
    bad = [(float((g0[n] - g1[n]).abs().max() / g0[n].abs().max().clamp_min(1e-12)), n) for n in g0]
    bad.sort(reverse=True)
    for e, n in bad[:8]: print(f"   {e:.3e} {n}")
