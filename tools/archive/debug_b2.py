import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
import deltaconv_amd as dc
from deltaconv_amd.nn import fused
from deltaconv_amd.data import synthetic_batch, Batch

def nod(m):
    for x in m.modules():
        if isinstance(x, torch.nn.Dropout): x.eval()
    return m

for B, N in ((2, 1024), (2, 512), (4, 1024)):
    b = synthetic_batch(B, N, seed=40)
    torch.manual_seed(1)
    ref64 = nod(oracle.models.DeltaNetClassification(3, 40, num_neighbors=20).double().train())
    sd = {k: v.float() for k, v in ref64.state_dict().items()}
    l64 = ref64(Batch(b.pos.double(), b.batch, b.norm.double(), None, b.y))
    oracle.loss.calc_loss(l64, b.y).backward()
    g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
    gmax = max(float(g.abs().max()) for g in g64.values())
    for own in (True, False):
        fused.USE_OWN_GEMM = own
        model = dc.models.DeltaNetClassification(3, 40, num_neighbors=20)
        model.load_state_dict(sd)
        model = nod(model.cuda().train())
        bd = b.to("cuda")
        ld = model(bd)
        oracle.loss.calc_loss(ld, bd.y).backward()
        errs = sorted(((float((p.grad.cpu().double() - g64[n]).abs().max()) / gmax, n) for n, p in model.named_parameters() if p.grad is not None), reverse=True)
        print(f"B={B} N={N} own={own}: logits err {float((ld.cpu().double()-l64).abs().max()/l64.abs().max()):.2e}; worst:", [(f"{e:.1e}", n) for e, n in errs[:4]])
