import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd as dc, oracle
from deltaconv_amd.data import synthetic_batch
from deltaconv_amd.geometry import Graph, estimate_basis, build_grad_div
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def P(*a):
    torch.cuda.synchronize(); print(*a, flush=True)
b = synthetic_batch(B, 2048, seed=70, normals=False, outlier_frac=0.05, jitter=0.005).to("cuda")
torch.manual_seed(1)
m = dc.models.DeltaNetClassification(3, 15, conv_channels=[64, 64, 64, 128], num_neighbors=20, grad_regularizer=1e-2).to("cuda").train()
info = dc.models.deltanet_base._ptr_info(b); P("ptr", info[1], info[2])
g = Graph.knn(b.pos, 20, ptr_info=info); P("knn20")
g10 = Graph.knn(b.pos, 10, ptr_info=info); P("knn10")
n_, xb, yb = estimate_basis(b.pos, g10, orientation=b.pos); P("basis", float(n_.abs().sum()))
G, D = build_grad_div(b.pos, n_, xb, yb, g, b.batch, regularizer=1e-2); P("mls", float(G.coef.abs().max()), float(D.coef.abs().max()))
g.csc(); P("csc")
x = b.pos; v = G @ x; P("v0")
outs = []
for li, conv in enumerate(m.deltanet_base.convs):
    x, v = conv(x, v, G, D, g); P("conv", li, float(x.abs().mean()))
    outs.append(x)
h = m.lin_embedding(torch.cat(outs, 1)); P("embed")
loss = h.sum(); loss.backward(); P("backward ok")
m.zero_grad()
out = m(b); P("model fwd")
l = oracle.loss.calc_loss(out, b.y); l.backward(); P("model bwd", float(l))
