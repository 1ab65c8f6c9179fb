"""Debug: C4 (2 clouds) -- gradients at the inputs / outputs of the segmentation-head modules, HIP vs the fp32 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
import deltaconv_amd as dc
from deltaconv_amd.data import synthetic_batch
from tests.test_gpu_configs import CONFIGS, _build, _no_dropout
B, N, k, normals, kind, kw, bkw = CONFIGS["C4_shapenet"]
b = synthetic_batch(2, N, seed=73, normals=normals, **dict(bkw))
model = _build(kind, kw, k)
ref = oracle.models.DeltaNetSegmentation(num_neighbors=k, **kw)
ref.load_state_dict(model.state_dict())
ref = _no_dropout(ref.train())
model = _no_dropout(model.to("cuda").train())
def tap(seq, store):
    hs = []
    for i, m in enumerate(seq):
        def fh(mod, inp, out, i=i):
            store[f"out{i}"] = out.detach().cpu()
            out.register_hook(lambda g, i=i: store.__setitem__(f"dout{i}", g.detach().cpu().clone()))
        hs.append(m.register_forward_hook(fh))
    return hs
sh, so = {}, {}
tap(model.segmentation_head, sh); tap(ref.segmentation_head, so)
bd = b.to("cuda")
ld = model(bd); oracle.loss.calc_loss(ld, bd.y, smoothing=False).backward()
lo = ref(b); oracle.loss.calc_loss(lo, b.y, smoothing=False).backward()
def rel(a, c): return float((a.double() - c.double()).abs().max() / c.double().abs().max())
for key in sorted(so):
    if key in sh:
        print(key, tuple(so[key].shape), f"{rel(sh[key], so[key]):.2e}")
for (n1, p1), (n2, p2) in zip(model.named_parameters(), ref.named_parameters()):
    if "segmentation_head" in n1 and p2.grad is not None:
        print(n1, f"{rel(p1.grad.cpu(), p2.grad):.2e}")

# --- is the HIP backward of head.2 consistent with ITS OWN inputs?  recompute the block in fp64 from HIP's tensors
print("---- head.2 recomputed in fp64 from the HIP tensors")
blk = model.segmentation_head[2][0]
x1 = sh["out1"].double().requires_grad_(True)           # output of Dropout-1 = input of head.2 (HIP)
W = blk[0].weight.detach().cpu().double().requires_grad_(True)
ga = blk[1].bn.weight.detach().cpu().double().requires_grad_(True); be = blk[1].bn.bias.detach().cpu().double().requires_grad_(True)
h = x1 @ W.t()
z = torch.nn.functional.batch_norm(h, None, None, ga, be, True, 0.1, 1e-5)
y = torch.nn.functional.leaky_relu(z, 0.2)
y.backward(sh["dout2"].double())
print("out2 vs recompute", f"{rel(sh['out2'], y.detach()):.2e}")
print("dout1 (HIP) vs recompute", f"{rel(sh['dout1'], x1.grad):.2e}", " oracle dout1 vs recompute", f"{rel(so['dout1'], x1.grad):.2e}")
print("dW", f"{rel(blk[0].weight.grad.cpu(), W.grad):.2e}", "dgamma", f"{rel(blk[1].bn.weight.grad.cpu(), ga.grad):.2e}",
      "dbeta", f"{rel(blk[1].bn.bias.grad.cpu(), be.grad):.2e}")
hv = h.detach()
print("column std of h: min", float(hv.std(0).min()), "median", float(hv.std(0).median()), " |z| < 1e-5 count", int((z.detach().abs() < 1e-5).sum()))
d = (sh["dout1"].double() - x1.grad).abs()
print("worst rows of dout1 error:", torch.topk(d.max(1).values, 5))
print("rows identical to another row in x1:", int((torch.unique(sh["out1"], dim=0).shape[0])), "unique of", sh["out1"].shape[0])

print("---- which input moves dout1?")
def block_dx(x1_, dy_):
    x = x1_.double().requires_grad_(True)
    hh = x @ W.detach().t()
    zz = torch.nn.functional.batch_norm(hh, None, None, ga.detach(), be.detach(), True, 0.1, 1e-5)
    torch.nn.functional.leaky_relu(zz, 0.2).backward(dy_.double())
    return x.grad, zz.detach()
dx_hh, z_h = block_dx(sh["out1"], sh["dout2"])
dx_oo, z_o = block_dx(so["out1"], so["dout2"])
dx_ho, _ = block_dx(sh["out1"], so["dout2"])
dx_oh, _ = block_dx(so["out1"], sh["dout2"])
print("recompute(hip x, hip dy) vs recompute(orc x, orc dy)", f"{rel(dx_hh, dx_oo):.2e}")
print("  swap dy only: (hip x, orc dy) vs (hip x, hip dy)", f"{rel(dx_ho, dx_hh):.2e}", "  swap x only: (orc x, hip dy) vs (hip x, hip dy)", f"{rel(dx_oh, dx_hh):.2e}")
flip = ((z_h > 0) != (z_o > 0))
print("sign flips of z between the two inputs:", int(flip.sum()), "of", z_h.numel(), " rows with flips:", int(flip.any(1).sum()))
dyo = so["dout2"].double()
print("sum |dy| on flipped elements / max column sum |dy|:", float((dyo.abs() * flip).sum(0).max() / dyo.abs().sum(0).max()))
print("x1 diff: max", float((sh['out1'].double() - so['out1'].double()).abs().max()), " max |x1|", float(so['out1'].abs().max()))
