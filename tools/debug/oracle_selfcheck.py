import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, oracle
from deltaconv_amd.data import synthetic_batch
kw = dict(in_channels=3, num_classes=50, categorical_vector=True)
bkw = dict(dup_frac=0.03, per_point_labels=True, categories=16, num_classes=50)
b = synthetic_batch(2, 2048, seed=73, normals=True, **bkw)
torch.manual_seed(1)
ref = oracle.models.DeltaNetSegmentation(num_neighbors=20, **kw).train()
for m in ref.modules():
    if isinstance(m, torch.nn.Dropout): m.eval()
so = {}
for i, m in enumerate(ref.segmentation_head):
    def fh(mod, inp, out, i=i):
        so[f"in{i}"] = inp[0].detach().clone()
        so[f"out{i}"] = out.detach().clone()
        out.register_hook(lambda g, i=i: so.__setitem__(f"dout{i}", g.detach().clone()))
    m.register_forward_hook(fh)
lo = ref(b); oracle.loss.calc_loss(lo, b.y, smoothing=False).backward()
def rel(a, c): return float((a.double() - c.double()).abs().max() / c.double().abs().max())
blk = ref.segmentation_head[2][0]
x1 = so["in2"].double().requires_grad_(True)
W = blk[0].weight.detach().double().requires_grad_(True)
ga = blk[1].bn.weight.detach().double().requires_grad_(True); be = blk[1].bn.bias.detach().double().requires_grad_(True)
h = x1 @ W.t()
z = torch.nn.functional.batch_norm(h, None, None, ga, be, True, 0.1, 1e-5)
y = torch.nn.functional.leaky_relu(z, 0.2)
y.backward(so["dout2"].double())
print("out2 vs recompute", rel(so["out2"], y.detach()), "in2 == out1", torch.equal(so["in2"], so["out1"]), "out0 is out1", torch.equal(so["out0"], so["out1"]))
print("oracle dout1 vs recompute", rel(so["dout1"], x1.grad), " dout0 vs recompute", rel(so["dout0"], x1.grad))
print("dW", rel(blk[0].weight.grad, W.grad), "dbeta", rel(blk[1].bn.bias.grad, be.grad))
print("threads", torch.get_num_threads())
# plain torch: batch_norm backward on CPU, [R, C] input, fp32 vs a manual fp64 formula, for several thread counts
for nt in (1, 8, 32, torch.get_num_threads()):
    torch.set_num_threads(nt)
    for R in (4096, 8192):
        g = torch.Generator().manual_seed(R)
        h = torch.randn(R, 256, generator=g).requires_grad_(True)
        ga = torch.randn(256, generator=g).requires_grad_(True); be = torch.randn(256, generator=g).requires_grad_(True)
        dy = torch.randn(R, 256, generator=g)
        z = torch.nn.functional.batch_norm(h, None, None, ga, be, True, 0.1, 1e-5); z.backward(dy)
        h64 = h.detach().double(); mu = h64.mean(0); var = h64.var(0, unbiased=False); inv = (var + 1e-5).rsqrt()
        xh = (h64 - mu) * inv; dz = dy.double()
        dh = ga.detach().double() * inv * (dz - dz.mean(0) - xh * (dz * xh).mean(0))
        print("threads", nt, "R", R, "dh", rel(h.grad, dh), "dbeta", rel(be.grad, dz.sum(0)), "dgamma", rel(ga.grad, (dz * xh).sum(0)))
