"""GPU parity of the fused MLP-stream blocks (GEMM + HIP BatchNorm/activation/vector non-linearity)
vs the oracle's torch-CPU modules carrying the same weights: forward, input/parameter gradients,
running statistics, eval mode; plus the reference's own nn property tests
(test/nn/test_mlp.py, test/nn/test_nonlin.py)."""
import pytest
import torch

import oracle
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(make_ours, make_ref):
    torch.manual_seed(3)
    ref = make_ref()
    ours = make_ours()
    ours.load_state_dict(ref.state_dict())
    with torch.no_grad():                       # non-trivial affine parameters
        for (n1, p1), (n2, p2) in zip(ours.named_parameters(), ref.named_parameters()):
            if n1.endswith("bn.weight"):
                v = torch.linspace(0.4, 1.6, p1.numel()); v[::5] *= -1
                p1.copy_(v); p2.copy_(v)
            if n1.endswith("bn.bias") or n1.endswith(".bias"):
                v = torch.linspace(-0.3, 0.3, p1.numel())
                p1.copy_(v); p2.copy_(v)
    return ours.to(DEV), ref


def _compare(ours, ref, x, train=True, tol=2e-4):
    ours.train(train); ref.train(train)
    xo = x.clone().requires_grad_(True)
    xd = x.to(DEV).requires_grad_(True)
    yo, yd = ref(xo), ours(xd)
    assert rel_err(yd, yo) < tol
    w = torch.randn_like(yo)
    yo.backward(w); yd.backward(w.to(DEV))
    assert rel_err(xd.grad, xo.grad) < 5 * tol
    for (n1, p1), (n2, p2) in zip(ours.named_parameters(), ref.named_parameters()):
        if p2.grad is None:
            assert p1.grad is None or float(p1.grad.abs().max()) == 0, n1
        else:
            assert rel_err(p1.grad, p2.grad) < 5 * tol, n1
    for (n1, b1), (n2, b2) in zip(ours.named_buffers(), ref.named_buffers()):
        if b2.dtype.is_floating_point:
            assert rel_err(b1, b2) < tol, n1
        else:
            assert int(b1) == int(b2), n1


@pytest.mark.parametrize("channels,rows", [((16, 32), 1000), ((12, 64, 64, 24), 4096), ((5, 7), 333)])
def test_mlp_blocks(channels, rows):
    import deltaconv_amd as dc
    ours, ref = _pair(lambda: dc.nn.MLP(channels), lambda: oracle.nn.MLP(channels))
    x = torch.randn(rows, channels[0]) * 1.5 + 0.3
    _compare(ours, ref, x, train=True)
    _compare(ours, ref, x, train=True)          # second step: running statistics keep tracking
    _compare(ours, ref, x, train=False)         # eval: running statistics


@pytest.mark.parametrize("channels,rows", [((16, 32), 1000), ((24, 64, 64), 2048), ((6, 5), 100)])
def test_vector_mlp_blocks(channels, rows):
    import deltaconv_amd as dc
    ours, ref = _pair(lambda: dc.nn.VectorMLP(channels), lambda: oracle.nn.VectorMLP(channels))
    v = torch.randn(2 * rows, channels[0])
    v[:4] = 0
    _compare(ours, ref, v, train=True)
    _compare(ours, ref, v, train=False)


def test_vector_block_vcat_equals_IJ_form():
    """forward_vcat(a) must equal forward(I_J(a)) (the P/Q fold of the 90-degree rotation)."""
    import deltaconv_amd as dc
    torch.manual_seed(0)
    blk = dc.nn.VectorMLP((2 * 20, 16))[0].to(DEV).train()
    a = torch.randn(2 * 500, 20, device=DEV, requires_grad=True)
    o1 = blk.forward_vcat(a)
    (g1,) = torch.autograd.grad(o1, a, torch.ones_like(o1))
    gw1 = torch.autograd.grad(blk.forward_vcat(a), blk[0].weight, torch.ones_like(o1))[0]
    o2 = blk(dc.geometry.I_J(a))
    (g2,) = torch.autograd.grad(o2, a, torch.ones_like(o2))
    gw2 = torch.autograd.grad(blk(dc.geometry.I_J(a)), blk[0].weight, torch.ones_like(o2))[0]
    assert rel_err(o1, o2) < 1e-5 and rel_err(g1, g2) < 1e-4 and rel_err(gw1, gw2) < 1e-4


def test_vectornonlin_without_batchnorm_and_repr():
    import deltaconv_amd as dc
    vn = dc.nn.VectorNonLin(8).to(DEV)
    assert repr(vn) == 'VectorNonLin(batchnorm=None)' and repr(dc.nn.BatchNorm1d(8)) == 'BatchNorm1d(8)'
    v = torch.rand(200, 8, device=DEV)
    assert torch.allclose(vn(v), v, atol=1e-6)                       # identity at init (test_nonlin.py)
    ref = oracle.nn.VectorNonLin(8)
    with torch.no_grad():
        b = torch.linspace(-0.5, 0.5, 8)
        vn.bias.copy_(b); ref.bias.copy_(b)
    x = torch.randn(200, 8)
    xd = x.to(DEV).requires_grad_(True); xo = x.clone().requires_grad_(True)
    yd, yo = vn(xd), ref(xo)
    w = torch.randn_like(yo)
    yd.backward(w.to(DEV)); yo.backward(w)
    assert rel_err(yd, yo) < 1e-5 and rel_err(xd.grad, xo.grad) < 1e-4 and rel_err(vn.bias.grad, ref.bias.grad) < 1e-4


def test_batchnorm_shift_scale_invariance():
    """reference test_batchnorm1d (test/nn/test_nonlin.py:7-38)."""
    import deltaconv_amd as dc
    bn = dc.nn.BatchNorm1d(16).to(DEV).train()
    x = torch.rand(100, 16, device=DEV)
    out = bn(x)
    assert out.shape == x.shape and not out.isnan().any()
    assert torch.allclose(bn(x + 3.0), out, atol=1e-4) and torch.allclose(bn(x * 5.0), out, atol=1e-3)


def test_vector_mlp_equivariance():
    """reference test_vectormlp (test/nn/test_mlp.py:22-78): MLP(T v) == T MLP(v) for per-point
    rotations / reflections T, train-mode BN."""
    import deltaconv_amd as dc
    torch.manual_seed(0)
    N, ci, co = 1000, 16, 32
    v = torch.rand(N, ci, device=DEV)
    mlp1 = dc.nn.VectorMLP((ci, co)).to(DEV)
    mlp2 = dc.nn.VectorMLP((ci, co, co, co)).to(DEV)
    ang = torch.rand(N // 2, device=DEV) * 2 * torch.pi
    c, s = torch.cos(ang), torch.sin(ang)
    Rm = torch.stack([torch.stack([c, s], 1), torch.stack([-s, c], 1)], 1)
    refl = torch.where(torch.rand(N // 2, device=DEV) > 0.1, 1.0, -1.0)
    Fm = torch.stack([torch.stack([torch.ones_like(refl), torch.zeros_like(refl)], 1),
                      torch.stack([torch.zeros_like(refl), refl], 1)], 1)
    T = Fm @ Rm
    for mlp in (mlp1, mlp2):
        t_out = (T @ mlp(v).view(-1, 2, co)).view(-1, co)
        out_t = mlp((T @ v.view(-1, 2, ci)).view(-1, ci))
        assert torch.allclose(t_out, out_t, atol=1e-5)
