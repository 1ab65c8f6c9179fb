"""GPU parity of the fused MLP-stream blocks (GEMM + HIP BatchNorm/activation/vector non-linearity)
vs the oracle's torch-CPU modules carrying the same weights: forward, input/parameter gradients,
running statistics, eval mode; plus the reference's own nn property tests
(test/nn/test_mlp.py, test/nn/test_nonlin.py)."""
import pytest
import torch

import oracle
from tests.helpers import rel_err
from deltaconv_amd.data import synthetic_batch

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


@pytest.mark.parametrize("channels,rows", [((16, 32), 1000), ((12, 64, 64, 24), 4096), ((5, 7), 333), ((32, 16), 2)])
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


def test_batchnorm_two_rows_no_cancellation():
    """R = 2 with nearly equal rows (the classification head at batch 2): var = (dx/2)^2 must not be
    lost to E[x^2] - E[x]^2 cancellation -> statistics are accumulated in fp64."""
    import deltaconv_amd as dc
    torch.manual_seed(5)
    base = torch.randn(1, 64) * 3
    x = torch.cat([base, base + 2e-3 * torch.randn(1, 64)], 0)
    bn = dc.nn.BatchNorm1d(64).to(DEV).train()
    ref = torch.nn.BatchNorm1d(64).double().train()
    out = bn(x.to(DEV))
    assert rel_err(out, ref(x.double())) < 1e-3


@pytest.mark.parametrize("depth,train", [(1, True), (1, False), (2, True)])
def test_centralized_layer_vs_oracle(depth, train):
    """Layer 0: analytic (no [E,C] tensor) path for depth 1, materialised fused path for depth 2."""
    import deltaconv_amd as dc
    from deltaconv_amd.data import synthetic_batch
    from oracle import geometry as geo
    b = synthetic_batch(3, 0, seed=50, sizes=[400, 256, 300], dup_frac=0.03)
    ours, ref = _pair(lambda: dc.nn.DeltaConv(3, 32, depth=depth, centralized=True, vector=True),
                      lambda: oracle.nn.DeltaConv(3, 32, depth=depth, centralized=True, vector=True))
    ours.train(train); ref.train(train)
    ptr = geo.cloud_ptr(b.batch)
    nbr = geo.knn(b.pos, 20, ptr)
    xb, yb = geo.build_tangent_basis(b.norm)
    Go, Do = geo.build_grad_div(b.pos, b.norm, xb, yb, nbr, ptr)
    bd = b.to(DEV)
    graph = dc.geometry.Graph.knn(bd.pos, 20, bd.batch)
    assert torch.equal(graph.nbr.cpu().long(), nbr)
    G, D = dc.geometry.build_grad_div(bd.pos, bd.norm, xb.to(DEV), yb.to(DEV), graph, bd.batch)
    x = b.pos.clone().requires_grad_(True)
    xd = bd.pos.clone().requires_grad_(True)
    xo_, vo_ = ref(x, Go @ x, Go, Do, nbr)
    xd_, vd_ = ours(xd, G @ xd, G, D, graph)
    assert rel_err(xd_, xo_) < 1e-3 and rel_err(vd_, vo_) < 1e-3
    if train:
        wx, wv = torch.randn_like(xo_), torch.randn_like(vo_)
        ((xo_ * wx).sum() + (vo_ * wv).sum()).backward()
        ((xd_ * wx.to(DEV)).sum() + (vd_ * wv.to(DEV)).sum()).backward()
        assert rel_err(xd.grad, x.grad) < 5e-3
        for (n1, p1), (n2, p2) in zip(ours.named_parameters(), ref.named_parameters()):
            if p2.grad is not None:
                assert rel_err(p1.grad, p2.grad) < 5e-3, n1
        for (n1, b1), (n2, b2) in zip(ours.named_buffers(), ref.named_buffers()):
            if b2.dtype.is_floating_point:
                assert rel_err(b1, b2) < 1e-3, n1


@pytest.mark.parametrize("centralized,vector,ci,co", [(True, True, 3, 32), (False, True, 32, 64), (False, False, 64, 64),
                                                      (False, True, 6, 10)])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_fused_layer_matches_composed(centralized, vector, ci, co, train, depth):
    """nn/layer.py (one autograd node, in-place accumulation, no cat) vs the chain of small nodes."""
    import deltaconv_amd as dc
    from deltaconv_amd.data import synthetic_batch
    b = synthetic_batch(2, 512, seed=60).to(DEV)
    graph = dc.geometry.Graph.knn(b.pos, 20, b.batch)
    xb, yb = dc.geometry.build_tangent_basis(b.norm)
    G, D = dc.geometry.build_grad_div(b.pos, b.norm, xb, yb, graph, b.batch)
    torch.manual_seed(7)
    conv = dc.nn.DeltaConv(ci, co, depth=depth, centralized=centralized, vector=vector).to(DEV)
    with torch.no_grad():
        for n_, p_ in conv.named_parameters():
            if n_.endswith("bn.weight"):
                v_ = torch.linspace(0.4, 1.6, p_.numel(), device=DEV); v_[::5] *= -1
                p_.copy_(v_)
            if n_.endswith("bn.bias"):
                p_.copy_(torch.linspace(-0.3, 0.3, p_.numel(), device=DEV))
    conv.train(train)
    assert conv._fusable() is not None
    x0 = torch.randn(graph.n, ci, device=DEV)
    v0 = torch.randn(2 * graph.n, ci, device=DEV)
    res = []
    sd0 = {k_: t.clone() for k_, t in conv.state_dict().items()}
    for fuse in (True, False):
        conv.load_state_dict(sd0)
        conv.zero_grad()
        conv.fuse_layer = fuse
        x = x0.clone().requires_grad_(True)
        v = v0.clone().requires_grad_(True)
        xo, vo = conv(x, v, G, D, graph)
        gen = torch.Generator(device=DEV).manual_seed(1)
        loss = (xo * torch.randn(xo.shape, device=DEV, generator=gen)).sum() + \
               (vo * torch.randn(vo.shape, device=DEV, generator=gen)).sum()
        if train:
            loss.backward()
        res.append((xo.detach(), vo.detach(), x.grad, v.grad,
                    {n_: (p_.grad.clone() if p_.grad is not None else None) for n_, p_ in conv.named_parameters()},
                    {n_: b_.clone() for n_, b_ in conv.named_buffers()}))
    (x1, v1, gx1, gv1, gp1, bf1), (x2, v2, gx2, gv2, gp2, bf2) = res
    assert rel_err(x1, x2) < 1e-5 and rel_err(v1, v2) < 1e-5
    if train:
        assert rel_err(gx1, gx2) < 1e-4 and rel_err(gv1, gv2) < 1e-4
        for n_ in gp1:
            if gp2[n_] is None:
                assert gp1[n_] is None, n_
            else:
                assert rel_err(gp1[n_], gp2[n_]) < 1e-4, n_
    for n_ in bf1:
        if bf1[n_].dtype.is_floating_point:
            assert rel_err(bf1[n_], bf2[n_]) < 1e-5, n_
        else:
            assert int(bf1[n_]) == int(bf2[n_]), n_


@pytest.mark.parametrize("R,M,N", [(32768, 64, 64), (32768, 256, 128), (65536, 128, 192), (65536, 256, 256),
                                   (32768, 256, 512), (10000, 96, 32), (8193, 32, 160),
                                   (32768, 128, 70), (9000, 64, 12), (4097, 64, 3), (65536, 70, 128), (33, 5, 7), (8192, 256, 448),
                                   (20480, 50, 128)])
def test_mfma_gemm_tn(R, M, N):
    """Hand-written fp32-MFMA tall-skinny weight-gradient GEMM vs fp64 matmul (exact fp32 fma chain:
    error of the fp32 class, ~1e-7 * sum|a b|), strided operands (column blocks of concat buffers),
    bit-reproducible."""
    from deltaconv_amd.nn import fused
    torch.manual_seed(R % 97)
    abig = torch.randn(R, M + 8, device=DEV)
    bbig = torch.randn(R, N + 12, device=DEV)
    a, b = abig[:, 4:4 + M], bbig[:, 8:8 + N]
    from deltaconv_amd._lib import lib

    def mfma_tn(a_, b_):
        out = torch.empty(M, N, device=DEV)
        nb = lib.raw("dc_gemm_tn_workspace_bytes")(R, M, N)
        ws = torch.empty((nb + 3) // 4, device=DEV)
        lib.call("dc_gemm_tn", a_, a_.stride(0), b_, b_.stride(0), R, M, N, out, N, 0, ws, ws.numel() * 4)
        return out

    c = mfma_tn(a, b)
    ref = a.double().t() @ b.double()
    scale = float((a.abs().double().t() @ b.abs().double()).max())
    assert float((c.double() - ref).abs().max()) < 2e-6 * scale
    assert torch.equal(c, mfma_tn(a, b))
    assert float((fused.gemm_tn(a, b).double() - ref).abs().max()) < 1e-5 * scale   # dispatcher (either path)
    lib_res = a.t() @ b
    assert float((lib_res.double() - ref).abs().max()) < 1e-5 * scale   # the library is no closer


@pytest.mark.parametrize("with_mean,train,B,N,C", [(True, True, 4, 256, 64), (False, True, 3, 100, 20), (True, False, 2, 64, 8)])
def test_bn_act_pool_fused(with_mean, train, B, N, C):
    """Embedding head fused with per-cloud pooling vs the plain composition (oracle modules on CPU)."""
    import deltaconv_amd as dc
    from deltaconv_amd.nn import fused
    ours, ref = _pair(lambda: dc.nn.MLP((12, C)), lambda: oracle.nn.MLP((12, C)))
    ours.train(train); ref.train(train)
    x = torch.randn(B * N, 12)
    x[5] = x[3]                                   # duplicate row -> tie in the max: lowest row wins
    xo = x.clone().requires_grad_(True)
    xd = x.to(DEV).requires_grad_(True)
    y = ref(xo).view(B, N, C)
    po = torch.cat([y.max(1).values, y.mean(1)], 1) if with_mean else y.max(1).values
    info = (torch.arange(0, (B + 1) * N, N, dtype=torch.int32, device=DEV), B, N)
    pd = dc.models.pool.embed_and_pool(ours, xd, info, with_mean)
    assert rel_err(pd, po) < 2e-4
    if train:
        w = torch.randn_like(po)
        po.backward(w); pd.backward(w.to(DEV))
        assert rel_err(xd.grad, xo.grad) < 1e-3
        for (n1, p1), (n2, p2) in zip(ours.named_parameters(), ref.named_parameters()):
            assert rel_err(p1.grad, p2.grad) < 1e-3, n1
        for (n1, b1), (n2, b2) in zip(ours.named_buffers(), ref.named_buffers()):
            if b2.dtype.is_floating_point:
                assert rel_err(b1, b2) < 2e-4, n1


@pytest.mark.parametrize("R,C,smoothing,strided", [(32, 40, True, False), (32768, 50, False, False), (1, 2, True, False),
                                                    (100, 15, True, True), (7, 8, False, True), (64, 40, True, False),
                                                    (65, 40, True, True), (33, 130, False, False)])
def test_fused_loss_matches_oracle(R, C, smoothing, strided):
    """dc_ce_loss (value + gradient, scaled by the incoming gradient) == experiments/utils.py:7-24 as restated
    in oracle/loss.py, incl. logits that are a column slice of a wider matrix."""
    from deltaconv_amd.utils import calc_loss
    torch.manual_seed(R * 31 + C)
    wide = torch.randn(R, C + 5) * 3
    x_cpu = (wide[:, 2:2 + C] if strided else wide[:, :C].contiguous()).clone().requires_grad_(True)
    y = torch.randint(0, C, (R,))
    ref = oracle.loss.calc_loss(x_cpu, y, smoothing=smoothing) * 1.7
    ref.backward()
    wide_d = wide.to(DEV).requires_grad_(True)
    x = wide_d[:, 2:2 + C] if strided else wide_d[:, :C].contiguous()
    if not strided:
        x.retain_grad()
    loss = calc_loss(x, y.to(DEV), smoothing=smoothing) * 1.7
    loss.backward()
    g = wide_d.grad[:, 2:2 + C] if strided else x.grad
    assert abs(float(loss) - float(ref)) <= 3e-6 * max(1.0, abs(float(ref)))        # fp32 exp/log, fp64 row sum
    assert rel_err(g.cpu(), x_cpu.grad) < 1e-5
    # repeatability: the ordered reduction gives the same bits every time
    again = calc_loss(x.detach(), y.to(DEV), smoothing=smoothing) * 1.7
    assert float(again) == float(loss)


def test_fused_loss_rejects_bad_shapes():
    from deltaconv_amd.utils import calc_loss
    with pytest.raises((RuntimeError, AssertionError)):
        calc_loss(torch.zeros(0, 4, device=DEV), torch.zeros(0, dtype=torch.long, device=DEV))
    bad = calc_loss(torch.zeros(3, 4, device=DEV), torch.tensor([0, 9, 1], device=DEV))   # label out of range
    assert torch.isnan(bad)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("n,co", [(300, 8), (257, 5)])
def test_vector_nonlin_combine_layouts(mode, n, co):
    """[P | Q] blocked (combine=1) and (P_c, Q_c) interleaved (combine=2) inputs give the same result and
    input gradient as the plain layout (combine=0) applied to y = (P_u - Q_v, P_v + Q_u)."""
    from deltaconv_amd.nn import fused
    torch.manual_seed(n + co + mode)
    vn = oracle.nn.VectorNonLin(co, batchnorm=oracle.nn.BatchNorm1d(co))
    import deltaconv_amd as dc
    ours = dc.nn.VectorNonLin(co, batchnorm=dc.nn.BatchNorm1d(co)).to(DEV).train()
    ours.load_state_dict(vn.state_dict())
    P, Q = torch.randn(2 * n, co, device=DEV), torch.randn(2 * n, co, device=DEV)
    y = torch.empty(2 * n, co, device=DEV)
    y[0::2] = P[0::2] - Q[1::2]
    y[1::2] = P[1::2] + Q[0::2]
    y.requires_grad_(True)
    pq = (torch.cat([P, Q], 1) if mode == 1 else torch.stack([P, Q], 2).reshape(2 * n, 2 * co)).requires_grad_(True)
    w = torch.randn(2 * n, co, device=DEV)
    out0 = fused.vector_nonlin(y, 0, ours)
    (out0 * w).sum().backward()
    ours2 = dc.nn.VectorNonLin(co, batchnorm=dc.nn.BatchNorm1d(co)).to(DEV).train()
    ours2.load_state_dict(vn.state_dict())
    out = fused.vector_nonlin(pq, mode, ours2)
    (out * w).sum().backward()
    assert rel_err(out, out0) < 1e-6
    g = pq.grad.view(2 * n, 2, co) if mode == 1 else pq.grad.view(2 * n, co, 2).transpose(1, 2)
    dP, dQ = g[:, 0], g[:, 1]
    # y_u = P_u - Q_v, y_v = P_v + Q_u  =>  dP = dy, dQ_u = dy_v, dQ_v = -dy_u
    assert rel_err(dP, y.grad) < 1e-5
    dq_ref = torch.empty_like(y.grad)
    dq_ref[0::2] = y.grad[1::2]
    dq_ref[1::2] = -y.grad[0::2]
    assert rel_err(dQ, dq_ref) < 1e-5


@pytest.mark.parametrize("n_clouds,N,k,C,slope", [(2, 300, 20, 64, 0.2), (1, 257, 10, 5, 0.2), (2, 128, 30, 128, 0.0)])
def test_knn_max_affine_equals_bn_act_then_max(n_clouds, N, k, C, slope):
    """dc_knn_max_affine (BatchNorm scale/shift + LeakyReLU evaluated inside the gather) == dc_bn_act followed by
    dc_knn_max, bit for bit: values AND first-maximal slots (negative, zero and tiny scales included)."""
    from deltaconv_amd._lib import lib
    from deltaconv_amd.geometry import Graph
    from deltaconv_amd.data import synthetic_batch
    b = synthetic_batch(n_clouds, N, seed=77).to(DEV)
    g = Graph.knn(b.pos, k, b.batch)
    n = b.pos.shape[0]
    torch.manual_seed(C)
    h = torch.randn(n, C, device=DEV).round(decimals=1)              # plenty of exact ties
    scale = torch.randn(C, device=DEV)
    scale[0], scale[1 % C], scale[2 % C] = 0.0, -0.7, 1e-9            # constant column, min instead of max, rounding ties
    shift = torch.randn(C, device=DEV)
    y = torch.empty_like(h)
    lib.call("dc_bn_act", h, n, C, C, scale, shift, slope, None, C, y, C)
    o1, a1 = torch.empty_like(h), torch.empty(n, C, dtype=torch.uint8, device=DEV)
    lib.call("dc_knn_max", g.nbr, n, k, y, C, C, o1, C, a1)
    o2, a2 = torch.empty_like(h), torch.empty(n, C, dtype=torch.uint8, device=DEV)
    lib.call("dc_knn_max_affine", g.nbr, n, k, h, C, C, scale, shift, slope, o2, C, a2)
    assert torch.equal(o1, o2) and torch.equal(a1, a2)


# ---------------------------------------------------------------------------------- blocks on a handful of rows
@pytest.mark.parametrize("M", [2, 7, 32, 33, 64])
@pytest.mark.parametrize("K,N", [(2048, 512), (512, 256), (256, 40), (64, 10)])
def test_rowblock_block_vs_torch_fp64(M, K, N):
    """csrc/rowblock.hip (the classification head: one row per cloud): [Linear -> BatchNorm over the rows -> LeakyReLU] as
    one kernel per direction against torch in fp64: output, input / weight / gamma / beta gradients, running statistics;
    train and eval mode."""
    from deltaconv_amd.nn import fused
    from deltaconv_amd.nn.nonlin import BatchNorm1d
    torch.manual_seed(M * 1000 + N)
    x = torch.randn(M, K)
    lin = torch.nn.Linear(K, N, bias=False)
    bn = BatchNorm1d(N)
    with torch.no_grad():
        bn.bn.weight.uniform_(0.5, 1.5)
        bn.bn.bias.uniform_(-0.5, 0.5)
    g = torch.randn(M, N)
    for training in (True, False):
        # fp64 reference
        xr = x.double().requires_grad_(True)
        wr = lin.weight.detach().double().requires_grad_(True)
        gr, br = bn.bn.weight.detach().double().requires_grad_(True), bn.bn.bias.detach().double().requires_grad_(True)
        rm, rv = bn.bn.running_mean.double().clone(), bn.bn.running_var.double().clone()
        yr = torch.nn.functional.leaky_relu(
            torch.nn.functional.batch_norm(xr @ wr.t(), rm, rv, gr, br, training, 0.1, 1e-5), 0.2)
        yr.backward(g.double())
        # device
        lin_d = torch.nn.Linear(K, N, bias=False).to(DEV)
        lin_d.load_state_dict(lin.state_dict())
        bn_d = BatchNorm1d(N).to(DEV)
        bn_d.load_state_dict(bn.state_dict())
        bn_d.train(training)
        xd = x.to(DEV).requires_grad_(True)
        assert fused._rowblock_ok(xd, lin_d.weight)
        yd = fused.linear_bn_act(xd, lin_d, bn_d.bn, 0.2)
        yd.backward(g.to(DEV))
        tol = 2e-5
        assert rel_err(yd, yr) < tol
        assert rel_err(xd.grad, xr.grad) < 5 * tol and rel_err(lin_d.weight.grad, wr.grad) < 5 * tol
        assert rel_err(bn_d.bn.weight.grad, gr.grad) < 5 * tol and rel_err(bn_d.bn.bias.grad, br.grad) < 5 * tol
        if training:
            assert rel_err(bn_d.bn.running_mean, rm) < tol and rel_err(bn_d.bn.running_var, rv) < tol
            assert int(bn_d.bn.num_batches_tracked) == 1


@pytest.mark.parametrize("M", [1, 32, 64])
def test_rowblock_linear_with_bias(M):
    from deltaconv_amd.nn import fused
    torch.manual_seed(M)
    K, N = 256, 40
    x, w, b, g = torch.randn(M, K), torch.randn(N, K) / 16, torch.randn(N), torch.randn(M, N)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    (xr @ wr.t() + br).backward(g.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = fused.linear(xd, wd, bd)
    y.backward(g.to(DEV))
    assert rel_err(y, (xr @ wr.t() + br)) < 1e-5
    assert rel_err(xd.grad, xr.grad) < 1e-5 and rel_err(wd.grad, wr.grad) < 1e-5 and rel_err(bd.grad, br.grad) < 1e-5


def test_rowblock_equals_composed_path_in_the_model():
    """The whole classification model with the head through csrc/rowblock.hip and through the composed path (library
    GEMM + statistics + finaliser + activation): logits and every parameter gradient agree to fp32 rounding."""
    import deltaconv_amd as dc
    from deltaconv_amd.nn import fused
    from deltaconv_amd.utils import calc_loss
    b = synthetic_batch(8, 256, seed=77).to(DEV)

    def run(use):
        fused.USE_ROWBLOCK = use
        try:
            torch.manual_seed(2)
            m = dc.models.DeltaNetClassification(3, 40).to(DEV).train()
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            out = m(b)
            calc_loss(out, b.y).backward()
            return out.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, \
                {n: t.clone() for n, t in m.named_buffers()}
        finally:
            fused.USE_ROWBLOCK = True

    o1, g1, b1 = run(True)
    o0, g0, b0 = run(False)
    assert rel_err(o1, o0) < 1e-5
    assert g1.keys() == g0.keys()
    gmax = max(float(v.abs().max()) for v in g0.values())
    for n in g0:
        assert float((g1[n] - g0[n]).abs().max()) < 2e-5 * max(float(g0[n].abs().max()), 1e-3 * gmax), n
    for n in b0:
        assert rel_err(b1[n].float(), b0[n].float()) < 1e-5, n


@pytest.mark.parametrize("rows,k,c,slope", [(4096, 256, 128, 0.2), (3000, 128, 50, 1.0), (513, 64, 7, 0.0)])
def test_linear_bias_act_vs_torch(rows, k, c, slope):
    """The segmentation head's Linear(bias) -> LeakyReLU pair and its plain Linear(bias) on many rows
    (deltaconv/models/deltanet_segmentation.py:45-51): bias + activation in the BatchNorm/activation kernel's pass, d b from its
    ordered reduction -- against torch in fp64, forward and every gradient."""
    from deltaconv_amd.nn import fused
    gen = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, k, generator=gen, dtype=torch.float64)
    w = torch.randn(c, k, generator=gen, dtype=torch.float64) / k ** 0.5
    b = torch.randn(c, generator=gen, dtype=torch.float64)
    dy = torch.randn(rows, c, generator=gen, dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (x, w, b)]
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.linear(*leaves), slope) if slope != 1.0 else torch.nn.functional.linear(*leaves)
    ref.backward(dy)
    dl = [t.float().to(DEV).requires_grad_(True) for t in (x, w, b)]
    out = fused.linear_bias_act(*dl, slope)
    out.backward(dy.float().to(DEV))
    assert rel_err(out, ref) < 1e-5
    for got, want, name in zip(dl, leaves, "xwb"):
        assert rel_err(got.grad, want.grad) < 1e-4, name
    # and the module form: Linear with a bias runs the same kernels
    import deltaconv_amd as dc
    lin = dc.nn.mlp.Linear(k, c).to(DEV)
    with torch.no_grad():
        lin.weight.copy_(w); lin.bias.copy_(b)
    y = lin(x.float().to(DEV))
    assert rel_err(y, torch.nn.functional.linear(x, w, b)) < 1e-5


@pytest.mark.parametrize("R,C", [(2, 40), (32, 40), (61, 15), (64, 7)])
def test_fused_loss_one_launch_same_bits_as_two(R, C):
    """<= 64 logits rows: the one-launch cross-entropy (round 6) adds the row losses in the association of the two-launch form."""
    from deltaconv_amd._lib import lib
    from deltaconv_amd.utils import calc_loss
    torch.manual_seed(R + C)
    x = (torch.randn(R, C, device=DEV) * 4).requires_grad_(True)
    y = torch.randint(0, C, (R,), device=DEV)
    one = calc_loss(x, y)
    one.backward()
    g1 = x.grad.clone()
    x.grad = None
    lib.raw("dc_set_option")(10, 1)
    try:
        two = calc_loss(x, y)
        two.backward()
    finally:
        lib.raw("dc_set_option")(10, 0)
    assert float(one) == float(two) and torch.equal(g1, x.grad)


@pytest.mark.parametrize("M,K,N,p", [(32, 2048, 512, 0.5), (32, 512, 256, 0.5), (8, 64, 40, 0.25), (64, 256, 96, 0.1)])
def test_rowblock_dropout_fused_equals_block_then_dropout(M, K, N, p):
    """MLP block -> Dropout(p) of the classification head (deltanet_classification.py:34-36) inside the block's own kernels
    (dc_rowblock_forward_dropout / _backward_dropout): with the mask the fused call drew, forward and every gradient equal the
    un-fused block followed by mask * 1 / (1 - p); the mask keeps a fraction 1 - p and changes with the step counter and the
    layer salt."""
    import copy
    import deltaconv_amd as dc
    torch.manual_seed(M + N)
    blk = dc.nn.MLP([K, N]).to(DEV).train()[0]
    with torch.no_grad():
        blk[1].bn.weight.uniform_(0.5, 1.5); blk[1].bn.bias.uniform_(-0.3, 0.3)
    x = torch.randn(M, K, device=DEV)
    dy = torch.randn(M, N, device=DEV)
    a, b = copy.deepcopy(blk), copy.deepcopy(blk)
    xa = x.clone().requires_grad_(True)
    assert a.dropout_ok(xa)
    ya = a(xa, dropout=(p, 1))
    ya.backward(dy)
    mask = (ya != 0).float()
    assert abs(float(mask.mean()) - (1 - p)) < 5 * (p * (1 - p) / (M * N)) ** 0.5 + 0.02
    xb = x.clone().requires_grad_(True)
    yb = b(xb) * mask * (1.0 / (1.0 - p))
    yb.backward(dy)
    tol = 0.0 if p == 0.5 else 1e-6
    assert rel_err(ya, yb) <= tol
    assert rel_err(xa.grad, xb.grad) <= max(tol, 1e-6)       # (the input gradient runs through the split-product GEMM in both)
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert rel_err(p1.grad, p2.grad) <= max(tol, 1e-6), n1
    for (n1, b1), (_, b2) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.equal(b1, b2), n1                        # running statistics, num_batches_tracked: untouched by the dropout
    # next training step (the BatchNorm's step counter moved), another salt: other masks; same step + salt: the same mask
    m2 = (a(x, dropout=(p, 1)) != 0)
    assert float((m2 != mask.bool()).float().mean()) > 0.5 * p * (1 - p)
    c1, c2, c3 = copy.deepcopy(a), copy.deepcopy(a), copy.deepcopy(a)
    k1, k2, k3 = (c1(x, dropout=(p, 1)) != 0), (c2(x, dropout=(p, 1)) != 0), (c3(x, dropout=(p, 2)) != 0)
    assert torch.equal(k1, k2) and float((k1 != k3).float().mean()) > 0.5 * p * (1 - p)


def test_classification_head_dropout_runs_in_the_row_blocks():
    """The whole classification net in training mode: the two Dropout(0.5) of the head run inside the row-block kernels
    (no torch RNG consumed), draw a new mask every step -- also in replays of a captured step -- and vanish in eval mode;
    with the switch off the same net runs torch's dropout."""
    import deltaconv_amd as dc
    from deltaconv_amd.nn import fused
    from deltaconv_amd.graph_step import GraphedTrainStep
    from deltaconv_amd.utils import calc_loss
    torch.manual_seed(3)
    model = dc.models.DeltaNetClassification(3, 40).to(DEV).train()
    b = synthetic_batch(4, 256, seed=9).to(DEV)
    state = torch.cuda.get_rng_state()
    out1 = model(b).detach().clone()
    assert torch.equal(torch.cuda.get_rng_state(), state), "the fused dropout must not consume torch's generator"
    out2 = model(b).detach().clone()
    assert not torch.equal(out1, out2) and torch.isfinite(out1).all()          # new masks (and new running statistics)
    fused.USE_ROWBLOCK_DROPOUT = False
    try:
        model(b)
        assert not torch.equal(torch.cuda.get_rng_state(), state)            # torch's dropout drew from the generator
    finally:
        fused.USE_ROWBLOCK_DROPOUT = True
    model.eval()
    with torch.no_grad():
        e1, e2 = model(b).clone(), model(b).clone()
    assert torch.equal(e1, e2)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)                          # lr 0: only the masks / statistics change
    g = GraphedTrainStep(model, calc_loss, b, optimizer=opt)
    losses = [float(g()) for _ in range(4)]
    assert len(set(losses)) == 4 and all(l == l for l in losses), losses       # every replay: another mask


@pytest.mark.parametrize("nc,mx,c", [(2, 2048, 256), (16, 2048, 256), (1, 4096, 256), (3, 50, 7), (5, 17, 64)])
def test_cloud_bias_join_and_its_backward(nc, mx, c):
    """fused.cloud_bias (dc_cloud_bias_add / dc_cloud_colsum): h + g[batch] in place for equal-size clouds and the per-cloud
    column sums of its backward, vs the broadcast form in fp64 (reference: x_max[batch], deltanet_segmentation.py:59).
    Tolerance: fp32 rounding of an fp64-accumulated sum, 1e-6 relative; repeated calls give the same bits."""
    from deltaconv_amd.nn import fused
    g = torch.Generator().manual_seed(nc * 1000 + mx + c)
    h0 = torch.randn(nc * mx, c, generator=g).to(DEV)
    b0 = torch.randn(nc, c, generator=g).to(DEV)
    dy = torch.randn(nc * mx, c, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        h = (h0.clone() * 1.0).requires_grad_(True)
        b = b0.clone().requires_grad_(True)
        y = fused.cloud_bias(h * 1.0, b, mx)              # (a fresh non-leaf tensor, as the product is in the model)
        y.backward(dy)
        outs.append((y.detach().clone(), h.grad.clone(), b.grad.clone()))
    ref_y = h0.double().view(nc, mx, c) + b0.double().unsqueeze(1)
    assert rel_err(outs[0][0], ref_y.view(nc * mx, c)) < 1e-7
    assert torch.equal(outs[0][1], dy)
    assert rel_err(outs[0][2], dy.double().view(nc, mx, c).sum(1)) < 1e-6
    assert all(torch.equal(a, b_) for a, b_ in zip(*outs))
    with pytest.raises(ValueError):
        fused.cloud_bias(h0.clone(), b0, mx + 1)


def test_split_cols_views_and_one_launch_backward():
    from deltaconv_amd.nn import fused
    w = torch.randn(40, 100, device=DEV, requires_grad=True)
    a, b = fused.split_cols(w, 36)
    assert a.shape == (40, 36) and b.shape == (40, 64) and a.data_ptr() == w.data_ptr()
    ga, gb = torch.randn(40, 36, device=DEV), torch.randn(40, 64, device=DEV)
    (a * ga).sum().backward(retain_graph=True)            # only one half used: the other block is zero
    assert torch.equal(w.grad[:, :36], ga) and float(w.grad[:, 36:].abs().max()) == 0
    w.grad = None
    ((a * ga).sum() + (b * gb).sum()).backward()
    assert torch.equal(w.grad, torch.cat([ga, gb], 1))
