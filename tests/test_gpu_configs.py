"""The BASELINE.json configurations at their FULL sizes on the GPU (the oracle would need minutes
per step there), checked through size-independent properties:
  * kNN: every row starts from a zero-distance point, distances ascending, ids inside the cloud;
  * operators: grad(const) ~ 0 (de Rham), transposed applies are exact adjoints (<A x, y> = <x, A^T y>);
  * CSC: a permutation of the edges, sorted columns;
  * one full train step: finite loss / gradients, bit-identical when repeated (no fp atomics anywhere),
    BatchNorm running statistics move;
and, at a reduced batch of the same per-cloud shape, against the CPU oracle."""
import pytest
import torch

import oracle
from tests.helpers import rel_err
from deltaconv_amd.data import synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"

CONFIGS = {
    # name: (B, N, k, normals, model kind, model kwargs, batch kwargs)
    "C2_modelnet40": (32, 1024, 20, True, "cls", dict(in_channels=3, num_classes=40), {}),
    "C3_scanobjectnn": (32, 2048, 20, False, "cls",
                        dict(in_channels=3, num_classes=15, conv_channels=[64, 64, 64, 128], grad_regularizer=1e-2),
                        dict(outlier_frac=0.05, jitter=0.005, num_classes=15)),
    "C4_shapenet": (16, 2048, 20, True, "seg", dict(in_channels=3, num_classes=50, categorical_vector=True),
                    dict(dup_frac=0.03, per_point_labels=True, categories=16, num_classes=50)),
    "C5_shapeseg": (8, 4096, 30, True, "seg",
                    dict(in_channels=3, num_classes=8, conv_channels=[128] * 8, mlp_depth=1, embedding_size=512),
                    dict(per_point_labels=True, num_classes=8)),
}


def _build(kind, kw, k):
    import deltaconv_amd as dc
    torch.manual_seed(1)
    cls = dc.models.DeltaNetSegmentation if kind == "seg" else dc.models.DeltaNetClassification
    return cls(num_neighbors=k, **kw)


def _no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.eval()
    return m


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_properties(name):
    import deltaconv_amd as dc
    B, N, k, normals, kind, kw, bkw = CONFIGS[name]
    b = synthetic_batch(B, N, seed=70, normals=normals, **bkw).to(DEV)
    model = _no_dropout(_build(kind, kw, k).to(DEV).train())
    graph, grad, div = model.deltanet_base.build_operators(b)
    n = B * N
    # kNN structure
    nbr = graph.nbr.long()
    assert nbr.shape == (n, k)
    d = (b.pos[nbr] - b.pos[:, None, :]).pow(2).sum(-1)
    assert float(d[:, 0].max()) == 0.0 and bool((d[:, 1:] >= d[:, :-1] - 1e-6).all())
    cloud = torch.arange(n, device=DEV) // N
    assert bool((nbr // N == cloud[:, None]).all())
    # CSC
    tptr, tedge = graph.csc()
    assert int(tptr[-1]) == n * k and torch.equal(torch.sort(tedge).values,
                                                  torch.arange(n * k, dtype=torch.int32, device=DEV))
    # operators: grad of a constant ~ 0; adjointness of the transposed kernels
    one = torch.ones(n, 4, device=DEV)
    assert float((grad @ one).abs().max()) < 5e-2
    x = torch.randn(n, 16, device=DEV, requires_grad=True)
    v = torch.randn(2 * n, 16, device=DEV, requires_grad=True)
    y = torch.randn(2 * n, 16, device=DEV)
    z = torch.randn(n, 16, device=DEV)
    (gx,) = torch.autograd.grad(grad @ x, x, y)
    (gv,) = torch.autograd.grad(div @ v, v, z)
    lhs1, rhs1 = ((grad @ x).detach().double() * y.double()).sum(), (x.detach().double() * gx.double()).sum()
    lhs2, rhs2 = ((div @ v).detach().double() * z.double()).sum(), (v.detach().double() * gv.double()).sum()
    assert abs(float(lhs1 - rhs1)) < 1e-4 * float(lhs1.abs() + 1) and abs(float(lhs2 - rhs2)) < 1e-4 * float(lhs2.abs() + 1)
    # one full step, twice: finite and bit-identical
    snaps = []
    rm0 = {n_: b_.clone() for n_, b_ in model.named_buffers() if "running_mean" in n_}
    sd0 = {k_: t.clone() for k_, t in model.state_dict().items()}
    for _ in range(2):
        model.load_state_dict(sd0)
        model.zero_grad(set_to_none=True)
        out = model(b)
        loss = oracle.loss.calc_loss(out, b.y, smoothing=(kind != "seg"))
        loss.backward()
        assert torch.isfinite(loss) and out.shape[0] == (n if kind == "seg" else B)
        g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
        assert torch.isfinite(g).all()
        snaps.append((out.detach().clone(), g.clone()))
    assert torch.equal(snaps[0][0], snaps[1][0]), "forward is not bit-reproducible"
    assert torch.equal(snaps[0][1], snaps[1][1]), "backward is not bit-reproducible"
    moved = [float((b_ - rm0[n_]).abs().max()) for n_, b_ in model.named_buffers() if "running_mean" in n_]
    assert min(moved) > 0


@pytest.mark.parametrize("name", ["C3_scanobjectnn", "C5_shapeseg"])
def test_reduced_batch_vs_oracle(name):
    """Same per-cloud shape (N, k, channels), 2 clouds: logits against the CPU oracle."""
    B, N, k, normals, kind, kw, bkw = CONFIGS[name]
    bkw = dict(bkw)
    b = synthetic_batch(2, N, seed=71, normals=normals, **bkw)
    model = _build(kind, kw, k)
    okw = dict(kw)
    ocls = oracle.models.DeltaNetSegmentation if kind == "seg" else oracle.models.DeltaNetClassification
    ref = ocls(num_neighbors=k, **okw)
    ref.load_state_dict(model.state_dict())
    ref = _no_dropout(ref.train())
    model = _no_dropout(model.to(DEV).train())
    with torch.no_grad():
        lo = ref(b)
        ld = model(b.to(DEV))
    # no-normals: SVD-sign gauge + ill-defined x-axis -> looser
    assert rel_err(ld, lo) < (5e-3 if not normals else 1e-3)


@pytest.mark.parametrize("name", ["C3_scanobjectnn", "C4_shapenet", "C5_shapeseg"])
def test_reduced_batch_step_vs_oracle(name):
    """Train-mode forward + loss + backward at the per-cloud shape of the BASELINE configuration (N, k, channels, MLP depth;
    2 clouds): logits AND every parameter gradient against the CPU oracle in fp64, with the oracle's own fp32 run (= the
    reference's numerics) as the yardstick -- max-aggregation / pooling make gradients piecewise, so the bound is
    self-calibrating: no further from fp64 than 3x the fp32 oracle (+ a floor).  These are the shapes at which the
    ragged-K guarded GEMM loads (K = 6 / 12 / 70, N = 3 / 50), the depth-2 fused nodes and the centralised [E, 64] layer 0
    (C4), the k = 30 / 8 x 128 layers from the P = 32 tile plan (C5) and the estimate_basis path (C3) actually run.
    No-normals (C3): the tangent frames come from an SVD whose x-axis is a gauge choice, features
    agree up to fp rounding through the gauge -> looser floor."""
    from deltaconv_amd.data import Batch
    B, N, k, normals, kind, kw, bkw = CONFIGS[name]
    b = synthetic_batch(2, N, seed=72, normals=normals, **dict(bkw))
    seg = kind == "seg"
    model = _build(kind, kw, k)
    ocls = oracle.models.DeltaNetSegmentation if seg else oracle.models.DeltaNetClassification
    ref32 = ocls(num_neighbors=k, **kw)
    ref32.load_state_dict(model.state_dict())
    ref64 = ocls(num_neighbors=k, **kw).double()
    ref64.load_state_dict(model.state_dict())
    ref32, ref64 = _no_dropout(ref32.train()), _no_dropout(ref64.train())
    model = _no_dropout(model.to(DEV).train())
    b64 = Batch(b.pos.double(), b.batch, None if b.norm is None else b.norm.double(), None, b.y,
                None if b.category is None else b.category.double(), b.num_graphs)
    l32 = ref32(b)
    oracle.loss.calc_loss(l32, b.y, smoothing=not seg).backward()
    l64 = ref64(b64)
    oracle.loss.calc_loss(l64, b.y, smoothing=not seg).backward()
    bd = b.to(DEV)
    ld = model(bd)
    oracle.loss.calc_loss(ld, bd.y, smoothing=not seg).backward()
    floor = 1e-3 if normals else 5e-3
    e_hip, e_ref = rel_err(ld, l64), rel_err(l32, l64)
    print(f"{name} logits: hip-vs-f64 {e_hip:.2e}  oracle32-vs-f64 {e_ref:.2e}")
    assert e_hip < 3 * e_ref + floor
    gmax = max(float(p.grad.abs().max()) for p in ref64.parameters() if p.grad is not None)
    worst_hip = worst_ref = 0.0
    worst_name = ""
    for (n1, p1), (n2, p2), (n3, p3) in zip(model.named_parameters(), ref32.named_parameters(), ref64.named_parameters()):
        assert n1 == n2 == n3
        if p3.grad is None:
            assert p1.grad is None, n1
            continue
        assert p1.grad is not None, n1
        scale = max(float(p3.grad.abs().max()), 1e-3 * gmax)
        eh = float((p1.grad.cpu().double() - p3.grad).abs().max()) / scale
        if eh > worst_hip:
            worst_hip, worst_name = eh, n1
        worst_ref = max(worst_ref, float((p2.grad.double() - p3.grad).abs().max()) / scale)
    print(f"{name} worst per-parameter gradient error: hip-vs-f64 {worst_hip:.2e} ({worst_name})  oracle32-vs-f64 {worst_ref:.2e}")
    assert worst_hip < 3 * worst_ref + floor


@pytest.mark.parametrize("name,clouds", [("C2_modelnet40", 8), ("C3_scanobjectnn", 2), ("C4_shapenet", 2), ("C5_shapeseg", 2)])
def test_pinned_slots_step_vs_oracle(name, clouds, monkeypatch):
    """The same train-mode step with the max-aggregation SELECTION pinned: the HIP layers report the slot they selected
    per (point, channel), the CPU oracle takes exactly those slots instead of its own arg-max
    (/root/reference/deltaconv/nn/deltaconv.py:52,54: scatter(..., reduce='max')), in fp64 and in fp32, so all three runs
    differentiate the same aggregation branch.  C3 (no normals, round 6): the tangent frames of estimate_basis carry a
    gauge (the sign of the SVD's x-axis, the x-axis itself on near-isotropic neighbourhoods: SURVEY.md section 8(a) a3), so
    the oracle is handed the frames the HIP path computed (dc_estimate_basis, itself pinned to the reference's fp64 frames by
    tests/test_gpu_geometry.py::test_estimate_basis) -- with gauge and selection fixed, C3 sits under the same FLAT bounds
    as the configurations with given normals.  What still separates them: rounding, and the KINKS of LeakyReLU / ReLU -- a
    pre-activation within rounding of zero flips its slope between two fp32 runs, and with a mean loss over a few thousand
    rows ONE flipped element moves one row of a weight gradient by ~1/sqrt(rows) of its largest entry (measured on the
    2-cloud ShapeNet step: a single flip in the head = 7e-3 in max norm; tools/debug/c4_head_grads.py).  A max-norm bound
    therefore cannot be tighter than the reference's own fp32-vs-fp64 gap whatever is pinned.  The criterion that CAN be
    flat is the L1 error of every parameter gradient relative to its own L1 norm (a flipped element touches one row /
    column, a sign or scale error in a tensor touches all of it: L1 error of order 1): <= 3e-3 for every parameter, small
    tensors judged on their own scale; logits flat 1e-4.  The max-norm errors are printed beside the fp32 oracle's."""
    from deltaconv_amd.data import Batch
    from deltaconv_amd.nn import layer as L
    B, N, k, normals, kind, kw, bkw = CONFIGS[name]
    b = synthetic_batch(clouds, N, seed=73, normals=normals, **dict(bkw))
    seg = kind == "seg"
    model = _build(kind, kw, k)
    ocls = oracle.models.DeltaNetSegmentation if seg else oracle.models.DeltaNetClassification
    ref32 = ocls(num_neighbors=k, **kw)
    ref32.load_state_dict(model.state_dict())
    ref64 = ocls(num_neighbors=k, **kw).double()
    ref64.load_state_dict(model.state_dict())
    ref32, ref64 = _no_dropout(ref32.train()), _no_dropout(ref64.train())
    model = _no_dropout(model.to(DEV).train())
    bd = b.to(DEV)
    if not normals:
        import deltaconv_amd as dc
        frames = [t.cpu() for t in dc.geometry.estimate_basis(bd.pos, dc.geometry.Graph.knn(bd.pos, 10, bd.batch),
                                                              orientation=bd.pos)]
        monkeypatch.setattr(oracle.geometry, "estimate_basis",
                            lambda pos, nbr, orientation=None: tuple(f.to(pos.dtype) for f in frames))
    L.SLOT_TAP[0] = []
    try:
        ld = model(bd)
        slots = [s.cpu() for s in L.SLOT_TAP[0]]
    finally:
        L.SLOT_TAP[0] = None
    oracle.loss.calc_loss(ld, bd.y, smoothing=not seg).backward()
    convs64, convs32 = list(ref64.deltanet_base.convs), list(ref32.deltanet_base.convs)
    assert len(slots) == len(convs64), (len(slots), len(convs64))
    for c64, c32, s in zip(convs64, convs32, slots):
        assert s.shape == (clouds * N, c64.out_channels) and int(s.max()) < k
        c64.pinned_slots = c32.pinned_slots = s
    b64 = Batch(b.pos.double(), b.batch, None if b.norm is None else b.norm.double(), None, b.y,
                None if b.category is None else b.category.double(), b.num_graphs)
    l64 = ref64(b64)
    oracle.loss.calc_loss(l64, b.y, smoothing=not seg).backward()
    l32 = ref32(b)
    oracle.loss.calc_loss(l32, b.y, smoothing=not seg).backward()
    # the pinned selection IS (up to rounding-level ties) the oracle's own maximum: pinning changes nothing but the branch
    for conv in convs64:
        conv.pinned_slots = None
    with torch.no_grad():
        free = ref64(b64)
    assert rel_err(l64.detach(), free) < 1e-4
    e_log = rel_err(ld, l64)
    gmax = max(float(p.grad.abs().max()) for p in ref64.parameters() if p.grad is not None)
    rows = []
    for (n1, p1), (n2, p2), (n3, p3) in zip(model.named_parameters(), ref32.named_parameters(), ref64.named_parameters()):
        assert n1 == n2 == n3
        if p3.grad is None:
            assert p1.grad is None, n1
            continue
        assert p1.grad is not None, n1
        g1, g2, g3 = p1.grad.cpu().double(), p2.grad.double(), p3.grad
        scale = max(float(g3.abs().max()), 1e-6 * gmax)
        l1 = max(float(g3.abs().sum()), 1e-6 * gmax * g3.numel())
        rows.append((float((g1 - g3).abs().sum()) / l1, float((g2 - g3).abs().sum()) / l1, float((g1 - g3).abs().max()) / scale,
                     float((g2 - g3).abs().max()) / scale, scale / gmax, n1))
    rows.sort(reverse=True)
    e_log32 = rel_err(l32, l64)
    print(f"{name} pinned slots: logits {e_log:.2e} (oracle32 {e_log32:.2e}); worst parameters by L1 error (hip L1, oracle32 L1, hip max, oracle32 max, own scale / largest):")
    for r in rows[:5]:
        print(f"   hip {r[0]:.2e}  oracle32 {r[1]:.2e}  | max norm: hip {r[2]:.2e}  oracle32 {r[3]:.2e}  scale {r[4]:.1e}  {r[5]}")
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    print(f"   median hip L1 {med:.2e}; worst max-norm: hip {max(r[2] for r in rows):.2e}  oracle32 {max(r[3] for r in rows):.2e}")
    # logits: flat 1e-4 with given normals (measured 2e-6 .. 5e-6); the noisy no-normals configuration (5 % outliers, jitter,
    # lambda = 1e-2) is an order more rounding-sensitive in ANY fp32 implementation -- the oracle's own fp32 run sits at
    # ~1e-4 of its fp64 run there (printed) -- flat 5e-4 (measured 1.7e-4)
    assert e_log < (1e-4 if normals else 5e-4)
    assert rows[0][0] < 3e-3, rows[0]
