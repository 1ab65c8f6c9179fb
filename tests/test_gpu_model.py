"""GPU parity of the layer / model level: deltaconv_amd (HIP kernels behind the reference's
nn / models API) vs the reference's golden vectors and vs the CPU oracle run on the same inputs
and weights.  Tolerances (scale-relative, tests/helpers.rel_err): layer outputs / grads 2e-3 vs
the reference's fp32 numbers (bounded by the reference's own fp32 LU in build_grad_div), 1e-3 vs
the fp64 run; whole-model logits 1e-3 (5e-3 without normals); measured deviations are 100x smaller
(profiles/r01l_parity_report.txt)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import geometry as geo
from tests.helpers import load_golden, rel_err
from tests.golden.probes import probe_vec, param_summaries, state_checksum
from deltaconv_amd.data import Batch, synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"

CONV_CFGS = {"cent": dict(ci=3, co=8, centralized=True, vector=True),
             "plain": dict(ci=8, co=16, centralized=False, vector=True),
             "last": dict(ci=8, co=8, centralized=False, vector=False),
             # DeltaConv(aggr=...) other than the default (fixture deltaconv_layers_aggr.npz)
             "mean": dict(ci=8, co=16, centralized=False, vector=True, aggr="mean"),
             "min": dict(ci=8, co=8, centralized=False, vector=False, aggr="min"),
             "sumc": dict(ci=3, co=8, centralized=True, vector=True, aggr="sum")}


def _layer_fixture(cname):
    return load_golden("deltaconv_layers_aggr" if "aggr" in CONV_CFGS[cname] else "deltaconv_layers")


@pytest.mark.parametrize("cname", list(CONV_CFGS))
def test_deltaconv_layer_golden(cname):
    import deltaconv_amd as dc
    g = _layer_fixture(cname)
    cfg = CONV_CFGS[cname]
    pos, normal, batch = g["pos"].to(DEV), g["normal"].to(DEV), g["batch"].to(DEV)
    ei = g["edge_index"].to(DEV)
    xb, yb = dc.geometry.build_tangent_basis(normal)
    grad, div = dc.geometry.build_grad_div(pos, normal, xb, yb, ei, batch, regularizer=g["lam"])
    conv = dc.nn.DeltaConv(cfg["ci"], cfg["co"], 1, cfg["centralized"], cfg["vector"], cfg.get("aggr", "max"))
    assert repr(conv) == f'DeltaConv({cfg["ci"]}, {cfg["co"]})'
    sd = {k[len(cname) + 4:]: v for k, v in g.items() if k.startswith(cname + "_sd_")}
    res = conv.load_state_dict(sd, strict=False)
    assert all("num_batches" in m for m in res.missing_keys) and not res.unexpected_keys
    conv = conv.to(DEV).train()
    x = g[f"{cname}_x"].to(DEV).requires_grad_(True)
    v = g[f"{cname}_v"].to(DEV).requires_grad_(True)
    xo, vo = conv(x, v, grad, div, ei)
    loss = (xo * probe_vec(tuple(xo.shape), 11).float().to(DEV)).sum()
    if cfg["vector"]:
        loss = loss + (vo * probe_vec(tuple(vo.shape), 12).float().to(DEV)).sum()
    else:
        assert vo is v
    loss.backward()
    for tag, tol in (("f32", 2e-3), ("f64", 1e-3)):
        assert rel_err(xo, g[f"{cname}_xo_{tag}"]) < tol
        assert rel_err(vo, g[f"{cname}_vo_{tag}"]) < tol
        assert rel_err(x.grad, g[f"{cname}_dx_{tag}"]) < tol
        assert rel_err(v.grad, g[f"{cname}_dv_{tag}"]) < tol
        for n_, p_ in conv.named_parameters():
            key = f"{cname}_g_{n_}_{tag}"
            if key in g:
                assert rel_err(p_.grad, g[key]) < 5 * tol, n_
            else:
                assert p_.grad is None, n_
        for n_, b_ in conv.named_buffers():
            if "running" in n_:
                assert rel_err(b_, g[f"{cname}_buf_{n_}_{tag}"]) < tol, n_


MODELS = {
    "model_cls_B4_N256_k20": ("cls", dict(in_channels=3, num_classes=40), True),
    "model_seg_B2_N256_k20": ("seg", dict(in_channels=3, num_classes=50, categorical_vector=True), True),
    "model_cls_nonormals_B2_N256_k20": ("cls", dict(in_channels=3, num_classes=15, conv_channels=[64, 64, 64, 128]), False),
    "model_seg_depth1_B2_N1024_k30": ("seg", dict(in_channels=3, num_classes=8, conv_channels=[32] * 4, mlp_depth=1,
                                                  embedding_size=128), True),
}


def _model(kind, kw, k, lam):
    import deltaconv_amd as dc
    torch.manual_seed(1)
    cls = dc.models.DeltaNetSegmentation if kind == "seg" else dc.models.DeltaNetClassification
    return cls(num_neighbors=k, grad_regularizer=lam, **kw)


def _no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    return model


@pytest.mark.parametrize("name", list(MODELS))
def test_model_step_golden(name):
    """Same seed -> same init as the reference (checksum), then one train-mode step."""
    kind, kw, normals = MODELS[name]
    g = load_golden(name)
    model = _model(kind, kw, g["k"], g["lam"])
    # |w| sums in double: identical weights, but the reduction order depends on the host thread count
    assert np.allclose(state_checksum(model), g["state_checksum"].numpy(), rtol=1e-9, atol=0)
    model = _no_dropout(model.to(DEV).train())
    data = Batch(g["pos"], g["batch"], g["normal"] if normals else None, None, g["y"],
                 g["category"] if "category" in g else None).to(DEV)
    logits = model(data)
    loss = oracle.loss.calc_loss(logits, data.y, smoothing=(kind != "seg"))
    loss.backward()
    # measured (profiles/r01l_parity_report.txt): 9e-6 with normals, 2e-4 without (SVD-sign gauge +
    # ill-defined x axis; the reference's own fp32-vs-fp64 gap there is 3e-4)
    tol = 1e-3 if normals else 5e-3
    assert rel_err(logits, g["logits_f64"]) < tol
    assert rel_err(logits, g["logits_f32"]) < tol
    assert abs(float(loss) - float(g["loss_f64"])) < tol * abs(float(g["loss_f64"]))
    names, norms, dots = param_summaries(model)
    assert names == [str(s) for s in g["gnames"]]
    gn = g["gnorm_f64"].numpy()
    den = gn + 1e-3 * gn.max()
    # fp64 = truth; the bound is self-calibrated like the element-wise one below: the reference's OWN fp32 run sits up to
    # 2.6e-3 from its fp64 run on these summaries (k = 30 fixture: max-aggregation argmax choices flip with any rounding)
    ref_gap = lambda key: float(np.max(np.abs(g[key + "_f32"].numpy() - g[key + "_f64"].numpy()) / den)) if key + "_f32" in g else 0.0
    assert np.max(np.abs(np.array(norms) - gn) / den) < 5 * tol + ref_gap("gnorm")
    # signed probe dot products (a sign / permutation error in a weight gradient changes these, not the norms)
    gd = g["gdot_f64"].numpy()
    assert np.max(np.abs(np.array(dots) - gd) / den) < 5 * tol + ref_gap("gdot")
    # and two weight gradients element by element (first edge MLP, first vector MLP)
    params = dict(model.named_parameters())
    # (fp64 = truth.  The reference's own fp32 run is only as close to truth as its max-aggregation argmax
    # choices allow -- 1.2e-2 on the edge-MLP weight of the B4 fixture -- so against it the bound is self-calibrated.)
    for pkey in ("deltanet_base.convs.0.s_mlp_max.0.0.weight", "deltanet_base.convs.1.v_mlp.0.0.weight"):
        g64, g32 = g[f"g_{pkey}_f64"], g[f"g_{pkey}_f32"]
        assert rel_err(params[pkey].grad, g64) < 5 * tol, pkey
        assert rel_err(params[pkey].grad, g32) < 5 * tol + rel_err(g32, g64), pkey
    key = ("lin_global" if kind == "seg" else "lin_embedding") + ".0.1.bn.running_mean"
    assert rel_err(dict(model.named_buffers())[key], g["rm_embed_f64"]) < tol
    assert rel_err(dict(model.named_buffers())[key.replace("running_mean", "running_var")], g["rv_embed_f64"]) < tol


@pytest.mark.parametrize("name", list(MODELS))
def test_model_step_golden_exact_chain_fixed_bound(name):
    """The same whole-model step with every dense product on the exact fp32 MFMA chain (dc_set_option(3, 1) = DC_GEMM_EXACT=1),
    gradient summaries against the reference's fp64 run at the FIXED bound 5 tol -- no self-calibration term: a real regression
    in the kernels cannot hide behind the reference's own fp32-vs-fp64 gap (round-4 verdict, weak point 1)."""
    from deltaconv_amd._lib import lib
    kind, kw, normals = MODELS[name]
    g = load_golden(name)
    model = _no_dropout(_model(kind, kw, g["k"], g["lam"]).to(DEV).train())
    data = Batch(g["pos"], g["batch"], g["normal"] if normals else None, None, g["y"],
                 g["category"] if "category" in g else None).to(DEV)
    lib.raw("dc_set_option")(3, 1)
    try:
        logits = model(data)
        oracle.loss.calc_loss(logits, data.y, smoothing=(kind != "seg")).backward()
        torch.cuda.synchronize()
    finally:
        lib.raw("dc_set_option")(3, 0)
    tol = 1e-3 if normals else 5e-3
    assert rel_err(logits, g["logits_f64"]) < tol
    names, norms, dots = param_summaries(model)
    gn = g["gnorm_f64"].numpy()
    den = gn + 1e-3 * gn.max()
    e_norm = float(np.max(np.abs(np.array(norms) - gn) / den))
    e_dot = float(np.max(np.abs(np.array(dots) - g["gdot_f64"].numpy()) / den))
    print(f"exact chain, {name}: gradient norms {e_norm:.2e}, probe dot products {e_dot:.2e} (fixed bound {5 * tol:.0e})")
    assert e_norm < 5 * tol and e_dot < 5 * tol


@pytest.mark.parametrize("B,N,k", [(2, 512, 20), (8, 1024, 20)])
def test_model_step_vs_oracle(B, N, k):
    """Beyond the fixtures: a bigger batch against the CPU oracle on identical inputs and weights,
    every parameter gradient compared tensor by tensor.  The tolerance is self-calibrating: the
    oracle is run in fp64 ("truth") and in fp32 (= the reference's own numerics); max-aggregation
    and max-pooling make gradients piecewise (an argmax can flip under fp32 rounding), so the HIP
    path is required to be no further from truth than 3x the reference's fp32 run (+1e-3)."""
    b = synthetic_batch(B, N, seed=40)
    torch.manual_seed(1)
    ref32 = _no_dropout(oracle.models.DeltaNetClassification(3, 40, num_neighbors=k).train())
    ref64 = _no_dropout(oracle.models.DeltaNetClassification(3, 40, num_neighbors=k).double().train())
    ref64.load_state_dict(ref32.state_dict())
    model = _model("cls", dict(in_channels=3, num_classes=40), k, 1e-3)
    model.load_state_dict(ref32.state_dict())
    model = _no_dropout(model.to(DEV).train())
    l32 = ref32(b)
    oracle.loss.calc_loss(l32, b.y).backward()
    b64 = Batch(b.pos.double(), b.batch, b.norm.double(), None, b.y)
    l64 = ref64(b64)
    oracle.loss.calc_loss(l64, b.y).backward()
    bd = b.to(DEV)
    ld = model(bd)
    oracle.loss.calc_loss(ld, bd.y).backward()
    e_hip, e_ref = rel_err(ld, l64), rel_err(l32, l64)
    print(f"logits: hip-vs-f64 {e_hip:.2e}  oracle32-vs-f64 {e_ref:.2e}")
    assert e_hip < 3 * e_ref + 1e-3
    gmax = max(float(p.grad.abs().max()) for p in ref64.parameters() if p.grad is not None)
    worst_hip = worst_ref = 0.0
    for (n1, p1), (n2, p2), (n3, p3) in zip(model.named_parameters(), ref32.named_parameters(),
                                            ref64.named_parameters()):
        assert n1 == n2 == n3
        if p3.grad is None:
            assert p1.grad is None, n1
            continue
        scale = max(float(p3.grad.abs().max()), 1e-3 * gmax)
        worst_hip = max(worst_hip, float((p1.grad.cpu().double() - p3.grad).abs().max()) / scale)
        worst_ref = max(worst_ref, float((p2.grad.double() - p3.grad).abs().max()) / scale)
    print(f"worst per-parameter grad error: hip-vs-f64 {worst_hip:.2e}  oracle32-vs-f64 {worst_ref:.2e}")
    assert worst_hip < 3 * worst_ref + 1e-3


def test_training_trajectory_vs_oracle():
    """Five SGD steps (momentum / weight decay of train_modelnet.py:67; a tenth of its learning rate, so that fp32
    rounding noise is not amplified into O(1) weight differences by the chaotic early steps -- at lr = 0.1 the
    reference's own fp32 run ends 2.0 away from its fp64 run; a new batch every step) from identical weights: the loss
    trajectory and the weights after the last step, HIP path vs the oracle in fp64 ("truth"), with the oracle in fp32
    (= the reference's numerics) as the yardstick: BatchNorm running statistics, momentum buffers and the weight decay
    all take part, so an error in any update rule compounds over the steps."""
    k, steps = 20, 5
    torch.manual_seed(1)
    ref32 = _no_dropout(oracle.models.DeltaNetClassification(3, 40, num_neighbors=k).train())
    ref64 = _no_dropout(oracle.models.DeltaNetClassification(3, 40, num_neighbors=k).double().train())
    ref64.load_state_dict(ref32.state_dict())
    model = _model("cls", dict(in_channels=3, num_classes=40), k, 1e-3)
    model.load_state_dict(ref32.state_dict())
    model = _no_dropout(model.to(DEV).train())
    mk = lambda m: torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    o32, o64, od = mk(ref32), mk(ref64), mk(model)
    losses = []
    for i in range(steps):
        b = synthetic_batch(4, 512, seed=60 + i)
        b64 = Batch(b.pos.double(), b.batch, b.norm.double(), None, b.y)
        row = []
        for m, o, bb in ((ref64, o64, b64), (ref32, o32, b), (model, od, b.to(DEV))):
            o.zero_grad()
            loss = oracle.loss.calc_loss(m(bb), bb.y)
            loss.backward()
            o.step()
            row.append(float(loss.detach()))
        losses.append(row)
    l64, l32, lhip = (torch.tensor([r[j] for r in losses], dtype=torch.float64) for j in range(3))
    e_ref, e_hip = float((l32 - l64).abs().max()), float((lhip - l64).abs().max())
    print(f"loss trajectory {lhip.tolist()}: hip-vs-f64 {e_hip:.2e}  oracle32-vs-f64 {e_ref:.2e}")
    assert e_hip < 3 * e_ref + 2e-3 * float(l64.abs().max())
    worst_hip = worst_ref = 0.0
    for (n1, p1), (n2, p2), (n3, p3) in zip(model.state_dict().items(), ref32.state_dict().items(),
                                            ref64.state_dict().items()):
        assert n1 == n2 == n3
        if not p3.dtype.is_floating_point:
            assert int(p1) == int(p3), n1                    # num_batches_tracked
            continue
        scale = max(float(p3.abs().max()), 1e-3)
        worst_hip = max(worst_hip, float((p1.cpu().double() - p3).abs().max()) / scale)
        worst_ref = max(worst_ref, float((p2.double() - p3).abs().max()) / scale)
    print(f"weights / buffers after {steps} steps: hip-vs-f64 {worst_hip:.2e}  oracle32-vs-f64 {worst_ref:.2e}")
    assert worst_hip < 3 * worst_ref + 2e-3


def test_eval_mode_and_determinism():
    b = synthetic_batch(2, 512, seed=41).to(DEV)
    model = _model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).eval()
    with torch.no_grad():
        a1 = model(b)
        a2 = model(b)
    assert torch.equal(a1, a2) and a1.shape == (2, 40) and not torch.isnan(a1).any()
    model.train()
    _no_dropout(model)
    grads = []
    for _ in range(2):
        model.zero_grad()
        oracle.loss.calc_loss(model(b), b.y).backward()
        grads.append(torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).clone())
    # our kernels are atomics-free and ordered; any residual difference would come from torch's GEMM/BN
    assert rel_err(grads[0], grads[1]) < 1e-6


def test_gauge_invariance_of_parameter_grads():
    """Reference test_deltaconv (test/nn/test_deltaconv.py:42-74): parameter gradients must not
    depend on the choice of tangent basis (atol relaxed to 1e-4: fails at 1e-5 on the reference
    itself in fp32)."""
    import deltaconv_amd as dc
    import torch.nn.functional as F
    torch.manual_seed(1)
    N = 1000
    x = torch.rand(N, 3, device=DEV)
    ei = dc.geometry.knn_graph(x, 20)
    normal, xb, yb = dc.geometry.estimate_basis(x, ei)
    grad, div = dc.geometry.build_grad_div(x, normal, xb, yb, ei, regularizer=1e-8)
    xr = geo.rotate_around(xb.cpu(), normal.cpu(), torch.rand(N) * 2 * torch.pi).to(DEV)
    yr = torch.linalg.cross(normal, xr)
    grad_r, div_r = dc.geometry.build_grad_div(x, normal, xr, yr, ei, regularizer=1e-8)
    conv = dc.nn.DeltaConv(3, 1, depth=1, centralized=False).to(DEV)
    target = torch.rand(N, 1, device=DEV)
    outs = []
    for G, D in ((grad, div), (grad_r, div_r)):
        conv.zero_grad()
        out, _ = conv(x, G @ x, G, D, ei)
        F.l1_loss(out, target).backward()
        outs.append(torch.cat([p.grad.flatten() for p in conv.parameters() if p.grad is not None]).clone())
    assert torch.allclose(outs[0], outs[1], atol=1e-4)


def test_operator_cache_opt_in():
    """Section 8(f)-3: operators are geometry-only, so a static batch can keep them across eval passes;
    the cache is keyed on the position tensor (identity + version) and is off by default."""
    b = synthetic_batch(2, 256, seed=42).to(DEV)
    model = _model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).eval()
    base = model.deltanet_base
    with torch.no_grad():
        ref = model(b)
        assert not hasattr(b, "_dc_ops")
        base.cache_operators = True
        o1 = model(b)
        g1 = b._dc_ops[1][0]
        o2 = model(b)
        assert b._dc_ops[1][0] is g1 and torch.equal(o1, ref) and torch.equal(o2, ref)
        b.pos.mul_(1.1)                       # geometry changed in place -> rebuilt
        model(b)
        assert b._dc_ops[1][0] is not g1


def test_graphed_step_matches_eager():
    """HIP-graph replay of forward+loss+backward == the eager step (same kernels, same order): logits and
    every gradient bit-identical, also after loading a different batch into the captured inputs."""
    from deltaconv_amd.graph_step import GraphedTrainStep
    from deltaconv_amd.utils import calc_loss
    b1 = synthetic_batch(4, 256, seed=43).to(DEV)
    b2 = synthetic_batch(4, 256, seed=44).to(DEV)
    model = _no_dropout(_model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).train())
    sd0 = {k_: t.clone() for k_, t in model.state_dict().items()}

    def eager(batch):
        model.load_state_dict(sd0)
        model.zero_grad(set_to_none=True)
        out = model(batch)
        calc_loss(out, batch.y).backward()
        return out.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]).clone()

    e1, e2 = eager(b1), eager(b2)
    model.load_state_dict(sd0)
    static = synthetic_batch(4, 256, seed=43).to(DEV)
    step = GraphedTrainStep(model, calc_loss, static)
    for batch, ref in ((b1, e1), (b2, e2), (b1, e1)):
        model.load_state_dict(sd0)
        step(batch)
        g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
        assert torch.equal(step.out, ref[0]) and torch.equal(g, ref[1])


def test_graphed_training_matches_eager():
    """Captured step INCLUDING the SGD update, replayed on changing batches with eager kernels in between
    (the situation in which ROCm 7.2's AQL-packet capture replays stale arguments): parameters, BatchNorm
    statistics and losses bit-identical to the same steps run eagerly."""
    from deltaconv_amd.graph_step import GraphedTrainStep
    from deltaconv_amd.utils import calc_loss
    batches = [synthetic_batch(4, 256, seed=50 + i).to(DEV) for i in range(3)]

    def make():
        torch.manual_seed(5)
        m = _no_dropout(_model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).train())
        return m, torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)

    def eager_step(m, opt, b):
        for p in m.parameters():
            p.grad = None
        loss = calc_loss(m(b), b.y)
        loss.backward()
        opt.step()
        return float(loss)

    m1, o1 = make()
    ref_losses = [eager_step(m1, o1, batches[0]) for _ in range(3)]          # = the warm-up steps of the capture
    ref_losses += [eager_step(m1, o1, b) for b in (batches[1], batches[2], batches[1], batches[0])]

    m2, o2 = make()
    static = synthetic_batch(4, 256, seed=50).to(DEV)
    step = GraphedTrainStep(m2, calc_loss, static, optimizer=o2, warmup=3)
    scratch = [torch.zeros(128, device=DEV) for _ in range(16)]
    losses = []
    for b in (batches[1], batches[2], batches[1], batches[0]):
        losses.append(float(step(b)))
        for t in scratch:                                                     # small eager launches between replays
            t.add_(1.0)
    assert losses == ref_losses[3:]
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for key in sd1:
        assert torch.equal(sd1[key], sd2[key]), key


def test_graph_capture_does_not_reuse_weight_planes_cut_before_it():
    """One eager forward/backward pass (no update) registers the weights with the pre-split plane cache and cuts their planes
    on the spot, but the one-launch table only exists from the SECOND forward pass on.  A capture right after that pass
    (warmup=0) finds those planes current -- and must still not bake them into the graph: replays would multiply by the weights
    of the capture instant while the captured SGD update moves the real ones (tools/debug/stale_planes_control.py: with the
    rule defeated this test fails).  Checked without following a trajectory (at this learning rate two runs drift apart from
    rounding alone): the logits a replay produces == an eager forward pass of a second model holding the weights and
    BatchNorm state the captured model had right before that replay (1e-4 of the largest logit; stale planes are the
    weights of four updates earlier)."""
    from deltaconv_amd.graph_step import GraphedTrainStep
    from deltaconv_amd.nn import fused
    from deltaconv_amd.utils import calc_loss
    batches = [synthetic_batch(4, 256, seed=60 + i).to(DEV) for i in range(2)]

    def make():
        torch.manual_seed(6)
        m = _no_dropout(_model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).train())
        return m, torch.optim.SGD(m.parameters(), lr=0.05, weight_decay=1e-4)     # (no state to create inside the capture)

    fused._planes_reset()
    m2, o2 = make()
    static = synthetic_batch(4, 256, seed=60).to(DEV)
    calc_loss(m2(static), static.y).backward()                  # the one eager pass: planes cut, weights untouched since
    step = GraphedTrainStep(m2, calc_loss, static, optimizer=o2, warmup=0)    # the capture itself does not execute a step
    w0 = m2.deltanet_base.convs[1].s_mlp[0][0].weight.detach().clone()
    for b in (batches[1], batches[0], batches[1]):
        step(b)
    sd = {k_: t.detach().clone() for k_, t in m2.state_dict().items()}
    moved = float((sd["deltanet_base.convs.1.s_mlp.0.0.weight"] - w0).abs().max() / w0.abs().max())
    assert moved > 1e-3, moved                                  # the weights did move under the replays
    step(batches[0])
    got = step.out.clone()
    m3, _ = make()
    m3.load_state_dict(sd)
    with torch.no_grad():
        ref = m3(batches[0])
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, err


# ---- C1 (ModelNet40, 1024 points, k = 20, batch 1: BASELINE.json configs[0]) ----------------------------
def _oracle_pair(kind, kw, k, seed=1):
    """(oracle model fp32, deltaconv_amd model on the GPU) with identical weights."""
    torch.manual_seed(seed)
    ocls = oracle.models.DeltaNetSegmentation if kind == "seg" else oracle.models.DeltaNetClassification
    ref = ocls(num_neighbors=k, **kw)
    import deltaconv_amd as dc
    cls = dc.models.DeltaNetSegmentation if kind == "seg" else dc.models.DeltaNetClassification
    model = cls(num_neighbors=k, **kw)
    model.load_state_dict(ref.state_dict())
    return ref, model.to(DEV)


def _randomize_bn(ref):
    """Non-trivial running statistics / affine parameters, so an eval-mode pass really tests them."""
    gen = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=gen))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=gen))
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))


def test_c1_eval_forward_b1():
    """C1 (i): eval-mode forward of ONE cloud of 1024 points, k = 20, logits vs the oracle."""
    b = synthetic_batch(1, 1024, seed=80)
    ref, model = _oracle_pair("cls", dict(in_channels=3, num_classes=40), 20)
    _randomize_bn(ref)
    model.load_state_dict(ref.state_dict())
    ref.eval(); model.eval()
    with torch.no_grad():
        lo = ref(b)
        ld = model(b.to(DEV))
    assert ld.shape == (1, 40)
    assert rel_err(ld, lo) < 1e-3


def test_c1_train_b1_raises_like_reference():
    """C1 caveat: the reference cannot run a train-mode step at B = 1 -- the head's
    BatchNorm sees one row and torch raises ValueError (nn/nonlin.py:29-30 -> F.batch_norm).  Same here."""
    b = synthetic_batch(1, 1024, seed=81).to(DEV)
    _, model = _oracle_pair("cls", dict(in_channels=3, num_classes=40), 20)
    model.train()
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        model(b)


def test_c1_backbone_train_b1():
    """C1 (ii): forward + backward through DeltaNetBase + lin_embedding at B = 1 (train-mode BatchNorm over the
    1024 rows is well defined) vs the oracle."""
    b = synthetic_batch(1, 1024, seed=82)
    ref, model = _oracle_pair("cls", dict(in_channels=3, num_classes=40), 20)
    ref.train(); model.train()

    def run(m, data):
        xs = m.deltanet_base(data)
        e = m.lin_embedding(torch.cat(list(xs), dim=1))
        w = probe_vec(tuple(e.shape), 31).float().to(e.device)
        (e * w).sum().backward()
        return e
    eo = run(ref, b)
    ed = run(model, b.to(DEV))
    assert rel_err(ed, eo) < 1e-3
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        if p2.grad is None:
            assert p1.grad is None, n1
            continue
        scale = max(float(p2.grad.abs().max()), 1e-3 * gmax)
        assert float((p1.grad.cpu() - p2.grad).abs().max()) / scale < 2e-2, n1


def test_c1_train_step_b2():
    """C1 (iii): the smallest batch the reference can train on (B = 2), one full step vs the oracle (fp64 truth,
    self-calibrated against the oracle's own fp32 run).  With two rows the head's BatchNorm maps its input to
    (+-1) * gamma + beta: the gradient that reaches everything upstream of it is an exact cancellation
    (dz - mean dz - xhat mean(dz xhat) = 0 for xhat = +-1), i.e. rounding noise 1e-3 below the head's own gradients in
    ANY fp32 implementation (the oracle's fp32 run is 5e-3 off per parameter there).  So gradients are compared on
    the scale of the largest gradient of the model, logits and loss as everywhere else."""
    b = synthetic_batch(2, 1024, seed=40)
    torch.manual_seed(1)
    ref32 = _no_dropout(oracle.models.DeltaNetClassification(3, 40, num_neighbors=20).train())
    ref64 = _no_dropout(oracle.models.DeltaNetClassification(3, 40, num_neighbors=20).double().train())
    ref64.load_state_dict(ref32.state_dict())
    model = _model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3)
    model.load_state_dict(ref32.state_dict())
    model = _no_dropout(model.to(DEV).train())
    l32 = ref32(b)
    oracle.loss.calc_loss(l32, b.y).backward()
    l64 = ref64(Batch(b.pos.double(), b.batch, b.norm.double(), None, b.y))
    loss64 = oracle.loss.calc_loss(l64, b.y)
    loss64.backward()
    bd = b.to(DEV)
    ld = model(bd)
    loss = oracle.loss.calc_loss(ld, bd.y)
    loss.backward()
    assert rel_err(ld, l64) < 3 * rel_err(l32, l64) + 1e-3
    assert abs(float(loss) - float(loss64)) < 1e-4 * abs(float(loss64))
    gmax = max(float(p.grad.abs().max()) for p in ref64.parameters() if p.grad is not None)
    worst_hip = worst_ref = 0.0
    for (n1, p1), (n2, p2), (n3, p3) in zip(model.named_parameters(), ref32.named_parameters(), ref64.named_parameters()):
        if p3.grad is None:
            assert p1.grad is None, n1
            continue
        worst_hip = max(worst_hip, float((p1.grad.cpu().double() - p3.grad).abs().max()) / gmax)
        worst_ref = max(worst_ref, float((p2.grad.double() - p3.grad).abs().max()) / gmax)
    print(f"worst grad error / largest gradient: hip-vs-f64 {worst_hip:.2e}  oracle32-vs-f64 {worst_ref:.2e}")
    # max-aggregation / max-pooling are piecewise: ONE arg-max that flips under fp32 rounding moves a few gradient
    # entries by ~1e-2 of the largest gradient, in either implementation (measured with tools/archive_r01_r04.tar.gz:archive/debug_b2.py: vendor-GEMM
    # path 1.1e-2 at B = 4, hand-written-GEMM path 1.8e-2 at B = 2, 8e-5 where nothing flips) -- hence the floor
    assert worst_hip < max(3 * worst_ref + 1e-3, 3e-2)


@pytest.mark.parametrize("kind,kw,bkw", [
    ("cls", dict(in_channels=3, num_classes=40), {}),
    ("seg", dict(in_channels=3, num_classes=50, categorical_vector=True),
     dict(per_point_labels=True, categories=16, num_classes=50)),
    ("cls", dict(in_channels=3, num_classes=15, conv_channels=[64, 64, 64, 128], grad_regularizer=1e-2),
     dict(normals=False, num_classes=15)),
])
def test_eval_mode_logits_vs_oracle(kind, kw, bkw):
    """Whole-model eval-mode parity (running statistics, dropout off): classification, segmentation (depth-2
    MLPs + categorical vector), and the no-normals (estimate_basis) variant."""
    b = synthetic_batch(3, 512, seed=83, **bkw)
    ref, model = _oracle_pair(kind, kw, 20)
    _randomize_bn(ref)
    model.load_state_dict(ref.state_dict())
    ref.eval(); model.eval()
    with torch.no_grad():
        lo = ref(b)
        ld = model(b.to(DEV))
    assert ld.shape == lo.shape
    assert rel_err(ld, lo) < (5e-3 if bkw.get("normals") is False else 1e-3)


def test_eval_coefficient_cache_follows_the_checkpoint():
    """Inference folds every BatchNorm into a per-channel affine map applied by the consuming kernel; under
    torch.no_grad() the maps are cached on the running-mean buffers (nn/fused.py: eval_coeffs).  The cache must notice
    every way the checkpoint can change: in-place updates, load_state_dict, a train step in between."""
    b = synthetic_batch(2, 256, seed=90).to(DEV)
    model = _model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).eval()
    with torch.no_grad():
        a1 = model(b)
        a2 = model(b)                                                      # served from the cache
        assert torch.equal(a1, a2)
        bn = model.deltanet_base.convs[1].s_mlp[0][1].bn
        assert getattr(bn.running_mean, "_dc_eval_coeffs", None) is not None
        bn.running_mean.add_(0.25)                                         # in-place: version bump
        a3 = model(b)
        assert not torch.equal(a1, a3)
        sd = {k_: v.clone() for k_, v in model.state_dict().items()}
        sd["deltanet_base.convs.1.s_mlp.0.1.bn.running_mean"] -= 0.25
        model.load_state_dict(sd)
        assert torch.equal(model(b), a1)
    model.train()
    _no_dropout(model)
    oracle.loss.calc_loss(model(b), b.y).backward()                        # running statistics move
    model.eval()
    fresh = _model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).eval()
    fresh.load_state_dict(model.state_dict())
    with torch.no_grad():
        assert torch.equal(model(b), fresh(b))


def test_eval_coefficient_cache_survives_graph_replays():
    """A HIP-graph replay of a training step (optimizer captured) moves parameters and running statistics through raw
    pointers: no tensor version changes.  train (replay) -> eval -> train (replay) -> eval must not serve the second
    eval from the first one's cached BatchNorm maps (round-2 advisor finding): compare with a fresh model loaded from
    the state_dict."""
    from deltaconv_amd.graph_step import GraphedTrainStep
    from deltaconv_amd.utils import calc_loss
    b = synthetic_batch(4, 256, seed=91).to(DEV)
    torch.manual_seed(7)
    model = _no_dropout(_model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).train())
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    step = GraphedTrainStep(model, calc_loss, synthetic_batch(4, 256, seed=91).to(DEV), optimizer=opt, warmup=2)

    def eval_pair():
        model.eval()
        fresh = _model("cls", dict(in_channels=3, num_classes=40), 20, 1e-3).to(DEV).eval()
        fresh.load_state_dict(model.state_dict())
        with torch.no_grad():
            got, want = model(b), fresh(b)
        model.train()
        return got, want

    step(b)
    g1, w1 = eval_pair()
    assert torch.equal(g1, w1)
    step(b)
    g2, w2 = eval_pair()
    assert torch.equal(g2, w2)
    assert not torch.equal(g1, g2)                                         # the checkpoint did move
