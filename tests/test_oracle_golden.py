"""Pin the CPU oracle against outputs of the REAL reference (tests/golden/*.npz, produced by
tests/golden/make_golden.py from /root/reference).  CPU-only; no HIP involved."""
import numpy as np
import pytest
import torch
from types import SimpleNamespace as NS

import oracle
from oracle import geometry as geo
from tests.helpers import load_golden, rel_err

GEOM = ["geom_normals_B2_N128_k20", "geom_ragged_dups_k30", "geom_nonormals_N200_k10"]


def _ptr(g):
    return geo.cloud_ptr(g["batch"], g["pos"].shape[0])


@pytest.mark.parametrize("name", GEOM)
def test_knn_bit_exact(name):
    g = load_golden(name)
    nbr = geo.knn(g["pos"], g["k"], _ptr(g))
    assert torch.equal(geo.edge_index_from_nbr(nbr), g["edge_index"])
    assert torch.equal(geo.nbr_from_edge_index(g["edge_index"]), nbr)


@pytest.mark.parametrize("name", GEOM)
@pytest.mark.parametrize("tag,tol", [("f32", 2e-3), ("f64", 1e-6)])
def test_geometry_stages(name, tag, tol):
    """Rows a2-a10 of SURVEY.md section 8(a): each stage fed the reference's own upstream values."""
    g = load_golden(name)
    dt = torch.float32 if tag == "f32" else torch.float64
    pos, ptr, k = g["pos"].to(dt), _ptr(g), g["k"]
    nbr = geo.nbr_from_edge_index(g["edge_index"], k)
    normal, xb, yb = g[f"normal_{tag}"], g[f"x_basis_{tag}"], g[f"y_basis_{tag}"]
    if "nonormals" not in name:
        x2, y2 = geo.build_tangent_basis(normal)
        assert rel_err(x2, xb) < 1e-6 and rel_err(y2, yb) < 1e-6
    else:
        n3, x3, y3 = geo.estimate_basis(pos, geo.nbr_from_edge_index(g["edge_index10"], 10), orientation=pos)
        assert rel_err(n3, normal) < 1e-4
        sgn = torch.sign((x3 * xb).sum(1, keepdim=True))          # U[:,0] is defined up to sign
        assert rel_err(x3 * sgn, xb) < 1e-3 and rel_err(y3 * sgn, yb) < 1e-3
    nt = pos.shape[0]
    coords = geo.coords_projected(pos, normal, xb, yb, nbr)
    assert rel_err(coords.reshape(-1, 2), g[f"coords_{tag}"]) < 1e-6
    dist = (pos[nbr] - pos[:, None]).norm(dim=-1)
    w = geo.gaussian_weights(dist, ptr, g["h"])
    assert rel_err(w.reshape(-1), g[f"weights_{tag}"]) < 1e-5
    wls = geo.weighted_least_squares(g[f"coords_{tag}"].view(nt, k, 2), g[f"weights_{tag}"].view(nt, k), g["lam"])
    assert rel_err(wls.reshape(-1, 6), g[f"wls_{tag}"]) < tol
    vmap = geo.fit_vector_mapping(pos, normal, xb, yb, nbr, g[f"wls_{tag}"].view(nt, k, 6), coords)
    assert rel_err(vmap.reshape(-1, 2, 2), g[f"vmap_{tag}"]) < 1e-5
    grad, div = geo.build_grad_div(pos, normal, xb, yb, nbr, ptr, g["h"], g["lam"])
    for op, nm in ((grad, "grad"), (div, "div")):
        row, col, val = op.to_coo()
        assert torch.equal(row, g[f"{nm}_row_{tag}"]) and torch.equal(col, g[f"{nm}_col_{tag}"])
        assert rel_err(val, g[f"{nm}_val_{tag}"]) < tol
        assert op.size(0) == (2 * nt if nm == "grad" else nt) and op.size(1) == (nt if nm == "grad" else 2 * nt)
    # operator algebra on the reference's own coefficient values
    G = geo.EllOp("grad", nbr, g[f"grad_val_{tag}"].view(nt, k, 2))
    D = geo.EllOp("div", nbr, g[f"div_val_{tag}"].view(nt, k, 2))
    x, v = g["x_in"].to(dt), g["v_in"].to(dt)
    checks = dict(grad_x=G @ x, div_v=D @ v, curl_v=geo.curl(v, D), lap_x=geo.laplacian(x, G, D),
                  hodge_v=geo.hodge_laplacian(v, G, D), norm_v=geo.norm(v), IJ_v=geo.I_J(v))
    for key, val in checks.items():
        assert rel_err(val, g[f"{key}_{tag}"]) < (1e-5 if tag == "f32" else 1e-12), key


@pytest.mark.parametrize("tag,tol", [("f32", 2e-3), ("f64", 1e-7)])
def test_build_grad_div_shape_regularizer(tag, tol):
    """build_grad_div(shape_regularizer=...) (grad_div_mls.py:241-244,266-267) against the reference's values."""
    g = load_golden("geom_shape_regularizer")
    dt = torch.float32 if tag == "f32" else torch.float64
    pos, normal = g["pos"].to(dt), g["normal"].to(dt)
    nt, k = pos.shape[0], int(g["k"])
    ptr = geo.cloud_ptr(g["batch"])
    nbr = geo.nbr_from_edge_index(g["edge_index"], k)
    xb, yb = geo.build_tangent_basis(normal)
    grad, div = geo.build_grad_div(pos, normal, xb, yb, nbr, ptr, 1.0, float(g["lam"]),
                                   shape_regularizer=float(g["lam_shape"]))
    assert rel_err(grad.coef.reshape(-1), g[f"grad_val_{tag}"]) < tol
    assert rel_err(div.coef.reshape(-1), g[f"div_val_{tag}"]) < tol
    _, div0 = geo.build_grad_div(pos, normal, xb, yb, nbr, ptr, 1.0, float(g["lam"]))
    assert rel_err(div0.coef.reshape(-1), g[f"div_val_{tag}"]) > 1e-3          # the second regulariser matters here


def _load_conv(g, cname, cfg, dt):
    conv = oracle.nn.DeltaConv(cfg["ci"], cfg["co"], 1, cfg["centralized"], cfg["vector"], cfg.get("aggr", "max"))
    sd = {k[len(cname) + 4:]: v for k, v in g.items() if k.startswith(cname + "_sd_")}
    missing = conv.load_state_dict(sd, strict=False)
    assert all("num_batches" in m for m in missing.missing_keys) and not missing.unexpected_keys
    return conv.to(dt).train()


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
@pytest.mark.parametrize("tag,tol", [("f32", 2e-3), ("f64", 1e-6)])
def test_deltaconv_layer(cname, tag, tol):
    g = _layer_fixture(cname)
    dt = torch.float32 if tag == "f32" else torch.float64
    cfg = CONV_CFGS[cname]
    pos, normal = g["pos"].to(dt), g["normal"].to(dt)
    ptr = geo.cloud_ptr(g["batch"])
    nbr = geo.nbr_from_edge_index(g["edge_index"], g["k"])
    xb, yb = geo.build_tangent_basis(normal)
    grad, div = geo.build_grad_div(pos, normal, xb, yb, nbr, ptr, 1.0, g["lam"])
    conv = _load_conv(g, cname, cfg, dt)
    x = g[f"{cname}_x"].to(dt).requires_grad_(True)
    v = g[f"{cname}_v"].to(dt).requires_grad_(True)
    xo, vo = conv(x, v, grad, div, nbr)
    from tests.golden.probes import probe_vec
    loss = (xo * probe_vec(tuple(xo.shape), 11).to(dt)).sum()
    if cfg["vector"]:
        loss = loss + (vo * probe_vec(tuple(vo.shape), 12).to(dt)).sum()
    loss.backward()
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


def build_oracle_model(kind, kw, k, lam, dt):
    torch.manual_seed(1)
    cls = oracle.models.DeltaNetSegmentation if kind == "seg" else oracle.models.DeltaNetClassification
    m = cls(num_neighbors=k, grad_regularizer=lam, **kw)
    return m


@pytest.mark.parametrize("name", list(MODELS))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_model_step(name, tag):
    """Whole-model train-mode step (dropout off): same seed => same init as the reference
    (checked by checksum), then logits / loss / gradient summaries / BN running stats."""
    from tests.golden.probes import param_summaries, state_checksum
    kind, kw, normals = MODELS[name]
    g = load_golden(name)
    dt = torch.float32 if tag == "f32" else torch.float64
    model = build_oracle_model(kind, kw, g["k"], g["lam"], dt)
    assert np.allclose(state_checksum(model), g["state_checksum"].numpy(), rtol=0, atol=0)
    model = model.to(dt).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    data = NS(pos=g["pos"].to(dt), batch=g["batch"], norm=(g["normal"].to(dt) if normals else None), x=None,
              category=(g["category"].to(dt) if "category" in g else None))
    logits = model(data)
    loss = oracle.loss.calc_loss(logits, g["y"], smoothing=(kind != "seg"))
    loss.backward()
    # the no-normals model goes through LAPACK SVD sign conventions + an ill-conditioned x-axis:
    # features are gauge-dependent only through fp rounding and lambda
    tol = (2e-2 if tag == "f32" else 1e-5) if normals else 5e-2
    assert rel_err(logits, g[f"logits_{tag}"]) < tol
    assert abs(float(loss.detach()) - float(g[f"loss_{tag}"])) < tol * abs(float(g[f"loss_{tag}"]))
    names, norms, dots = param_summaries(model)
    assert names == [str(s) for s in g["gnames"]]
    gn = g[f"gnorm_{tag}"].numpy()
    assert np.max(np.abs(np.array(norms) - gn) / (gn + 1e-12 + 1e-3 * gn.max())) < 5 * tol
    assert rel_err(dict(model.named_buffers())[("lin_global" if kind == "seg" else "lin_embedding")
                                               + ".0.1.bn.running_mean"], g[f"rm_embed_{tag}"]) < tol
