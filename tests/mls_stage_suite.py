"""One suite for the four public stages of the moving-least-squares assembly, run against two builds of the SAME
device functions (deltaconv_amd/csrc/point_math.h):

  * tests/test_hostcheck_stages.py  -- the g++ build, in the GPU-less container;
  * tests/test_gpu_mls_stages.py    -- the HIP entry points dc_mls_{coords,gaussian_weights,wls,vector_mapping} behind
                                       deltaconv_amd.geometry.{coords_projected, gaussian_weights,
                                       weighted_least_squares, fit_vector_mapping}.

`impl` offers those four functions with the reference's signatures (grad_div_mls.py:72,100,119,155) plus
`fused(pos, normal, xb, yb, edge_index, batch, k, h, lam, normalized)` -> (G, D) [Nt, k, 2] of the fused product
path (dc_mls_assemble), `dev` (where the tensors go) and `knn(pos, k)` -> edge_index.

Tolerances (rel_err = max|a-b| / max|b|): against the reference's fp64 run 2e-5 or tighter (fp32 in / out, fp64
interior), against its native fp32 run what ITS fp32 arithmetic allows (LU inverse: 2e-3)."""
import torch

from tests import mls_scenes as scenes
from tests.helpers import load_golden, rel_err

GEOM = ["geom_normals_B2_N128_k20", "geom_ragged_dups_k30", "geom_nonormals_N200_k10"]


def golden_stages(impl, name):
    """Each stage fed the reference's own upstream values, compared with the reference's output of that stage
    (tests/golden/make_golden.py::geometry_case: coords_*, weights_*, wls_*, vmap_*)."""
    g = load_golden(name)
    d = impl.dev
    k = int(g["k"])
    pos, ei, batch = g["pos"].to(d), g["edge_index"].to(d), g["batch"].to(d)
    dist = (g["pos"][g["edge_index"][1]] - g["pos"][g["edge_index"][0]]).norm(dim=1).to(d)
    out = {}
    for tag, tol_c, tol_w, tol_wls, tol_v in (("f64", 1e-6, 1e-6, 2e-5, 2e-5), ("f32", 1e-5, 1e-5, 2e-3, 1e-4)):
        normal, xb, yb = (g[f"{n}_{tag}"].float().to(d) for n in ("normal", "x_basis", "y_basis"))
        coords = impl.coords_projected(pos, normal, xb, yb, ei, k)
        assert coords.shape == (ei.shape[1], 2)
        out[f"coords_{tag}"] = rel_err(coords, g[f"coords_{tag}"])
        assert out[f"coords_{tag}"] < tol_c, (tag, out)
        w = impl.gaussian_weights(dist, k, batch, g["h"])
        out[f"weights_{tag}"] = rel_err(w, g[f"weights_{tag}"])
        assert out[f"weights_{tag}"] < tol_w, (tag, out)
        wls = impl.weighted_least_squares(g[f"coords_{tag}"].float().to(d), g[f"weights_{tag}"].float().to(d), k,
                                          g["lam"])
        assert wls.shape == (ei.shape[1], 6)
        out[f"wls_{tag}"] = rel_err(wls, g[f"wls_{tag}"])
        assert out[f"wls_{tag}"] < tol_wls, (tag, out)
        # column by column: column 0 (the constant term) reaches no operator value, only this test sees it
        for c in range(6):
            assert rel_err(wls[:, c], g[f"wls_{tag}"][:, c]) < 5 * tol_wls, (tag, c)
        vmap = impl.fit_vector_mapping(pos, normal, xb, yb, (ei[0], ei[1]), g[f"wls_{tag}"].float().to(d),
                                       g[f"coords_{tag}"].float().to(d))
        assert vmap.shape == (ei.shape[1], 2, 2)
        out[f"vmap_{tag}"] = rel_err(vmap, g[f"vmap_{tag}"])
        assert out[f"vmap_{tag}"] < tol_v, (tag, out)
    return out


def staged_equals_fused(impl, name, normalized=True):
    """coords -> weights -> wls -> mapping chained on the stage entry points, composed as build_grad_div composes them
    (grad_div_mls.py:253-260,271-272), against the fused product path on the same inputs."""
    g = load_golden(name)
    d = impl.dev
    k = int(g["k"])
    nt = g["pos"].shape[0]
    pos, ei, batch = g["pos"].to(d), g["edge_index"].to(d), g["batch"].to(d)
    normal, xb, yb = (g[f"{n}_f32"].to(d) for n in ("normal", "x_basis", "y_basis"))
    coords = impl.coords_projected(pos, normal, xb, yb, ei, k)
    dist = (pos[ei[1]] - pos[ei[0]]).norm(dim=1)
    w = impl.gaussian_weights(dist, k, batch, g["h"])
    wls = impl.weighted_least_squares(coords, w, k, g["lam"])
    vmap = impl.fit_vector_mapping(pos, normal, xb, yb, ei, wls, coords).cpu().double()
    G = wls[:, 1:3].cpu().double().view(nt, k, 2).clone()
    if normalized:
        rows = G.abs().sum(1).norm(dim=1)
        for b in range(int(g["batch"].max()) + 1):
            m = g["batch"] == b
            if rows[m].max() > 1e-5:
                G[m] = G[m] / rows[m].max()
    D = (G.view(-1, 1, 2) @ vmap).view(nt, k, 2)
    Gf, Df = impl.fused(pos, normal, xb, yb, ei, batch, k, g["h"], g["lam"], normalized)
    eg, ed = rel_err(Gf, G), rel_err(Df, D)
    # the staged chain rounds coords / weights / wls to fp32 between the stages, the fused kernels do not
    assert eg < 2e-5 and ed < 2e-5, (eg, ed)
    return eg, ed


def coords_scene(impl):
    """test_grad_div_mls.py:58-84."""
    s = scenes.rotated_paraboloid()
    d = impl.dev
    ei = impl.knn(s["pos"].to(d), 20)
    coords = impl.coords_projected(s["pos"].to(d), s["normal"].to(d), s["x_basis"].to(d), s["y_basis"].to(d), ei)  # k=None
    assert int(ei[0, 0]) == 0 and int(ei[1, 0]) == 0
    assert torch.allclose(coords[:20].cpu(), s["xy"][ei[1, :20].cpu()], atol=1e-6)


def weights_scene(impl):
    """test_grad_div_mls.py:87-104."""
    d = impl.dev
    g = torch.Generator().manual_seed(0)
    w = impl.gaussian_weights(torch.rand(1000, generator=g).to(d), 20)                 # batch=None, kernel_width=1
    assert not torch.isnan(w).any() and w.shape == (1000,)
    assert torch.allclose(w.view(-1, 20).sum(1).cpu(), torch.ones(50))
    w = impl.gaussian_weights(torch.tensor([0.1, 0.5, 1.0, 1.5, 2.0]).to(d), 5).cpu()
    assert bool((w[:-1] > w[1:]).all())
    # two clouds: each scaled by ITS mean edge length (scatter_mean over batch, grad_div_mls.py:112)
    dist = torch.rand(40 * 20, generator=g)
    dist[20 * 20:] *= 7.0
    batch = torch.arange(2).repeat_interleave(20)
    w2 = impl.gaussian_weights(dist.to(d), 20, batch.to(d), 1.5).cpu()
    for b in range(2):
        part = dist.view(40, 20)[b * 20:(b + 1) * 20].double()
        ref = torch.exp(-part ** 2 / (1.5 * part.mean()) ** 2)
        ref = ref / ref.sum(1, keepdim=True)
        assert rel_err(w2.view(40, 20)[b * 20:(b + 1) * 20], ref) < 1e-6


def wls_scene(impl):
    """test_grad_div_mls.py:107-145 (bounds of the reference; measured values are far inside)."""
    s = scenes.quadratic_patches()
    d = impl.dev
    n, k = s["n"], s["k"]
    w = impl.gaussian_weights(s["dist"].to(d), k)
    wls0 = impl.weighted_least_squares(s["coords"].to(d), w, k, 0)
    e0 = float((scenes.recovered_coefficients(wls0.cpu(), s["f"], n, k) - s["coefficients"].double()).abs().max())
    assert e0 < 1e-3, e0
    assert e0 < 2e-5, e0         # the fp64 interior: the reference's own fp32 inverse measures 3.5e-6 .. 5.9e-6
    wls = impl.weighted_least_squares(s["coords"].to(d), w, k, 1e-5)
    assert torch.allclose(scenes.recovered_coefficients(wls.cpu(), s["f"], n, k).float(), s["coefficients"], atol=5e-2)
    for key, bound in (("f_noise", 1e-1), ("f_outliers", 5e-1)):
        c = scenes.recovered_coefficients(wls.cpu(), s[key], n, k).float()
        assert torch.allclose(c, s["coefficients"], atol=bound)
        assert (c - s["coefficients"]).abs().mean() < 5e-2
    # shape_regularizer: the pair (wls, wls_shape), each the single-regulariser result (grad_div_mls.py:146-151)
    a, b = impl.weighted_least_squares(s["coords"].to(d), w, k, 1e-5, shape_regularizer=1e-2)
    assert torch.equal(a, wls) and torch.equal(b, impl.weighted_least_squares(s["coords"].to(d), w, k, 1e-2))
    assert rel_err(a, b) > 1e-3


def vmap_scene(impl):
    """test_grad_div_mls.py:148-275: centres are the points i*k of the scene, not 0 .. n-1."""
    s = scenes.height_field_patches()
    d = impl.dev
    k = s["k"]
    w = impl.gaussian_weights(s["dist"].to(d), k)
    wls = impl.weighted_least_squares(s["coords"].to(d), w, k, regularizer=0)
    vmap = impl.fit_vector_mapping(s["pos"].to(d), s["normal"].to(d), s["x_basis"].to(d), s["y_basis"].to(d),
                                   s["edge_index"].to(d), wls, s["coords"].to(d))
    assert vmap.size() == (s["n"] * k, 2, 2)
    return scenes.check_vector_mapping(s, vmap, atol=1e-6)
