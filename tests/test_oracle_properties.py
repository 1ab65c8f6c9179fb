"""The reference's analytic property tests of the moving-least-squares stages, put to the ORACLE
(/root/reference/test/geometry/test_grad_div_mls.py:58-275 -> scenes in tests/mls_scenes.py).  Together with
tests/test_oracle_golden.py (values of the imported reference) this pins oracle/geometry.py; the same scenes are
put to the kernels' device functions in tests/test_hostcheck.py (CPU) and tests/test_gpu_mls_stages.py (HIP)."""
import torch

from oracle import geometry as geo
from tests import mls_scenes as scenes


def test_coords_projected_recovers_planar_coordinates():
    s = scenes.rotated_paraboloid()
    nbr = geo.knn(s["pos"], 20, [0, 100])
    coords = geo.coords_projected(s["pos"], s["normal"], s["x_basis"], s["y_basis"], nbr)
    assert nbr[0, 0] == 0                                   # centre-major, self first (test_grad_div_mls.py:83)
    assert torch.allclose(coords[0], s["xy"][nbr[0]], atol=1e-6)


def test_gaussian_weights_normalised_and_monotone():
    g = torch.Generator().manual_seed(0)
    w = geo.gaussian_weights(torch.rand(50, 20, generator=g), [0, 50])
    assert not torch.isnan(w).any() and torch.allclose(w.sum(1), torch.ones(50))
    w = geo.gaussian_weights(torch.tensor([[0.1, 0.5, 1.0, 1.5, 2.0]]), [0, 1])[0]
    assert bool((w[:-1] > w[1:]).all())


def test_weighted_least_squares_recovers_quadratic():
    s = scenes.quadratic_patches()
    n, k = s["n"], s["k"]
    w = geo.gaussian_weights(s["dist"].view(n, k), [0, n])
    wls0 = geo.weighted_least_squares(s["coords"].view(n, k, 2), w, 0.0)
    assert torch.allclose(scenes.recovered_coefficients(wls0, s["f"], n, k).float(), s["coefficients"], atol=1e-3)
    wls = geo.weighted_least_squares(s["coords"].view(n, k, 2), w, 1e-5)
    assert torch.allclose(scenes.recovered_coefficients(wls, s["f"], n, k).float(), s["coefficients"], atol=5e-2)
    for key, bound in (("f_noise", 1e-1), ("f_outliers", 5e-1)):
        c = scenes.recovered_coefficients(wls, s[key], n, k).float()
        assert torch.allclose(c, s["coefficients"], atol=bound)
        assert (c - s["coefficients"]).abs().mean() < 5e-2


def test_fit_vector_mapping_expresses_neighbour_frames():
    s = scenes.height_field_patches()
    n, k = s["n"], s["k"]
    w = geo.gaussian_weights(s["dist"].view(n, k), [0, n])
    wls = geo.weighted_least_squares(s["coords"].view(n, k, 2), w, 0.0)
    # the oracle's ELL form has centre i = row i of nbr: the scene's centres are its points i*k, every other point
    # gets k self loops and a zero fit (its rows of the result are not looked at)
    total = n * k
    centre = torch.arange(n) * k
    nbr = torch.arange(total)[:, None].repeat(1, k)
    nbr[centre] = s["edge_index"][1].view(n, k)
    wls_all, coords_all = torch.zeros(total, k, 6), torch.zeros(total, k, 2)
    wls_all[centre], coords_all[centre] = wls, s["coords"].view(n, k, 2)
    vmap = geo.fit_vector_mapping(s["pos"], s["normal"], s["x_basis"], s["y_basis"], nbr, wls_all, coords_all)
    scenes.check_vector_mapping(s, vmap[centre], atol=5e-6)      # fp32 oracle = the reference's numerics
