"""GPU parity of the four public stages of the moving-least-squares assembly through the C ABI:
deltaconv_amd.geometry.{coords_projected, gaussian_weights, weighted_least_squares, fit_vector_mapping}
(dc_mls_coords / dc_mls_gaussian_weights / dc_mls_wls / dc_mls_vector_mapping, csrc/mls.hip) against

  * the reference's per-stage golden values `coords_* / weights_* / wls_* / vmap_*` of the three geometry fixtures
    (fp64 run 2e-5 or tighter, native fp32 run 2e-3 = its own LU error), every wls column on its own;
  * the reference's analytic property tests (test/geometry/test_grad_div_mls.py:58-275) as scenes;
  * the fused product path: the chained stages reproduce dc_mls_assemble's G and D.

The suite itself is tests/mls_stage_suite.py (also run on the g++ build of the same device functions)."""
import types

import pytest
import torch

from tests import mls_stage_suite as suite
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def impl():
    from deltaconv_amd import geometry as G

    def fused(pos, normal, xb, yb, edge_index, batch, k, h, lam, normalized):
        grad, div = G.build_grad_div(pos, normal, xb, yb, edge_index, batch, kernel_width=h, regularizer=lam,
                                     normalized=normalized)
        return grad.coef, div.coef

    def knn(pos, k):
        return G.knn_graph(pos, k, loop=True, flow='target_to_source')

    return types.SimpleNamespace(dev=DEV, coords_projected=G.coords_projected, gaussian_weights=G.gaussian_weights,
                                 weighted_least_squares=G.weighted_least_squares,
                                 fit_vector_mapping=G.fit_vector_mapping, fused=fused, knn=knn)


@pytest.mark.parametrize("name", suite.GEOM)
def test_stages_vs_reference_golden(impl, name):
    suite.golden_stages(impl, name)


@pytest.mark.parametrize("name", suite.GEOM)
@pytest.mark.parametrize("normalized", [True, False])
def test_staged_chain_equals_fused_path(impl, name, normalized):
    suite.staged_equals_fused(impl, name, normalized)


def test_coords_projected_scene(impl):
    suite.coords_scene(impl)


def test_gaussian_weights_scene(impl):
    suite.weights_scene(impl)


def test_weighted_least_squares_scene(impl):
    suite.wls_scene(impl)


def test_fit_vector_mapping_scene(impl):
    suite.vmap_scene(impl)


def test_stages_at_full_size_vs_oracle():
    """BASELINE config 2 geometry (32 x 1024 points, k = 20): every stage against the oracle in fp64."""
    from oracle import geometry as geo
    from deltaconv_amd import geometry as G
    from deltaconv_amd.data import synthetic_batch
    b = synthetic_batch(32, 1024, seed=41)
    k, ptr = 20, geo.cloud_ptr(b.batch)
    nbr = geo.knn(b.pos, k, ptr)
    xb, yb = geo.build_tangent_basis(b.norm)
    pos64, n64, x64, y64 = b.pos.double(), b.norm.double(), xb.double(), yb.double()
    _, _, parts = geo.build_grad_div(pos64, n64, x64, y64, nbr, ptr, 1.0, 1e-3, return_parts=True)
    ei = geo.edge_index_from_nbr(nbr).to(DEV)
    pos, normal, xbd, ybd = b.pos.to(DEV), b.norm.to(DEV), xb.to(DEV), yb.to(DEV)
    coords = G.coords_projected(pos, normal, xbd, ybd, ei, k)
    assert rel_err(coords, parts["coords"].reshape(-1, 2)) < 1e-6
    w = G.gaussian_weights(parts["dist"].reshape(-1).float().to(DEV), k, b.batch.to(DEV), 1.0)
    assert rel_err(w, parts["weights"].reshape(-1)) < 1e-6
    wls = G.weighted_least_squares(coords, w, k, 1e-3)
    assert rel_err(wls, parts["wls"].reshape(-1, 6)) < 1e-5
    vmap = G.fit_vector_mapping(pos, normal, xbd, ybd, ei, wls, coords)
    assert rel_err(vmap, parts["vmap"].reshape(-1, 2, 2)) < 1e-5


def test_stage_argument_errors():
    """Error behaviour: Python exceptions, like the rest of the operator API."""
    from deltaconv_amd import geometry as G
    pos = torch.rand(10, 3, device=DEV)
    ei = torch.stack([torch.arange(10).repeat_interleave(3), torch.randint(0, 10, (30,))]).to(DEV)
    with pytest.raises(ValueError):
        G.coords_projected(pos, pos[:5], pos[:5], pos[:5], ei, 3)                      # 5 frames for 10 groups
    with pytest.raises(ValueError):
        G.gaussian_weights(torch.rand(31, device=DEV), 3)
    with pytest.raises(ValueError):
        G.gaussian_weights(torch.rand(30, device=DEV), 3, torch.tensor([1] * 5 + [0] * 5, device=DEV))
    with pytest.raises(ValueError):
        G.weighted_least_squares(torch.rand(30, 2, device=DEV), torch.rand(29, device=DEV), 3, 0.1)
    bad = ei.clone()
    bad[0, 1] = 7                                                                      # a group with two centres
    with pytest.raises(ValueError):
        G.fit_vector_mapping(pos, pos, pos, pos, bad, torch.rand(30, 6, device=DEV), torch.rand(30, 2, device=DEV))
    with pytest.raises(RuntimeError):
        G.coords_projected(pos.cpu(), pos.cpu(), pos.cpu(), pos.cpu(), ei.cpu(), 3)    # no CPU path
    assert G.coords_projected(pos, pos[:0], pos[:0], pos[:0], ei[:, :0]).shape == (0, 2)
