"""The reference's own checks of the vector-field algebra, in its own form (polar construction of the field, closed-form
answers), run on the oracle's and on the product's `norm`, `J`, `I_J`, `batch_dot` (reference: test/geometry/test_operators.py:6-44,
test/geometry/test_utils.py:6-13).  These four are tensor algebra over torch storage in both packages -- no kernel -- so they run
wherever the tensors live; the GPU suite checks the same functions against the oracle on device tensors (test_gpu_geometry.py)."""
import pytest
import torch


def packages():
    from oracle import geometry as oracle_geo
    from deltaconv_amd import geometry as product_geo
    return {"oracle": oracle_geo, "product": product_geo}


def polar_field(N=1024, C=16, seed=0):
    """Interleaved field [2N, C] from random lengths and angles (test_operators.py:6-15)."""
    g = torch.Generator().manual_seed(seed)
    length = torch.rand(N, C, generator=g) * 5
    angle = torch.rand(N, C, generator=g) * 2 * torch.pi
    vx, vy = length * torch.cos(angle), length * torch.sin(angle)
    return torch.stack([vx, vy], dim=1).view(-1, C), length, vx, vy


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_norm_recovers_the_lengths(which):
    geo = packages()[which]
    v, length, _, _ = polar_field()
    assert torch.allclose(geo.norm(v), length)


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_J_is_a_quarter_turn(which):
    geo = packages()[which]
    v, _, vx, vy = polar_field(seed=1)
    C = v.shape[1]
    out = geo.J(v)
    assert torch.equal(out, torch.stack([-vy, vx], dim=1).view(-1, C))
    assert torch.allclose((v.view(-1, 2, C) * out.view(-1, 2, C)).sum(dim=1), torch.zeros_like(vx))
    assert torch.equal(geo.J(out), -v)                         # two quarter turns: the half turn, exactly


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_I_J_concatenates_field_and_turn(which):
    geo = packages()[which]
    v, _, _, _ = polar_field(seed=2)
    C = v.shape[1]
    out = geo.I_J(v)
    assert out.shape == (v.shape[0], 2 * C)
    assert torch.equal(out[:, :C], v) and torch.equal(out[:, C:], geo.J(v))


def test_batch_dot_is_the_row_dot_product():
    geo = packages()["product"]                                # the oracle writes the sum inline and has no such helper
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(1024, 10, generator=g), torch.rand(1024, 10, generator=g)
    out = geo.batch_dot(a, b)
    assert out.shape == (1024, 1)
    assert torch.allclose(out, (a * b).sum(dim=1, keepdim=True))


def test_oracle_and_product_agree_bit_for_bit():
    pk = packages()
    v, _, _, _ = polar_field(N=257, C=7, seed=4)
    for fn in ("norm", "J", "I_J"):
        assert torch.equal(getattr(pk["oracle"], fn)(v), getattr(pk["product"], fn)(v)), fn
