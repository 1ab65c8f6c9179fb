"""Analytic scenes with known answers for the four stages of the moving-least-squares assembly
(coords_projected, gaussian_weights, weighted_least_squares, fit_vector_mapping).

They restate the testing strategy of the reference's own property tests
(/root/reference/test/geometry/test_grad_div_mls.py:58-275) as data: every scene returns the inputs of a stage and
what the stage must produce, so that the same scene can be put to the oracle (tests/test_oracle_properties.py), to
the g++ build of the kernels' device functions (tests/test_hostcheck.py) and to the HIP entry points
(tests/test_gpu_mls_stages.py).  Test infrastructure; CPU tensors, fp32 unless stated.
"""
import torch


def _unit(a):
    return a / a.norm(dim=-1, keepdim=True).clamp(1e-8)


def _frame_of(normal):
    """x, y with (x, y, normal) right-handed -- build_tangent_basis semantics (grad_div_mls.py:50-69)."""
    t = torch.tensor([1.0, 0.0, 0.0]).expand_as(normal).clone()
    t[(normal[:, 0].abs() > 0.9)] = torch.tensor([0.0, 1.0, 0.0])
    x = _unit(torch.linalg.cross(t, normal))
    y = _unit(torch.linalg.cross(normal, x))
    return x, y


def rotated_paraboloid(n=100, seed=0):
    """test_grad_div_mls.py:58-84: the graph of z = x^2 + y^2 over [-1, 1]^2, point 0 at its apex, moved by a random
    offset and expressed in a random orthonormal frame.  In the tangent plane of the apex (the frame itself) the
    neighbours of point 0 have exactly their (x, y) as coordinates.
    -> pos [n, 3], one frame (normal, x, y) repeated n times, the planar coordinates xy [n, 2]."""
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g) * 2 - 1
    xy[0] = 0
    local = torch.cat([xy, (xy ** 2).sum(1, keepdim=True)], 1)
    normal = _unit(torch.rand(1, 3, generator=g))
    x, y = _frame_of(normal)
    offset = torch.rand(3, generator=g)
    # local coordinates (a, b, c) -> a x + b y + c n
    pos = (local + offset) @ torch.cat([x, y, normal], 0)
    return dict(pos=pos.contiguous(), normal=normal.repeat(n, 1), x_basis=x.repeat(n, 1), y_basis=y.repeat(n, 1), xy=xy)


def quadratic_patches(n=1000, k=20, seed=0):
    """test_grad_div_mls.py:107-128: n neighbourhoods of k planar points in [-1, 1]^2 (slot 0 = the centre at 0) and a
    random quadratic c0 + c1 u + c2 v + c3 u^2 + c4 uv + c5 v^2 sampled on them.
    -> coords [n*k, 2], dist [n*k], coefficients [n, 6], f [n, k] (+ a noisy and an outlier-ridden copy)."""
    g = torch.Generator().manual_seed(seed)
    coords = torch.rand(n, k, 2, generator=g) * 2 - 1
    coords[:, 0] = 0
    u, v = coords[..., 0], coords[..., 1]
    rows = torch.stack([torch.ones_like(u), u, v, u * u, u * v, v * v], -1)
    coefficients = torch.rand(n, 6, generator=g)
    f = (rows * coefficients[:, None, :]).sum(-1)
    noise = torch.rand(n, k, generator=g) * 0.01 - 0.005
    outliers = (torch.rand(n, k, generator=g) > 0.95) * torch.rand(n, k, generator=g) * 0.1
    return dict(coords=coords.reshape(-1, 2), dist=coords.reshape(-1, 2).norm(dim=1), coefficients=coefficients, f=f,
                f_noise=f + noise, f_outliers=f + outliers, k=k, n=n)


def recovered_coefficients(wls, f, n, k):
    """c = sum_s wls[s, :] f_s per neighbourhood (what fit_vector_mapping does with the heights, grad_div_mls.py:165)."""
    return (wls.reshape(n, k, 6).double() * f.reshape(n, k, 1).double()).sum(1)


def height_field_patches(n=1000, k=20, seed=0):
    """test_grad_div_mls.py:148-275: n separate patches z = c0 u^2 + c1 uv + c2 v^2 of k points each (slot 0 = the
    centre), every point carrying its exact surface frame turned in-plane by a random angle (the centre keeps the
    coordinate frame).  The map M that fit_vector_mapping returns for an edge must express the neighbour's frame
    in the centre's coordinate tangents: M[0, a] d_u + M[1, a] d_v = e_a at the neighbour.
    -> pos [n*k, 3], frames [n*k, 3] x 3, edge_index (row = centre of the patch, col = the point), coords [n*k, 2],
       dist [n*k], tangents dfdu / dfdv [n*k, 3]."""
    g = torch.Generator().manual_seed(seed)
    uv = torch.rand(n, k, 2, generator=g) * 2 - 1
    uv[:, 0] = 0
    c = torch.rand(n, 3, generator=g)
    u, v = uv[..., 0], uv[..., 1]
    z = c[:, None, 0] * u * u + c[:, None, 1] * u * v + c[:, None, 2] * v * v
    pos = torch.stack([u, v, z], -1).reshape(-1, 3)
    zu = (2 * c[:, None, 0] * u + c[:, None, 1] * v).reshape(-1)
    zv = (c[:, None, 1] * u + 2 * c[:, None, 2] * v).reshape(-1)
    one, zero = torch.ones_like(zu), torch.zeros_like(zu)
    dfdu = torch.stack([one, zero, zu], 1)
    dfdv = torch.stack([zero, one, zv], 1)
    normal = _unit(torch.linalg.cross(dfdu, dfdv))
    # in-plane turn: x = normalise(a d_u + b d_v) with random signs, |a|, |b| >= 0.01; centres keep (1, 0)
    mix = torch.rand(n * k, 2, generator=g) + 1e-2
    mix = mix * (torch.randint(0, 2, (n * k, 2), generator=g) * 2 - 1)
    mix = _unit(mix).reshape(n, k, 2)
    mix[:, 0] = torch.tensor([1.0, 0.0])
    mix = mix.reshape(-1, 2)
    x_basis = _unit(mix[:, :1] * dfdu + mix[:, 1:] * dfdv)
    y_basis = torch.linalg.cross(normal, x_basis)
    centre = (torch.arange(n) * k).repeat_interleave(k)
    edge_index = torch.stack([centre, torch.arange(n * k)])
    coords = uv.reshape(-1, 2)
    return dict(pos=pos.contiguous(), normal=normal.contiguous(), x_basis=x_basis.contiguous(),
                y_basis=y_basis.contiguous(), edge_index=edge_index, coords=coords.contiguous(),
                dist=coords.norm(dim=1), dfdu=dfdu, dfdv=dfdv, k=k, n=n)


def check_vector_mapping(scene, mapping, atol=1e-6, rtol=1e-5):
    """The two identities of test_grad_div_mls.py:269-270 (torch.allclose, the reference's atol = 1e-6 and the default
    rtol), evaluated in double from the (fp32) mapping.  Returns the largest absolute deviation.  Yardstick: the
    imported reference itself (fp32 LU inverse at lambda = 0) deviates by 1.2e-6 .. 1.7e-6 on seeds 0-2 of the scene,
    so an fp32 restatement is held to atol = 5e-6; implementations with an fp64 interior meet the reference's 1e-6."""
    m = mapping.detach().cpu().double().reshape(-1, 2, 2)
    assert m.shape[0] == scene["n"] * scene["k"] and not torch.isnan(m).any()
    du, dv = scene["dfdu"].double(), scene["dfdv"].double()
    ex = m[:, 0, 0, None] * du + m[:, 1, 0, None] * dv
    ey = m[:, 0, 1, None] * du + m[:, 1, 1, None] * dv
    xb, yb = scene["x_basis"].double(), scene["y_basis"].double()
    assert torch.allclose(ex, xb, atol=atol, rtol=rtol) and torch.allclose(ey, yb, atol=atol, rtol=rtol)
    return max(float((ex - xb).abs().max()), float((ey - yb).abs().max()))
