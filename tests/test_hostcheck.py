"""CPU check of the arithmetic the HIP geometry kernels run (deltaconv_amd/csrc/point_math.h):
tests/hostcheck builds the same header with g++ and loops it over points; here it is compared
with the oracle and with the reference's golden vectors.  Validates the math before GPU time is
spent; the GPU parity tests (tests/test_gpu_*.py) repeat the comparison through the C-ABI."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import geometry as geo
from tests.helpers import load_golden, rel_err, ROOT

HC_DIR = os.path.join(ROOT, "tests", "hostcheck")


@pytest.fixture(scope="module")
def hc():
    subprocess.run(["make", "-s", "-C", HC_DIR], check=True)
    return ctypes.CDLL(os.path.join(HC_DIR, "libhostcheck.so"))


def fptr(t):
    return ctypes.c_void_p(t.data_ptr())


def run_mls(hc, pos, normal, xb, yb, nbr, ptr, k, h, lam, normalized=True):
    nt = pos.shape[0]
    G = torch.zeros(nt, k, 2)
    D = torch.zeros(nt, k, 2)
    nbr32 = nbr.to(torch.int32).contiguous()
    ptr32 = torch.tensor(ptr, dtype=torch.int32)
    hc.hc_mls_assemble(fptr(pos), fptr(normal), fptr(xb), fptr(yb), fptr(nbr32), fptr(ptr32), len(ptr) - 1, k,
                       ctypes.c_float(h), ctypes.c_float(lam), int(normalized), fptr(G), fptr(D))
    return G, D


GEOM = ["geom_normals_B2_N128_k20", "geom_ragged_dups_k30", "geom_nonormals_N200_k10"]


@pytest.mark.parametrize("name", GEOM)
def test_mls_vs_reference_and_oracle(hc, name):
    g = load_golden(name)
    pos, k = g["pos"].contiguous(), g["k"]
    nt = pos.shape[0]
    ptr = geo.cloud_ptr(g["batch"], nt)
    nbr = geo.nbr_from_edge_index(g["edge_index"], k)
    normal, xb, yb = (g[f"{n}_f32"].contiguous() for n in ("normal", "x_basis", "y_basis"))
    G, D = run_mls(hc, pos, normal, xb, yb, nbr, ptr, k, g["h"], g["lam"])
    # (1) against fp64 "truth": the reference's own code run in double, fed the (fp32-rounded)
    #     frames of that double run -- the SVD sign may differ between its fp32 and fp64 runs
    n64, x64, y64 = (g[f"{n}_f64"].float().contiguous() for n in ("normal", "x_basis", "y_basis"))
    G64, D64 = run_mls(hc, pos, n64, x64, y64, nbr, ptr, k, g["h"], g["lam"])
    assert rel_err(G64.reshape(-1), g["grad_val_f64"]) < 2e-5
    assert rel_err(D64.reshape(-1), g["div_val_f64"]) < 2e-5
    # (2) against the reference's native fp32 numerics: limited by ITS fp32 LU inverse
    e32 = max(rel_err(G.reshape(-1), g["grad_val_f32"]), rel_err(D.reshape(-1), g["div_val_f32"]))
    assert e32 < 2e-3
    # (3) against the oracle in double, fed identical fp32 inputs: tight
    Go, Do = geo.build_grad_div(pos.double(), normal.double(), xb.double(), yb.double(), nbr, ptr, g["h"], g["lam"])
    assert rel_err(G, Go.coef) < 1e-6 and rel_err(D, Do.coef) < 1e-6


def test_mls_shape_regularizer_vs_reference(hc):
    """mls_fit_point<true> (dc_mls_assemble_shape): build_grad_div(shape_regularizer=...) of the reference."""
    g = load_golden("geom_shape_regularizer")
    pos, k = g["pos"].contiguous(), int(g["k"])
    nt = pos.shape[0]
    ptr = geo.cloud_ptr(g["batch"], nt)
    nbr = geo.nbr_from_edge_index(g["edge_index"], k)
    normal = g["normal"].contiguous()
    xb, yb = geo.build_tangent_basis(normal)
    G, D = torch.zeros(nt, k, 2), torch.zeros(nt, k, 2)
    nbr32 = nbr.to(torch.int32).contiguous()
    ptr32 = torch.tensor(ptr, dtype=torch.int32)
    hc.hc_mls_assemble_shape(fptr(pos), fptr(normal), fptr(xb.contiguous()), fptr(yb.contiguous()), fptr(nbr32),
                             fptr(ptr32), len(ptr) - 1, k, ctypes.c_float(1.0), ctypes.c_float(float(g["lam"])),
                             ctypes.c_float(float(g["lam_shape"])), 1, fptr(G), fptr(D))
    assert rel_err(G.reshape(-1), g["grad_val_f64"]) < 2e-5 and rel_err(D.reshape(-1), g["div_val_f64"]) < 2e-5
    assert rel_err(G.reshape(-1), g["grad_val_f32"]) < 2e-3 and rel_err(D.reshape(-1), g["div_val_f32"]) < 2e-3
    G0, D0 = run_mls(hc, pos, normal, xb.contiguous(), yb.contiguous(), nbr, ptr, k, 1.0, float(g["lam"]))
    assert torch.equal(G0, G) and rel_err(D0.reshape(-1), g["div_val_f64"]) > 1e-3


@pytest.mark.parametrize("lam,normalized", [(1e-8, False), (1e-8, True), (0.0, False)])
def test_mls_small_lambda(hc, lam, normalized):
    """The reference's tests use regularizer=1e-8 (test_grad_div_mls.py:332,390): ill-conditioned in
    fp32; the fp64 interior must still track an fp64 oracle."""
    from deltaconv_amd.data import synthetic_batch
    b = synthetic_batch(2, 300, seed=11)
    ptr = geo.cloud_ptr(b.batch)
    nbr = geo.knn(b.pos, 20, ptr)
    xb, yb = geo.build_tangent_basis(b.norm)
    G, D = run_mls(hc, b.pos, b.norm, xb, yb, nbr, ptr, 20, 1.0, lam, normalized)
    Go, Do = geo.build_grad_div(b.pos.double(), b.norm.double(), xb.double(), yb.double(), nbr, ptr, 1.0, lam,
                                normalized=normalized)
    assert rel_err(G, Go.coef) < 1e-4 and rel_err(D, Do.coef) < 1e-4


def test_tangent_basis(hc):
    torch.manual_seed(0)
    n = torch.randn(1000, 3)
    n = n / n.norm(dim=1, keepdim=True)
    n[:5] = torch.tensor([[1., 0, 0], [-1., 0, 0], [0.9, 0.43589, 0], [0.90001, 0.4358, 0], [0, 0, 1.]])
    n = (n / n.norm(dim=1, keepdim=True)).contiguous()
    xb, yb = torch.zeros_like(n), torch.zeros_like(n)
    hc.hc_tangent_basis(fptr(n), n.shape[0], fptr(xb), fptr(yb))
    xo, yo = geo.build_tangent_basis(n)
    assert rel_err(xb, xo) < 1e-6 and rel_err(yb, yo) < 1e-6
    # reference test_build_tangent_basis (test_grad_div_mls.py:12-24): orthonormal, right-handed
    basis = torch.stack([n, xb, yb], -1)
    assert torch.allclose(basis.transpose(1, 2) @ basis, torch.eye(3).expand(1000, 3, 3), atol=1e-6)
    assert (torch.linalg.cross(xb, yb) * n).sum(1).min() > 0


def test_estimate_basis(hc):
    g = load_golden("geom_nonormals_N200_k10")
    pos = g["pos"].contiguous()
    nbr10 = geo.nbr_from_edge_index(g["edge_index10"], 10).to(torch.int32).contiguous()
    n = pos.shape[0]
    normal, xb, yb = torch.zeros(n, 3), torch.zeros(n, 3), torch.zeros(n, 3)
    hc.hc_estimate_basis(fptr(pos), fptr(nbr10), n, 10, fptr(pos), fptr(normal), fptr(xb), fptr(yb))
    assert rel_err(normal, g["normal_f64"]) < 1e-5                     # oriented normal: unique
    sgn = torch.sign((xb * g["x_basis_f64"].float()).sum(1, keepdim=True))
    assert rel_err(xb * sgn, g["x_basis_f64"]) < 1e-4                  # x: unique up to sign
    assert rel_err(yb * sgn, g["y_basis_f64"]) < 1e-4
    # reference test_estimate_basis (test_grad_div_mls.py:27-55): plane -> normal = plane normal
    torch.manual_seed(1)
    p = torch.cat([torch.rand(100, 2), torch.zeros(100, 1)], 1)
    nrm = torch.rand(1, 3); nrm = nrm / nrm.norm()
    xbt, ybt = geo.build_tangent_basis(nrm)
    T = torch.stack([xbt[0], ybt[0], nrm[0]], -1)
    p = (p @ T.T).contiguous()
    nb = geo.knn(p, 20, [0, 100]).to(torch.int32).contiguous()
    normal, xb, yb = torch.zeros(100, 3), torch.zeros(100, 3), torch.zeros(100, 3)
    hc.hc_estimate_basis(fptr(p), fptr(nb), 100, 20, None, fptr(normal), fptr(xb), fptr(yb))
    basis = torch.stack([normal, xb, yb], -1)
    assert torch.allclose(basis.transpose(1, 2) @ basis, torch.eye(3).expand(100, 3, 3), atol=1e-5)
    assert (torch.linalg.cross(xb, yb) * normal).sum(1).min() > 0
    assert torch.allclose((nrm * normal).sum(1).abs(), torch.ones(100), atol=1e-5)
