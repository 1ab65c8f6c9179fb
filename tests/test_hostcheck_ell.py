"""CPU check of the per-thread bodies of the operator-apply / max-aggregation kernels
(deltaconv_amd/csrc/ell_math.h, built for the host by tests/hostcheck) against the oracle:
forward applies vs oracle EllOp algebra, transposed applies vs oracle autograd."""
import ctypes
import os
import subprocess

import pytest
import torch

from oracle import geometry as geo
from tests.helpers import rel_err, ROOT
from deltaconv_amd.data import synthetic_batch

HC_DIR = os.path.join(ROOT, "tests", "hostcheck")


@pytest.fixture(scope="module")
def hc():
    subprocess.run(["make", "-s", "-C", HC_DIR], check=True)
    lib = ctypes.CDLL(os.path.join(HC_DIR, "libhostcheck.so"))
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    lib.hc_csc_build.argtypes = [vp, ci, ci, vp, vp]
    lib.hc_ell_fwd.argtypes = [ci, ci, vp, vp, ci, ci, vp, ci, cl, vp, cl]
    lib.hc_ell_T.argtypes = [ci, ci, vp, vp, vp, ci, ci, vp, ci, cl, vp, cl, ci, vp, cl]
    lib.hc_knn_max.argtypes = [ci, vp, ci, ci, vp, ci, cl, vp, cl, vp]
    lib.hc_knn_max_bwd.argtypes = [ci, vp, vp, ci, ci, vp, vp, ci, cl, vp, cl, ci]
    lib.hc_knn_max_affine.argtypes = [ci, vp, ci, ci, vp, ci, cl, vp, vp, ctypes.c_float, vp, cl, vp]
    lib.hc_knn_max_affine_residual.argtypes = [ci, vp, ci, ci, vp, ci, cl, vp, vp, ctypes.c_float, vp, cl, vp, vp, ctypes.c_float, vp, cl,
                                               vp, cl, vp]
    lib.hc_knn_sum.argtypes = [ci, vp, ci, ci, vp, ci, cl, ctypes.c_float, vp, cl]
    lib.hc_knn_sum_bwd.argtypes = [ci, vp, vp, ci, ci, vp, ci, cl, ctypes.c_float, vp, cl, ci]
    lib.hc_grad_T_sum.argtypes = [ci, vp, vp, vp, ci, ci, vp, ci, cl, vp, cl, vp, cl, vp, cl]
    return lib


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def graph():
    b = synthetic_batch(3, 0, seed=21, sizes=[70, 90, 64], dup_frac=0.05)
    ptr = geo.cloud_ptr(b.batch)
    k = 12
    nbr = geo.knn(b.pos, k, ptr)
    xb, yb = geo.build_tangent_basis(b.norm)
    G, D = geo.build_grad_div(b.pos, b.norm, xb, yb, nbr, ptr, 1.0, 1e-3)
    return dict(nbr=nbr, nbr32=nbr.to(torch.int32).contiguous(), k=k, n=b.pos.shape[0], G=G, D=D)


def csc(hc, g):
    n, k = g["n"], g["k"]
    tptr = torch.zeros(n + 1, dtype=torch.int32)
    tedge = torch.zeros(n * k, dtype=torch.int32)
    hc.hc_csc_build(P(g["nbr32"]), n, k, P(tptr), P(tedge))
    return tptr, tedge


def test_csc(hc, graph):
    tptr, tedge = csc(hc, graph)
    n, k = graph["n"], graph["k"]
    flat = graph["nbr"].reshape(-1)
    assert int(tptr[-1]) == n * k and int(tptr[0]) == 0
    assert torch.equal(torch.sort(tedge).values, torch.arange(n * k, dtype=torch.int32))  # a permutation
    for j in (0, 7, n - 1):
        col = tedge[tptr[j]:tptr[j + 1]].long()
        assert bool((flat[col] == j).all()) and bool((col[1:] > col[:-1]).all())
    assert torch.equal(torch.bincount(flat, minlength=n).to(torch.int32), tptr[1:] - tptr[:-1])


@pytest.mark.parametrize("C,pad", [(8, 0), (8, 4), (5, 0), (3, 2), (64, 0)])
def test_forward_and_transposed(hc, graph, C, pad):
    """pad > 0 exercises leading dimensions larger than the row (writes into concat buffers)."""
    torch.manual_seed(C)
    n, k, nbr32 = graph["n"], graph["k"], graph["nbr32"]
    G, D = graph["G"], graph["D"]
    Gc, Dc = G.coef.contiguous(), D.coef.contiguous()
    tptr, tedge = csc(hc, graph)
    V = 4 if (C % 4 == 0 and pad % 4 == 0) else 1
    ld = C + pad

    def buf(rows, cols, ldc):
        t = torch.full((rows, ldc), 7.0)
        return t

    x = torch.randn(n, C, requires_grad=True)
    v = torch.randn(2 * n, C, requires_grad=True)
    xin = torch.zeros(n, ld); xin[:, :C] = x.detach()
    vin = torch.zeros(2 * n, ld); vin[:, :C] = v.detach()

    # grad
    out = buf(2 * n, C, ld)
    hc.hc_ell_fwd(0, V, P(Gc), P(nbr32), n, k, P(xin), C, ld, P(out), ld)
    ref = G @ x
    assert rel_err(out[:, :C], ref) < 1e-5 and bool((out[:, C:] == 7).all())
    dy = torch.randn(2 * n, C)
    (dx_ref,) = torch.autograd.grad(ref, x, dy)
    dyb = torch.zeros(2 * n, ld); dyb[:, :C] = dy
    dx = buf(n, C, ld)
    hc.hc_ell_T(0, V, P(Gc), P(tptr), P(tedge), n, k, P(dyb), C, ld, P(dx), ld, 0, None, 0)
    assert rel_err(dx[:, :C], dx_ref) < 1e-5
    hc.hc_ell_T(0, V, P(Gc), P(tptr), P(tedge), n, k, P(dyb), C, ld, P(dx), ld, 1, None, 0)   # accumulate
    assert rel_err(dx[:, :C], 2 * dx_ref) < 1e-5

    # div
    out = buf(n, C, ld)
    hc.hc_ell_fwd(1, V, P(Dc), P(nbr32), n, k, P(vin), C, ld, P(out), ld)
    ref = D @ v
    assert rel_err(out[:, :C], ref) < 1e-5
    dy = torch.randn(n, C)
    (dv_ref,) = torch.autograd.grad(ref, v, dy)
    dyb = torch.zeros(n, ld); dyb[:, :C] = dy
    dv = buf(2 * n, C, ld)
    hc.hc_ell_T(1, V, P(Dc), P(tptr), P(tedge), n, k, P(dyb), C, ld, P(dv), ld, 0, None, 0)
    assert rel_err(dv[:, :C], dv_ref) < 1e-5

    # fused div | curl | norm  (output row = 3C, own leading dimension)
    ldo = 3 * C + pad
    out = buf(n, 3 * C, ldo)
    hc.hc_ell_fwd(2, V, P(Dc), P(nbr32), n, k, P(vin), C, ld, P(out), ldo)
    ref = torch.cat([D @ v, geo.curl(v, D), geo.norm(v)], 1)
    assert rel_err(out[:, :3 * C], ref) < 1e-5
    dy = torch.randn(n, 3 * C)
    (dv_ref,) = torch.autograd.grad(ref, v, dy)
    dyb = torch.zeros(n, ldo); dyb[:, :3 * C] = dy
    dv = buf(2 * n, C, ld)
    hc.hc_ell_T(2, V, P(Dc), P(tptr), P(tedge), n, k, P(dyb), C, ldo, P(dv), ld, 0, P(vin), ld)
    assert rel_err(dv[:, :C], dv_ref) < 1e-5

    # fused hodge from [div | curl]
    dc = torch.randn(n, 2 * C, requires_grad=True)
    ldd = 2 * C + pad
    dcb = torch.zeros(n, ldd); dcb[:, :2 * C] = dc.detach()
    out = buf(2 * n, C, ld)
    hc.hc_ell_fwd(3, V, P(Gc), P(nbr32), n, k, P(dcb), C, ldd, P(out), ld)
    ref = -(G @ dc[:, :C] + geo.J(G @ dc[:, C:]))
    assert rel_err(out[:, :C], ref) < 1e-5
    vv = torch.randn(2 * n, C)                                           # and the composed identity
    dcv = torch.cat([D @ vv, geo.curl(vv, D)], 1)
    assert rel_err(-(G @ dcv[:, :C] + geo.J(G @ dcv[:, C:])), geo.hodge_laplacian(vv, G, D)) < 1e-5
    dy = torch.randn(2 * n, C)
    (ddc_ref,) = torch.autograd.grad(ref, dc, dy)
    dyb = torch.zeros(2 * n, ld); dyb[:, :C] = dy
    ddc = buf(n, 2 * C, ldd)
    hc.hc_ell_T(3, V, P(Gc), P(tptr), P(tedge), n, k, P(dyb), C, ld, P(ddc), ldd, 0, None, 0)
    assert rel_err(ddc[:, :2 * C], ddc_ref) < 1e-5

    # max aggregation (+ ties through the duplicate points of the fixture)
    h = torch.randn(n, C, requires_grad=True)
    hb = torch.zeros(n, ld); hb[:, :C] = h.detach()
    out = buf(n, C, ld)
    arg = torch.zeros(n, C, dtype=torch.uint8)
    hc.hc_knn_max(V, P(nbr32), n, k, P(hb), C, ld, P(out), ld, P(arg))
    ref, ref_arg = h[graph["nbr"]].max(dim=1)
    assert torch.equal(out[:, :C], ref.detach()) and torch.equal(arg.long(), ref_arg)
    dy = torch.randn(n, C)
    (dh_ref,) = torch.autograd.grad(ref, h, dy)
    dyb = torch.zeros(n, ld); dyb[:, :C] = dy
    dh = buf(n, C, ld)
    hc.hc_knn_max_bwd(V, P(tptr), P(tedge), n, k, P(arg), P(dyb), C, ld, P(dh), ld, 0)
    assert rel_err(dh[:, :C], dh_ref) < 1e-6

    # the same with BatchNorm scale/shift + LeakyReLU folded into the gather (dc_knn_max_affine): dyadic values
    # make every fmaf exact, so values AND first-maximal slots must equal "transform, then max"
    gq = torch.Generator().manual_seed(11)
    hq = torch.randint(-16, 17, (n, C), generator=gq) / 8.0
    sc = torch.randint(-8, 9, (C,), generator=gq) / 4.0
    sc[0] = 0.0                                                            # constant column: slot 0 wins
    sh = torch.randint(-8, 9, (C,), generator=gq) / 8.0
    z = sc * hq + sh
    yq = torch.where(z > 0, z, 0.25 * z)
    yg = yq[graph["nbr"]]                                                  # [n,k,C]
    mx = yg.max(dim=1).values
    eq = yg == mx[:, None, :]
    first = (eq.float().cumsum(1) == 1) & eq
    slot = (first.float() * torch.arange(k).view(1, k, 1)).sum(1).long()
    hb2 = torch.zeros(n, ld); hb2[:, :C] = hq
    out2 = buf(n, C, ld)
    arg2 = torch.zeros(n, C, dtype=torch.uint8)
    hc.hc_knn_max_affine(V, P(nbr32), n, k, P(hb2), C, ld, P(sc.contiguous()), P(sh.contiguous()), 0.25, P(out2), ld, P(arg2))
    assert torch.equal(out2[:, :C], mx) and torch.equal(arg2.long(), slot)

    # round 6: the layer's last s_mlp block in the epilogue (dc_knn_max_affine_residual): act2(scale2 h2 + shift2) + max, the same
    # two addends as dc_bn_act2 with the maximum as residual -> same bits as "affine max, then block + residual" on ARBITRARY values
    hr, h2 = torch.randn(n, C), torch.randn(n, C)
    s1, t1, s2, t2 = torch.randn(C), torch.randn(C), torch.randn(C), torch.randn(C)
    hb3 = torch.zeros(n, ld); hb3[:, :C] = hr
    hb4 = torch.zeros(n, ld); hb4[:, :C] = h2
    mx3, arg3 = buf(n, C, ld), torch.zeros(n, C, dtype=torch.uint8)
    hc.hc_knn_max_affine(V, P(nbr32), n, k, P(hb3), C, ld, P(s1), P(t1), 0.2, P(mx3), ld, P(arg3))
    z2 = torch.addcmul(t2, s2, h2)                                       # fmaf(scale2, h2, shift2): one rounding
    want = torch.where(z2 > 0, z2, 0.2 * z2) + mx3[:, :C]
    out3, dup3, arg4 = buf(n, C, ld), buf(n, C, ld), torch.zeros(n, C, dtype=torch.uint8)
    hc.hc_knn_max_affine_residual(V, P(nbr32), n, k, P(hb3), C, ld, P(s1), P(t1), 0.2, P(hb4), ld, P(s2), P(t2),
                                  0.2, P(out3), ld, P(dup3), ld, P(arg4))
    assert torch.equal(arg4, arg3) and torch.equal(out3[:, :C], dup3[:, :C])
    assert rel_err(out3[:, :C], want) < 2e-7                              # (torch's addcmul may or may not fuse: bits on the GPU test)


@pytest.mark.parametrize("C,pad", [(8, 0), (8, 4), (5, 0)])
def test_sum_aggregation_and_folded_accumulation(hc, graph, C, pad):
    """dc_knn_sum / dc_knn_sum_backward (DeltaConv(aggr='sum' | 'mean')) and dc_apply_grad_T_sum (transposed gradient
    apply + the other gradients of x') on the CPU build of their per-thread bodies."""
    torch.manual_seed(C + pad)
    n, k, nbr32, G = graph["n"], graph["k"], graph["nbr32"], graph["G"]
    Gc = G.coef.contiguous()
    tptr, tedge = csc(hc, graph)
    V = 4 if (C % 4 == 0 and pad % 4 == 0) else 1
    ld = C + pad
    h = torch.randn(n, C, requires_grad=True)
    hb = torch.zeros(n, ld); hb[:, :C] = h.detach()
    for scale in (1.0, 1.0 / k):
        out = torch.full((n, ld), 7.0)
        hc.hc_knn_sum(V, P(nbr32), n, k, P(hb), C, ld, scale, P(out), ld)
        ref = h[graph["nbr"]].sum(1) * scale
        assert rel_err(out[:, :C], ref) < 1e-6 and bool((out[:, C:] == 7).all())
        dy = torch.randn(n, C)
        (dh_ref,) = torch.autograd.grad(ref, h, dy)
        dyb = torch.zeros(n, ld); dyb[:, :C] = dy
        dh = torch.full((n, ld), 7.0)
        hc.hc_knn_sum_bwd(V, P(tptr), P(tedge), n, k, P(dyb), C, ld, scale, P(dh), ld, 0)
        assert rel_err(dh[:, :C], dh_ref) < 1e-6
    # out = (a + b) + grad^T dy, and with b absent
    x = torch.randn(n, C, requires_grad=True)
    dy = torch.randn(2 * n, C)
    (gt,) = torch.autograd.grad(G @ x, x, dy)
    dyb = torch.zeros(2 * n, ld); dyb[:, :C] = dy
    a, b = torch.randn(n, ld), torch.randn(n, ld)
    out = torch.full((n, ld), 7.0)
    hc.hc_grad_T_sum(V, P(Gc), P(tptr), P(tedge), n, k, P(dyb), C, ld, P(a), ld, P(b), ld, P(out), ld)
    assert rel_err(out[:, :C], a[:, :C] + b[:, :C] + gt) < 1e-5 and bool((out[:, C:] == 7).all())
    hc.hc_grad_T_sum(V, P(Gc), P(tptr), P(tedge), n, k, P(dyb), C, ld, P(a), ld, None, 0, P(out), ld)
    assert rel_err(out[:, :C], a[:, :C] + gt) < 1e-5
