"""CPU check of the fused BatchNorm/activation and vector non-linearity formulas
(deltaconv_amd/csrc/nn_math.h via tests/hostcheck) against torch autograd on the reference's own
module definitions restated in oracle/nn.py."""
import ctypes
import os
import subprocess

import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import geometry as geo
from tests.helpers import rel_err, ROOT

HC_DIR = os.path.join(ROOT, "tests", "hostcheck")


@pytest.fixture(scope="module")
def hc():
    subprocess.run(["make", "-s", "-C", HC_DIR], check=True)
    lib = ctypes.CDLL(os.path.join(HC_DIR, "libhostcheck.so"))
    vp, ci, cl, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
    lib.hc_bn_act.argtypes = [vp, cl, ci, vp, vp, cf, cf, vp, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.hc_vn.argtypes = [vp, cl, ci, ci, vp, vp, cf, ci, vp, vp, vp, vp, vp]
    lib.hc_ce_loss.argtypes = [vp, vp, cl, ci, cf, vp]
    lib.hc_ce_loss.restype = ctypes.c_double
    return lib


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("slope,training,use_res", [(0.2, 1, False), (0.2, 1, True), (1.0, 1, False), (0.2, 0, False)])
def test_bn_act(hc, slope, training, use_res):
    torch.manual_seed(0)
    R, C = 500, 12
    h = (torch.randn(R, C) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C) + 0.5).requires_grad_(True)
    gamma.data[3] = -0.7
    beta = (torch.randn(C) * 0.3).requires_grad_(True)
    res = torch.randn(R, C) if use_res else None
    rm, rv = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    z = F.batch_norm(h, rm.clone(), rv.clone(), gamma, beta, training=bool(training), momentum=0.1, eps=1e-5)
    y_ref = F.leaky_relu(z, slope) if slope != 1.0 else z
    if use_res:
        y_ref = y_ref + res
    dy = torch.randn(R, C)
    gh, gg, gb = torch.autograd.grad(y_ref, [h, gamma, beta], dy)
    y, dh, dg, db = torch.zeros(R, C), torch.zeros(R, C), torch.zeros(C), torch.zeros(C)
    hc.hc_bn_act(P(h.detach()), R, C, P(gamma.detach()), P(beta.detach()), 1e-5, slope, P(res), training, P(rm), P(rv),
                 P(y), P(dy), P(dh), P(dg), P(db))
    assert rel_err(y, y_ref) < 1e-5 and rel_err(dh, gh) < 1e-4 and rel_err(dg, gg) < 1e-4 and rel_err(db, gb) < 1e-4


@pytest.mark.parametrize("combine,training", [(1, 1), (0, 1), (0, 0), (1, 0)])
def test_vector_nonlin(hc, combine, training):
    torch.manual_seed(1)
    n, co, K = 300, 8, 10
    if combine:
        a = torch.randn(2 * n, K, requires_grad=True)                 # v_cat
        W = torch.randn(co, 2 * K) * 0.3                              # Linear on I_J(v_cat)
        y_ref_in = F.linear(geo.I_J(a), W)
        pq = F.linear(a, torch.cat([W[:, :K], W[:, K:]], 0)).detach() # [2n, 2co] = [P | Q]
        inp = pq.contiguous()
    else:
        a = torch.randn(2 * n, co, requires_grad=True)
        y_ref_in = a
        inp = a.detach().contiguous()
    inp[:2] = 0                                                       # a zero vector: |y| = 0 path
    if combine:
        with torch.no_grad():
            a[:2] = 0
        y_ref_in = F.linear(geo.I_J(a), W)
    else:
        with torch.no_grad():
            a[:2] = 0
        y_ref_in = a
    bn = oracle.nn.BatchNorm1d(co) if training else None
    vn = oracle.nn.VectorNonLin(co, batchnorm=bn)
    if training:
        with torch.no_grad():
            bn.bn.weight.copy_(torch.rand(co) + 0.5)
            bn.bn.bias.copy_(torch.randn(co) * 0.3)
        gamma, beta = bn.bn.weight, bn.bn.bias
        vn.train()
    else:
        with torch.no_grad():
            vn.bias.copy_(torch.randn(co) * 0.3)
        gamma, beta = torch.ones(co), vn.bias
    out_ref = vn(y_ref_in)
    dout = torch.randn(2 * n, co)
    params = [a] + ([bn.bn.weight, bn.bn.bias] if training else [vn.bias])
    grads = torch.autograd.grad(out_ref, params, dout)
    ld = 2 * co if combine else co
    out, din, dg, db = torch.zeros(2 * n, co), torch.zeros(2 * n, ld), torch.zeros(co), torch.zeros(co)
    hc.hc_vn(P(inp), n, co, combine, P(gamma.detach().contiguous()), P(beta.detach().contiguous()), 1e-5, training,
             P(out), P(dout), P(din), P(dg), P(db))
    assert rel_err(out, out_ref) < 1e-5
    if combine:   # chain d[P|Q] back to v_cat through the stacked weights
        da = din @ torch.cat([W[:, :K], W[:, K:]], 0)
    else:
        da = din
    assert rel_err(da, grads[0]) < 1e-4
    if training:
        assert rel_err(dg, grads[1]) < 1e-4 and rel_err(db, grads[2]) < 1e-4
    else:
        assert rel_err(db, grads[1]) < 1e-4


@pytest.mark.parametrize("slope,training", [(0.2, 1), (0.0, 1), (0.2, 0)])
def test_edge_mlp_without_edge_tensor(hc, slope, training):
    """edge_math.h (analytic max + moment statistics + closed-form BN backward) vs the materialised
    reference formulation: BN over all E edge rows of y_j - y_i, LeakyReLU, max over the k-list."""
    from deltaconv_amd.data import synthetic_batch
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    hc.hc_edge.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, cf, cf, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    hc.hc_csc_build.argtypes = [vp, ci, ci, vp, vp]
    torch.manual_seed(2)
    b = synthetic_batch(2, 0, seed=13, sizes=[60, 45], dup_frac=0.05)
    n, k, C = b.pos.shape[0], 9, 6
    nbr = geo.knn(b.pos, k, geo.cloud_ptr(b.batch))
    nbr32 = nbr.to(torch.int32).contiguous()
    tptr, tedge = torch.zeros(n + 1, dtype=torch.int32), torch.zeros(n * k, dtype=torch.int32)
    hc.hc_csc_build(P(nbr32), n, k, P(tptr), P(tedge))
    y = torch.randn(n, C, requires_grad=True)
    gamma = torch.tensor([1.0, -0.8, 0.5, 1.5, -1.2, 0.7], requires_grad=True)
    beta = (torch.randn(C) * 0.3).requires_grad_(True)
    rm, rv = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    a = (y[nbr] - y[:, None, :]).reshape(n * k, C)
    z = F.batch_norm(a, rm.clone(), rv.clone(), gamma, beta, training=bool(training), eps=1e-5)
    h = F.leaky_relu(z, slope).view(n, k, C)
    out_ref, arg_ref = h.max(dim=1)
    dout = torch.randn(n, C)
    gy, gg, gb = torch.autograd.grad(out_ref, [y, gamma, beta], dout)
    out, arg = torch.zeros(n, C), torch.zeros(n, C, dtype=torch.uint8)
    dy, dg, db = torch.zeros(n, C), torch.zeros(C), torch.zeros(C)
    hc.hc_edge(P(y.detach()), P(nbr32), P(tptr), P(tedge), n, k, C, P(gamma.detach()), P(beta.detach()), 1e-5, slope,
               training, P(rm), P(rv), P(out), P(arg), P(dout), P(dy), P(dg), P(db))
    assert rel_err(out, out_ref) < 1e-5
    if slope > 0:
        assert torch.equal(arg.long(), arg_ref)
    assert rel_err(dy, gy) < 1e-4 and rel_err(dg, gg) < 1e-4 and rel_err(db, gb) < 1e-4


@pytest.mark.parametrize("R,C,smoothing", [(32, 40, True), (257, 50, False), (1, 2, True), (5, 1, False), (64, 15, True)])
def test_ce_loss_row_formula(hc, R, C, smoothing):
    """csrc/loss_math.h == the oracle's restatement of experiments/utils.py:7-24, value and gradient."""
    torch.manual_seed(R + C)
    x = (torch.randn(R, C) * 3).requires_grad_(True)
    y = torch.randint(0, C, (R,))
    ref = oracle.loss.calc_loss(x, y, smoothing=smoothing)
    ref.backward()
    dx = torch.empty(R, C)
    got = hc.hc_ce_loss(P(x.detach()), P(y), R, C, 0.2 if smoothing else 0.0, P(dx))
    assert abs(got - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert rel_err(dx, x.grad) < 5e-6 or float((dx - x.grad).abs().max()) < 1e-7


def test_philox_known_answers_and_dropout_keep(hc):
    """The dropout draws of the row-block kernels (csrc/nn_math.h): Philox-4x32-10 against the known-answer vectors of the
    Random123 distribution (kat_vectors: counter / key all zero, all ones, digits of pi), and the keep decision."""
    import ctypes
    import numpy as np
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        c = (ctypes.c_uint32 * 4)(*ctr)
        out = (ctypes.c_uint32 * 4)()
        hc.hc_philox(c, ctypes.c_uint32(key[0]), ctypes.c_uint32(key[1]), out)
        assert tuple(out) == want, [hex(v) for v in out]
    n = 1 << 16
    keep = np.zeros(n, dtype=np.uint8)
    for p in (0.5, 0.1):
        hc.hc_dropout_keep(ctypes.c_uint32(1234), ctypes.c_longlong(7), ctypes.c_uint32(1), n, ctypes.c_float(p),
                           keep.ctypes.data_as(ctypes.c_void_p))
        assert abs(keep.mean() - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-3          # 4 sigma
    a = keep.copy()
    hc.hc_dropout_keep(ctypes.c_uint32(1234), ctypes.c_longlong(8), ctypes.c_uint32(1), n, ctypes.c_float(0.1),
                       keep.ctypes.data_as(ctypes.c_void_p))
    assert (a != keep).mean() > 0.1                   # another step: another mask
    b = keep.copy()
    hc.hc_dropout_keep(ctypes.c_uint32(1234), ctypes.c_longlong(8), ctypes.c_uint32(2), n, ctypes.c_float(0.1),
                       keep.ctypes.data_as(ctypes.c_void_p))
    assert (b != keep).mean() > 0.1                   # another layer: another mask
