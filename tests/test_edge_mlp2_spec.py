"""Executable specification of the FUSED depth-2 centralised edge MLP (the one composed block left in C4's layer 0,
/root/reference/deltaconv/nn/deltaconv.py:50-52 with mlp_depth = 2, experiments/train_shapenet.py:77-89):

    out[i] = max_s  act(bn2( W2 act(bn1( W1 (x_j - x_i) )) )),   j = nbr[i, s],  BatchNorm statistics over all E = N k edges.

The reference (and today's product path, nn/deltaconv.py::_centralized_max) materialises two [E, C] activations.  The fused
form below never holds an [E, C] tensor across a pass -- every pass walks the edges in chunks of points and keeps only [N, C]
and [C] quantities -- and is what a gathered-operand MFMA kernel has to compute.  This file pins its algebra on the CPU in
fp64 against autograd of the composed formulation (oracle blocks), forward and every gradient:

  forward, ONE pass over the edges:   z = x W1^T  [N, C]  (linear: W1 (x_j - x_i) = z_j - z_i);
      statistics of y1 = z_j - z_i (pass 0, as the depth-1 analytic form does today);
      per chunk: h1 = act(bn1(y1)), y2 = h1 W2^T; accumulate sum / sum of squares of y2; per point and channel keep
      max_s y2, min_s y2 and their first slots.  act(bn2(.)) is monotone per channel (non-decreasing for gamma2 >= 0,
      non-increasing below), so  out = act(bn2(sel)),  sel = max or min by the sign of the scale -- no second pass.
  backward, TWO recompute passes:  d z2 lives on the selected edges only, so the two BatchNorm-2 sums come from [N, C] data;
      pass A recomputes h1, y2 per chunk: dy2 = g2 (dz2 - m1 - xhat2 m2), dW2 += dy2^T h1, du1 = (dy2 W2) act'(.),
      accumulates the two BatchNorm-1 sums;  pass B recomputes du1 the same way and forms dy1 = g1 (du1 - n1 - xhat1 n2),
      scattered to dz (+ at j, - at i);  then dW1 = dz^T x, dx = dz W1.
"""
import pytest
import torch

SLOPE = 0.2
EPS = 1e-5


def act(t):
    return torch.where(t > 0, t, SLOPE * t)


def dact(t):
    return torch.where(t > 0, torch.ones_like(t), torch.full_like(t, SLOPE))


def composed(x, nbr, W1, g1, b1, W2, g2, b2):
    """The reference formulation on [E, C] tensors (train-mode BatchNorm, biased variance)."""
    n, k = nbr.shape
    xe = (x[nbr] - x[:, None, :]).reshape(n * k, -1)
    y1 = xe @ W1.t()
    h1 = act((y1 - y1.mean(0)) / torch.sqrt(y1.var(0, unbiased=False) + EPS) * g1 + b1)
    y2 = h1 @ W2.t()
    h2 = act((y2 - y2.mean(0)) / torch.sqrt(y2.var(0, unbiased=False) + EPS) * g2 + b2)
    return h2.view(n, k, -1).max(dim=1).values


def chunks(n, size):
    return [(a, min(a + size, n)) for a in range(0, n, size)]


def fused_forward(x, nbr, W1, g1, b1, W2, g2, b2, chunk):
    n, k = nbr.shape
    E = n * k
    z = x @ W1.t()
    c = z.shape[1]
    # pass 0: statistics of y1 = z_j - z_i (today's dc_edge_gather_stats)
    s0 = torch.zeros(c, dtype=x.dtype)
    s1 = torch.zeros(c, dtype=x.dtype)
    for a, b in chunks(n, chunk):
        y1 = z[nbr[a:b]] - z[a:b, None, :]
        s0 += y1.sum((0, 1))
        s1 += (y1 * y1).sum((0, 1))
    mu1 = s0 / E
    is1 = 1.0 / torch.sqrt(s1 / E - mu1 * mu1 + EPS)
    sc1, sh1 = g1 * is1, b1 - mu1 * g1 * is1
    # pass 1: y2 per chunk -> statistics + per-point extrema
    t0 = torch.zeros(c, dtype=x.dtype)
    t1 = torch.zeros(c, dtype=x.dtype)
    ymax = torch.empty(n, c, dtype=x.dtype)
    ymin = torch.empty(n, c, dtype=x.dtype)
    amax = torch.empty(n, c, dtype=torch.long)
    amin = torch.empty(n, c, dtype=torch.long)
    for a, b in chunks(n, chunk):
        y1 = z[nbr[a:b]] - z[a:b, None, :]
        y2 = act(sc1 * y1 + sh1) @ W2.t()                                  # [p, k, C]: the MFMA product of the kernel
        t0 += y2.sum((0, 1))
        t1 += (y2 * y2).sum((0, 1))
        ymax[a:b], amax[a:b] = y2.max(dim=1)
        ymin[a:b], amin[a:b] = y2.min(dim=1)
    mu2 = t0 / E
    is2 = 1.0 / torch.sqrt(t1 / E - mu2 * mu2 + EPS)
    sc2, sh2 = g2 * is2, b2 - mu2 * g2 * is2
    up = sc2 >= 0
    sel = torch.where(up, ymax, ymin)
    arg = torch.where(up, amax, amin)
    out = act(sc2 * sel + sh2)
    saved = dict(z=z, mu1=mu1, is1=is1, sc1=sc1, sh1=sh1, mu2=mu2, is2=is2, sc2=sc2, sh2=sh2, sel=sel, arg=arg)
    return out, saved


def fused_backward(dout, x, nbr, W1, g1, W2, g2, sv, chunk):
    n, k = nbr.shape
    E = n * k
    z, c = sv["z"], sv["z"].shape[1]
    # BatchNorm-2 backward sums: d z2 is non-zero on the selected edge of every (point, channel) only
    dz2 = dout * dact(sv["sc2"] * sv["sel"] + sv["sh2"])                   # [N, C], sits at slot arg[i, c]
    xh_sel = (sv["sel"] - sv["mu2"]) * sv["is2"]
    db2 = dz2.sum(0)
    dg2 = (dz2 * xh_sel).sum(0)
    m1, m2 = db2 / E, dg2 / E
    gi2 = g2 * sv["is2"]

    def du1_of(a, b):
        y1 = z[nbr[a:b]] - z[a:b, None, :]
        pre1 = sv["sc1"] * y1 + sv["sh1"]
        h1 = act(pre1)
        y2 = h1 @ W2.t()
        hot = torch.zeros_like(y2)
        hot.scatter_(1, sv["arg"][a:b, None, :], dz2[a:b, None, :])        # d z2 of the chunk's edges
        dy2 = gi2 * (hot - m1 - (y2 - sv["mu2"]) * sv["is2"] * m2)
        return y1, h1, dy2, (dy2 @ W2) * dact(pre1)

    # pass A: d W2 and the two BatchNorm-1 sums
    dW2 = torch.zeros_like(W2)
    db1 = torch.zeros(c, dtype=x.dtype)
    dg1 = torch.zeros(c, dtype=x.dtype)
    for a, b in chunks(n, chunk):
        y1, h1, dy2, du1 = du1_of(a, b)
        dW2 += dy2.reshape(-1, c).t() @ h1.reshape(-1, c)
        db1 += du1.sum((0, 1))
        dg1 += (du1 * (y1 - sv["mu1"]) * sv["is1"]).sum((0, 1))
    n1, n2 = db1 / E, dg1 / E
    gi1 = g1 * sv["is1"]
    # pass B: d y1 -> d z (transposed scatter: + at the neighbour, - at the centre)
    dz = torch.zeros_like(z)
    for a, b in chunks(n, chunk):
        y1, _, _, du1 = du1_of(a, b)
        dy1 = gi1 * (du1 - n1 - (y1 - sv["mu1"]) * sv["is1"] * n2)
        dz.index_add_(0, nbr[a:b].reshape(-1), dy1.reshape(-1, c))
        dz[a:b] -= dy1.sum(1)
    return dict(x=dz @ W1, W1=dz.t() @ x, g1=dg1, b1=db1, W2=dW2, g2=dg2, b2=db2)


@pytest.mark.parametrize("seed,n,k,ci,c,chunk", [(0, 96, 8, 3, 16, 7), (1, 64, 20, 6, 32, 64), (2, 50, 5, 3, 8, 1)])
def test_fused_depth2_edge_mlp_equals_composed(seed, n, k, ci, c, chunk):
    gen = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
    x = rnd(n, ci)
    nbr = torch.stack([torch.randperm(n, generator=gen)[:k] for _ in range(n)])
    nbr[:, 0] = torch.arange(n)                                            # the self loop of the kNN graph
    W1, W2 = rnd(c, ci) * 0.7, rnd(c, c) * 0.4
    g1, b1 = rnd(c), rnd(c) * 0.3                                          # both signs of gamma: max AND min selections
    g2, b2 = rnd(c), rnd(c) * 0.3
    assert bool((g2 < 0).any()) and bool((g2 > 0).any())
    leaves = [t.clone().requires_grad_(True) for t in (x, W1, g1, b1, W2, g2, b2)]
    ref = composed(leaves[0], nbr, *leaves[1:])
    dout = rnd(n, c)
    ref.backward(dout)
    out, sv = fused_forward(x, nbr, W1, g1, b1, W2, g2, b2, chunk)
    assert torch.allclose(out, ref.detach(), rtol=1e-11, atol=1e-11)
    got = fused_backward(dout, x, nbr, W1, g1, W2, g2, sv, chunk)
    for name, leaf in zip(("x", "W1", "g1", "b1", "W2", "g2", "b2"), leaves):
        scale = float(leaf.grad.abs().max()) + 1e-30
        err = float((got[name] - leaf.grad).abs().max()) / scale
        assert err < 1e-9, (name, err)
