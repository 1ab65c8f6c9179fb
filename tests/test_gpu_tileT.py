"""GPU: the TRANSPOSED tile plan (deltaconv_amd/csrc/tile_plan.h second half, tileplan.hip: tileT_build_kernel) and the
transposed applies / max-aggregation backward that run from it (csrc/ell_tileT.h) through the C ABI.

Like the forward plan this is an acceleration structure (the reference has none: torch_sparse's autograd spmm with A^T
and torch_scatter's arg-indexed backward, /root/reference/deltaconv/nn/deltaconv.py:52-57,66); what must hold:
  * structure: the targets of a tile are the tile's points ordered by in-degree (descending, ties by position); the
    tile's records are the CSC columns of its targets in that order, every record's local index points at the edge's
    source in the ascending unique list; tile ranges start at multiples of 4 entries and do not overlap;
  * results: every tiled transposed entry point returns the SAME BITS as its gather-path counterpart (same FMAs, ascending
    edge id per target; the gather kernels are the ones checked against the oracle and the reference's golden vectors),
    with and without accumulation, on coherent kNN graphs, ragged clouds, duplicates, k = 10 / 20 / 30, strided operands,
    and on adversarial graphs whose tiles overflow the LDS capacity in rows and in edge records;
  * whole model: logits and every parameter gradient identical with the transposed plan switched on and off.
"""
import pytest
import torch

from deltaconv_amd.data import synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(sizes, k, seed=3, dup_frac=0.0):
    from deltaconv_amd.geometry import Graph, build_grad_div, build_tangent_basis
    b = synthetic_batch(len(sizes), 0, seed=seed, sizes=list(sizes), dup_frac=dup_frac).to(DEV)
    gr = Graph.knn(b.pos, k, b.batch)
    xb, yb = build_tangent_basis(b.norm)
    grad, div = build_grad_div(b.pos, b.norm, xb, yb, gr, b.batch)
    gr.tile_plan(force_P=64 if k <= 24 else 32)
    return b, gr, grad, div


CASES = [((256, 256, 256, 256), 20), ((512, 700, 300), 20), ((1024, 1024), 30), ((200, 64, 333), 10), ((4096,), 20), ((700, 90), 64)]


@pytest.mark.parametrize("sizes,k", CASES)
def test_planT_structure(sizes, k):
    b, gr, _, _ = _setup(sizes, k, dup_frac=0.03)
    fwd, pt = gr.tile_plan(), gr.tile_plan_T()
    assert pt is not None and pt.P == fwd.P
    P, n = pt.P, gr.n
    tg, hdr, uniq, rec, edge = (pt.section(s).cpu() for s in ("tg", "hdr", "uniq", "rec", "edge"))
    pts = fwd.section("pts").cpu()
    tptr, tedge = (t.cpu().long() for t in gr.csc())
    deg_all = tptr[1:] - tptr[:-1]
    seen_edges = []
    prev_end = 0
    for t in range(pts.shape[0]):
        ids = pts[t][pts[t] >= 0].long()
        m = ids.numel()
        U, start, Et = (int(x) for x in hdr[t, :3])
        if m == 0:
            assert Et == 0 and U == 0 and bool((tg[t, :, 0] < 0).all())
            continue
        # targets = the tile's points, by (degree descending, position ascending)
        d = deg_all[ids]
        order = sorted(range(m), key=lambda p: (-int(d[p]), p))
        assert tg[t, :m, 0].tolist() == ids[order].tolist()
        assert bool((tg[t, m:, 0] < 0).all()) and bool((tg[t, m:, 2] == 0).all())
        assert tg[t, :m, 2].tolist() == d[order].tolist()
        assert tg[t, :m, 1].tolist() == (torch.cumsum(d[order], 0) - d[order]).tolist()
        assert Et == int(d.sum()) and start % 4 == 0 and start >= prev_end
        prev_end = start + ((Et + 3) & ~3)
        assert prev_end <= pt.edges
        # records: the CSC columns of the targets, in that order
        want_e = torch.cat([tedge[tptr[j]:tptr[j + 1]] for j in ids[order].tolist()]) if Et else torch.zeros(0, dtype=torch.long)
        got_e = edge[start:start + Et].long()
        assert torch.equal(got_e, want_e)
        seen_edges.append(got_e)
        src, slot = got_e // k, got_e % k
        r = rec[start:start + Et].long() & 0xffffffff
        assert torch.equal(r >> 16, slot)
        want_u = torch.unique(src)
        assert U == want_u.numel()
        u = uniq[t, :min(U, 256)].long()
        assert torch.equal(u, want_u[:256])
        if U < 256:
            assert bool((uniq[t, U:] == uniq[t, U - 1]).all())
        assert torch.equal(want_u[r & 0xffff], src)
        # padding entries are valid edge ids
        pad = edge[start + Et:prev_end].long()
        assert bool(((pad >= 0) & (pad < n * k)).all())
    allE = torch.cat(seen_edges)
    assert allE.numel() == n * k and torch.equal(torch.sort(allE).values, torch.arange(n * k))
    # degree ordering pays: wave-level loop count (max of 4 consecutive lane groups) close to the mean
    dg = tg[:, :, 2].float().view(-1, P // 4, 4)
    full = (pts >= 0).all(1)
    if full.any():
        waves = dg[full].max(2).values.sum(1)
        ideal = dg[full].sum((1, 2)) / 4
        assert float((waves / ideal).mean()) < 1.12


def _rand(*shape):
    return torch.randn(*shape, device=DEV)


def _pairT(fn_plain, fn_tiled, shape, accumulate):
    base = _rand(*shape) if accumulate else torch.full(shape, float("nan"), device=DEV)
    ref, got = base.clone(), base.clone()
    fn_plain(ref)
    fn_tiled(got)
    torch.cuda.synchronize()
    assert torch.equal(ref, got)
    assert bool(torch.isfinite(got).all())


def _check_all(gr, grad, div, C, seed=0):
    from deltaconv_amd._lib import lib
    pt = gr.tile_plan_T()
    n, k = gr.n, gr.k
    tptr, tedge = gr.csc()
    a = pt.args
    torch.manual_seed(C + k + seed)
    GT, DT, GTt, DTt = grad.coefT(), div.coefT(), grad.coefTt(), div.coefTt()
    dyv, dys, dout, v = _rand(2 * n, C), _rand(n, C), _rand(n, 3 * C), _rand(2 * n, C)
    v[5] = 0.0
    v[4] = 0.0                                                             # |v| = 0: subgradient 0
    for acc in (0, 1):
        _pairT(lambda o: lib.call("dc_apply_grad_T", GT, tptr, tedge, n, k, dyv, C, C, o, C, acc),
               lambda o: lib.call("dc_apply_grad_T_tiled", GTt, pt.blob, *a, dyv, C, C, o, C, acc), (n, C), acc)
        _pairT(lambda o: lib.call("dc_apply_div_T", DT, tptr, tedge, n, k, dys, C, C, o, C, acc),
               lambda o: lib.call("dc_apply_div_T_tiled", DTt, pt.blob, *a, dys, C, C, o, C, acc), (2 * n, C), acc)
        _pairT(lambda o: lib.call("dc_apply_hodge_T", GT, tptr, tedge, n, k, dyv, C, C, o, 2 * C, acc),
               lambda o: lib.call("dc_apply_hodge_T_tiled", GTt, pt.blob, *a, dyv, C, C, o, 2 * C, acc), (n, 2 * C), acc)
        _pairT(lambda o: lib.call("dc_apply_div_curl_norm_T", DT, tptr, tedge, n, k, dout, C, 3 * C, v, C, o, C, acc),
               lambda o: lib.call("dc_apply_div_curl_norm_T_tiled", DTt, pt.blob, *a, dout, C, 3 * C, v, C, o, C, acc),
               (2 * n, C), acc)
        arg = torch.randint(0, k, (n, C), device=DEV).to(torch.uint8)
        _pairT(lambda o: lib.call("dc_knn_max_backward", tptr, tedge, n, k, arg, dys, C, C, o, C, acc),
               lambda o: lib.call("dc_knn_max_backward_tiled", pt.blob, *a, arg, dys, C, C, o, C, acc), (n, C), acc)
    ga, gb = _rand(n, C), _rand(n, C)
    for b_ in (None, gb):
        _pairT(lambda o: lib.call("dc_apply_grad_T_sum", GT, tptr, tedge, n, k, dyv, C, C, ga, C, b_, C, o, C),
               lambda o: lib.call("dc_apply_grad_T_sum_tiled", GTt, pt.blob, *a, dyv, C, C, ga, C, b_, C, o, C), (n, C), 0)


@pytest.mark.parametrize("sizes,k", CASES)
@pytest.mark.parametrize("C", [64, 128, 256])
def test_tiledT_equals_gather(sizes, k, C):
    _, gr, grad, div = _setup(sizes, k, dup_frac=0.03)
    _check_all(gr, grad, div, C)


def test_tiledT_strided_operands():
    """Operands and results as column blocks of wider buffers, as the layer node passes them."""
    from deltaconv_amd._lib import lib
    _, gr, grad, div = _setup((512, 512), 20)
    pt = gr.tile_plan_T()
    n, k, ci, co = gr.n, gr.k, 64, 128
    K = 2 * ci + co
    tptr, tedge = gr.csc()
    a = pt.args
    dv_cat, d_xcat, v = _rand(2 * n, K), _rand(n, 4 * ci), _rand(2 * n, 2 * ci)
    # hodge^T accumulates into d[div | curl] of the s_mlp operand gradient
    x1, x2 = d_xcat.clone(), d_xcat.clone()
    lib.call("dc_apply_hodge_T", grad.coefT(), tptr, tedge, n, k, dv_cat[:, ci:], ci, K, x1[:, ci:], 4 * ci, 1)
    lib.call("dc_apply_hodge_T_tiled", grad.coefTt(), pt.blob, *a, dv_cat[:, ci:], ci, K, x2[:, ci:], 4 * ci, 1)
    assert torch.equal(x1, x2)
    # [div | curl | norm]^T accumulates into the v block of the v_mlp operand gradient
    d1, d2 = dv_cat.clone(), dv_cat.clone()
    lib.call("dc_apply_div_curl_norm_T", div.coefT(), tptr, tedge, n, k, x1[:, ci:], ci, 4 * ci, v[:, :ci], 2 * ci, d1[:, :ci], K, 1)
    lib.call("dc_apply_div_curl_norm_T_tiled", div.coefTt(), pt.blob, *a, x2[:, ci:], ci, 4 * ci, v[:, :ci], 2 * ci, d2[:, :ci], K, 1)
    assert torch.equal(d1, d2)
    # grad^T sum reads the `grad @ x'` block
    ga = _rand(n, co)
    o1, o2 = torch.empty(n, co, device=DEV), torch.empty(n, co, device=DEV)
    lib.call("dc_apply_grad_T_sum", grad.coefT(), tptr, tedge, n, k, dv_cat[:, 2 * ci:], co, K, ga, co, None, 0, o1, co)
    lib.call("dc_apply_grad_T_sum_tiled", grad.coefTt(), pt.blob, *a, dv_cat[:, 2 * ci:], co, K, ga, co, None, 0, o2, co)
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("sizes,k", [((512, 700, 300), 20), ((1024, 1024), 30)])
@pytest.mark.parametrize("training", [1, 0])
def test_edge_backward_tiled_equals_gather(sizes, k, training):
    """Backward of the layer-0 edge MLP (dc_edge_max_backward: /root/reference/deltaconv/nn/deltaconv.py:50-52): the CSC
    pass from the transposed plan gives the same bits as the gather kernel, for both BatchNorm modes and mixed-sign scales."""
    from deltaconv_amd._lib import lib
    from deltaconv_amd.nn import fused
    _, gr, _, _ = _setup(sizes, k, dup_frac=0.03)
    pt = gr.tile_plan_T()
    n, C = gr.n, 64
    tptr, tedge = gr.csc()
    torch.manual_seed(k + training)
    y, dout = _rand(n, C), _rand(n, C)
    gamma, beta = _rand(C), _rand(C)                                       # mixed signs: max and min selections
    stat = torch.empty(3, n, C, device=DEV)
    args = torch.empty(2, n, C, dtype=torch.uint8, device=DEV)
    coef = torch.empty(4, C, device=DEV)
    ws, nb = fused._ws(n, C, DEV)
    lib.call("dc_edge_gather_stats", y, C, gr.nbr, n, k, C, 1, gamma, beta, 1e-5, 0.1, None, None, stat[0], stat[1], args[0], args[1],
             stat[2], coef[0], coef[1], coef[2], coef[3], ws, nb)
    xmax, argsel = torch.empty(n, C, device=DEV), torch.empty(n, C, dtype=torch.uint8, device=DEV)
    lib.call("dc_edge_max_apply", stat[0], stat[1], args[0], args[1], n, C, coef[2], coef[3], 0.2, xmax, C, argsel)
    assert torch.equal(argsel, torch.where(coef[2] >= 0, args[0], args[1]))
    outs = []
    for tiled in (False, True):
        dzs, dy = torch.full((n, C), float("nan"), device=DEV), torch.full((n, C), float("nan"), device=DEV)
        dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        if tiled:
            lib.call("dc_edge_max_backward_tiled", dout, C, y, pt.blob, *pt.args, C, stat[0], stat[1], argsel, stat[2], coef[2],
                     coef[3], coef[0], coef[1], 0.2, training, dzs, dy, C, dg, db, ws, nb)
        else:
            lib.call("dc_edge_max_backward", dout, C, y, C, tptr, tedge, n, k, C, stat[0], stat[1], args[0], args[1], stat[2],
                     coef[2], coef[3], coef[0], coef[1], 0.2, training, dzs, dy, C, dg, db, ws, nb)
        torch.cuda.synchronize()
        outs.append((dzs, dy, dg, db))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_) and bool(torch.isfinite(b_).all())


def test_tiledT_rejects_what_it_cannot_do():
    from deltaconv_amd._lib import lib
    _, gr, grad, _ = _setup((256,), 20)
    pt = gr.tile_plan_T()
    n, C = gr.n, 48
    dy, o = _rand(2 * n, C), torch.empty(n, C, device=DEV)
    with pytest.raises(RuntimeError, match="C % 64"):
        lib.call("dc_apply_grad_T_tiled", grad.coefTt(), pt.blob, *pt.args, dy, C, C, o, C, 0)


@pytest.mark.parametrize("k,mode", [(20, "random"), (30, "random"), (20, "hub")])
def test_tiledT_overflow_tiles(k, mode):
    """Graphs without spatial coherence: 'random' = random neighbours inside the cloud (the unique sources of a tile exceed
    the LDS capacity: the excess rows come from global memory by source id); 'hub' = every point lists the same few
    points (one tile collects most in-edges of the cloud: its edge list exceeds the LDS capacity, other targets have no
    in-edge at all).  Also > k coincident points."""
    from deltaconv_amd.geometry import Graph
    from deltaconv_amd.geometry.grad_div_mls import SparseOp
    torch.manual_seed(k)
    sizes = [1024, 777]
    n = sum(sizes)
    pos = torch.randn(n, 3)
    pos[100:100 + k + 9] = pos[100]
    batch = torch.repeat_interleave(torch.arange(2), torch.tensor(sizes))
    if mode == "random":
        nbr = torch.cat([torch.randint(0, s, (s, k)) + o for s, o in zip(sizes, (0, sizes[0]))]).to(torch.int32)
    else:
        nbr = torch.cat([torch.randint(0, 40, (s, k)) + o for s, o in zip(sizes, (0, sizes[0]))]).to(torch.int32)
    ei = torch.stack([torch.arange(n).repeat_interleave(k), nbr.reshape(-1).long()]).to(DEV)
    gr = Graph.from_edge_index(ei, n, k=k, batch=batch.to(DEV))
    gr.pos = pos.to(DEV)
    gr.tile_plan(force_P=64 if k <= 24 else 32)
    pt = gr.tile_plan_T()
    hdr = pt.section("hdr").cpu()
    if mode == "random":
        assert int(hdr[:, 0].max()) > 248
    else:
        assert int(hdr[:, 2].max()) > 2048 and int((pt.section("tg")[:, :, 2] == 0).sum()) > 0
    coef = _rand(n, k, 2)
    grad, div = SparseOp("grad", gr, coef), SparseOp("div", gr, coef.flip(2).contiguous())
    _check_all(gr, grad, div, 64)


@pytest.mark.parametrize("kind", ["cls", "seg"])
def test_model_identical_with_and_without_planT(kind):
    """Train-mode forward + backward of whole models: logits and every parameter gradient identical bit for bit with the
    transposed applies running from the transposed plan and over the CSC through the gather path."""
    from deltaconv_amd.geometry import graph as G
    from deltaconv_amd.models import DeltaNetClassification, DeltaNetSegmentation
    from deltaconv_amd.utils import calc_loss
    b = synthetic_batch(3, 512, seed=12, per_point_labels=(kind == "seg"), num_classes=8 if kind == "seg" else 40).to(DEV)

    def run(use):
        import os
        G.USE_TILE_PLAN_T[0] = use
        os.environ["DC_TILE_P"] = "64"                     # a batch this small would not get a plan by itself
        try:
            torch.manual_seed(4)
            m = (DeltaNetClassification(3, 40) if kind == "cls" else DeltaNetSegmentation(3, 8, mlp_depth=1)).to(DEV).train()
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            out = m(b)
            calc_loss(out, b.y, smoothing=(kind == "cls")).backward()
            return out.detach(), [p.grad.clone() for p in m.parameters() if p.grad is not None]
        finally:
            G.USE_TILE_PLAN_T[0] = True
            os.environ.pop("DC_TILE_P", None)

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert torch.equal(o1, o0)
    assert len(g1) == len(g0) and all(torch.equal(a_, b_) for a_, b_ in zip(g1, g0))
