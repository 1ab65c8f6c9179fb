"""GPU: the tile plan (deltaconv_amd/csrc/tile_plan.h, tileplan.hip) and the forward applies / max aggregation that run
from it (csrc/ell_tile.h) through the C ABI.

The plan is an acceleration structure (the reference has none); what must hold:
  * structure: every point sits in exactly one tile of its own cloud, the unique list of a tile is ascending and holds
    every neighbour of the tile's points and the points themselves, the tile-local indices point at the right rows;
  * results: every tiled entry point returns the SAME BITS as its plain counterpart (same FMAs in the same slot order;
    the plain kernels are the ones checked against the oracle and the reference's golden vectors in
    test_gpu_geometry.py), on coherent kNN graphs, ragged clouds, duplicate points, k = 10 / 20 / 30, strided operands,
    and on an adversarial graph whose tiles overflow the LDS capacity (rows fetched from global memory by id);
  * whole model: logits and every parameter gradient identical with the plan switched on and off.
"""
import pytest
import torch

from deltaconv_amd.data import synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(sizes, k, seed=3, dup_frac=0.0, normals=True):
    from deltaconv_amd.geometry import Graph, build_grad_div, build_tangent_basis
    b = synthetic_batch(len(sizes), 0, seed=seed, sizes=list(sizes), dup_frac=dup_frac).to(DEV)
    gr = Graph.knn(b.pos, k, b.batch)
    xb, yb = build_tangent_basis(b.norm)
    grad, div = build_grad_div(b.pos, b.norm, xb, yb, gr, b.batch)
    return b, gr, grad, div


CASES = [((256, 256, 256, 256), 20), ((512, 700, 300), 20), ((1024, 1024), 30), ((200, 64, 333), 10), ((4096,), 20), ((700, 90), 64)]


@pytest.mark.parametrize("sizes,k", CASES)
def test_plan_structure(sizes, k):
    b, gr, _, _ = _setup(sizes, k, dup_frac=0.03)
    plan = gr.tile_plan(force_P=64 if k <= 24 else 32)
    assert plan is not None and plan.P == (64 if k <= 24 else 32)
    P, n = plan.P, gr.n
    pts, nu, uniq, loc, slf = (plan.section(s).cpu() for s in ("pts", "nu", "uniq", "loc", "self"))
    nbr = gr.nbr.cpu().long()
    batch = b.batch.cpu()
    valid = pts >= 0
    # every point exactly once
    flat = pts[valid].long()
    assert flat.numel() == n and torch.equal(torch.sort(flat).values, torch.arange(n))
    # the occupied tile ids are a dense prefix (unused ids trail: a launch meets its empty workgroups last), and equal-sized
    # clouds leave no unused id at all
    occ = valid.any(1)
    assert bool(occ[:int(occ.sum())].all())
    if len(set(sizes)) == 1:
        assert bool(occ.all())
    # a tile never mixes clouds; empty tiles have no unique rows; padding only at the end of a tile
    for t in range(pts.shape[0]):
        row = pts[t]
        m = int(valid[t].sum())
        assert bool((row[:m] >= 0).all()) and bool((row[m:] < 0).all())
        if m == 0:
            assert int(nu[t]) == 0
            continue
        ids = row[:m].long()
        assert len(set(batch[ids].tolist())) == 1
        U = int(nu[t])
        u = uniq[t, :U].long()
        assert bool((u[1:] > u[:-1]).all())                               # ascending, unique
        assert bool((uniq[t, U:] == uniq[t, U - 1]).all())                # tail repeats the last id
        want = torch.unique(torch.cat([nbr[ids].reshape(-1), ids]))
        assert torch.equal(u, want)
        l = loc[t].view(P, k)[:m].long()
        assert torch.equal(u[l], nbr[ids])                                # local index -> the neighbour's row
        assert torch.equal(u[slf[t, :m].long()], ids)                     # and the point's own row
    # Morton tiles are spatially coherent: far fewer unique rows than P * k
    full = nu[valid.sum(1) == P].float()
    if full.numel():
        assert float(full.mean()) < 0.35 * P * k


def _rand(*shape):
    return torch.randn(*shape, device=DEV)


def _pair(fn_plain, fn_tiled, *outs):
    """Run both, compare bits."""
    ref = [torch.full_like(o, float("nan")) if o.dtype.is_floating_point else torch.full_like(o, 255) for o in outs]
    got = [r.clone() for r in ref]
    fn_plain(*ref)
    fn_tiled(*got)
    torch.cuda.synchronize()
    for r, g_ in zip(ref, got):
        assert torch.equal(r, g_)


@pytest.mark.parametrize("sizes,k", CASES)
@pytest.mark.parametrize("C", [64, 128, 256])
def test_tiled_equals_plain(sizes, k, C):
    from deltaconv_amd._lib import lib
    _, gr, grad, div = _setup(sizes, k, dup_frac=0.03)
    plan = gr.tile_plan(force_P=64 if k <= 24 else 32)
    n = gr.n
    torch.manual_seed(C + k)
    x, v, dcn = _rand(n, C), _rand(2 * n, C), _rand(n, 3 * C)
    GP, DP = grad.coef, div.coef
    a = plan.args
    _pair(lambda o: lib.call("dc_apply_grad", grad.coef, gr.nbr, n, k, x, C, C, o, C),
          lambda o: lib.call("dc_apply_grad_tiled", GP, plan.blob, gr.nbr, *a, x, C, C, o, C), torch.empty(2 * n, C, device=DEV))
    _pair(lambda o: lib.call("dc_apply_div", div.coef, gr.nbr, n, k, v, C, C, o, C),
          lambda o: lib.call("dc_apply_div_tiled", DP, plan.blob, gr.nbr, *a, v, C, C, o, C), torch.empty(n, C, device=DEV))
    _pair(lambda o: lib.call("dc_apply_div_curl_norm", div.coef, gr.nbr, n, k, v, C, C, o, 3 * C),
          lambda o: lib.call("dc_apply_div_curl_norm_tiled", DP, plan.blob, gr.nbr, *a, v, C, C, o, 3 * C),
          torch.empty(n, 3 * C, device=DEV))
    _pair(lambda o: lib.call("dc_apply_hodge", grad.coef, gr.nbr, n, k, dcn, C, 3 * C, o, C),
          lambda o: lib.call("dc_apply_hodge_tiled", GP, plan.blob, gr.nbr, *a, dcn, C, 3 * C, o, C),
          torch.empty(2 * n, C, device=DEV))
    h = _rand(n, C)
    h[::7] = h[3]                                                          # ties: first maximal slot
    _pair(lambda o, ar: lib.call("dc_knn_max", gr.nbr, n, k, h, C, C, o, C, ar),
          lambda o, ar: lib.call("dc_knn_max_tiled", plan.blob, gr.nbr, *a, h, C, C, o, C, ar),
          torch.empty(n, C, device=DEV), torch.empty(n, C, dtype=torch.uint8, device=DEV))
    scale, shift = _rand(C), _rand(C)
    _pair(lambda o, ar: lib.call("dc_knn_max_affine", gr.nbr, n, k, h, C, C, scale, shift, 0.2, o, C, ar),
          lambda o, ar: lib.call("dc_knn_max_affine_tiled", plan.blob, gr.nbr, *a, h, C, C, scale, shift, 0.2, o, C, ar),
          torch.empty(n, C, device=DEV), torch.empty(n, C, dtype=torch.uint8, device=DEV))


def test_tiled_strided_operands():
    """Operands and results as column blocks of wider buffers (the layer writes straight into the next GEMM's operand)."""
    from deltaconv_amd._lib import lib
    _, gr, grad, div = _setup((512, 512), 20)
    plan = gr.tile_plan(force_P=64)
    n, k, C = gr.n, gr.k, 64
    a = plan.args
    vbuf, obuf = _rand(2 * n, 3 * C), torch.zeros(n, 4 * C, device=DEV)
    v = vbuf[:, C:2 * C]
    o1, o2 = obuf.clone(), obuf.clone()
    lib.call("dc_apply_div_curl_norm", div.coef, gr.nbr, n, k, v, C, 3 * C, o1[:, C:], 4 * C)
    lib.call("dc_apply_div_curl_norm_tiled", div.coef, plan.blob, gr.nbr, *a, v, C, 3 * C, o2[:, C:], 4 * C)
    assert torch.equal(o1, o2) and bool((o2[:, :C] == 0).all())
    hb1, hb2 = torch.zeros(2 * n, 3 * C, device=DEV), torch.zeros(2 * n, 3 * C, device=DEV)
    lib.call("dc_apply_hodge", grad.coef, gr.nbr, n, k, o1[:, C:], C, 4 * C, hb1[:, C:2 * C], 3 * C)
    lib.call("dc_apply_hodge_tiled", grad.coef, plan.blob, gr.nbr, *a, o2[:, C:], C, 4 * C, hb2[:, C:2 * C], 3 * C)
    assert torch.equal(hb1, hb2)


def test_tiled_rejects_what_it_cannot_do():
    from deltaconv_amd._lib import lib
    _, gr, grad, _ = _setup((256,), 20)
    plan = gr.tile_plan(force_P=64)
    n, C = gr.n, 48                                                        # not a multiple of the 64-channel slab
    x, o = _rand(n, C), torch.empty(2 * n, C, device=DEV)
    with pytest.raises(RuntimeError, match="C % 64"):
        lib.call("dc_apply_grad_tiled", grad.coef, plan.blob, gr.nbr, *plan.args, x, C, C, o, C)


@pytest.mark.parametrize("k", [20, 30])
def test_tiled_overflow_tiles(k):
    """A graph without spatial coherence (random neighbours inside the cloud): the unique rows of a tile exceed the LDS
    capacity, the excess is fetched from global memory by neighbour id.  Also > k coincident points (a point that is not
    among its own neighbours)."""
    from deltaconv_amd._lib import lib
    from deltaconv_amd.geometry import Graph
    from deltaconv_amd.geometry.grad_div_mls import SparseOp
    torch.manual_seed(k)
    sizes = [1024, 777]
    n = sum(sizes)
    pos = torch.randn(n, 3)
    pos[100:100 + k + 9] = pos[100]
    batch = torch.repeat_interleave(torch.arange(2), torch.tensor(sizes))
    nbr = torch.cat([torch.randint(0, s, (s, k)) + o for s, o in zip(sizes, (0, sizes[0]))]).to(torch.int32)
    ei = torch.stack([torch.arange(n).repeat_interleave(k), nbr.reshape(-1).long()]).to(DEV)
    gr = Graph.from_edge_index(ei, n, k=k, batch=batch.to(DEV))
    gr.pos = pos.to(DEV)
    plan = gr.tile_plan(force_P=64 if k <= 24 else 32)
    assert plan is not None and int(plan.section("nu").max()) > 248
    C = 64
    coef = _rand(n, k, 2)
    grad, div = SparseOp("grad", gr, coef), SparseOp("div", gr, coef.flip(2).contiguous())
    x, v = _rand(n, C), _rand(2 * n, C)
    a = plan.args
    _pair(lambda o: lib.call("dc_apply_grad", grad.coef, gr.nbr, n, k, x, C, C, o, C),
          lambda o: lib.call("dc_apply_grad_tiled", grad.coef, plan.blob, gr.nbr, *a, x, C, C, o, C),
          torch.empty(2 * n, C, device=DEV))
    _pair(lambda o: lib.call("dc_apply_div_curl_norm", div.coef, gr.nbr, n, k, v, C, C, o, 3 * C),
          lambda o: lib.call("dc_apply_div_curl_norm_tiled", div.coef, plan.blob, gr.nbr, *a, v, C, C, o, 3 * C),
          torch.empty(n, 3 * C, device=DEV))
    _pair(lambda o, ar: lib.call("dc_knn_max", gr.nbr, n, k, x, C, C, o, C, ar),
          lambda o, ar: lib.call("dc_knn_max_tiled", plan.blob, gr.nbr, *a, x, C, C, o, C, ar),
          torch.empty(n, C, device=DEV), torch.empty(n, C, dtype=torch.uint8, device=DEV))


@pytest.mark.parametrize("k,P", [(2, 64), (4, 32)])
def test_tiled_more_unique_rows_than_the_plan_lists(k, P):
    """A user graph WITHOUT self loops and a small k: a tile of P points can have up to P + P*k unique rows while the plan
    lists P*k (< the LDS capacity here: 128 rows).  The plan stores the true count; such tiles take the by-id path for the
    rows the list does not hold (round-3 advisor finding: the count used to be clamped and the kernels read stale LDS)."""
    from deltaconv_amd._lib import lib
    from deltaconv_amd.geometry import Graph
    from deltaconv_amd.geometry.grad_div_mls import SparseOp
    torch.manual_seed(k)
    n = 1024
    pos = torch.randn(n, 3)
    idx = torch.arange(n)
    nbr = torch.stack([(idx + 37 * (s + 1) + (idx % 5)) % n for s in range(k)], 1).to(torch.int32)   # never the point itself
    assert not bool((nbr == idx[:, None]).any())
    ei = torch.stack([idx.repeat_interleave(k), nbr.reshape(-1).long()]).to(DEV)
    gr = Graph.from_edge_index(ei, n, k=k)
    gr.pos = pos.to(DEV)
    plan = gr.tile_plan(force_P=P)
    assert int(plan.section("nu").max()) > P * k                       # more unique rows than the list holds
    C = 64
    coef = _rand(n, k, 2)
    grad, div = SparseOp("grad", gr, coef), SparseOp("div", gr, coef.flip(2).contiguous())
    x, v = _rand(n, C), _rand(2 * n, C)
    a = plan.args
    _pair(lambda o: lib.call("dc_apply_grad", grad.coef, gr.nbr, n, k, x, C, C, o, C),
          lambda o: lib.call("dc_apply_grad_tiled", grad.coef, plan.blob, gr.nbr, *a, x, C, C, o, C),
          torch.empty(2 * n, C, device=DEV))
    _pair(lambda o: lib.call("dc_apply_div_curl_norm", div.coef, gr.nbr, n, k, v, C, C, o, 3 * C),
          lambda o: lib.call("dc_apply_div_curl_norm_tiled", div.coef, plan.blob, gr.nbr, *a, v, C, C, o, 3 * C),
          torch.empty(n, 3 * C, device=DEV))
    _pair(lambda o, ar: lib.call("dc_knn_max", gr.nbr, n, k, x, C, C, o, C, ar),
          lambda o, ar: lib.call("dc_knn_max_tiled", plan.blob, gr.nbr, *a, x, C, C, o, C, ar),
          torch.empty(n, C, device=DEV), torch.empty(n, C, dtype=torch.uint8, device=DEV))


@pytest.mark.parametrize("kind", ["cls", "seg"])
def test_model_identical_with_and_without_plan(kind):
    """Train-mode forward + backward of whole models: logits and every parameter gradient identical bit for bit with the
    forward applies running from the tile plan and through the gather path."""
    from deltaconv_amd.geometry import graph as G
    from deltaconv_amd.models import DeltaNetClassification, DeltaNetSegmentation
    from deltaconv_amd.utils import calc_loss
    b = synthetic_batch(3, 512, seed=12, per_point_labels=(kind == "seg"), num_classes=8 if kind == "seg" else 40).to(DEV)

    def run(use):
        import os
        G.USE_TILE_PLAN[0] = use
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
            G.USE_TILE_PLAN[0] = True
            os.environ.pop("DC_TILE_P", None)

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert torch.equal(o1, o0)
    assert len(g1) == len(g0) and all(torch.equal(a_, b_) for a_, b_ in zip(g1, g0))


def test_device_clock_stamps_of_the_tiled_applies():
    """dc_stamp_buffer (bench.py's in-step `roofline.frac`): armed, every tiled two-piece forward apply and tiled transposed
    apply takes the next record at enqueue time -- [earliest entry, latest exit] on a 100 MHz clock, tag = 1000 * kind +
    channels -- and computes the same bits as without stamps; disarmed, launches take no record."""
    from deltaconv_amd import _ops
    from deltaconv_amd._lib import lib
    b, gr, grad, div = _setup((1024,) * 8, 20)
    assert gr.tile_plan() is not None and gr.tile_plan_T() is not None
    n, C = gr.n, 64
    v = torch.randn(2 * n, C, device=DEV)
    ref = torch.empty(n, 3 * C, device=DEV)
    _ops.fwd_apply("div_curl_norm", div, v, C, C, ref, 3 * C)
    stamps = torch.zeros(8, 4, dtype=torch.int64, device=DEV)
    stamps[:, 0] = 2 ** 62
    assert lib.raw("dc_stamp_buffer")(stamps.data_ptr(), 8) == 0
    try:
        out = torch.empty_like(ref)
        _ops.fwd_apply("div_curl_norm", div, v, C, C, out, 3 * C)
        dcn = torch.randn(n, 3 * C, device=DEV)
        dv = torch.zeros(2 * n, C, device=DEV)
        _ops.bwd_div_curl_norm(div, dcn, C, 3 * C, v, C, dv, C, 0)
        used = lib.raw("dc_stamp_count")()
        tags = [lib.raw("dc_stamp_tag")(i) for i in range(used)]
    finally:
        lib.raw("dc_stamp_buffer")(None, 0)
    torch.cuda.synchronize()
    assert tags == [1064, 11064] and lib.raw("dc_stamp_tag")(5) == -1
    assert torch.equal(out, ref)
    rec = stamps.cpu()
    us = (rec[:2, 1] - rec[:2, 0]).double() * 1e-2
    assert bool((us > 1.0).all()) and bool((us < 2000.0).all()), us           # a few us each (first launches: cold)
    assert int(rec[2, 0]) == 2 ** 62 and int(rec[2, 1]) == 0                  # untouched records
    _ops.fwd_apply("div_curl_norm", div, v, C, C, out, 3 * C)                 # disarmed: no record taken, same bits
    assert lib.raw("dc_stamp_count")() == 0 and torch.equal(out, ref)


@pytest.mark.parametrize("plan", [True, False])
@pytest.mark.parametrize("kind,train", [("cls", True), ("seg2", True), ("cls", False)])
def test_max_aggregation_with_residual_block_epilogue_is_bit_identical(kind, train, plan):
    """Round 6: from the tile plan the max aggregation takes the layer's last s_mlp block (BatchNorm + activation + the residual
    add of deltaconv.py:59) into its epilogue (dc_knn_max_affine_residual_tiled; plan = False: the gather-path twin
    dc_knn_max_affine_residual, what small batches and k = 30 run).  Same addends as the two-launch form: logits,
    every gradient and every BatchNorm buffer identical bit for bit; depth-2 blocks (the part-segmentation net) and inference
    coefficients included."""
    import os
    from deltaconv_amd.models import DeltaNetClassification, DeltaNetSegmentation
    from deltaconv_amd.nn import layer as L
    from deltaconv_amd.utils import calc_loss
    seg = kind != "cls"
    b = synthetic_batch(3, 512, seed=13, per_point_labels=seg, num_classes=8 if seg else 40).to(DEV)

    def run(fuse):
        L.FUSE_MAX_RESIDUAL[0] = fuse
        if plan:
            os.environ["DC_TILE_P"] = "64"
        try:
            torch.manual_seed(5)
            m = (DeltaNetSegmentation(3, 8, mlp_depth=2) if seg else DeltaNetClassification(3, 40)).to(DEV).train(train)
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            if not train:
                with torch.no_grad():
                    return m(b).clone(), [], []
            out = m(b)
            calc_loss(out, b.y, smoothing=not seg).backward()
            return (out.detach(), [p.grad.clone() for p in m.parameters() if p.grad is not None],
                    [t.clone() for t in m.buffers()])
        finally:
            L.FUSE_MAX_RESIDUAL[0] = True
            os.environ.pop("DC_TILE_P", None)

    o1, g1, b1 = run(True)
    o0, g0, b0 = run(False)
    assert torch.equal(o1, o0)
    assert len(g1) == len(g0) and all(torch.equal(a_, b_) for a_, b_ in zip(g1, g0))
    assert all(torch.equal(a_, b_) for a_, b_ in zip(b1, b0))
