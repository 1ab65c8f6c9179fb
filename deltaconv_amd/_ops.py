"""torch.autograd wrappers around the C-ABI kernels (forward + hand-written backward each).

Operators (G, D) depend on geometry only and never require grad (reference: backward never
differentiates through build_grad_div), so every backward is a transposed
apply over the graph's CSC.
"""
import torch

from ._lib import lib


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


def copy_many(pairs):
    """dst.copy_(src) for every (src, dst) of `pairs` in ONE launch (csrc/optim.hip: dc_copy_many): 1-D / 2-D device tensors of
    equal shape and dtype (4- or 8-byte elements), unit stride along the last dimension, rows may be strided.  Pairs it
    does not take (other layouts / dtypes / devices) go through torch's copy, one launch each."""
    table = []
    for src, dst in pairs:
        es = src.element_size()
        ok = (src.is_cuda and dst.is_cuda and src.device == dst.device and src.dtype == dst.dtype and src.shape == dst.shape
              and src.dim() in (1, 2) and es in (4, 8) and src.numel() < (1 << 30)
              and (src.numel() == 0 or (src.stride(-1) == 1 and dst.stride(-1) == 1)))
        if not ok:
            dst.copy_(src, non_blocking=True)
            continue
        if src.numel() == 0:
            continue
        w = es // 4
        rows, cols = (1, src.shape[0]) if src.dim() == 1 else src.shape
        lds, ldd = (cols, cols) if src.dim() == 1 else (src.stride(0), dst.stride(0))
        if rows > 1 and (lds < cols or ldd < cols):
            dst.copy_(src, non_blocking=True)           # broadcast / overlapping rows
            continue
        table.append((src.data_ptr(), dst.data_ptr(), lds * w, ldd * w, rows, cols * w))
    if table:
        import ctypes
        n = len(table)
        i64, i32 = ctypes.c_int64 * n, ctypes.c_int32 * n
        col = lambda j: [e[j] for e in table]
        lib.call("dc_copy_many", i64(*col(0)), i64(*col(1)), i64(*col(2)), i64(*col(3)), i32(*col(4)), i32(*col(5)), n)


# ---- forward applies / max aggregation: from the graph's tile plan when it applies (neighbour rows in LDS,
# csrc/ell_tile.h), else through the gather path.  Same results bit for bit; `a`, `out` may be column blocks of wider
# buffers (leading dimensions lda / ldo).
def _tiled(graph, c, *tensors_lds):
    """The graph's TilePlan if the 16-byte tiled path applies to these operands, else None."""
    if c <= 0 or c % 64:
        return None
    for t, ld in tensors_lds:
        if ld % 4 or t.data_ptr() % 16:
            return None
    return graph.tile_plan()


def fwd_apply(name, op, a, c, lda, out, ldo):
    """name in {'grad', 'div', 'div_curl_norm', 'hodge'}: dc_apply_<name>[_tiled](op, a[., c] -> out)."""
    g = op.graph
    plan = _tiled(g, c, (a, lda), (out, ldo))
    if plan is not None:
        lib.call(f"dc_apply_{name}_tiled", op.coef, plan.blob, g.nbr, *plan.args, a, c, lda, out, ldo)
    else:
        lib.call(f"dc_apply_{name}", op.coef, g.nbr, g.n, g.k, a, c, lda, out, ldo)


def fwd_knn_max(g, h, c, ldh, out, ldo, arg, affine=None):
    """max over the k neighbours (+ first maximal slot); affine = (scale, shift, slope) folds BatchNorm + activation."""
    plan = _tiled(g, c, (h, ldh), (out, ldo), (arg, 4))
    if plan is not None:
        if affine is None:
            lib.call("dc_knn_max_tiled", plan.blob, g.nbr, *plan.args, h, c, ldh, out, ldo, arg)
        else:
            lib.call("dc_knn_max_affine_tiled", plan.blob, g.nbr, *plan.args, h, c, ldh, affine[0], affine[1], affine[2],
                     out, ldo, arg)
    elif affine is None:
        lib.call("dc_knn_max", g.nbr, g.n, g.k, h, c, ldh, out, ldo, arg)
    else:
        lib.call("dc_knn_max_affine", g.nbr, g.n, g.k, h, c, ldh, affine[0], affine[1], affine[2], out, ldo, arg)


def fwd_knn_max_residual(g, h, c, ldh, affine, h2, ldh2, affine2, out, ldo, out2, ldo2, arg):
    """out = act2(bn2(h2)) + max_j act(bn(h_j)) (+ second copy out2): the max aggregation with the layer's last s_mlp block in its
    epilogue (deltaconv.py:54-59): from the tile plan when it applies, else through the gather path.  Returns False only for an
    activation without a slope (the caller then runs the two steps)."""
    if affine2[2] is None:
        return False
    ops = [(h, ldh), (out, ldo), (arg, 4), (h2, ldh2), (affine2[0], 4), (affine2[1], 4)] + ([(out2, ldo2)] if out2 is not None else [])
    plan = _tiled(g, c, *ops)
    if plan is None:            # the gather-path twin: any C / alignment / k
        lib.call("dc_knn_max_affine_residual", g.nbr, g.n, g.k, h, c, ldh, affine[0], affine[1], affine[2], h2, ldh2, affine2[0],
                 affine2[1], float(affine2[2]), out, ldo, out2, ldo2, arg)
        return True
    lib.call("dc_knn_max_affine_residual_tiled", plan.blob, g.nbr, *plan.args, h, c, ldh, affine[0], affine[1], affine[2], h2, ldh2,
             affine2[0], affine2[1], float(affine2[2]), out, ldo, out2, ldo2, arg)
    return True


# ---- transposed applies / max-aggregation backward: from the graph's transposed tile plan when it applies (source rows in
# LDS, csrc/ell_tileT.h), else over the CSC through the gather path.  Same results bit for bit.
def _tiledT(graph, c, *tensors_lds):
    if c <= 0 or c % 64:
        return None
    for t, ld in tensors_lds:
        if t is not None and (ld % 4 or t.data_ptr() % 16):
            return None
    return graph.tile_plan_T()


def bwd_apply(name, op, dy, c, ldy, out, ldo, accumulate):
    """name in {'grad', 'div', 'hodge'}: dc_apply_<name>_T[_tiled](op^T, dy[., c]) (+)-> out."""
    g = op.graph
    plan = _tiledT(g, c, (dy, ldy), (out, ldo))
    if plan is not None:
        lib.call(f"dc_apply_{name}_T_tiled", op.coefTt(), plan.blob, *plan.args, dy, c, ldy, out, ldo, int(accumulate))
    else:
        tptr, tedge = g.csc()
        lib.call(f"dc_apply_{name}_T", op.coefT(), tptr, tedge, g.n, g.k, dy, c, ldy, out, ldo, int(accumulate))


def bwd_grad_sum(op, dy, c, ldy, a, lda, b, ldb, out, ldo):
    """out = a (+ b) + grad^T dy (dc_apply_grad_T_sum[_tiled])."""
    g = op.graph
    plan = _tiledT(g, c, (dy, ldy), (a, lda), (b, ldb), (out, ldo))
    if plan is not None:
        lib.call("dc_apply_grad_T_sum_tiled", op.coefTt(), plan.blob, *plan.args, dy, c, ldy, a, lda, b, ldb, out, ldo)
    else:
        tptr, tedge = g.csc()
        lib.call("dc_apply_grad_T_sum", op.coefT(), tptr, tedge, g.n, g.k, dy, c, ldy, a, lda, b, ldb, out, ldo)


def bwd_div_curl_norm(op, dout, c, ldo, v, ldv, dv, lddv, accumulate):
    g = op.graph
    plan = _tiledT(g, c, (dout, ldo), (v, ldv), (dv, lddv))
    if plan is not None:
        lib.call("dc_apply_div_curl_norm_T_tiled", op.coefTt(), plan.blob, *plan.args, dout, c, ldo, v, ldv, dv, lddv,
                 int(accumulate))
    else:
        tptr, tedge = g.csc()
        lib.call("dc_apply_div_curl_norm_T", op.coefT(), tptr, tedge, g.n, g.k, dout, c, ldo, v, ldv, dv, lddv,
                 int(accumulate))


def bwd_knn_max(g, arg, dout, c, ldo, dh, ldh, accumulate=0):
    plan = _tiledT(g, c, (dout, ldo), (dh, ldh), (arg, 4))
    if plan is not None:
        lib.call("dc_knn_max_backward_tiled", plan.blob, *plan.args, arg, dout, c, ldo, dh, ldh, int(accumulate))
    else:
        tptr, tedge = g.csc()
        lib.call("dc_knn_max_backward", tptr, tedge, g.n, g.k, arg, dout, c, ldo, dh, ldh, int(accumulate))


class _Apply(torch.autograd.Function):
    """kind in {'grad','div'}: y = A @ x with A in ELL form."""

    @staticmethod
    def forward(ctx, x, op, graph, kind):
        coef = op.coef
        x = _f32c(x)
        n, k, c = graph.n, graph.k, x.shape[1]
        ctx.graph, ctx.kind, ctx.op = graph, kind, op
        if kind == 'grad':
            assert x.shape[0] == n, f"grad @ x: x has {x.shape[0]} rows, graph has {n} points"
            out = torch.empty(2 * n, c, dtype=torch.float32, device=x.device)
            fwd_apply("grad", op, x, c, c, out, c)
        else:
            assert x.shape[0] == 2 * n, f"div @ v: v has {x.shape[0]} rows, graph has {n} points"
            out = torch.empty(n, c, dtype=torch.float32, device=x.device)
            fwd_apply("div", op, x, c, c, out, c)
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        g, c = ctx.graph, dy.shape[1]
        if ctx.kind == 'grad':
            dx = torch.empty(g.n, c, dtype=torch.float32, device=dy.device)
            bwd_apply("grad", ctx.op, dy, c, c, dx, c, 0)
        else:
            dx = torch.empty(2 * g.n, c, dtype=torch.float32, device=dy.device)
            bwd_apply("div", ctx.op, dy, c, c, dx, c, 0)
        return dx, None, None, None


class _DivCurlNorm(torch.autograd.Function):
    """v[2Nt,C] -> [div v | curl v | norm v] [Nt,3C] in one gather pass."""

    @staticmethod
    def forward(ctx, v, op, graph):
        v = _f32c(v)
        coef = op.coef
        n, k, c = graph.n, graph.k, v.shape[1]
        assert v.shape[0] == 2 * n
        out = torch.empty(n, 3 * c, dtype=torch.float32, device=v.device)
        fwd_apply("div_curl_norm", op, v, c, c, out, 3 * c)
        ctx.graph, ctx.op = graph, op
        ctx.save_for_backward(v)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32c(dout)
        (v,) = ctx.saved_tensors
        g, c = ctx.graph, v.shape[1]
        dv = torch.empty_like(v)
        bwd_div_curl_norm(ctx.op, dout, c, 3 * c, v, c, dv, c, 0)
        return dv, None, None


class _Hodge(torch.autograd.Function):
    """dcn[Nt, >=2C] holding [div v | curl v | ...] -> hodge_laplacian(v) [2Nt,C]."""

    @staticmethod
    def forward(ctx, dcn, op, graph, c):
        dcn = _f32c(dcn)
        coef = op.coef
        n, k = graph.n, graph.k
        ld = dcn.shape[1]
        assert dcn.shape[0] == n and ld >= 2 * c
        out = torch.empty(2 * n, c, dtype=torch.float32, device=dcn.device)
        fwd_apply("hodge", op, dcn, c, ld, out, c)
        ctx.graph, ctx.op, ctx.c, ctx.ld = graph, op, c, ld
        return out

    @staticmethod
    def backward(ctx, dh):
        dh = _f32c(dh)
        g, c, ld = ctx.graph, ctx.c, ctx.ld
        ddcn = torch.zeros(g.n, ld, dtype=torch.float32, device=dh.device) if ld > 2 * c else \
            torch.empty(g.n, ld, dtype=torch.float32, device=dh.device)
        bwd_apply("hodge", ctx.op, dh, c, c, ddcn, ld, 0)
        return ddcn, None, None, None


class _KnnMax(torch.autograd.Function):
    """out[i,c] = max over the k neighbours of h[.,c]; first maximal slot takes the gradient."""

    @staticmethod
    def forward(ctx, h, graph):
        h = _f32c(h)
        n, k, c = graph.n, graph.k, h.shape[1]
        assert h.shape[0] == n
        out = torch.empty(n, c, dtype=torch.float32, device=h.device)
        arg = torch.empty(n, c, dtype=torch.uint8, device=h.device)
        fwd_knn_max(graph, h, c, c, out, c, arg)
        ctx.graph = graph
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, dout, _darg):
        dout = _f32c(dout)
        (arg,) = ctx.saved_tensors
        g, c = ctx.graph, dout.shape[1]
        dh = torch.empty(g.n, c, dtype=torch.float32, device=dout.device)
        bwd_knn_max(g, arg, dout, c, c, dh, c, 0)
        return dh, None


class _KnnSum(torch.autograd.Function):
    """out[i,c] = scale * sum over the k neighbours of h[.,c] (aggr = 'sum' / 'add' / 'mean'); slots in order,
    backward over the CSC in ascending edge order."""

    @staticmethod
    def forward(ctx, h, graph, scale):
        h = _f32c(h)
        n, k, c = graph.n, graph.k, h.shape[1]
        assert h.shape[0] == n
        out = torch.empty(n, c, dtype=torch.float32, device=h.device)
        lib.call("dc_knn_sum", graph.nbr, n, k, h, c, c, float(scale), out, c)
        ctx.graph, ctx.scale = graph, float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32c(dout)
        g, c = ctx.graph, dout.shape[1]
        tptr, tedge = g.csc()
        dh = torch.empty(g.n, c, dtype=torch.float32, device=dout.device)
        lib.call("dc_knn_sum_backward", tptr, tedge, g.n, g.k, dout, c, c, ctx.scale, dh, c, 0)
        return dh, None, None


AGGREGATIONS = ("max", "min", "sum", "add", "mean")      # torch_scatter reduce names DeltaConv(aggr=...) accepts


def knn_aggregate(h, graph, aggr):
    """torch_scatter.scatter(h[col], row, dim=0, reduce=aggr) over the kNN graph (nn/deltaconv.py:52,54)."""
    if aggr == "max":
        return knn_max(h, graph)
    if aggr == "min":
        return -knn_max(-h, graph)
    if aggr in ("sum", "add"):
        return _KnnSum.apply(h, graph, 1.0)
    if aggr == "mean":
        return _KnnSum.apply(h, graph, 1.0 / graph.k)
    raise ValueError(f"aggr must be one of {AGGREGATIONS}, got {aggr!r}")


def apply_op(x, op):
    return _Apply.apply(x, op, op.graph, op.kind)


def div_curl_norm(v, div):
    return _DivCurlNorm.apply(v, div, div.graph)


def hodge_from_dcn(dcn, grad, c):
    return _Hodge.apply(dcn, grad, grad.graph, c)


def knn_max(h, graph):
    return _KnnMax.apply(h, graph)[0]
