"""Fused blocks of the scalar / vector MLP stream: hand-written fp32-MFMA products (deltaconv_amd/csrc/gemm.hip, with
the BatchNorm statistics in their epilogue and the BatchNorm backward in their operand loaders) + hand-written HIP
BatchNorm / activation / vector non-linearity kernels (deltaconv_amd/csrc/nn.hip), each with its own backward; the centralised
edge MLPs of the first layer (csrc/edge.hip, edge2.hip); blocks on a handful of rows (the classification head) on
csrc/rowblock.hip.  No vendor-library GEMM is reachable for fp32 GPU inputs.

Reference semantics: deltaconv/nn/mlp.py:7-17 and nn/nonlin.py:11-86 (Linear(no bias) ->
BatchNorm1d over rows -> LeakyReLU(0.2);  Linear(no bias) -> VectorNonLin(BatchNorm1d))."""
import contextlib
import ctypes
import os
import threading

import torch
import torch.nn.functional as F

from .._lib import lib, require_gpu


def _c(t):
    return t if (t is None or (t.dtype == torch.float32 and t.is_contiguous())) else t.contiguous().float()


def _ws(rows, c, device):
    nbytes = lib.raw("dc_bn_workspace_bytes")(rows, c)
    return torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device), nbytes


_PENDING, _DEFER = [], [0]


def bump_counter(bn):
    """num_batches_tracked += 1; inside ``defer_counters()`` the increments of a whole forward pass
    are applied by one multi-tensor launch instead of one tiny kernel per BatchNorm."""
    # the statistics kernels write the running buffers through raw pointers (no autograd version bump): drop the
    # cached inference coefficients of this layer (eval_coeffs) here, the one place every training-mode pass goes through
    if getattr(bn.running_mean, "_dc_eval_coeffs", None) is not None:
        del bn.running_mean._dc_eval_coeffs
    if _DEFER[0] and bn.momentum is not None:
        _PENDING.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


def flush_counters():
    if _PENDING:
        torch._foreach_add_(_PENDING, 1)
        _PENDING.clear()


class defer_counters:
    def __enter__(self):
        _DEFER[0] += 1

    def __exit__(self, *exc):
        _DEFER[0] -= 1
        if _DEFER[0] == 0:
            flush_counters()
        return False


def sync_group():
    """Process group over which BatchNorm statistics are synchronised (deltaconv_amd/dp.py), or None."""
    from .. import dp
    return dp.sync_state()


class BatchStats(int):
    """The `use_batch_stats` flag of a block (truthy, int() == 1) that also remembers the data-parallel group its FORWARD
    statistics were reduced over: the backward pass reduces over the same group whatever `dp.set_sync_bn` says by then
    (advisor, round 5: a toggle between forward and backward gave global statistics forward and local means backward)."""

    def __new__(cls, group):
        self = int.__new__(cls, 1)
        self.group = group
        return self


def group_of(use):
    """Group of a block's forward statistics: carried by the flag (BatchStats), else the current state (plain True)."""
    if not use:
        return None
    return use.group if isinstance(use, BatchStats) else sync_group()


def _sync_stats(kernel_args, c, rows, dev, group):
    """Local fp64 sums via a dc_*_sums entry point -> all-reduced over `group`.  The kernels write TWO identical records
    [sum_0 | sum_1 | rows] (csrc/colreduce.h: SumsFin): the first is all-reduced in place, the second stays this rank's own --
    no fill, no clone launch.  Returns (global record, local record), views of one buffer.  (`rows` is written by the kernel.)"""
    from .. import dp
    buf = torch.empty(2, 2 * c + 1, dtype=torch.float64, device=dev)
    name, args = kernel_args
    lib.call(name, *args(buf))
    dp.all_reduce_stats(buf[0], group)
    return buf[0], buf[1]


_EVAL_EPOCH = [0]


def invalidate_eval_coeffs():
    """Drop every cached inference-mode BatchNorm map.  Needed whenever parameters / running statistics change through
    raw pointers, i.e. without a tensor version bump: a HIP-graph replay of a training step (graph_step.py)."""
    _EVAL_EPOCH[0] += 1
    _PL["epoch"] += 1        # the pre-split weight planes as well: a replayed optimizer update moved the weights under them


def eval_coeffs(gamma, beta, rm, rv, eps, c):
    """-> coef[4, c] = (mean, invstd, scale, shift) of an inference-mode BatchNorm: the layer folded to ONE per-channel
    affine map (scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale) that the CONSUMING kernel
    applies (max-aggregation gather, activation + residual, pooling) -- in eval mode no BatchNorm pass exists.  The
    rows depend only on the checkpoint: under torch.no_grad() they are computed once and kept on the running-mean
    buffer until any of the tensors changes (in-place update = version bump, .to(device) / load = new storage, graph
    replay of a training step = invalidate_eval_coeffs(): a replay writes through raw pointers, no version moves)."""
    key = (_EVAL_EPOCH[0], rm._version, rv._version, rm.data_ptr(), rv.data_ptr(), float(eps),
           None if gamma is None else (gamma._version, gamma.data_ptr()),
           None if beta is None else (beta._version, beta.data_ptr()))
    cacheable = not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()
    hit = getattr(rm, "_dc_eval_coeffs", None) if cacheable else None
    if hit is not None and hit[0] == key:
        return hit[1]
    coef = torch.empty(4, c, dtype=torch.float32, device=rm.device)
    lib.call("dc_bn_eval_coeffs", gamma, beta, rm, rv, eps, c, coef[0], coef[1], coef[2], coef[3])
    if cacheable:
        rm._dc_eval_coeffs = (key, coef)
    return coef


def check_bn_rows(bn, rows):
    """torch.nn.functional.batch_norm refuses a train-mode batch with one value per channel (the reference
    hits this at B = 1 in its head: deltaconv/nn/nonlin.py:29-30, SURVEY.md section 8(d) C1 caveat); same here."""
    if bn.training and int(rows) <= 1:
        group = sync_group()
        if group is not None:
            # synchronised statistics span the GLOBAL batch: a rank may hold a single row (one cloud per rank in the
            # categorical head of the segmentation net, deltaconv_amd/dp.py); the all-reduced count is on the device,
            # so only the case that is decidable without a host sync is refused here
            import torch.distributed as dist
            if dist.get_world_size(group) > 1:
                return
        raise ValueError(f"Expected more than 1 value per channel when training, got input size [1, "
                         f"{bn.num_features}, {int(rows)}]")


def slope_of(act):
    """negative slope of a piecewise-linear activation module, or None if it is something else."""
    if isinstance(act, torch.nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, torch.nn.ReLU):
        return 0.0
    if isinstance(act, torch.nn.Identity):
        return 1.0
    return None


class _BNAct(torch.autograd.Function):
    """y = leaky_slope(batch_norm(h)) (+ residual) on an [R,C] matrix."""

    @staticmethod
    def forward(ctx, h, gamma, beta, rm, rv, use_batch_stats, momentum, eps, slope, residual):
        h = _c(h)
        r, c = h.shape
        dev = h.device
        coef = torch.empty(4, c, dtype=torch.float32, device=dev)     # mean, invstd, scale, shift
        ctx.group = sync_group() if use_batch_stats else None
        if ctx.group is not None:                                     # statistics of the global batch
            ws, nb = _ws(r, c, dev)
            stats, _ = _sync_stats(("dc_bn_sums", lambda out: (h, r, c, c, out, ws, nb)), c, r, dev, ctx.group)
            lib.call("dc_bn_coeffs_from_sums", stats, 0, c, gamma, beta, eps, momentum, rm, rv, coef[0], coef[1],
                     coef[2], coef[3])
        elif use_batch_stats:
            ws, nb = _ws(r, c, dev)
            lib.call("dc_bn_stats", h, r, c, c, gamma, beta, eps, momentum, rm, rv, coef[0], coef[1], coef[2],
                     coef[3], ws, nb)
        else:
            coef = eval_coeffs(gamma, beta, rm, rv, eps, c)
        y = torch.empty_like(h)
        res = _c(residual)
        lib.call("dc_bn_act", h, r, c, c, coef[2], coef[3], slope, res, c, y, c)
        ctx.save_for_backward(h, coef, gamma)
        ctx.cfg = (use_batch_stats, slope, gamma is not None, beta is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, coef, gamma = ctx.saved_tensors
        training, slope, has_g, has_b, has_res = ctx.cfg
        dy = _c(dy)
        r, c = h.shape
        dh = torch.empty_like(h)
        dgamma = torch.empty(c, dtype=torch.float32, device=h.device) if has_g else None
        dbeta = torch.empty(c, dtype=torch.float32, device=h.device) if has_b else None
        ws, nb = _ws(r, c, h.device)
        if ctx.group is not None:
            stats, _ = _sync_stats(("dc_bn_act_backward_sums", lambda out: (dy, c, h, c, r, c, coef[2], coef[3], coef[0],
                                                                         coef[1], slope, out, ws, nb)), c, r, h.device, ctx.group)
            m = torch.empty(2, c, dtype=torch.float32, device=h.device)
            lib.call("dc_sync_means", stats, c, m[0], m[1], dgamma, dbeta)      # global means + this rank's dgamma / dbeta
            lib.call("dc_bn_act_backward_apply", dy, c, h, c, r, c, coef[2], coef[3], coef[0], coef[1], gamma, slope,
                     int(training), m[0], m[1], dh, c)
        else:
            lib.call("dc_bn_act_backward", dy, c, h, c, r, c, coef[2], coef[3], coef[0], coef[1], gamma, slope,
                     int(training), dh, c, dgamma, dbeta, ws, nb)
        return dh, dgamma, dbeta, None, None, None, None, None, None, (dy if has_res else None)


def bn_act(h, bn, slope, residual=None):
    """bn: torch.nn.BatchNorm1d holding the parameters / running statistics."""
    require_gpu()
    check_bn_rows(bn, h.shape[0])
    use_batch = bn.training or bn.running_mean is None
    mom = 0.0 if bn.momentum is None else float(bn.momentum)
    rm, rv = (bn.running_mean, bn.running_var) if (bn.training and bn.track_running_stats) or not use_batch else (None, None)
    if bn.training and bn.track_running_stats:
        bump_counter(bn)
        if bn.momentum is None:
            mom = 1.0 / float(bn.num_batches_tracked)
    return _BNAct.apply(h, bn.weight, bn.bias, rm, rv, use_batch, mom, float(bn.eps), float(slope), residual)


class _VectorNonLin(torch.autograd.Function):
    """out = y * relu(scale*|y| + shift) / max(|y|, 1e-8); `inp` = y [2n,co] or [P|Q] [2n,2co]."""

    @staticmethod
    def forward(ctx, inp, combine, gamma, beta, rm, rv, mode, momentum, eps):
        # mode: 2 = batch statistics, 1 = running statistics, 0 = no batch norm (shift = beta = bias)
        inp = _c(inp)
        ld = inp.shape[1]
        co = ld // 2 if combine else ld
        n = inp.shape[0] // 2
        dev = inp.device
        coef = torch.empty(4, co, dtype=torch.float32, device=dev)
        ctx.group = sync_group() if mode == 2 else None
        if ctx.group is not None:
            ws, nb = _ws(n, co, dev)
            stats, _ = _sync_stats(("dc_vn_sums", lambda out: (inp, n, co, ld, int(combine), out, ws, nb)), co, n, dev,
                                   ctx.group)
            lib.call("dc_bn_coeffs_from_sums", stats, 0, co, gamma, beta, eps, momentum, rm, rv, coef[0], coef[1],
                     coef[2], coef[3])
        elif mode == 2:
            ws, nb = _ws(n, co, dev)
            lib.call("dc_vn_stats", inp, n, co, ld, int(combine), gamma, beta, eps, momentum, rm, rv, coef[0],
                     coef[1], coef[2], coef[3], ws, nb)
        elif mode == 1:
            coef = eval_coeffs(gamma, beta, rm, rv, eps, co)
        else:
            coef[0].zero_(); coef[1].fill_(1.0); coef[2].fill_(1.0); coef[3].copy_(beta)
        out = torch.empty(2 * n, co, dtype=torch.float32, device=dev)
        lib.call("dc_vn_apply", inp, n, co, ld, int(combine), coef[2], coef[3], out, co)
        ctx.save_for_backward(inp, coef, gamma)
        ctx.cfg = (int(combine), mode, gamma is not None, beta is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        inp, coef, gamma = ctx.saved_tensors
        combine, mode, has_g, has_b = ctx.cfg
        dout = _c(dout)
        ld = inp.shape[1]
        co = ld // 2 if combine else ld
        n = inp.shape[0] // 2
        dev = inp.device
        din = torch.empty_like(inp)
        dgamma = torch.empty(co, dtype=torch.float32, device=dev) if has_g else None
        dbeta = torch.empty(co, dtype=torch.float32, device=dev) if has_b else None
        ws, nb = _ws(n, co, dev)
        if ctx.group is not None:
            stats, _ = _sync_stats(("dc_vn_backward_sums", lambda out: (dout, co, inp, ld, combine, n, co, coef[2], coef[3],
                                                                     coef[0], coef[1], out, ws, nb)), co, n, dev, ctx.group)
            m = torch.empty(2, co, dtype=torch.float32, device=dev)
            lib.call("dc_sync_means", stats, co, m[0], m[1], dgamma, dbeta)
            lib.call("dc_vn_backward_apply", dout, co, inp, ld, combine, n, co, coef[2], coef[3], coef[0], coef[1], gamma,
                     1, m[0], m[1], din, ld)
        else:
            lib.call("dc_vn_backward", dout, co, inp, ld, combine, n, co, coef[2], coef[3], coef[0], coef[1],
                     gamma if mode else None, int(mode == 2), din, ld, dgamma, dbeta, ws, nb)
        return din, None, dgamma, dbeta, None, None, None, None, None


def vector_nonlin(inp, combine, vn):
    """vn: VectorNonLin module (bias, optional batchnorm wrapper with .bn)."""
    require_gpu()
    if vn.batchnorm is None:
        return _VectorNonLin.apply(inp, combine, None, vn.bias, None, None, 0, 0.0, 0.0)
    bn = vn.batchnorm.bn
    check_bn_rows(bn, inp.shape[0] // 2)
    use_batch = bn.training or bn.running_mean is None
    mom = 0.0 if bn.momentum is None else float(bn.momentum)
    track = bn.training and bn.track_running_stats
    if track:
        bump_counter(bn)
        if bn.momentum is None:
            mom = 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if (track or not use_batch) else (None, None)
    return _VectorNonLin.apply(inp, combine, bn.weight, bn.bias, rm, rv, 2 if use_batch else 1, mom, float(bn.eps))


class _EdgeMaxBN(torch.autograd.Function):
    """x_max[i] = max_s leaky(bn(y_j - y_i)) over the k-list of i, BN statistics over all E edges,
    without materialising any [E,C] tensor (csrc/edge_math.h)."""

    @staticmethod
    def forward(ctx, y, graph, gamma, beta, rm, rv, use_batch_stats, momentum, eps, slope):
        y = _c(y)
        n, c = y.shape
        k, dev = graph.k, y.device
        f32 = dict(dtype=torch.float32, device=dev)
        stat = torch.empty(3, n, c, **f32)                   # amax, amin, s1pt
        args = torch.empty(2, n, c, dtype=torch.uint8, device=dev)
        coef = torch.empty(4, c, **f32)                      # mean, invstd, scale, shift
        ws, nb = _ws(n, c, dev)
        if not use_batch_stats:
            coef = eval_coeffs(gamma, beta, rm, rv, eps, c)
        lib.call("dc_edge_gather_stats", y, c, graph.nbr, n, k, c, int(use_batch_stats), gamma, beta, eps, momentum,
                 rm if use_batch_stats else None, rv if use_batch_stats else None, stat[0], stat[1], args[0], args[1],
                 stat[2], coef[0], coef[1], coef[2], coef[3], ws, nb)
        out = torch.empty(n, c, **f32)
        lib.call("dc_edge_max_apply", stat[0], stat[1], args[0], args[1], n, c, coef[2], coef[3], slope, out, c, None)
        ctx.save_for_backward(y, stat, args, coef)
        ctx.graph, ctx.cfg = graph, (use_batch_stats, slope, gamma is not None, beta is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, stat, args, coef = ctx.saved_tensors
        training, slope, has_g, has_b = ctx.cfg
        g = ctx.graph
        dout = _c(dout)
        n, c = y.shape
        dev = y.device
        tptr, tedge = g.csc()
        dzs = torch.empty(n, c, dtype=torch.float32, device=dev)
        dy = torch.empty(n, c, dtype=torch.float32, device=dev)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev) if has_g else None
        dbeta = torch.empty(c, dtype=torch.float32, device=dev) if has_b else None
        ws, nb = _ws(n, c, dev)
        lib.call("dc_edge_max_backward", dout, c, y, c, tptr, tedge, n, g.k, c, stat[0], stat[1], args[0], args[1],
                 stat[2], coef[2], coef[3], coef[0], coef[1], slope, int(training), dzs, dy, c, dgamma, dbeta, ws, nb)
        return dy, None, dgamma, dbeta, None, None, None, None, None, None


def edge_max_bn(y, graph, bn, slope):
    """bn: torch.nn.BatchNorm1d of the (single) s_mlp_max block; y = Linear(x)."""
    require_gpu()
    use_batch = bn.training or bn.running_mean is None
    mom = 0.0 if bn.momentum is None else float(bn.momentum)
    track = bn.training and bn.track_running_stats
    if track:
        bump_counter(bn)
        if bn.momentum is None:
            mom = 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if (track or not use_batch) else (None, None)
    return _EdgeMaxBN.apply(y, graph, bn.weight, bn.bias, rm, rv, use_batch, mom, float(bn.eps), float(slope))


class _EdgeDiff(torch.autograd.Function):
    """x_edge[e] = x[nbr[e]] - x[e // k]  (deltaconv/nn/deltaconv.py:50) as an [n k, C] tensor: the general (materialised)
    form of the centralised edge MLP; its transpose sums the in-edges of a point in ascending edge id (no atomics)."""

    @staticmethod
    def forward(ctx, x, graph):
        x = _c(x)
        n, k, c = graph.n, graph.k, x.shape[1]
        out = torch.empty(n * k, c, dtype=torch.float32, device=x.device)
        lib.call("dc_edge_diff", x, c, graph.nbr, n, k, c, out)
        ctx.graph, ctx.c = graph, c
        return out

    @staticmethod
    def backward(ctx, d_edge):
        g, c = ctx.graph, ctx.c
        d_edge = _c(d_edge)
        tptr, tedge = g.csc()
        dx = torch.empty(g.n, c, dtype=torch.float32, device=d_edge.device)
        lib.call("dc_edge_diff_backward", d_edge, tptr, tedge, g.n, g.k, c, dx, c)
        return dx, None


def edge_diff(x, graph):
    require_gpu()
    return _EdgeDiff.apply(x, graph)


_SEG_MODES = {"max": 0, "min": 1, "sum": 2, "add": 2, "mean": 3}


class _SegReduce(torch.autograd.Function):
    """scatter(h, row, reduce=aggr) for centre-major edges (deltaconv.py:52): reduction over the k consecutive rows of every
    point; max / min keep the first extremal slot (uint8), the backward writes the [n k, C] gradient in one pass."""

    @staticmethod
    def forward(ctx, h, n, k, mode):
        h = _c(h)
        c = h.shape[1]
        out = torch.empty(n, c, dtype=torch.float32, device=h.device)
        arg = torch.empty(n, c, dtype=torch.uint8, device=h.device) if mode < 2 else None
        lib.call("dc_seg_reduce", h, n, k, c, mode, out, arg)
        ctx.cfg = (n, k, c, mode)
        ctx.arg = arg
        ctx.save_for_backward(*(() if arg is None else (arg,)))
        if arg is None:
            arg = torch.empty(0, dtype=torch.uint8, device=h.device)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, dout, _darg):
        n, k, c, mode = ctx.cfg
        dout = _c(dout)
        arg = ctx.saved_tensors[0] if mode < 2 else None
        dh = torch.empty(n * k, c, dtype=torch.float32, device=dout.device)
        lib.call("dc_seg_reduce_backward", dout, c, arg, n, k, c, mode, dh)
        return dh, None, None, None


def seg_reduce(h, n, k, aggr):
    """-> (reduced [n, C], first extremal slot uint8 [n, C] for 'max' / 'min', else an empty tensor)."""
    require_gpu()
    return _SegReduce.apply(h, n, k, _SEG_MODES[aggr])


USE_EDGE2 = True    # A/B switch: False = the materialised [E, C] path for the depth-2 centralised edge MLP
EDGE2_DIRECT = True # A/B switch: False = BatchNorm-1 statistics and closed forms through rows of z = x W1^T even for ci <= 3
EDGE2_CH = 64       # csrc/edge2.hip is specialised to 64 channels in both blocks (the part-segmentation net's first layer)


class _EdgeMLP2(torch.autograd.Function):
    """x_max[i] = max_s act2(bn2(W2 act1(bn1(W1 (x_j - x_i))))) -- the depth-2 centralised edge MLP of the first layer
    (deltaconv/nn/deltaconv.py:50-52 with mlp_depth = 2) on csrc/edge2.hip: no [E, C] tensor in the forward pass, one
    recompute pass + a CSC closing pass backward.  Bit-reproducible.  ci <= 3 (positions): everything from the input channels
    themselves -- no z = x W1^T at all, BatchNorm-1 from the moments of the edge differences; otherwise through rows of z."""

    @staticmethod
    def forward(ctx, x, graph, W1, g1, b1, W2, g2, b2, mode1, mode2, slope1, slope2):
        # mode = (use_batch_stats, momentum, running_mean, running_var, eps)
        n, k, dev, c = graph.n, graph.k, x.device, EDGE2_CH
        ci = x.shape[1]
        f32 = dict(dtype=torch.float32, device=dev)
        use1, mom1, rm1, rv1, eps1 = mode1
        use2, mom2, rm2, rv2, eps2 = mode2
        direct = ci <= 3 and EDGE2_DIRECT
        nb2 = lib.raw("dc_edge2_workspace_bytes")(n, k, 0)
        ws2 = torch.empty((nb2 + 7) // 8, dtype=torch.float64, device=dev)
        coef1 = torch.empty(4, c, **f32) if use1 else eval_coeffs(g1, b1, rm1, rv1, eps1, c)
        z = s1 = None
        if direct:
            if use1:                                               # BatchNorm-1 from the moments of x_j - x_i
                s1 = torch.empty(n, ci, **f32)
                lib.call("dc_edge2_bn1_stats", x, x.stride(0), ci, W1, graph.nbr, n, k, g1, b1, eps1, mom1, rm1, rv1, s1, coef1[0],
                         coef1[1], coef1[2], coef1[3], ws2, nb2)
        else:
            z = mm_nt(x, W1)                                       # [n, 64]: W1 (x_j - x_i) = z_j - z_i
            stat = torch.empty(3, n, c, **f32)                     # amax, amin (unused here), s1pt
            args = torch.empty(2, n, c, dtype=torch.uint8, device=dev)
            ws, nb = _ws(n, c, dev)
            lib.call("dc_edge_gather_stats", z, c, graph.nbr, n, k, c, int(use1), g1, b1, eps1, mom1, rm1 if use1 else None,
                     rv1 if use1 else None, stat[0], stat[1], args[0], args[1], stat[2], coef1[0], coef1[1], coef1[2], coef1[3],
                     ws, nb)
            s1 = stat[2]
        coef2 = torch.empty(4, c, **f32) if use2 else eval_coeffs(g2, b2, rm2, rv2, eps2, c)
        ysel = torch.empty(n, c, **f32)
        arg = torch.empty(n, c, dtype=torch.uint8, device=dev)
        lib.call("dc_edge2_forward", z, x, x.stride(0), ci, W1, graph.nbr, n, k, W2, coef1[2], coef1[3], slope1, int(use2), g2, b2,
                 eps2, mom2, rm2 if use2 else None, rv2 if use2 else None, ysel, arg, coef2[0], coef2[1], coef2[2], coef2[3], None,
                 ws2, nb2)
        out = torch.empty(n, c, **f32)
        lib.call("dc_bn_act", ysel, n, c, c, coef2[2], coef2[3], slope2, None, 0, out, c)
        ctx.save_for_backward(x, z, s1, coef1, coef2, ysel, arg, W1, W2, g2)
        ctx.graph, ctx.cfg = graph, (use1, use2, slope1, slope2, g1 is not None, b1 is not None, g2 is not None, b2 is not None)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, dout, _darg):
        x, z, s1, coef1, coef2, ysel, arg, W1, W2, g2 = ctx.saved_tensors
        use1, use2, slope1, slope2, hg1, hb1, hg2, hb2 = ctx.cfg
        g = ctx.graph
        n, k, dev, c = g.n, g.k, x.device, EDGE2_CH
        f32 = dict(dtype=torch.float32, device=dev)
        dout = _c(dout)
        tptr, tedge = g.csc()
        dz = torch.empty(n, c, **f32)
        dW2 = torch.empty(c, c, **f32)
        dg1, db1, dg2, db2 = (torch.empty(c, **f32) for _ in range(4))
        nb = lib.raw("dc_edge2_workspace_bytes")(n, k, 1)
        ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
        lib.call("dc_edge2_backward", dout, c, z, x, x.stride(0), x.shape[1], W1, g.nbr, tptr, tedge, n, k, W2, coef1, coef2, g2,
                 slope1, slope2, int(use1), int(use2), ysel, arg, s1, dz, c, dW2, dg1, db1, dg2, db2, ws, nb)
        dW1 = gemm_tn(dz, x if x.stride(1) == 1 else x.contiguous()) if ctx.needs_input_grad[2] else None
        dx = mm_nn(dz, W1) if ctx.needs_input_grad[0] else None
        return (dx, None, dW1, dg1 if hg1 else None, db1 if hb1 else None, dW2, dg2 if hg2 else None, db2 if hb2 else None,
                None, None, None, None)


def _bn_mode_of(bn):
    use_batch = bn.training or bn.running_mean is None
    mom = 0.0 if bn.momentum is None else float(bn.momentum)
    track = bn.training and bn.track_running_stats
    if track:
        bump_counter(bn)
        if bn.momentum is None:
            mom = 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if (track or not use_batch) else (None, None)
    return use_batch, mom, rm, rv, float(bn.eps)


def edge_mlp2_ok(x, lin1, bn1, lin2, bn2, slope1, slope2):
    """The shapes csrc/edge2.hip is specialised for: two bias-free blocks of 64 channels, a monotone second activation."""
    return (USE_EDGE2 and x.is_cuda and x.dtype == torch.float32 and lin1.bias is None and lin2.bias is None
            and lin1.weight.shape[0] == EDGE2_CH and tuple(lin2.weight.shape) == (EDGE2_CH, EDGE2_CH)
            and slope1 is not None and slope2 is not None and slope2 >= 0 and sync_group() is None)


def edge_mlp2(x, graph, lin1, bn1, slope1, lin2, bn2, slope2):
    """bn1 / bn2: torch.nn.BatchNorm1d of the two blocks; returns (x_max [n, 64], selected slots uint8 [n, 64])."""
    require_gpu()
    check_bn_rows(bn1, graph.n * graph.k)
    m1, m2 = _bn_mode_of(bn1), _bn_mode_of(bn2)
    return _EdgeMLP2.apply(_c(x), graph, lin1.weight, bn1.weight, bn1.bias, _c(lin2.weight), bn2.weight, bn2.bias, m1, m2,
                           float(slope1), float(slope2))


# ---- pre-split weight planes (csrc/gemm.hip: dc_presplit_weights, BP kernels) -----------------------------------------------
# The split products cut every fp32 operand into three bf16 planes; for a WEIGHT matrix that work is identical in every
# workgroup of every product of a step.  `presplit_begin()` (called at the start of a model forward) cuts all weights that
# products have asked for -- as stored [N, K] for the forward products and transposed [K, N] for the input gradients -- in ONE
# launch; `_hint_planes(w, transposed)` in front of a product hands the library the planes of its weight operand.  Same bits
# as the in-loop split.  A weight the cache does not know yet is registered at its first use and served from the next step on
# (never during a graph capture: registration copies a table to the device); planes are used only when they were cut after
# the last modification of the weight (version counter + step epoch), else the product splits in its K loop as before.
USE_WEIGHT_PLANES = os.environ.get("DC_WEIGHT_PLANES", "1") != "0"
# DC_WEIGHT_PLANES_CHECK=1 (advisor, round 4): cut the planes again in front of EVERY eager product -- the cache watches the tensor
# version counter and the step epoch, which a write through `p.data` (manual EMA / clipping idioms) moves neither of; with this
# switch stale planes are impossible (one extra tiny launch per product: a debugging mode, not a fast path).  The supported ways to
# change a weight behind autograd's back stay `invalidate_planes()` after the write, or tracked in-place ops (`p.copy_`, optimizers).
PLANES_ALWAYS_RECUT = os.environ.get("DC_WEIGHT_PLANES_CHECK", "0") == "1"
_PL = {"entries": {}, "order": [], "table": None, "chunks": None, "n_chunks": 0, "epoch": 0, "dirty": False}


def _planes_reset():
    _PL.update(entries={}, order=[], table=None, chunks=None, n_chunks=0, dirty=False)


def invalidate_planes():
    """Declare every pre-split plane stale.  Only needed after weights were written behind autograd's back (`p.data...`,
    raw pointers): in-place updates through tracked tensors (optimizers, load_state_dict, `copy_`) bump the version counter
    the cache watches, and graph replays bump the epoch themselves."""
    _PL["epoch"] += 1


def planes_snapshot():
    """Every device tensor a captured `presplit_begin()` / plane-fed product refers to by raw address: the record table, the
    chunk table and the plane buffers.  A HIP graph that captured them must keep this list alive (graph_step.py): an eager
    `presplit_begin()` after the capture REPLACES the tables when the set of weights changed (another model, a garbage-collected
    one) and entries of dead weights drop their planes -- without a holder the old tensors would return to the allocator and the
    next replay would read recycled memory as records.  Tables are never modified in place, only replaced."""
    keep = [t for t in (_PL["table"], _PL["chunks"]) if t is not None]
    for e in _PL["entries"].values():
        keep.extend(t for t in (e["fwd"], e["bwd"]) if t is not None)
    return keep


def presplit_begin():
    """Cut every registered weight into its bf16 planes (one launch).  Call once per forward pass, before its first product."""
    if not USE_WEIGHT_PLANES or not _PL["entries"]:
        return
    capturing = torch.cuda.is_current_stream_capturing()
    ents = _PL["entries"]
    if capturing:
        _PL["captured_epoch"] = None             # set below once this capture has cut its own planes
    if not capturing:
        dead = [k for k, e in ents.items() if e["wref"]() is None]
        for k in dead:
            del ents[k]
        if dead or _PL["dirty"] or _PL["table"] is None:
            rows, starts, total = [], [0], 0
            cur_dev = torch.device("cuda", torch.cuda.current_device())
            order = []
            for key_, e in ents.items():
                if e["dev"] != cur_dev:
                    # (advisor, round 4) one table per launch, launched on the current device: weights that live on ANOTHER device
                    # stay out of it -- their planes are cut on the spot at each use (_hint_planes -> _split_one), never through
                    # cross-device pointers
                    continue
                order.append(key_)
                n, k = e["n"], e["k"]
                for need in ("fwd", "bwd"):
                    if e["want_" + need] and e[need] is None:
                        e[need] = torch.empty(3 * n * k, dtype=torch.int16, device=e["dev"])
                rows.append([e["ptr"], e["fwd"].data_ptr() if e["fwd"] is not None else 0,
                             e["bwd"].data_ptr() if e["bwd"] is not None else 0, n, k, k])
                total += (n * k + 1023) // 1024
                starts.append(total)
            if not rows:
                _PL.update(table=None, chunks=None, n_chunks=0, dirty=False)
                return
            _PL["table"] = torch.tensor(rows, dtype=torch.int64).to(cur_dev)
            _PL["chunks"] = torch.tensor(starts, dtype=torch.int32).to(cur_dev)
            _PL["n_chunks"], _PL["dirty"] = total, False
            _PL["order"] = order
    if _PL["table"] is None:
        return
    if not capturing:        # nothing changed since the last cut (inference, repeated forward passes): the planes stand
        cur = True
        for key in _PL["order"]:
            e = ents.get(key)
            w = e["wref"]() if e is not None else None
            if w is None or e["epoch"] != _PL["epoch"] or e["version"] != w._version:
                cur = False
                break
        if cur:
            return
    lib.call("dc_presplit_weights", _PL["table"], _PL["chunks"], len(_PL["order"]), _PL["n_chunks"])
    _PL["epoch"] += 1
    if capturing:
        _PL["captured_epoch"] = _PL["epoch"]     # planes cut INSIDE the capture: the only ones a captured product may use
    for key in _PL["order"]:
        e = ents.get(key)
        if e is not None:
            w = e["wref"]()
            e["epoch"], e["version"] = _PL["epoch"], (w._version if w is not None else -1)


def _split_one(e, base):
    """Cut ONE weight now (its first use, or a use after it changed without a presplit_begin): a one-record table."""
    n, k = e["n"], e["k"]
    for need in ("fwd", "bwd"):
        if e["want_" + need] and e[need] is None:
            e[need] = torch.empty(3 * n * k, dtype=torch.int16, device=e["dev"])
    row = [[e["ptr"], e["fwd"].data_ptr() if e["fwd"] is not None else 0, e["bwd"].data_ptr() if e["bwd"] is not None else 0, n, k, k]]
    chunks = (n * k + 1023) // 1024
    lib.call("dc_presplit_weights", torch.tensor(row, dtype=torch.int64).to(e["dev"]),
             torch.tensor([0, chunks], dtype=torch.int32).to(e["dev"]), 1, chunks)
    e["epoch"], e["version"] = _PL["epoch"], base._version


def _hint_planes(w, transposed):
    """In front of a dc_linear_* call: hand the library the pre-split planes of its weight operand `w` [N, K] (transposed: the
    planes of w^T, for the input-gradient product).  Outside a graph capture the planes are ALWAYS current when this returns
    (a weight seen for the first time, or changed since the last cut, is cut on the spot), so a product's arithmetic never
    depends on the call history; inside a capture only planes cut earlier in the capture (or still current) are offered."""
    if not USE_WEIGHT_PLANES or w.dim() != 2 or not w.is_cuda or w.dtype != torch.float32 or not w.is_contiguous():
        return
    base = w._base if w._base is not None else w
    if not (isinstance(base, torch.nn.Parameter) or (base.is_leaf and base.requires_grad)):
        return                                            # temporaries would churn the table
    n, k = w.shape
    if n % 32 or k % 32 or w.data_ptr() % 16:
        return    # the fragment-major plane layout wants whole 32 x 16 blocks; presplit_kernel loads the source 16 bytes at a time
    key = (w.device.index, w.data_ptr(), n, k)
    need = "bwd" if transposed else "fwd"
    e = _PL["entries"].get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if e is not None and e["wref"]() is not base:
        # the address was recycled: ANOTHER tensor with this shape lives here now (the parameter the entry was made for is
        # gone).  Same key, same version counter value are possible -- never trust the old planes.
        if capturing:
            return
        import weakref
        e["wref"], e["version"], e["epoch"] = weakref.ref(base), -1, -1
    if e is None or e[need] is None:
        if capturing:
            return
        if e is None:
            import weakref
            e = dict(wref=weakref.ref(base), ptr=w.data_ptr(), n=n, k=k, dev=w.device, fwd=None, bwd=None, want_fwd=False,
                     want_bwd=False, epoch=-1, version=-1)
            _PL["entries"][key] = e
        e["want_" + need] = True
        _PL["dirty"] = True                               # joins the one-launch table at the next presplit_begin()
        _split_one(e, base)
    elif e["epoch"] != _PL["epoch"] or e["version"] != base._version or (PLANES_ALWAYS_RECUT and not capturing):
        if capturing:
            return
        _split_one(e, base)
    if capturing and e["epoch"] != _PL.get("captured_epoch"):
        # cut before the capture began: a replay would read them again after a captured optimizer update moved the weight --
        # only planes that the captured presplit_begin() rewrites in every replay are safe inside a graph
        return
    lib.raw("dc_gemm_next_b_planes")(w.data_ptr(), e[need].data_ptr(), n * k, n if transposed else k, 1 if transposed else 0)


OWN_TN_MAX_OUTPUTS = 1 << 21


def _require_fp32_gpu(what, *tensors):
    """The dense products exist as hand-written HIP kernels for fp32 device tensors only: no library GEMM, no CPU path."""
    for t in tensors:
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
            raise TypeError(f"{what}: 2-D float32 tensors on a HIP device only (got {t.dtype}, {t.device.type}, {t.dim()}-D); "
                            "deltaconv_amd has no library / CPU product path")


# ---- deferred slab reductions of one autograd node ------------------------------------------------------------------
# A weight gradient is a split-K product: partial tiles per row slab, then their ordered sum.  A node that forms several
# (a DeltaConv layer: v_mlp, s_mlp, max-aggregation Linear) runs the products as they come and ALL the sums in one launch
# at the end of its backward (`with tn_batch():`): the same bits (csrc/gemm_tn.hip: gemm_tn_reduce_many_kernel), two or
# three launches of ~4.7 us less per layer inside a replayed step.  DC_TN_BATCH=0: every weight reduces at once (A/B).
USE_TN_BATCH = [os.environ.get("DC_TN_BATCH", "1") != "0"]
_TN = threading.local()


class _TnBatch:
    def __init__(self):
        self.items = []           # (workspace tensor, slabs, rows, cols, out tensor / view, ldc)

    def add(self, ws, slabs, rows, cols, out, ldc):
        self.items.append((ws, int(slabs), int(rows), int(cols), out, int(ldc)))

    def flush(self):
        it, self.items = self.items, []
        if not it:
            return
        n = len(it)
        i64, i32 = ctypes.c_int64 * n, ctypes.c_int32 * n
        lib.call("dc_gemm_tn_reduce_many", i64(*[e[0].data_ptr() for e in it]), i64(*[e[4].data_ptr() for e in it]),
                 i64(*[e[5] for e in it]), i32(*[e[2] for e in it]), i32(*[e[3] for e in it]), i32(*[e[1] for e in it]), None, n)


@contextlib.contextmanager
def tn_batch():
    """Weight gradients formed inside are complete when the block exits (an inner block joins the outer one)."""
    if not USE_TN_BATCH[0] or getattr(_TN, "batch", None) is not None:
        yield
        return
    _TN.batch = b = _TnBatch()
    try:
        yield
        b.flush()
    finally:
        _TN.batch = None


# ---- deferred finalisers: the second stages of independent column reductions in ONE launch ----------------------------------
# (csrc/common.h, colreduce.h: dc_finalisers_begin / dc_finaliser_defer_next / dc_finalisers_end).  `with fin_batch():` opens the
# queue; `linear_stats(..., defer_final=True)` / `bn_block_reduce(..., defer_final=True)` inside it queue their finaliser; the
# coefficients are valid when the block exits.  DC_FIN_BATCH=0: every finaliser is its own launch (A/B: same bits).
USE_FIN_BATCH = [os.environ.get("DC_FIN_BATCH", "1") != "0"]
USE_GEMM_PAIR = [os.environ.get("DC_GEMM_PAIR", "1") != "0"]     # the two queued forward products of a layer node as one launch
_FIN = threading.local()


@contextlib.contextmanager
def fin_batch():
    if not USE_FIN_BATCH[0] or getattr(_FIN, "keep", None) is not None:
        yield
        return
    rc = lib.raw("dc_finalisers_begin")()
    if rc != 0:
        raise RuntimeError(f"dc_finalisers_begin failed (rc={rc}): {lib.last_error()}")
    _FIN.keep = []
    ok = False
    try:
        yield
        ok = True
    finally:
        _FIN.keep = None
        rc = lib.raw("dc_finalisers_end")(0 if ok else 1, torch.cuda.current_stream().cuda_stream)
        if ok and rc != 0:
            raise RuntimeError(f"dc_finalisers_end failed (rc={rc}): {lib.last_error()}")


def _defer_final(*keepalive):
    """Ask the next reduction call to queue its finaliser (only inside fin_batch()); its workspaces stay alive until the flush."""
    keep = getattr(_FIN, "keep", None)
    if keep is None:
        return False
    keep.extend(keepalive)
    lib.raw("dc_finaliser_defer_next")()
    if USE_GEMM_PAIR[0]:
        lib.raw("dc_gemm_defer_next")()     # (consumed by the call's product, if it has one of the pairable kind)
    return True


def gemm_tn(a, b):
    """a.t() @ b for a [R,M], b [R,N] (weight gradient dW = dY^T X) on the hand-written fp32-MFMA split-K kernel
    (csrc/gemm_tn.hip), whatever the shape.  Inside `tn_batch()` the result is complete when that block exits."""
    r, m = a.shape
    n = b.shape[1]
    # measured (profiles/r01i_kernels.log, r01n, gpurun r02c): the MFMA kernels win or tie against the TUNED library
    # up to 256K outputs (e.g. 256x128: 31 vs 62 us; 512x256: 91 vs 88 us; 64x64: 11 vs 10 us) and by 5-18x against
    # its default heuristic (448x256: 1.5 ms); the 1024x512 embedding is ~4 % behind the tuned library (312 vs ~300 us)
    # and ahead of the untuned one.  Round 4: EVERY row count runs here (the kernel guards ragged shapes): the library's fp32
    # product came back 7e-3 off at [4096, 128]^T [4096, 256] (a reduced-precision algorithm: the pinned-slot gradient test
    # of tests/test_gpu_configs.py caught it on the 2-cloud ShapeNet step) -- no vendor GEMM is reachable for fp32 GPU inputs.
    _require_fp32_gpu("gemm_tn", a, b)
    if a.stride(1) != 1 or b.stride(1) != 1:
        a, b = _rowmajor(a), _rowmajor(b)
    out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    if r == 0:                      # no rows: the empty sum (the kernels want r >= 1)
        return out.zero_()
    # outputs beyond the kernel's workspace budget: column blocks of b, each its own launch into its column block of dW
    # (round 5: the library fallback above 2^21 outputs is gone -- no vendor GEMM for fp32 GPU inputs, whatever the size)
    nblk = max(1, min(n, OWN_TN_MAX_OUTPUTS // max(m, 1)))
    for j0 in range(0, n, nblk):
        nj = min(nblk, n - j0)
        nbytes = lib.raw("dc_gemm_tn_workspace_bytes")(r, m, nj)
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=a.device)
        batch = getattr(_TN, "batch", None)
        if batch is not None:
            slabs = ctypes.c_int32(0)
            lib.call("dc_gemm_tn_slabs", a, a.stride(0), b[:, j0:j0 + nj], b.stride(0), r, m, nj, ws, ws.numel() * 4,
                     ctypes.byref(slabs))
            batch.add(ws, slabs.value, m, nj, out[:, j0:j0 + nj], n)
        else:
            lib.call("dc_gemm_tn", a, a.stride(0), b[:, j0:j0 + nj], b.stride(0), r, m, nj, out[:, j0:j0 + nj], n, 0, ws,
                     ws.numel() * 4)
    return out


# ---- dense products of the per-point Linear layers: hand-written fp32-MFMA kernels (csrc/gemm.hip) --------------
# Every per-point product runs on the hand-written kernels, the embedding MLP included: against the per-shape TUNED
# vendor library its input gradient is 276 vs 240 us and its weight gradient ~312 vs ~300 us at the ModelNet40 shape
# (profiles/r02f_step_timeline.txt, r02e_gemm_lab.txt) -- 1 % of the step -- but against the library's default
# heuristic (any shape without a shipped TunableOp entry: the other configurations) the hand-written kernels win by
# 1.3-1.5x (ShapeNet embedding: 376 + 241 us vs ~250 + ~250 us), and the path no longer depends on tuning files.
# Round 6: the library branches (and their A/B switches) are deleted -- anything but 2-D fp32 device tensors raises.


def _rowmajor(t):
    return t if (t.dtype == torch.float32 and t.stride(-1) == 1 and t.stride(0) >= t.shape[1]) else t.contiguous().float()


def mm_nt(x, w, out=None):
    """x [M,K] (row stride = ld) @ w[N,K]^T -> [M,N]: forward product of a Linear layer (nn/mlp.py:9,15)."""
    x, w = _rowmajor(x), _rowmajor(w)
    m, k = x.shape
    n = w.shape[0]
    _require_fp32_gpu("mm_nt", x, w)
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    _hint_planes(w, False)
    lib.call("dc_linear_forward", x, x.stride(0), w, w.stride(0), m, n, k, out, out.stride(0), 0)
    return out


def mm_nn(dy, w, out=None, accumulate=False):
    """dy [M,N] @ w[N,K] -> [M,K] (+= when accumulate): input gradient of a Linear layer."""
    dy, w = _rowmajor(dy), _rowmajor(w)
    m, n = dy.shape
    k = w.shape[1]
    _require_fp32_gpu("mm_nn", dy, w)
    if out is None:
        assert not accumulate
        out = torch.empty(m, k, dtype=torch.float32, device=dy.device)
    _hint_planes(w, True)
    lib.call("dc_linear_backward_input", dy, dy.stride(0), w, w.stride(0), m, n, k, out, out.stride(0), int(accumulate), 0)
    return out


def linear_grads(dh, x, w, dx_out=None, accumulate=False):
    """Both gradients of y = x w^T for the incoming dh [R, N]: -> (dW [N, K], dX [R, K]); dX lands in `dx_out`
    (+= when accumulate) if given."""
    dh, w = _rowmajor(dh), _rowmajor(w)
    xx = x if x.stride(1) == 1 else x.contiguous()
    return gemm_tn(dh, xx), mm_nn(dh, w, out=dx_out, accumulate=accumulate)


def linear_stats(x, w, bn, gamma, beta, vn=0, defer_final=False):
    """h = x w^T together with the BatchNorm coefficients of the layer behind it, from the GEMM epilogue:
    -> (h, coef[4, C] = mean / invstd / scale / shift, use_batch_stats).  vn = 2: w = the [2co, K] view of the first
    vector block, statistics of the per-point norms of the interleaved (P_c, Q_c) output (C = co); vn = 1: a deeper
    vector block, h = [2n, co], norms over the row pairs.  Running statistics and num_batches_tracked advance exactly
    as in bn_act / vector_nonlin."""
    x, w = _rowmajor(x), _rowmajor(w)
    m, k = x.shape
    n = w.shape[0]
    c = n // 2 if vn == 2 else n
    rows = m // 2 if vn else m
    dev = x.device
    check_bn_rows(bn, rows)
    use_batch = bn.training or bn.running_mean is None
    mom = 0.0 if bn.momentum is None else float(bn.momentum)
    track = bn.training and bn.track_running_stats
    if track:
        bump_counter(bn)
        if bn.momentum is None:
            mom = 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if (track or not use_batch) else (None, None)
    coef = torch.empty(4, c, dtype=torch.float32, device=dev)
    h = torch.empty(m, n, dtype=torch.float32, device=dev)
    _require_fp32_gpu("linear_stats", x, w)
    group = sync_group() if use_batch else None
    if group is not None:
        # statistics of the GLOBAL batch (deltaconv_amd/dp.py): the same GEMM epilogue, cut at the reduction -- this rank's fp64
        # column sums + its row count are all-reduced, dc_bn_coeffs_from_sums finishes (round 5: the fused layer nodes keep
        # their epilogues under synchronised BatchNorm instead of falling back to the composed blocks)
        from .. import dp
        nb = lib.raw("dc_linear_stats_workspace_bytes")(m, n, k, 0)
        ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
        sums = torch.empty(2, 2 * c + 1, dtype=torch.float64, device=dev)      # two records: [0] all-reduced, [1] local
        _hint_planes(w, False)
        if vn:
            lib.call("dc_linear_vn_sums_forward", x, x.stride(0), w, w.stride(0), rows, c, k, h, n, int(vn == 2), sums, 0, ws, nb)
        else:
            lib.call("dc_linear_bn_sums_forward", x, x.stride(0), w, w.stride(0), m, n, k, h, n, sums, 0, ws, nb)
        dp.all_reduce_stats(sums[0], group)
        lib.call("dc_bn_coeffs_from_sums", sums[0], 0, c, gamma, beta, float(bn.eps), mom, rm, rv, coef[0], coef[1], coef[2], coef[3])
        return h, coef, BatchStats(group)
    if use_batch:
        nb = lib.raw("dc_linear_stats_workspace_bytes")(m, n, k, 0)
        ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
        _hint_planes(w, False)
        if vn:
            lib.call("dc_linear_vn_stats_forward", x, x.stride(0), w, w.stride(0), rows, c, k, h, n, int(vn == 2), gamma,
                     beta, float(bn.eps), mom, rm, rv, coef[0], coef[1], coef[2], coef[3], 0, ws, nb)
        else:
            if defer_final:
                _defer_final(ws)           # (inside fin_batch(): coef is valid when that block exits)
            lib.call("dc_linear_bn_stats_forward", x, x.stride(0), w, w.stride(0), m, n, k, h, n, gamma, beta,
                     float(bn.eps), mom, rm, rv, coef[0], coef[1], coef[2], coef[3], 0, ws, nb)
        return h, coef, BatchStats(None)
    mm_nt(x, w, out=h)                 # inference: coefficients from the running statistics
    return h, eval_coeffs(gamma, beta, rm, rv, float(bn.eps), c), False


FUSE_BN_BWD = True     # A/B switch: BatchNorm/activation backward folded into the consuming GEMMs (no dh tensor)


def bn_block_reduce(dy, lddy, inp, h, coef, use_batch, gamma, slope, W, defer_final=False):
    """First half of bn_block_backward in its fused form: the one reduction over (dy, h) -> dgamma / dbeta and the five
    per-column coefficients of the GEMM prologue.  -> state for bn_block_products, or None when the fused form does not
    apply (the caller then uses bn_block_backward).  defer_final (inside fin_batch()): the finaliser is queued -- the
    coefficients are valid when that block exits, bn_block_products must come after it."""
    r, c = h.shape
    k = W.shape[1]
    dev = h.device
    inp = _rowmajor(inp)
    _require_fp32_gpu("bn_block_backward", h, inp)
    group = group_of(use_batch)        # the group of the FORWARD statistics
    if not ((FUSE_BN_BWD or group is not None) and c * k <= OWN_TN_MAX_OUTPUTS):
        return None
    dg = torch.empty(c, dtype=torch.float32, device=dev)
    db = torch.empty(c, dtype=torch.float32, device=dev)
    ws, nb = _ws(r, c, dev)
    coefs = torch.empty(5 * c, dtype=torch.float32, device=dev)
    if group is not None:     # sums of this rank -> all-reduce -> the prologue coefficients from the global sums
        stats, local = _sync_stats(("dc_bn_act_backward_sums", lambda out: (dy, lddy, h, c, r, c, coef[2], coef[3], coef[0],
                                                                         coef[1], slope, out, ws, nb)), c, r, dev, group)
        lib.call("dc_bn_backward_coefs_from_sums", stats, 0, local, c, gamma, coef[2], coef[3], coef[0], coef[1], 1, dg, db,
                 coefs)
    else:
        if defer_final:
            _defer_final(ws, coefs, dg, db, coef)
        lib.call("dc_bn_act_backward_reduce", dy, lddy, h, c, r, c, coef[2], coef[3], coef[0], coef[1], gamma, slope,
                 int(use_batch), dg, db, coefs, ws, nb)
    return dy, lddy, inp, h, coefs, slope, W, dg, db


def bn_block_products(state, want_dinp=True, dinp_out=None, accumulate=False):
    """Second half: both products rebuild dh = c_g dy act'(c_sc h + c_sh) + c_a h + c_b in their operand loaders.
    -> (dW [C, K], dgamma, dbeta, d_inp [R, K] or None)."""
    dy, lddy, inp, h, coefs, slope, W, dg, db = state
    r, c = h.shape
    k = W.shape[1]
    dev = h.device
    dW = torch.empty(c, k, dtype=torch.float32, device=dev)
    nb2 = lib.raw("dc_gemm_tn_workspace_bytes")(r, c, k)
    ws2 = torch.empty((nb2 + 3) // 4, dtype=torch.float32, device=dev)
    batch = getattr(_TN, "batch", None)
    if batch is not None:
        slabs = ctypes.c_int32(0)
        lib.call("dc_linear_bn_backward_weight_slabs", dy, lddy, h, c, coefs, slope, inp, inp.stride(0), r, c, k, ws2,
                 ws2.numel() * 4, ctypes.byref(slabs))
        batch.add(ws2, slabs.value, c, k, dW, k)
    else:
        lib.call("dc_linear_bn_backward_weight", dy, lddy, h, c, coefs, slope, inp, inp.stride(0), r, c, k, dW, k, 0, ws2,
                 ws2.numel() * 4)
    dinp = None
    if want_dinp:
        dinp = dinp_out if dinp_out is not None else torch.empty(r, k, dtype=torch.float32, device=dev)
        W = _rowmajor(W)
        _hint_planes(W, True)
        lib.call("dc_linear_bn_backward_input", dy, lddy, h, c, coefs, slope, W, W.stride(0), r, c, k, dinp,
                 dinp.stride(0), int(accumulate), 0)
    return dW, dg, db, dinp


def bn_block_backward(dy, lddy, inp, h, coef, use_batch, gamma, slope, W, want_dinp=True, dinp_out=None,
                      accumulate=False):
    """Backward of one block y = leaky(batch_norm(inp W^T)) for the incoming dy [R, C] (row stride lddy):
    -> (dW [C, K], dgamma, dbeta, d_inp [R, K] or None).  d_inp lands in `dinp_out` (+= when accumulate) if given.
    Fused form (csrc/gemm.hip prologue): one reduction over (dy, h) yields dgamma / dbeta and five per-column
    coefficients (bn_block_reduce); both products rebuild dh = c_g dy act'(c_sc h + c_sh) + c_a h + c_b in their operand
    loaders (bn_block_products), so the [R, C] tensor dh is never written or read.  Otherwise: dc_bn_act_backward, then the
    two products on dh."""
    state = bn_block_reduce(dy, lddy, inp, h, coef, use_batch, gamma, slope, W)
    if state is not None:
        return bn_block_products(state, want_dinp, dinp_out, accumulate)
    r, c = h.shape
    k = W.shape[1]
    dev = h.device
    group = group_of(use_batch)
    if group is not None:              # same shapes on every rank: every rank raises here, before any collective
        raise NotImplementedError(f"synchronised BatchNorm backward of a fused block with {c} x {k} > {OWN_TN_MAX_OUTPUTS} "
                                  "weight entries (the un-fused form has no split reduction)")
    dg = torch.empty(c, dtype=torch.float32, device=dev)
    db = torch.empty(c, dtype=torch.float32, device=dev)
    ws, nb = _ws(r, c, dev)
    inp = _rowmajor(inp)
    dh = torch.empty_like(h)
    lib.call("dc_bn_act_backward", dy, lddy, h, c, r, c, coef[2], coef[3], coef[0], coef[1], gamma, slope, int(use_batch),
             dh, c, dg, db, ws, nb)
    dW = gemm_tn(dh, inp)
    dinp = mm_nn(dh, W, out=dinp_out, accumulate=accumulate) if want_dinp else None
    return dW, dg, db, dinp


class _Linear(torch.autograd.Function):
    """y = x W^T (+ b) on 2-D row-major x: forward and input gradient through csrc/gemm.hip, the weight gradient
    dW = dY^T X through `gemm_tn` (own fp32-MFMA kernels; small / per-cloud problems stay with the library)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        y = mm_nt(x, w)
        if b is not None:
            y += b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dw, dx = linear_grads(dy, x, w)
        else:
            dx = mm_nn(dy, w) if ctx.needs_input_grad[0] else None
            dw = gemm_tn(dy, x if x.stride(1) == 1 else x.contiguous()) if ctx.needs_input_grad[1] else None
        db = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


USE_BIAS_ACT = True     # A/B switch: False = ATen add / leaky_relu / column sum around the product (round 4)
_CONST = {}


def _const(c, dev, value):
    """[c] fp32 constant vector on `dev` (ones / zeros: the identity BatchNorm map of the bias + activation form below)."""
    key = (dev.index, c, value)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((c,), value, dtype=torch.float32, device=dev)
    return t


class _LinearBiasAct(torch.autograd.Function):
    """y = leaky_slope(x W^T + b) on many rows (slope 1 = plain Linear with bias): the segmentation head's
    `Linear(256, 128) -> LeakyReLU(0.2) -> Linear(128, num_classes)` (deltaconv/models/deltanet_segmentation.py:45-51).
    Bias + activation are ONE pass of the BatchNorm/activation kernel with the identity map (scale 1, shift b); the backward
    pass takes d b and d h from its inference-mode backward (one ordered reduction + one pass) -- ATen ran add, leaky_relu,
    leaky_relu_backward and a [R, C] -> [C] sum here (round 5)."""

    @staticmethod
    def forward(ctx, x, w, b, slope):
        h = mm_nt(x, w)
        r, c = h.shape
        y = torch.empty_like(h)
        lib.call("dc_bn_act", h, r, c, c, _const(c, h.device, 1.0), b, slope, None, 0, y, c)
        ctx.save_for_backward(x, w, h if slope != 1.0 else None, b)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, h, b = ctx.saved_tensors
        slope = ctx.slope
        dy = _c(dy)
        r, c = dy.shape
        dev = dy.device
        one, zero = _const(c, dev, 1.0), _const(c, dev, 0.0)
        db = torch.empty(c, dtype=torch.float32, device=dev)
        ws, nb = _ws(r, c, dev)
        dh = torch.empty(r, c, dtype=torch.float32, device=dev)
        # (slope 1: act' = 1 whatever h is -- dy itself stands in for the pre-activation that was not kept)
        lib.call("dc_bn_act_backward", dy, c, h if h is not None else dy, c, r, c, one, b, zero, one, None, slope, 0, dh, c,
                 None, db, ws, nb)
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dw, dx = linear_grads(dh, x, w)
        else:
            dx = mm_nn(dh, w) if ctx.needs_input_grad[0] else None
            dw = gemm_tn(dh, x if x.stride(1) == 1 else x.contiguous()) if ctx.needs_input_grad[1] else None
        return dx, dw, (db if ctx.needs_input_grad[2] else None), None


def linear_bias_act(x, w, b, slope=1.0):
    """leaky_slope(x W^T + b) -- own kernels for 2-D fp32 GPU inputs with a bias, else torch."""
    if (USE_BIAS_ACT and b is not None and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32
            and b.dtype == torch.float32 and slope >= 0 and not _rowblock_ok(x, w)):
        return _LinearBiasAct.apply(_c(x), w, b, float(slope))
    y = linear(x, w, b)
    return y if slope == 1.0 else torch.nn.functional.leaky_relu(y, slope)


# ---- blocks on a handful of rows (classification head: one row per cloud): csrc/rowblock.hip ------------------------
ROWBLOCK_MAX_ROWS = 64
USE_ROWBLOCK = True        # A/B switch: False = the composed path (library GEMM + statistics + finaliser + activation)


def _rows16(t):
    """rows of a row-major fp32 matrix start on 16-byte boundaries (what dc_rowblock_* require of X and W)"""
    return t.stride(-1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def _rowblock_ok(x, w):
    return (USE_ROWBLOCK and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32
            and 1 <= x.shape[0] <= ROWBLOCK_MAX_ROWS and x.shape[1] % 4 == 0 and x.shape[1] >= 4 and sync_group() is None
            and _rows16(x) and _rows16(w))


def _rowblock_dx(dh, w):
    """d X = dH W on the MFMA GEMM (ragged M is guarded there)."""
    m, n = dh.shape
    k = w.shape[1]
    dx = torch.empty(m, k, dtype=torch.float32, device=dh.device)
    lib.call("dc_linear_backward_input", dh, dh.stride(0), w, w.stride(0), m, n, k, dx, k, 0, 0)
    return dx


class _RowBlock(torch.autograd.Function):
    """[Linear(no bias) -> BatchNorm1d -> leaky(slope)] on <= 64 rows as ONE kernel forward (product, statistics over
    the rows, finalisation, running statistics, activation) and one backward (BatchNorm / activation backward, d gamma,
    d beta, dW) + the input gradient on the MFMA GEMM.  (models/deltanet_classification.py:34-36; nn/mlp.py:7-11.)"""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn, slope, drop=None):
        # drop = (p, salt): torch.nn.Dropout(p) behind the block, applied in the block's own kernels (csrc/rowblock.hip)
        x, w = _rowmajor(x), _rowmajor(w)
        m, k = x.shape
        n = w.shape[0]
        check_bn_rows(bn, m)
        use_batch = bn.training or bn.running_mean is None
        mom = 0.0 if bn.momentum is None else float(bn.momentum)
        track = bn.training and bn.track_running_stats
        if track:
            bump_counter(bn)
            if bn.momentum is None:
                mom = 1.0 / float(bn.num_batches_tracked)
        rm, rv = (bn.running_mean, bn.running_var) if (track or not use_batch) else (None, None)
        dev = x.device
        h = torch.empty(m, n, dtype=torch.float32, device=dev)
        y = torch.empty(m, n, dtype=torch.float32, device=dev)
        coef = torch.empty(4, n, dtype=torch.float32, device=dev)
        mask = None
        if drop is not None:
            mask = torch.empty(m, n, dtype=torch.uint8, device=dev)
            lib.call("dc_rowblock_forward_dropout", x, x.stride(0), w, w.stride(0), m, n, k, gamma, beta, float(bn.eps), mom, rm,
                     rv, 1 if use_batch else 2, slope, h, n, coef, y, n, float(drop[0]), _seed32(), bn.num_batches_tracked,
                     int(drop[1]), mask)
        else:
            lib.call("dc_rowblock_forward", x, x.stride(0), w, w.stride(0), None, m, n, k, gamma, beta, float(bn.eps), mom, rm,
                     rv, 1 if use_batch else 2, slope, h, n, coef, y, n)
        ctx.save_for_backward(x, w, h, coef, gamma, mask)
        ctx.cfg = (use_batch, slope, gamma is not None, beta is not None, None if drop is None else float(drop[0]))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, h, coef, gamma, mask = ctx.saved_tensors
        use_batch, slope, has_g, has_b, drop_p = ctx.cfg
        dy = _c(dy)
        m, n = dy.shape
        k = x.shape[1]
        dev = dy.device
        dh = torch.empty(m, n, dtype=torch.float32, device=dev)
        dw = torch.empty(n, k, dtype=torch.float32, device=dev)
        dgamma = torch.empty(n, dtype=torch.float32, device=dev)
        dbeta = torch.empty(n, dtype=torch.float32, device=dev)
        if mask is not None:
            lib.call("dc_rowblock_backward_dropout", dy, dy.stride(0), h, n, coef, gamma, slope, 1 if use_batch else 2, x,
                     x.stride(0), m, n, k, dw, k, dgamma, dbeta, dh, n, mask, drop_p)
        else:
            lib.call("dc_rowblock_backward", dy, dy.stride(0), h, n, coef, gamma, slope, 1 if use_batch else 2, x, x.stride(0),
                     m, n, k, dw, k, None, dgamma, dbeta, dh, n)
        dx = _rowblock_dx(dh, w) if ctx.needs_input_grad[0] else None
        return dx, dw, dgamma if has_g else None, dbeta if has_b else None, None, None, None


class _RowLinear(torch.autograd.Function):
    """y = x W^T (+ b) on <= 64 rows (the last Linear of the classification head, deltanet_classification.py:36)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _rowmajor(x), _rowmajor(w)
        m, k = x.shape
        n = w.shape[0]
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        lib.call("dc_rowblock_forward", x, x.stride(0), w, w.stride(0), b, m, n, k, None, None, 0.0, 0.0, None, None, 0, 1.0,
                 y, n, None, y, n)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        m, n = dy.shape
        k = x.shape[1]
        dev = dy.device
        dw = torch.empty(n, k, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        db = torch.empty(n, dtype=torch.float32, device=dev) if ctx.has_bias else None
        dh = torch.empty(m, n, dtype=torch.float32, device=dev)
        lib.call("dc_rowblock_backward", dy, dy.stride(0), None, n, None, None, 1.0, 0, x, x.stride(0), m, n, k, dw, k, db,
                 None, None, dh, n)
        dx = _rowblock_dx(dh, w) if ctx.needs_input_grad[0] else None
        return dx, dw, db


def linear(x, w, b=None):
    """F.linear on the hand-written GEMM kernels: 2-D fp32 device tensors (anything else raises -- no library path)."""
    _require_fp32_gpu("linear", x, w)
    if _rowblock_ok(x, w):
        return _RowLinear.apply(x, w, b)
    if USE_BIAS_ACT and b is not None and b.dtype == torch.float32:
        return _LinearBiasAct.apply(_c(x), w, b, 1.0)        # bias in the activation kernel's pass, d b from its reduction
    return _Linear.apply(x, w, b)


class _CloudBias(torch.autograd.Function):
    """h[i] += g[cloud of i] in place (h: a fresh product nobody else reads; equal-size clouds of mx points): the per-cloud half of
    the segmentation head's first Linear joins the per-point half (csrc/nn.hip: dc_cloud_bias_add).  Backward: d h is the
    incoming gradient itself, d g its per-cloud column sums (dc_cloud_colsum: the index_add of `x_max[batch]`,
    deltaconv/models/deltanet_segmentation.py:59) -- an ordered fp64 two-stage reduction instead of ATen's `sum(1)`
    (29 us whatever the batch, profiles/r06_c4_per_rank_step_timeline.txt)."""

    @staticmethod
    def forward(ctx, h, g, mx):
        n, c = h.shape
        lib.call("dc_cloud_bias_add", h, h.stride(0), g, g.stride(0), n, c, mx, h, h.stride(0))
        ctx.mark_dirty(h)
        ctx.cfg = (g.shape[0], mx)
        return h

    @staticmethod
    def backward(ctx, dy):
        nc, mx = ctx.cfg
        dg = None
        if ctx.needs_input_grad[1]:
            d = _rowmajor(dy)
            c = d.shape[1]
            dg = torch.empty(nc, c, dtype=torch.float32, device=d.device)
            nb = lib.raw("dc_cloud_colsum_workspace_bytes")(nc, mx, c)
            ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=d.device)
            lib.call("dc_cloud_colsum", d, d.stride(0), nc, mx, c, dg, c, ws, ws.numel() * 8)
        return dy, dg, None


def cloud_bias(h, g, mx):
    """h [B * mx, C] (a fresh fp32 product, overwritten) + g [B, C] broadcast over each cloud's mx rows."""
    _require_fp32_gpu("cloud_bias", h, g)
    if h.shape[0] != g.shape[0] * mx or h.shape[1] != g.shape[1] or h.stride(1) != 1:
        raise ValueError(f"cloud_bias: h {tuple(h.shape)} against g {tuple(g.shape)} x {mx} rows")
    return _CloudBias.apply(h, _c(g), int(mx))


class _SplitCols(torch.autograd.Function):
    """(w[:, :p], w[:, p:]) as views; backward writes both gradients into ONE fresh [rows, cols] tensor with one copy launch
    (autograd's own slicing pays a fill and a copy per block and an add of the two full-size results: five launches)."""

    @staticmethod
    def forward(ctx, w, p):
        ctx.cfg = (tuple(w.shape), p)
        return w[:, :p], w[:, p:]

    @staticmethod
    def backward(ctx, da, db):
        (r, c), p = ctx.cfg
        src = da if da is not None else db
        if src is None:
            return None, None
        dw = torch.empty(r, c, dtype=src.dtype, device=src.device)
        pairs = []
        for d, blk in ((da, dw[:, :p]), (db, dw[:, p:])):
            if d is None:
                blk.zero_()
            else:
                pairs.append((d if d.stride(-1) == 1 else d.contiguous(), blk))
        from .. import _ops
        _ops.copy_many(pairs)
        return dw, None


def split_cols(w, p):
    """-> (w[:, :p], w[:, p:]) with a one-launch backward."""
    return _SplitCols.apply(w, int(p))


class _LinearBNAct(torch.autograd.Function):
    """One MLP block of nn/mlp.py:7-11 as a single node: y = leaky(batch_norm(x W^T)) (+ residual).  The batch
    statistics come out of the GEMM epilogue (no pass over the Linear output), the backward is
    BatchNorm/activation backward -> dW (gemm_tn) -> dX (gemm.hip)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn, slope, residual):
        h, coef, use_batch = linear_stats(x, w, bn, gamma, beta)
        r, c = h.shape
        y = torch.empty_like(h)
        res = _c(residual)
        lib.call("dc_bn_act", h, r, c, c, coef[2], coef[3], slope, res, c, y, c)
        ctx.save_for_backward(x, w, h, coef, gamma)
        ctx.cfg = (use_batch, slope, gamma is not None, beta is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, h, coef, gamma = ctx.saved_tensors
        training, slope, has_g, has_b, has_res = ctx.cfg
        dy = _c(dy)
        dw, dgamma, dbeta, dx = bn_block_backward(dy, dy.stride(0), x, h, coef, training, gamma, slope, w,
                                                  want_dinp=ctx.needs_input_grad[0])
        return (dx, dw, dgamma if has_g else None, dbeta if has_b else None, None, None, (dy if has_res else None))


def _seed32():
    """The process seed (torch.manual_seed) folded to 31 bits: the key of the row-block dropout streams."""
    s = int(torch.initial_seed())
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():      # data-parallel ranks share the seed: their dropout masks must not coincide
        s ^= (dist.get_rank() + 1) * 0x9E3779B1
    return (s ^ (s >> 31)) & 0x7fffffff


USE_ROWBLOCK_DROPOUT = True     # A/B switch: False = torch.nn.Dropout behind the block (ATen fused_dropout + masked_scale)


def rowblock_dropout_ok(x, lin, bn, residual=None):
    """Can `Dropout` behind the block [lin -> bn -> act] run inside the block's kernels?  (<= 64 rows, a BatchNorm whose
    num_batches_tracked advances every training step: the device-side counter of the mask stream.)"""
    return (USE_ROWBLOCK_DROPOUT and lin.bias is None and residual is None and _rowblock_ok(x, lin.weight) and bn.training
            and bn.track_running_stats and bn.num_batches_tracked is not None and bn.num_batches_tracked.is_cuda)


def linear_bn_act(x, lin, bn, slope, residual=None, dropout=None):
    """[Linear(no bias) -> BatchNorm1d -> leaky(slope)](x) (+ residual); lin: torch.nn.Linear, bn: torch.nn.BatchNorm1d.
    dropout = (p, salt): only with rowblock_dropout_ok(...) -- Dropout(p) behind the block, in the block's kernels."""
    if dropout is not None:
        assert rowblock_dropout_ok(x, lin, bn, residual)
        return _RowBlock.apply(x, lin.weight, bn.weight, bn.bias, bn, float(slope), (float(dropout[0]), int(dropout[1])))
    if lin.bias is None and residual is None and _rowblock_ok(x, lin.weight):
        return _RowBlock.apply(x, lin.weight, bn.weight, bn.bias, bn, float(slope))
    if lin.bias is None and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32:     # (synchronised statistics included)
        return _LinearBNAct.apply(x, lin.weight, bn.weight, bn.bias, bn, float(slope), residual)
    return bn_act(linear(x, lin.weight, lin.bias), bn, slope, residual)


class _LinearBNActPool(torch.autograd.Function):
    """pooled[B, (2)C] = [max | mean] over each cloud of leaky(batch_norm(x W^T)): the embedding MLP in front of the
    global pooling (models/deltanet_classification.py:42-49, deltanet_segmentation.py:58-61).  Statistics from the
    GEMM epilogue; the [B*N, C] activation is never materialised, in either direction (csrc/nn.hip: pool_fwd_kernel,
    PoolBwdF, PoolBwdBody)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn, slope, num_clouds, n_per, with_mean):
        h, coef, use_batch = linear_stats(x, w, bn, gamma, beta)
        r, c = h.shape
        dev = h.device
        width = 2 * c if with_mean else c
        pooled = torch.empty(num_clouds, width, dtype=torch.float32, device=dev)
        arg = torch.empty(num_clouds, c, dtype=torch.int32, device=dev)
        lib.call("dc_bn_act_pool", h, c, num_clouds, n_per, c, coef[2], coef[3], slope, int(with_mean), pooled, width,
                 arg)
        ctx.save_for_backward(x, w, h, coef, gamma, arg)
        ctx.cfg = (use_batch, slope, num_clouds, n_per, with_mean, gamma is not None, beta is not None)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x, w, h, coef, gamma, arg = ctx.saved_tensors
        training, slope, num_clouds, n_per, with_mean, has_g, has_b = ctx.cfg
        dpooled = _c(dpooled)
        r, c = h.shape
        dev = h.device
        dh = torch.empty_like(h)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev) if has_g else None
        dbeta = torch.empty(c, dtype=torch.float32, device=dev) if has_b else None
        ws, nb = _ws(r, c, dev)
        lib.call("dc_bn_act_pool_backward", dpooled, dpooled.shape[1], arg, h, c, num_clouds, n_per, c, coef[2], coef[3],
                 coef[0], coef[1], gamma, slope, int(with_mean), int(training), dh, c, dgamma, dbeta, ws, nb)
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dw, dx = linear_grads(dh, x, w)
        else:
            dw = gemm_tn(dh, x if x.stride(1) == 1 else x.contiguous()) if ctx.needs_input_grad[1] else None
            dx = mm_nn(dh, w) if ctx.needs_input_grad[0] else None
        return dx, dw, dgamma, dbeta, None, None, None, None, None


def linear_bn_act_pool(x, lin, bn, slope, num_clouds, n_per, with_mean):
    require_gpu()
    return _LinearBNActPool.apply(x, lin.weight, bn.weight, bn.bias, bn, float(slope), num_clouds, n_per, with_mean)


class _BNActPool(torch.autograd.Function):
    """pooled[B, (2)C] = [max | mean] over each cloud of leaky(batch_norm(h)); the [B*N, C] activation is
    never materialised, in either direction (csrc/nn.hip: pool_fwd_kernel, PoolBwdF, PoolBwdBody)."""

    @staticmethod
    def forward(ctx, h, gamma, beta, rm, rv, use_batch_stats, momentum, eps, slope, num_clouds, n_per, with_mean):
        h = _c(h)
        r, c = h.shape
        dev = h.device
        coef = torch.empty(4, c, dtype=torch.float32, device=dev)
        if use_batch_stats:
            ws, nb = _ws(r, c, dev)
            lib.call("dc_bn_stats", h, r, c, c, gamma, beta, eps, momentum, rm, rv, coef[0], coef[1], coef[2],
                     coef[3], ws, nb)
        else:
            coef = eval_coeffs(gamma, beta, rm, rv, eps, c)
        width = 2 * c if with_mean else c
        pooled = torch.empty(num_clouds, width, dtype=torch.float32, device=dev)
        arg = torch.empty(num_clouds, c, dtype=torch.int32, device=dev)
        lib.call("dc_bn_act_pool", h, c, num_clouds, n_per, c, coef[2], coef[3], slope, int(with_mean), pooled, width,
                 arg)
        ctx.save_for_backward(h, coef, gamma, arg)
        ctx.cfg = (use_batch_stats, slope, num_clouds, n_per, with_mean, gamma is not None, beta is not None)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        h, coef, gamma, arg = ctx.saved_tensors
        training, slope, num_clouds, n_per, with_mean, has_g, has_b = ctx.cfg
        dpooled = _c(dpooled)
        r, c = h.shape
        dev = h.device
        dh = torch.empty_like(h)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev) if has_g else None
        dbeta = torch.empty(c, dtype=torch.float32, device=dev) if has_b else None
        ws, nb = _ws(r, c, dev)
        lib.call("dc_bn_act_pool_backward", dpooled, dpooled.shape[1], arg, h, c, num_clouds, n_per, c, coef[2], coef[3],
                 coef[0], coef[1], gamma, slope, int(with_mean), int(training), dh, c, dgamma, dbeta, ws, nb)
        return dh, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def bn_act_pool(h, bn, slope, num_clouds, n_per, with_mean):
    require_gpu()
    check_bn_rows(bn, h.shape[0])
    use_batch = bn.training or bn.running_mean is None
    mom = 0.0 if bn.momentum is None else float(bn.momentum)
    track = bn.training and bn.track_running_stats
    if track:
        bump_counter(bn)
        if bn.momentum is None:
            mom = 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if (track or not use_batch) else (None, None)
    return _BNActPool.apply(h, bn.weight, bn.bias, rm, rv, use_batch, mom, float(bn.eps), float(slope), num_clouds,
                            n_per, with_mean)
