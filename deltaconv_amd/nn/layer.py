"""One DeltaConv layer as a single autograd node with a hand-written backward.

Same arithmetic as ``DeltaConv.forward_composed`` (reference: deltaconv/nn/deltaconv.py:44-70), but
instead of ~25 small autograd nodes glued by ATen ``cat`` / ``add`` / ``copy`` kernels, the layer
owns its buffers:

* ``div v | curl v | |v|`` is written straight into columns [ci, 4ci) of the s_mlp operand,
  ``hodge(v)`` and ``grad x'`` straight into columns [ci, 2ci+co) of the v_mlp operand (kernels take
  leading dimensions) -- no ``torch.cat``;
* in the backward pass every transposed apply accumulates in place (``accumulate=1``) into the
  gradient buffer of the tensor it belongs to -- no autograd ``add`` kernels, no zero fills;
* the residual ``x_max + s_mlp(...)`` is folded into the BatchNorm/LeakyReLU kernel, the BatchNorm +
  activation of ``s_mlp_max`` into the max-aggregation gather;
* consecutive layers chain their buffers: x' / v' are produced inside the NEXT layer's operand buffers and
  x' is written a second time into its column block of the concatenated embedding input (``LayerCfg.chain``,
  ``LayerCfg.dup``) -- no copies between layers, no ``torch.cat`` of the layer outputs;
* the I_J fold of the first VectorMLP layer uses the reference's [co, 2K] weight as a [2co, K] view (GEMM
  output = interleaved (P_c, Q_c) columns), so neither the weight nor its gradient is re-stacked.

Dense GEMMs: hand-written fp32-MFMA kernels (csrc/gemm.hip forward + input gradient, the forward ones with the
BatchNorm statistics of the block in their epilogue; csrc/gemm_tn.hip weight gradient).  Used when every MLP of the
layer has depth 1 and standard activations (all reference models except the depth-2 segmentation net, which takes
the composed path).
"""
import torch

from .._lib import lib
from . import fused

_F32 = torch.float32


def _c(t):
    return t if (t.dtype == _F32 and t.is_contiguous()) else t.contiguous().float()


def _rows(t):
    """(tensor, leading dimension) for a row-major matrix view whose rows may be strided."""
    if t.dtype != _F32 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous().float()
    return t, t.stride(0)


class LayerCfg:
    """Non-tensor state of one call: graph, operators, BatchNorm modules, flags.
    chain = (width of the next layer's s_mlp operand, width of its v_mlp operand or None): the layer then
    writes x' / v' straight into the left columns of those operands (allocated here, adopted by the next
    layer through `_CHAIN`), so the next layer's `[x | ...]` / `[v | ...]` buffers need no copy."""

    def __init__(self, graph, grad, div, bn_m, bn_s, bn_v, centralized, slope_m, slope_s, vector, chain=None,
                 dup=None):
        self.graph, self.grad, self.div = graph, grad, div
        self.bn_m, self.bn_s, self.bn_v = bn_m, bn_s, bn_v
        self.centralized, self.slope_m, self.slope_s, self.vector = centralized, slope_m, slope_s, vector
        self.chain = chain
        self.dup = dup     # (buffer [n, sum co], column offset): x' is written into that column block as well


# operand buffers handed from one layer to the next: data_ptr of the view -> buffer.  The consumer pops its
# entry; entries that are never consumed (the next layer was not called) are dropped by the size cap.
_CHAIN = {}


def _offer(view, buf):
    if len(_CHAIN) >= 16:
        _CHAIN.clear()
    _CHAIN[view.data_ptr()] = buf


def _adopt(t, rows, cols, width):
    """The chain buffer [rows, width] whose left `cols` columns ARE `t`, or None."""
    buf = _CHAIN.pop(t.data_ptr(), None)
    if (buf is not None and tuple(buf.shape) == (rows, width) and tuple(t.shape) == (rows, cols)
            and t.stride(0) == width and t.stride(1) == 1 and t.dtype == _F32 and buf.data_ptr() == t.data_ptr()):
        return buf
    return None


def _bn_mode(bn):
    """-> (use_batch_stats, momentum, running_mean, running_var) and bumps num_batches_tracked."""
    use_batch = bn.training or bn.running_mean is None
    mom = 0.0 if bn.momentum is None else float(bn.momentum)
    track = bn.training and bn.track_running_stats
    if track:
        fused.bump_counter(bn)
        if bn.momentum is None:
            mom = 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if (track or not use_batch) else (None, None)
    return use_batch, mom, rm, rv


def _bn_coeffs(h, rows, c, ld, bn, gamma, beta, dev, vn_combine=None):
    """Batch (or running) statistics of h -> coef[4,c] = mean, invstd, scale, shift."""
    fused.check_bn_rows(bn, rows)
    use_batch, mom, rm, rv = _bn_mode(bn)
    coef = torch.empty(4, c, dtype=_F32, device=dev)
    if use_batch:
        ws, nb = fused._ws(rows, c, dev)
        if vn_combine is None:
            lib.call("dc_bn_stats", h, rows, c, ld, gamma, beta, float(bn.eps), mom, rm, rv, coef[0], coef[1],
                     coef[2], coef[3], ws, nb)
        else:
            lib.call("dc_vn_stats", h, rows, c, ld, int(vn_combine), gamma, beta, float(bn.eps), mom, rm, rv,
                     coef[0], coef[1], coef[2], coef[3], ws, nb)
    else:
        lib.call("dc_bn_eval_coeffs", gamma, beta, rm, rv, float(bn.eps), c, coef[0], coef[1], coef[2], coef[3])
    return coef, use_batch


class DeltaConvLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, v, Wm, gm, bm, Ws, gs, bs, Wv, gv, bv, cfg):
        g = cfg.graph
        n, k = g.n, g.k
        (x, ldx), (v, ldv) = _rows(x), _rows(v)
        dev = x.device
        ci, co = x.shape[1], Wm.shape[0]
        f32 = dict(dtype=_F32, device=dev)
        call = lib.call
        G, D = cfg.grad.coef, cfg.div.coef

        # ---- scalar stream, max aggregation over the k neighbours (deltaconv.py:50-54)
        x_max = torch.empty(n, co, **f32)
        if cfg.centralized:
            y0 = fused.mm_nt(x, Wm)
            stat = torch.empty(3, n, co, **f32)
            args = torch.empty(2, n, co, dtype=torch.uint8, device=dev)
            use_m, mom, rm, rv = _bn_mode(cfg.bn_m)
            coef_m = torch.empty(4, co, **f32)
            ws, nb = fused._ws(n, co, dev)
            if not use_m:
                call("dc_bn_eval_coeffs", gm, bm, rm, rv, float(cfg.bn_m.eps), co, coef_m[0], coef_m[1], coef_m[2],
                     coef_m[3])
            call("dc_edge_gather_stats", y0, co, g.nbr, n, k, co, int(use_m), gm, bm, float(cfg.bn_m.eps), mom,
                 rm if use_m else None, rv if use_m else None, stat[0], stat[1], args[0], args[1], stat[2],
                 coef_m[0], coef_m[1], coef_m[2], coef_m[3], ws, nb)
            call("dc_edge_max_apply", stat[0], stat[1], args[0], args[1], n, co, coef_m[2], coef_m[3], cfg.slope_m,
                 x_max, co, None)
            max_saved = (y0, stat, args)
        else:
            hm, coef_m, use_m = fused.linear_stats(x, Wm, cfg.bn_m, gm, bm)      # GEMM + statistics epilogue
            arg = torch.empty(n, co, dtype=torch.uint8, device=dev)     # BN + activation folded into the gather
            call("dc_knn_max_affine", g.nbr, n, k, hm, co, co, coef_m[2], coef_m[3], cfg.slope_m, x_max, co, arg)
            max_saved = (hm, arg)

        # ---- [x | div v | curl v | |v|] -> s_mlp, residual x_max (deltaconv.py:57-59)
        x_cat = _adopt(x, n, ci, 4 * ci)
        if x_cat is None:
            x_cat = torch.empty(n, 4 * ci, **f32)
            x_cat[:, :ci].copy_(x)
        call("dc_apply_div_curl_norm", D, g.nbr, n, k, v, ci, ldv, x_cat[:, ci:], 4 * ci)
        hs, coef_s, use_s = fused.linear_stats(x_cat, Ws, cfg.bn_s, gs, bs)
        if cfg.chain is not None:
            xbuf = torch.empty(n, cfg.chain[0], **f32)
            x_new = xbuf[:, :co]
            _offer(x_new, xbuf)
        else:
            x_new = torch.empty(n, co, **f32)
        ldxn = x_new.stride(0)
        # the block view is created HERE (inside the node, like the chain views): an outside view object
        # must not be returned as an output of an autograd Function
        x_dup = cfg.dup[0][:, cfg.dup[1]:cfg.dup[1] + co] if cfg.dup is not None else None
        call("dc_bn_act2", hs, n, co, co, coef_s[2], coef_s[3], cfg.slope_s, x_max, co, x_new, ldxn, x_dup,
             x_dup.stride(0) if x_dup is not None else 0)

        # ---- vector stream: [v | hodge v | grad x'] and its 90-degree rotation -> v_mlp (deltaconv.py:64-68)
        if cfg.vector:
            K = 2 * ci + co
            v_cat = _adopt(v, 2 * n, ci, K)
            if v_cat is None:
                v_cat = torch.empty(2 * n, K, **f32)
                v_cat[:, :ci].copy_(v)
            call("dc_apply_hodge", G, g.nbr, n, k, x_cat[:, ci:], ci, 4 * ci, v_cat[:, ci:], K)
            call("dc_apply_grad", G, g.nbr, n, k, x_new, co, ldxn, v_cat[:, 2 * ci:], K)
            Wst = Wv.view(2 * co, K)                                 # rows (c, half): the I_J fold, a free view
            # [2n, 2co], columns interleaved (P_c, Q_c); statistics of the per-point norms from the GEMM epilogue
            PQ, coef_v, use_v = fused.linear_stats(v_cat, Wst, cfg.bn_v, gv, bv, vn=True)
            if cfg.chain is not None and cfg.chain[1] is not None:
                vbuf = torch.empty(2 * n, cfg.chain[1], **f32)
                v_new = vbuf[:, :co]
                _offer(v_new, vbuf)
            else:
                v_new = torch.empty(2 * n, co, **f32)
            call("dc_vn_apply", PQ, n, co, 2 * co, 2, coef_v[2], coef_v[3], v_new, v_new.stride(0))
        else:
            v_cat = PQ = coef_v = Wst = None
            use_v = False
            v_new = None                   # the caller passes v through untouched (deltaconv.py:64,70)

        ctx.cfg = cfg
        ctx.flags = (use_m, use_s, use_v, ci, co)
        ctx.max_saved = max_saved
        ctx.save_for_backward(x, v, Wm, gm, Ws, gs, Wst, gv, coef_m, x_cat, hs, coef_s, v_cat, PQ, coef_v)
        return x_new, v_new, x_dup

    @staticmethod
    def backward(ctx, dx_new, dv_new, dx_dup):
        cfg = ctx.cfg
        x, v, Wm, gm, Ws, gs, Wst, gv, coef_m, x_cat, hs, coef_s, v_cat, PQ, coef_v = ctx.saved_tensors
        use_m, use_s, use_v, ci, co = ctx.flags
        g = cfg.graph
        n, k = g.n, g.k
        dev = x.device
        f32 = dict(dtype=_F32, device=dev)
        call = lib.call
        tptr, tedge = g.csc()
        need_x, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dWv = dgv = dbv = None

        # d x' arrives from the next layer (dx_new) and / or from the concatenated embedding input (dx_dup)
        if dx_new is not None and dx_dup is not None:
            (dxn, lddx), private = _rows(dx_new + dx_dup), True
        elif dx_new is not None or dx_dup is not None:
            (dxn, lddx), private = _rows(dx_new if dx_new is not None else dx_dup), False
        else:
            dxn, lddx, private = torch.zeros(n, co, **f32), co, True
        dv_cat = None
        if cfg.vector and dv_new is not None:
            K = 2 * ci + co
            dvn, lddvn = _rows(dv_new)
            dPQ = torch.empty_like(PQ)
            dgv, dbv = torch.empty(co, **f32), torch.empty(co, **f32)
            ws, nb = fused._ws(n, co, dev)
            call("dc_vn_backward", dvn, lddvn, PQ, 2 * co, 2, n, co, coef_v[2], coef_v[3], coef_v[0], coef_v[1], gv,
                 int(use_v), dPQ, 2 * co, dgv, dbv, ws, nb)
            dWv = fused.gemm_tn(dPQ, v_cat).view(co, 2 * K)           # [2co, K] rows (c, half) = the [co, 2K] layout
            if need_v or ctx.needs_input_grad[0]:
                dv_cat = fused.mm_nn(dPQ, Wst)                        # [2n, K]
            else:   # first layer (x, v carry no gradient): only the `grad @ x'` block of d v_cat is consumed
                dv_cat = torch.empty(2 * n, K, **f32)
                fused.mm_nn(dPQ, Wst[:, 2 * ci:].contiguous(), out=dv_cat[:, 2 * ci:])
            # grad^T of the `grad @ x'` block accumulates into d x'
            if not private:                    # accumulated into below: never touch autograd's buffer
                dxn, lddx = (dxn.clone() if dxn.is_contiguous() else dxn.contiguous()), co
            call("dc_apply_grad_T", cfg.grad.coefT(), tptr, tedge, n, k, dv_cat[:, 2 * ci:], co, K, dxn, lddx, 1)

        # ---- s_mlp block (residual: d x_max = d x')
        dhs = torch.empty_like(hs)
        dgs, dbs = torch.empty(co, **f32), torch.empty(co, **f32)
        ws, nb = fused._ws(n, co, dev)
        call("dc_bn_act_backward", dxn, lddx, hs, co, n, co, coef_s[2], coef_s[3], coef_s[0], coef_s[1], gs, cfg.slope_s,
             int(use_s), dhs, co, dgs, dbs, ws, nb)
        dWs = fused.gemm_tn(dhs, x_cat)
        d_xcat = None
        if need_x or need_v:
            d_xcat = fused.mm_nn(dhs, Ws)                             # [n, 4ci] = d[x | div | curl | norm]
        if dv_cat is not None and d_xcat is not None:   # hodge^T accumulates into d[div | curl]
            call("dc_apply_hodge_T", cfg.grad.coefT(), tptr, tedge, n, k, dv_cat[:, ci:], ci, 2 * ci + co,
                 d_xcat[:, ci:], 4 * ci, 1)
        dv = None
        if need_v:
            if dv_cat is not None:          # accumulate on top of d v from the v_mlp operand, in place
                dv = dv_cat[:, :ci]
                call("dc_apply_div_curl_norm_T", cfg.div.coefT(), tptr, tedge, n, k, d_xcat[:, ci:], ci, 4 * ci, v, v.stride(0),
                     dv, 2 * ci + co, 1)
            else:
                dv = torch.empty(2 * n, ci, **f32)
                call("dc_apply_div_curl_norm_T", cfg.div.coefT(), tptr, tedge, n, k, d_xcat[:, ci:], ci, 4 * ci, v, v.stride(0),
                     dv, ci, 0)

        # ---- max-aggregation branch (d x_max = d x')
        dgm, dbm = torch.empty(co, **f32), torch.empty(co, **f32)
        ws, nb = fused._ws(n, co, dev)
        if cfg.centralized:
            y0, stat, args = ctx.max_saved
            dzs, dy0 = torch.empty(n, co, **f32), torch.empty(n, co, **f32)
            call("dc_edge_max_backward", dxn, lddx, y0, co, tptr, tedge, n, k, co, stat[0], stat[1], args[0], args[1],
                 stat[2], coef_m[2], coef_m[3], coef_m[0], coef_m[1], cfg.slope_m, int(use_m), dzs, dy0, co, dgm, dbm,
                 ws, nb)
            dpre = dy0
        else:
            hm, arg = ctx.max_saved
            dym = torch.empty(n, co, **f32)
            call("dc_knn_max_backward", tptr, tedge, n, k, arg, dxn, co, lddx, dym, co, 0)
            dpre = torch.empty_like(hm)
            call("dc_bn_act_backward", dym, co, hm, co, n, co, coef_m[2], coef_m[3], coef_m[0], coef_m[1], gm,
                 cfg.slope_m, int(use_m), dpre, co, dgm, dbm, ws, nb)
        dWm = fused.gemm_tn(dpre, x)
        dx = None
        if need_x:                            # d x = d_xcat[:, :ci] + dpre Wm, accumulated in place (GEMM with ldc = 4 ci)
            dx = d_xcat[:, :ci]
            fused.mm_nn(dpre, Wm, out=dx, accumulate=True)
        nz = lambda t, ref: t if ref is not None else None
        return (dx, dv, dWm, nz(dgm, gm), nz(dbm, gm), dWs, nz(dgs, gs), nz(dbs, gs), dWv, nz(dgv, gv), nz(dbv, gv),
                None)


# ---- concatenation of the layer outputs without a copy ---------------------------------------------
_CATBUF = {}   # data_ptr of the first block -> (buffer [n, sum co], [(data_ptr, width), ...])


def offer_cat(buf, widths):
    if len(_CATBUF) >= 8:
        _CATBUF.clear()
    offs = [sum(widths[:i]) for i in range(len(widths))]
    _CATBUF[buf.data_ptr()] = (buf, [(buf.data_ptr() + 4 * o, w) for o, w in zip(offs, widths)])


class _ViewCat(torch.autograd.Function):
    """cat(xs, dim=1) when xs are the consecutive column blocks of one buffer: returns the buffer."""

    @staticmethod
    def forward(ctx, holder, *xs):
        buf = holder[0]
        ctx.widths = [x.shape[1] for x in xs]
        return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(buf.untyped_storage(), buf.storage_offset(),
                                                                       buf.shape, buf.stride())

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for w in ctx.widths:
            outs.append(g[:, off:off + w])
            off += w
        return (None, *outs)


def cat_outputs(xs):
    """torch.cat(xs, dim=1) (deltanet_classification.py:42, deltanet_segmentation.py:58) -- free when the
    backbone already wrote every layer output into its column block of one buffer (`LayerCfg.dup`)."""
    hit = _CATBUF.pop(xs[0].data_ptr(), None) if len(xs) else None
    if hit is not None:
        buf, blocks = hit
        if len(blocks) == len(xs) and all(x.data_ptr() == p and x.shape[1] == w and x.stride(0) == buf.shape[1]
                                          for x, (p, w) in zip(xs, blocks)):
            return _ViewCat.apply((buf,), *xs)
    return torch.cat(xs, dim=1)
