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
  activation of the last ``s_mlp_max`` block into the max-aggregation gather;
* consecutive layers chain their buffers: x' / v' are produced inside the NEXT layer's operand buffers and
  x' is written a second time into its column block of the concatenated embedding input (``LayerCfg.chain``,
  ``LayerCfg.dup``) -- no copies between layers, no ``torch.cat`` of the layer outputs.  The hand-over needs no
  registry: a chained tensor is a view whose ``_base`` IS the next operand buffer;
* the I_J fold of the first VectorMLP block uses the reference's [co, 2K] weight as a [2co, K] view (GEMM
  output = interleaved (P_c, Q_c) columns), so neither the weight nor its gradient is re-stacked.

MLPs of any depth (reference default 1; the part-segmentation net uses 2, models/deltanet_segmentation.py:10): every
block but the last of a stream is [GEMM + statistics epilogue -> BatchNorm/activation kernel], the last block carries
the fusions above.  The edge MLP of a centralized first layer with depth > 1 (BatchNorm over the edges between two products) is
computed in front of the node -- depth 2 x 64 channels on csrc/edge2.hip (chained fp32-MFMA products over the edges), other shapes
through dc_edge_diff / dc_seg_reduce -- and enters it as ``x_max``; so does a depth-1 one under synchronised BatchNorm.
Under synchronised BatchNorm (deltaconv_amd/dp.py) the node is unchanged: fused.linear_stats / bn_block_backward / _vn_backward
all-reduce the fp64 sums their kernels hand out between a product and its finaliser.

Dense GEMMs: hand-written fp32-MFMA kernels (csrc/gemm.hip forward + input gradient, the forward ones with the
BatchNorm statistics of the block in their epilogue; csrc/gemm_tn.hip weight gradient).
"""
import os

import torch

from .._lib import lib
from .. import _ops
from . import fused

_F32 = torch.float32
# Test hook: a list here receives, per DeltaConv layer in call order, a copy of the slot the max aggregation selected per
# (point, channel) -- tests pin the CPU oracle to the same selection (tests/test_gpu_configs.py: pinned-slot gradients).
SLOT_TAP = [None]


def _rows(t):
    """(tensor, leading dimension) for a row-major matrix view whose rows may be strided."""
    if t.dtype != _F32 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous().float()
    return t, t.stride(0)


class LayerCfg:
    """Non-tensor state of one call: graph, operators, BatchNorm modules (one per block and stream), flags.
    chain = (width of the next layer's s_mlp operand, width of its v_mlp operand or None): the layer then
    writes x' / v' straight into the left columns of those operands, so the next layer's `[x | ...]` /
    `[v | ...]` buffers need no copy.
    bns_m = None: the max-aggregation branch was computed outside (x_max is an input of the node)."""

    def __init__(self, graph, grad, div, bns_m, bns_s, bns_v, centralized, slopes_m, slopes_s, chain=None, dup=None):
        self.graph, self.grad, self.div = graph, grad, div
        self.bns_m, self.bns_s, self.bns_v = bns_m, bns_s, bns_v
        self.centralized, self.slopes_m, self.slopes_s = centralized, slopes_m, slopes_s
        self.vector = bns_v is not None
        self.chain = chain
        self.dup = dup     # (buffer [n, sum co], column offset): x' is written into that column block as well


# A/B switch: the max aggregation takes the last s_mlp block's BatchNorm / activation / residual into its epilogue (tile plan only)
FUSE_MAX_RESIDUAL = [os.environ.get("DC_FUSE_MAX_RESIDUAL", "1") != "0"]
# A/B switch: the div|curl|norm apply of a layer runs before (1) / after (0) the max-aggregation stream's dense product
APPLY_FIRST = [os.environ.get("DC_APPLY_FIRST", "1") != "0"]


def _padded(rows, cols, block, f32):
    """[rows, cols] fp32 matrix whose column `block` starts on a 16-byte boundary and whose row stride is a multiple of
    4 floats: a view behind (-block) % 4 spare columns of a wider buffer when that is possible, else a plain matrix."""
    pad = (-block) % 4
    if pad and (cols + pad) % 4 == 0:
        return torch.empty(rows, cols + pad, **f32)[:, pad:]
    return torch.empty(rows, cols, **f32)


def _adopt(t, rows, cols, width):
    """The buffer [rows, width] whose left `cols` columns ARE `t` (t was produced by the previous layer inside this
    layer's operand buffer), or None.  Only buffers that the previous layer created FOR chaining qualify (they carry
    the `_dc_chain` mark set in forward): a user tensor that merely looks like the left columns of a wider buffer is
    copied, never overwritten in place."""
    buf = t._base
    if (buf is not None and getattr(buf, "_dc_chain", False) and buf.dim() == 2 and tuple(buf.shape) == (rows, width) and tuple(t.shape) == (rows, cols)
            and t.stride(0) == width and t.stride(1) == 1 and t.dtype == _F32 and buf.dtype == _F32
            and buf.is_contiguous() and buf.data_ptr() == t.data_ptr()):
        buf._dc_chain = False      # one adoption only: a second consumer of the same x' / v' gets its own copy
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


def _bn_backward(dy, lddy, h, coef, use, gamma, slope):
    """-> (dh, dgamma, dbeta) of y = leaky(bn(h)) for the incoming dy (row stride lddy)."""
    r, c = h.shape
    dev = h.device
    dh = torch.empty_like(h)
    dg, db = torch.empty(c, dtype=_F32, device=dev), torch.empty(c, dtype=_F32, device=dev)
    ws, nb = fused._ws(r, c, dev)
    lib.call("dc_bn_act_backward", dy, lddy, h, c, r, c, coef[2], coef[3], coef[0], coef[1], gamma, slope, int(use),
             dh, c, dg, db, ws, nb)
    return dh, dg, db


def _vn_backward(dout, lddo, h, combine, coef, use, gamma):
    """-> (dh, dgamma, dbeta) of the vector non-linearity on h ([2n, co], or interleaved (P, Q) [2n, 2co])."""
    ld = h.shape[1]
    co = ld // 2 if combine else ld
    n = h.shape[0] // 2
    dev = h.device
    dh = torch.empty_like(h)
    dg, db = torch.empty(co, dtype=_F32, device=dev), torch.empty(co, dtype=_F32, device=dev)
    ws, nb = fused._ws(n, co, dev)
    group = fused.group_of(use)      # the group the FORWARD statistics were reduced over (fused.BatchStats)
    if group is not None:        # synchronised statistics: this rank's sums -> all-reduce -> apply with the global means
        stats, local = fused._sync_stats(("dc_vn_backward_sums", lambda out: (dout, lddo, h, ld, combine, n, co, coef[2], coef[3],
                                                                              coef[0], coef[1], out, ws, nb)), co, n, dev, group)
        m = torch.empty(2, co, dtype=_F32, device=dev)
        lib.call("dc_sync_means", stats, co, m[0], m[1], dg, db)             # global means + this rank's dgamma / dbeta
        lib.call("dc_vn_backward_apply", dout, lddo, h, ld, combine, n, co, coef[2], coef[3], coef[0], coef[1], gamma, 1,
                 m[0], m[1], dh, ld)
        return dh, dg, db
    lib.call("dc_vn_backward", dout, lddo, h, ld, combine, n, co, coef[2], coef[3], coef[0], coef[1], gamma, int(use),
             dh, ld, dg, db, ws, nb)
    return dh, dg, db


class DeltaConvLayerFn(torch.autograd.Function):
    """apply(x, v, x_max_or_None, cfg, *params); params = (W, gamma, beta) per block: the s_mlp_max blocks (none when
    x_max is given), the s_mlp blocks, the v_mlp blocks (none for a layer without vector stream)."""

    @staticmethod
    def forward(ctx, x, v, x_max_ext, cfg, *params):
        # outputs nobody consumes (the chained x' of the LAST layer, v' of a layer without successor) must arrive as
        # None in backward, not as materialised zero tensors: the last layer otherwise pays a 33 MB zero fill + an add
        # of zeros per step (7 + 18 us at the bench shape)
        ctx.set_materialize_grads(False)
        g = cfg.graph
        n, k = g.n, g.k
        (x, ldx), (v, ldv) = _rows(x), _rows(v)
        dev = x.device
        ci = x.shape[1]
        f32 = dict(dtype=_F32, device=dev)
        call = lib.call
        nm = 0 if cfg.bns_m is None else len(cfg.bns_m)
        ns = len(cfg.bns_s)
        nv = len(cfg.bns_v) if cfg.vector else 0
        blk = lambda i: params[3 * i:3 * i + 3]
        pm, ps, pv = [blk(i) for i in range(nm)], [blk(nm + i) for i in range(ns)], [blk(nm + ns + i) for i in range(nv)]
        co = ps[-1][0].shape[0]
        saved_m, saved_s, saved_v = [], [], []      # per block: (input, h, coef, use_batch_stats)

        # ---- [x | div v | curl v | |v|] (deltaconv.py:57): formed FIRST (round 6, A/B switch DC_APPLY_FIRST): the apply reads v right
        # behind the vector non-linearity that wrote it, with no dense product's output in between (profiles/r06_labs.txt item 11)
        def build_cat():
            # (x / v that the previous layer did not already write into this layer's operand buffers -- the first layer -- are
            #  copied into their left columns: both copies in one launch)
            x_cat = _adopt(x, n, ci, 4 * ci)
            v_cat = _adopt(v, 2 * n, ci, 2 * ci + co) if cfg.vector else None
            copies = []
            if x_cat is None:
                x_cat = torch.empty(n, 4 * ci, **f32)
                copies.append((x, x_cat[:, :ci]))
            if cfg.vector and v_cat is None:
                # (first layer, ci = 3: two spare columns in front put the `grad @ x'` block on a 16-byte boundary with a
                #  row stride that is a multiple of 4 floats, so that block runs from the tile plans, forward and transposed)
                v_cat = _padded(2 * n, 2 * ci + co, 2 * ci, f32)
                copies.append((v, v_cat[:, :ci]))
            if copies:
                _ops.copy_many(copies)
            _ops.fwd_apply("div_curl_norm", cfg.div, v, ci, ldv, x_cat[:, ci:], 4 * ci)
            return x_cat, v_cat
        if APPLY_FIRST[0]:
            x_cat, v_cat = build_cat()

        # ---- [x | div v | curl v | |v|] -> s_mlp, residual x_max (deltaconv.py:57-59)
        if not APPLY_FIRST[0]:
            x_cat, v_cat = build_cat()
        # ---- scalar stream, max aggregation over the k neighbours (deltaconv.py:50-54)
        def run_m():
            max_saved = None
            pending_max = pending_edge = None
            if nm == 0:
                x_max, ldm = _rows(x_max_ext)
            else:
                x_max, ldm = torch.empty(n, co, **f32), co
                inp = x
                for (W, gm, bm), bn, slope in zip(pm[:-1], cfg.bns_m[:-1], cfg.slopes_m[:-1]):
                    h, coef, use = fused.linear_stats(inp, W, bn, gm, bm)
                    a = torch.empty_like(h)
                    call("dc_bn_act", h, n, h.shape[1], h.shape[1], coef[2], coef[3], slope, None, 0, a, h.shape[1])
                    saved_m.append((inp, h, coef, use))
                    inp = a
                Wm, gm, bm = pm[-1]
                bn_m, slope_m = cfg.bns_m[-1], cfg.slopes_m[-1]
                if cfg.centralized:                      # depth 1 only (DeltaConv.forward routes depth > 1 outside)
                    y0 = fused.mm_nt(inp, Wm)
                    stat = torch.empty(3, n, co, **f32)
                    args = torch.empty(2, n, co, dtype=torch.uint8, device=dev)
                    use_m, mom, rm, rv = _bn_mode(bn_m)
                    coef_m = torch.empty(4, co, **f32)
                    ws, nb = fused._ws(n, co, dev)
                    if not use_m:
                        coef_m = fused.eval_coeffs(gm, bm, rm, rv, float(bn_m.eps), co)
                    call("dc_edge_gather_stats", y0, co, g.nbr, n, k, co, int(use_m), gm, bm, float(bn_m.eps), mom,
                         rm if use_m else None, rv if use_m else None, stat[0], stat[1], args[0], args[1], stat[2],
                         coef_m[0], coef_m[1], coef_m[2], coef_m[3], ws, nb)
                    argsel = torch.empty(n, co, dtype=torch.uint8, device=dev)   # the selected slot: backward from the tile plan
                    # (applied below, behind the last s_mlp block, whose BatchNorm / activation / residual add it takes along)
                    pending_edge = (stat, args, coef_m, slope_m, argsel)
                    max_saved = (stat, args, argsel)
                    saved_m.append((inp, y0, coef_m, use_m))
                else:
                    hm, coef_m, use_m = fused.linear_stats(inp, Wm, bn_m, gm, bm, defer_final=True)   # GEMM + statistics epilogue
                    arg = torch.empty(n, co, dtype=torch.uint8, device=dev)     # BN + activation folded into the gather
                    # the gather itself waits for the last s_mlp block (below): from the tile plan it takes that block's BatchNorm /
                    # activation pass and the residual add into its epilogue (round 6: one launch and the x_max round trip less)
                    pending_max = (hm, coef_m, slope_m, arg)
                    max_saved = (arg,)
                    saved_m.append((inp, hm, coef_m, use_m))
            return x_max, ldm, pending_max, pending_edge, max_saved

        def run_s():
            inp = x_cat
            for (W, gs, bs), bn, slope in zip(ps[:-1], cfg.bns_s[:-1], cfg.slopes_s[:-1]):
                h, coef, use = fused.linear_stats(inp, W, bn, gs, bs)
                a = torch.empty_like(h)
                call("dc_bn_act", h, n, h.shape[1], h.shape[1], coef[2], coef[3], slope, None, 0, a, h.shape[1])
                saved_s.append((inp, h, coef, use))
                inp = a
            Ws, gs, bs = ps[-1]
            hs, coef_s, use_s = fused.linear_stats(inp, Ws, cfg.bns_s[-1], gs, bs, defer_final=True)
            saved_s.append((inp, hs, coef_s, use_s))
            return hs, coef_s

        # (round-6 lab: the two independent products in the other order -- the rows the aggregation gathers written last -- +0.3 % of
        #  the step, profiles/r06_labs.txt item 11: the max-aggregation stream's product stays first)
        # (the finalisers of the two streams' LAST products share one launch: their coefficients are first read by the max
        #  aggregation below -- fused.fin_batch, round 6)
        with fused.fin_batch():
            x_max, ldm, pending_max, pending_edge, max_saved = run_m()
            hs, coef_s = run_s()
        if cfg.chain is not None:
            xbuf = torch.empty(n, cfg.chain[0], **f32)
            xbuf._dc_chain = True
            x_new = xbuf[:, :co]
        else:
            x_new = torch.empty(n, co, **f32)
        ldxn = x_new.stride(0)
        # the block view is created HERE (inside the node, like the chain views): an outside view object
        # must not be returned as an output of an autograd Function
        x_dup = cfg.dup[0][:, cfg.dup[1]:cfg.dup[1] + co] if cfg.dup is not None else None
        fused_max = False
        if pending_max is not None:
            hm, coef_m, slope_m, arg = pending_max
            if FUSE_MAX_RESIDUAL[0]:
                fused_max = _ops.fwd_knn_max_residual(g, hm, co, co, (coef_m[2], coef_m[3], slope_m), hs, co,
                                                      (coef_s[2], coef_s[3], cfg.slopes_s[-1]), x_new, ldxn, x_dup,
                                                      x_dup.stride(0) if x_dup is not None else 0, arg)
            if not fused_max:
                _ops.fwd_knn_max(g, hm, co, co, x_max, co, arg, affine=(coef_m[2], coef_m[3], slope_m))
            if SLOT_TAP[0] is not None:
                SLOT_TAP[0].append(arg.clone())
        if pending_edge is not None:
            stat, args, coef_m, slope_m, argsel = pending_edge
            if FUSE_MAX_RESIDUAL[0] and cfg.slopes_s[-1] is not None:
                call("dc_edge_max_apply_residual", stat[0], stat[1], args[0], args[1], n, co, coef_m[2], coef_m[3], slope_m, hs, co,
                     coef_s[2], coef_s[3], float(cfg.slopes_s[-1]), x_new, ldxn, x_dup, x_dup.stride(0) if x_dup is not None else 0,
                     argsel)
                fused_max = True
            else:
                call("dc_edge_max_apply", stat[0], stat[1], args[0], args[1], n, co, coef_m[2], coef_m[3], slope_m, x_max, co, argsel)
            if SLOT_TAP[0] is not None:
                SLOT_TAP[0].append(argsel.clone())
        if not fused_max:
            call("dc_bn_act2", hs, n, co, co, coef_s[2], coef_s[3], cfg.slopes_s[-1], x_max, ldm, x_new, ldxn, x_dup,
                 x_dup.stride(0) if x_dup is not None else 0)

        # ---- vector stream: [v | hodge v | grad x'] and its 90-degree rotation -> v_mlp (deltaconv.py:64-68)
        v_new = None                       # without vector stream the caller passes v through (deltaconv.py:64,70)
        if cfg.vector:
            K = 2 * ci + co
            ldvc = v_cat.stride(0)
            # (round-6 lab: this apply right behind div|curl|norm, whose streamed-out block it reads: 14.2 -> 16.3 us in step, dropped)
            _ops.fwd_apply("hodge", cfg.grad, x_cat[:, ci:], ci, 4 * ci, v_cat[:, ci:], ldvc)
            _ops.fwd_apply("grad", cfg.grad, x_new, co, ldxn, v_cat[:, 2 * ci:], ldvc)
            if cfg.chain is not None and cfg.chain[1] is not None:
                vbuf = torch.empty(2 * n, cfg.chain[1], **f32)
                vbuf._dc_chain = True
                v_new = vbuf[:, :co]
            else:
                v_new = torch.empty(2 * n, co, **f32)
            inp = v_cat
            for j, ((W, gv, bv), bn) in enumerate(zip(pv, cfg.bns_v)):
                c1 = W.shape[0]
                if j == 0:
                    # rows (c, half) of the [c1, 2K] weight: the I_J fold, a free view.  Output [2n, 2c1], columns
                    # interleaved (P_c, Q_c); statistics of the per-point norms from the GEMM epilogue
                    h, coef, use = fused.linear_stats(inp, W.view(2 * c1, K), bn, gv, bv, vn=2)
                else:
                    h, coef, use = fused.linear_stats(inp, W, bn, gv, bv, vn=1)
                out = v_new if j == nv - 1 else torch.empty(2 * n, c1, **f32)
                call("dc_vn_apply", h, n, c1, h.shape[1], 2 if j == 0 else 0, coef[2], coef[3], out, out.stride(0))
                saved_v.append((inp, h, coef, use))
                inp = out

        ctx.cfg = cfg
        ctx.meta = (ci, co, nm, ns, nv, [s[3] for s in saved_m], [s[3] for s in saved_s], [s[3] for s in saved_v],
                    bool(cfg.centralized and nm > 0))
        ctx.max_saved_n = 0 if max_saved is None else len(max_saved)
        flat = [x, v, x_cat]
        for s in saved_m + saved_s + saved_v:
            flat.extend(s[:3])
        flat.extend(max_saved or ())
        flat.extend(params)
        ctx.save_for_backward(*flat)
        return x_new, v_new, x_dup

    @staticmethod
    def backward(ctx, dx_new, dv_new, dx_dup):
        # the slab sums of the node's three or four weight gradients run as one launch when the block exits (fused.tn_batch)
        with fused.tn_batch():
            return DeltaConvLayerFn._backward(ctx, dx_new, dv_new, dx_dup)

    @staticmethod
    def _backward(ctx, dx_new, dv_new, dx_dup):
        cfg = ctx.cfg
        ci, co, nm, ns, nv, use_m, use_s, use_v, centralized = ctx.meta
        sv = list(ctx.saved_tensors)
        x, v, x_cat = sv[0], sv[1], sv[2]
        pos = 3

        def take(cnt):
            nonlocal pos
            out = [tuple(sv[pos + 3 * i:pos + 3 * i + 3]) for i in range(cnt)]
            pos += 3 * cnt
            return out
        sm, ss, svv = take(nm), take(ns), take(nv)
        max_saved = sv[pos:pos + ctx.max_saved_n]
        params = sv[pos + ctx.max_saved_n:]
        blk = lambda i: params[3 * i:3 * i + 3]
        pm, ps, pv = [blk(i) for i in range(nm)], [blk(nm + i) for i in range(ns)], [blk(nm + ns + i) for i in range(nv)]
        g = cfg.graph
        n, k = g.n, g.k
        dev = x.device
        f32 = dict(dtype=_F32, device=dev)
        call = lib.call
        need_x, need_v, need_xmax = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gm_list, gs_list, gv_list = [None] * nm, [None] * ns, [None] * nv        # per block (dW, dgamma, dbeta)

        # d x' arrives from the next layer (dx_new) and / or from the concatenated embedding input (dx_dup); with a
        # vector stream, `grad^T d(grad @ x')` is a third term: all of them are summed by ONE kernel into a fresh
        # tensor (dc_apply_grad_T_sum, below) -- no add pass and no clone of autograd's buffers
        fold_sum = bool(nv and dv_new is not None and (dx_new is not None or dx_dup is not None))
        if fold_sum:
            dxn = lddx = None
            private = True
        elif dx_new is not None and dx_dup is not None:
            (dxn, lddx), private = _rows(dx_new + dx_dup), True
        elif dx_new is not None or dx_dup is not None:
            (dxn, lddx), private = _rows(dx_new if dx_new is not None else dx_dup), False
        else:
            dxn, lddx, private = torch.zeros(n, co, **f32), co, True

        # ---- vector stream
        dv_cat = None
        if nv and dv_new is not None:
            K = 2 * ci + co
            dcur, ldd = _rows(dv_new)
            for j in range(nv - 1, -1, -1):
                inp, h, coef = svv[j]
                W, gv, _ = pv[j]
                dh, dg, db = _vn_backward(dcur, ldd, h, 2 if j == 0 else 0, coef, use_v[j], gv)
                if j == 0:
                    Wst = W.view(2 * W.shape[0], K)
                    if need_v or need_x:
                        dW, dv_cat = fused.linear_grads(dh, inp, Wst)        # dv_cat [2n, K]
                        dW = dW.view(W.shape[0], 2 * K)                      # [2c, K] rows (c, half) = the [c, 2K] layout
                    else:   # first layer (x, v carry no gradient): only the `grad @ x'` block of d v_cat is consumed
                        dW = fused.gemm_tn(dh, inp).view(W.shape[0], 2 * K)
                        dv_cat = _padded(2 * n, K, 2 * ci, f32)
                        # (the strided, 8-byte-aligned weight block as it lies: the product's guarded scalar loads of it cost what the
                        #  copy launch that made it contiguous did -- round 6: 2.846 / 2.844 -> 2.849 / 2.844 ms, one launch less)
                        fused.mm_nn(dh, Wst[:, 2 * ci:], out=dv_cat[:, 2 * ci:])
                else:
                    dW, dcur = fused.linear_grads(dh, inp, W)
                    ldd = dcur.stride(0)
                gv_list[j] = (dW, dg, db)
            # grad^T of the `grad @ x'` block accumulates into d x'
            if fold_sum:
                ga, lda_ = _rows(dx_new if dx_new is not None else dx_dup)
                gb, ldb_ = _rows(dx_dup) if (dx_new is not None and dx_dup is not None) else (None, 0)
                dxn, lddx = torch.empty(n, co, **f32), co
                _ops.bwd_grad_sum(cfg.grad, dv_cat[:, 2 * ci:], co, dv_cat.stride(0), ga, lda_, gb, ldb_, dxn, lddx)
            else:
                if not private:                # accumulated into below: never touch autograd's buffer
                    dxn, lddx = (dxn.clone() if dxn.is_contiguous() else dxn.contiguous()), co
                _ops.bwd_apply("grad", cfg.grad, dv_cat[:, 2 * ci:], co, dv_cat.stride(0), dxn, lddx, 1)

        # ---- s_mlp blocks (residual: d x_max = d x')
        d_xcat = None
        dcur, ldd = dxn, lddx
        # (round 6) the BatchNorm-backward reductions of the LAST s_mlp block and of the max-aggregation stream's last block
        # both start from d x': the max aggregation's backward runs up here and the two finalisers share one launch
        # (fused.fin_batch; same sums, same bits); the products follow in the old order
        st_s = pair_m = None
        if nm > 0 and not centralized and fused.USE_FIN_BATCH[0]:
            (arg,) = max_saved
            dmax = torch.empty(n, co, **f32)
            _ops.bwd_knn_max(g, arg, dxn, co, lddx, dmax, co, 0)
            (inp_s, h_s, coef_s), (W_s, gs_s, _) = ss[ns - 1], ps[ns - 1]
            (inp_m, h_m, coef_m), (W_m, gm_m, _) = sm[nm - 1], pm[nm - 1]
            with fused.fin_batch():
                st_s = fused.bn_block_reduce(dcur, ldd, inp_s, h_s, coef_s, use_s[ns - 1], gs_s, cfg.slopes_s[ns - 1], W_s,
                                             defer_final=True)
                st_m = fused.bn_block_reduce(dmax, co, inp_m, h_m, coef_m, use_m[nm - 1], gm_m, cfg.slopes_m[nm - 1], W_m,
                                             defer_final=True)
            pair_m = (st_m, dmax)
        for j in range(ns - 1, -1, -1):
            inp, h, coef = ss[j]
            W, gs, _ = ps[j]
            # (j == 0: d_inp = [n, 4ci] = d[x | div | curl | norm])
            if j == ns - 1 and st_s is not None:
                dW, dg, db, dinp = fused.bn_block_products(st_s, want_dinp=(j > 0 or need_x or need_v))
            else:
                dW, dg, db, dinp = fused.bn_block_backward(dcur, ldd, inp, h, coef, use_s[j], gs, cfg.slopes_s[j], W,
                                                           want_dinp=(j > 0 or need_x or need_v))
            gs_list[j] = (dW, dg, db)
            if j > 0:
                dcur, ldd = dinp, dinp.stride(0)
            else:
                d_xcat = dinp
        if dv_cat is not None and d_xcat is not None:   # hodge^T accumulates into d[div | curl]
            _ops.bwd_apply("hodge", cfg.grad, dv_cat[:, ci:], ci, dv_cat.stride(0), d_xcat[:, ci:], 4 * ci, 1)
        dv = None
        if need_v:
            if dv_cat is not None:          # accumulate on top of d v from the v_mlp operand, in place
                dv = dv_cat[:, :ci]
                _ops.bwd_div_curl_norm(cfg.div, d_xcat[:, ci:], ci, 4 * ci, v, v.stride(0), dv, dv_cat.stride(0), 1)
            else:
                dv = torch.empty(2 * n, ci, **f32)
                _ops.bwd_div_curl_norm(cfg.div, d_xcat[:, ci:], ci, 4 * ci, v, v.stride(0), dv, ci, 0)

        # ---- max-aggregation branch (d x_max = d x')
        dx = d_xcat[:, :ci] if (need_x and d_xcat is not None) else None
        d_xmax = None
        if nm == 0:
            d_xmax = dxn if need_xmax else None
        else:
            inp, hm, coef_m = sm[-1]
            Wm, gm, _ = pm[-1]
            if centralized:
                stat, args, argsel = max_saved
                dzs, dpre = torch.empty(n, co, **f32), torch.empty(n, co, **f32)
                dg, db = torch.empty(co, **f32), torch.empty(co, **f32)
                ws, nb = fused._ws(n, co, dev)
                planT = _ops._tiledT(g, co, (dxn, lddx), (hm, co))
                if planT is not None and hm.is_contiguous():
                    call("dc_edge_max_backward_tiled", dxn, lddx, hm, planT.blob, *planT.args, co, stat[0], stat[1], argsel,
                         stat[2], coef_m[2], coef_m[3], coef_m[0], coef_m[1], cfg.slopes_m[-1], int(use_m[-1]), dzs, dpre, co,
                         dg, db, ws, nb)
                else:
                    tptr, tedge = g.csc()
                    call("dc_edge_max_backward", dxn, lddx, hm, co, tptr, tedge, n, k, co, stat[0], stat[1], args[0], args[1],
                         stat[2], coef_m[2], coef_m[3], coef_m[0], coef_m[1], cfg.slopes_m[-1], int(use_m[-1]), dzs, dpre, co,
                         dg, db, ws, nb)
                gm_list[-1] = (fused.gemm_tn(dpre, inp), dg, db)
                dcur = fused.mm_nn(dpre, Wm) if nm > 1 else None
                if nm == 1 and need_x:            # d x = d_xcat[:, :ci] + dpre Wm, accumulated in place (ldc = 4 ci)
                    fused.mm_nn(dpre, Wm, out=dx, accumulate=True)
                first = nm - 2
            else:
                (arg,) = max_saved
                if pair_m is not None:            # (done above, with the reduction of its last block)
                    dcur = pair_m[1]
                else:
                    dcur = torch.empty(n, co, **f32)
                    _ops.bwd_knn_max(g, arg, dxn, co, lddx, dcur, co, 0)
                first = nm - 1
            for j in range(first, -1, -1):        # [Linear -> BN -> act] blocks, last to first; block 0 feeds d x
                inp, h, coef = sm[j]
                W, gmj, _ = pm[j]
                into_dx = j == 0 and need_x
                if j == nm - 1 and pair_m is not None and pair_m[0] is not None:
                    dW, dg, db, dinp = fused.bn_block_products(pair_m[0], want_dinp=(j > 0 or need_x),
                                                               dinp_out=dx if into_dx else None, accumulate=into_dx)
                else:
                    dW, dg, db, dinp = fused.bn_block_backward(dcur, dcur.stride(0), inp, h, coef, use_m[j], gmj,
                                                               cfg.slopes_m[j], W, want_dinp=(j > 0 or need_x),
                                                               dinp_out=dx if into_dx else None, accumulate=into_dx)
                gm_list[j] = (dW, dg, db)
                dcur = dinp

        grads = []
        for lst, pl in ((gm_list, pm), (gs_list, ps), (gv_list, pv)):
            for got, (W, gamma, beta) in zip(lst, pl):
                if got is None:
                    grads.extend((None, None, None))
                else:
                    grads.extend((got[0], got[1] if gamma is not None else None, got[2] if beta is not None else None))
        return (dx, dv, d_xmax, None, *grads)


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
    backbone already wrote every layer output into its column block of one buffer (`LayerCfg.dup`): the tensors are
    then views of that buffer (`_base`) at consecutive column offsets."""
    buf = xs[0]._base if len(xs) else None
    if buf is not None and buf.dim() == 2 and buf.is_contiguous() and buf.dtype == _F32:
        off, ok = 0, True
        for x in xs:
            ok = ok and (x._base is buf and x.dim() == 2 and x.shape[0] == buf.shape[0] and x.stride(1) == 1
                         and x.stride(0) == buf.shape[1] and x.storage_offset() == buf.storage_offset() + off)
            off += x.shape[1] if x.dim() == 2 else 0
        if ok and off == buf.shape[1]:
            return _ViewCat.apply((buf,), *xs)
    return torch.cat(xs, dim=1)
