"""Data-parallel training over the GPUs of one node: one process per GPU, replicated weights,
clouds sharded across ranks (they are independent units: kNN, operators, convolutions and pools
never cross a ``batch`` boundary), ONE flat fp32 gradient all-reduce per step over RCCL/xGMI
(``torch.distributed`` backend "nccl"; "gloo" on CPU in the tests).

The reference has no multi-device code at all; this is the new capability
BASELINE.json asks for.  Design for xGMI: the whole gradient is 5.8-8.9 MB, so a single collective
on one flat buffer is latency- not bandwidth-bound; BatchNorm statistics stay per rank (standard DDP
semantics).  Gradients are produced by autograd as fresh tensors (``.grad = None`` before backward:
no per-parameter accumulate kernels), packed into the flat buffer by one multi-tensor copy, reduced,
and the parameters' ``.grad`` then alias the reduced buffer (no unflatten copy).  On one rank nothing
is copied at all.
"""
import torch
import torch.distributed as dist

# ---- synchronised BatchNorm (optional) -----------------------------------------------------
# Per-rank statistics are the default (standard DDP behaviour).  With `sync_bn=True` every BatchNorm of the reference
# (nn/nonlin.py:24-35, inside MLP and VectorNonLin) normalises with the statistics of the GLOBAL batch, so R ranks on B/R
# clouds each compute exactly what one process computes on B clouds -- needed when a rank holds a single cloud
# (`lin_categorical`, models/deltanet_segmentation.py:44,64, sees one row) and for accuracy-parity runs.  Protocol (the
# same for the HIP kernels in nn/fused.py and the device-agnostic torch module below): each rank reduces its rows to
# fp64 column sums [sum_0 (C) | sum_1 (C) | rows], the 2C+1 doubles are all-reduced (SUM), everything downstream uses
# the global sums; in backward the weight / bias gradients come from the LOCAL sums (they are averaged with all other
# gradients afterwards), the input gradient from the global means.
_SYNC = {"on": False, "group": None}


def sync_state():
    """None when BatchNorm statistics are per rank, else the process group to reduce them over."""
    return (_SYNC["group"] or dist.group.WORLD) if (_SYNC["on"] and dist.is_initialized()) else None


def set_sync_bn(on, group=None):
    _SYNC["on"], _SYNC["group"] = bool(on), group


def all_reduce_stats(stats, group):
    """stats: double [2C+1] = (sum_0, sum_1, rows) of this rank -> the same over all ranks (in place)."""
    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


class _SyncBNFn(torch.autograd.Function):
    """y = batch_norm(x) over the rows of ALL ranks; plain torch ops (any device), fp64 sums."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, eps, momentum, group):
        c = x.shape[1]
        xd = x.double()
        stats = torch.cat([xd.sum(0), (xd * xd).sum(0), torch.full((1,), float(x.shape[0]), dtype=torch.float64,
                                                                    device=x.device)])
        all_reduce_stats(stats, group)
        cnt = stats[2 * c]
        mean = stats[:c] / cnt
        var = (stats[c:2 * c] / cnt - mean * mean).clamp_min(0)
        invstd = torch.rsqrt(var + eps)
        if rm is not None:
            with torch.no_grad():
                unb = var * cnt / (cnt - 1).clamp_min(1)
                rm.mul_(1 - momentum).add_((momentum * mean).to(rm.dtype))
                rv.mul_(1 - momentum).add_((momentum * unb).to(rv.dtype))
        xhat = ((xd - mean) * invstd)
        y = xhat * (gamma.double() if gamma is not None else 1.0) + (beta.double() if beta is not None else 0.0)
        ctx.save_for_backward(xhat.to(x.dtype), invstd, gamma)
        ctx.group, ctx.cnt = group, cnt
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xhat, invstd, gamma = ctx.saved_tensors
        c = dy.shape[1]
        dyd, xh = dy.double(), xhat.double()
        dz = dyd * (gamma.double() if gamma is not None else 1.0)
        local = torch.cat([dz.sum(0), (dz * xh).sum(0), torch.zeros(1, dtype=torch.float64, device=dy.device)])
        dbeta, dgamma = dyd.sum(0), (dyd * xh).sum(0)
        stats = all_reduce_stats(local.clone(), ctx.group)
        m1, m2 = stats[:c] / ctx.cnt, stats[c:2 * c] / ctx.cnt
        dx = invstd * (dz - m1 - xh * m2)
        return (dx.to(dy.dtype), dgamma.to(dy.dtype) if gamma is not None else None,
                dbeta.to(dy.dtype) if gamma is not None else None, None, None, None, None, None)


class SyncBatchNorm1d(torch.nn.BatchNorm1d):
    """torch.nn.BatchNorm1d on [N, C] inputs whose train-mode statistics span all ranks while `sync_state()` is set
    (torch.nn.SyncBatchNorm refuses CPU tensors; this one is plain torch ops -- used with gloo in the tests and as the
    readable statement of the protocol).  Same parameters / buffers / state_dict keys as BatchNorm1d."""

    def forward(self, x):
        group = sync_state()
        if group is None or not self.training or x.dim() != 2:
            return super().forward(x)
        if self.track_running_stats:
            self.num_batches_tracked.add_(1)
        mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
        rm, rv = (self.running_mean, self.running_var) if self.track_running_stats else (None, None)
        return _SyncBNFn.apply(x, self.weight, self.bias, rm, rv, self.eps, mom, group)


def convert_sync_batchnorm(module):
    """Swap every torch.nn.BatchNorm1d below `module` for SyncBatchNorm1d, in place, keeping its tensors."""
    for name, child in list(module.named_children()):
        if type(child) is torch.nn.BatchNorm1d:
            new = SyncBatchNorm1d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            for key, buf in child._buffers.items():          # the SAME tensors (device, dtype, identity), not copies
                new._buffers[key] = buf
            if child.affine:
                new.weight, new.bias = child.weight, child.bias
            new.train(child.training)
            setattr(module, name, new)
        else:
            convert_sync_batchnorm(child)
    return module


class FlatGradDataParallel:
    """Wraps a module: ``zero_grad()`` -> forward/backward as usual -> ``reduce_gradients()``.
    sync_bn=True: BatchNorm statistics over the global batch (see above); the statistics collectives run inside
    forward and backward; GraphedTrainStep then captures the WHOLE step, collectives included, in one graph.
    With graph_step.GraphedTrainStep(..., optimizer=opt, reducer=this) a data-parallel step is three host calls:
    replay (forward + loss + backward + gradient pack) -> one all-reduce -> replay (scale + optimizer update)."""

    def __init__(self, module, process_group=None, broadcast=True, always_reduce=False, sync_bn=False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.flat = None
        self.views = None
        self.always_reduce = always_reduce      # exercise the collective path on a single rank (tests)
        self.sync_bn = bool(sync_bn) and dist.is_initialized()
        if self.sync_bn:
            set_sync_bn(True, process_group)    # stays on for this process: backward runs outside __call__
        if broadcast and (self.world > 1 or (always_reduce and dist.is_initialized())):   # identical weights / buffers
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.detach(), 0, group=self.group)     # (detach() shares the version counter, .data does not:
                #  caches keyed on it -- inference BatchNorm maps, pre-split weight planes -- see the new contents)

    def __call__(self, *a, **kw):
        return self.module(*a, **kw)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def _ensure_flat(self, live):
        total = sum(p.numel() for p in live)
        if self.flat is None or self.flat.numel() != total:
            ref = live[0]
            self.flat = torch.empty(total, dtype=ref.dtype, device=ref.device)
        views, off = [], 0
        for p in live:
            views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return views

    # ---- the three pieces of reduce_gradients(), separately, for the two-graph step (graph_step.GraphedTrainStep with
    # reducer=...): pack (captured behind the backward pass) -> all-reduce (eager: a collective) -> scale + re-point
    # (the scale is captured in front of the optimizer update)
    def active(self):
        return self.world > 1 or (self.always_reduce and dist.is_initialized())

    def pack(self):
        """Copy the gradients into the flat buffer (one multi-tensor copy) and re-point .grad at its views."""
        live = [p for p in self.params if p.grad is not None]
        self.views = self._ensure_flat(live)
        torch._foreach_copy_(self.views, [p.grad for p in live])
        self.live = live
        return live

    def repoint(self):
        for p, v in zip(self.live, self.views):
            p.grad = v

    def all_reduce(self):
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def scale(self):
        self.flat.div_(self.world)

    def reduce_gradients(self):
        """Average gradients over ranks: one all-reduce of the flat buffer (no-op on 1 rank).
        Parameters that received no gradient (e.g. VectorNonLin.bias under BatchNorm, reference
        nn/nonlin.py:74-77) are skipped -- identically on every rank, since the model is replicated."""
        if self.world == 1 and not (self.always_reduce and dist.is_initialized()):
            return None
        self.pack()
        self.all_reduce()
        self.scale()
        self.repoint()
        return self.flat
