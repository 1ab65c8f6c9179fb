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


class FlatGradDataParallel:
    """Wraps a module: ``zero_grad()`` -> forward/backward as usual -> ``reduce_gradients()``."""

    def __init__(self, module, process_group=None, broadcast=True, always_reduce=False):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.flat = None
        self.views = None
        self.always_reduce = always_reduce      # exercise the collective path on a single rank (tests)
        if broadcast and (self.world > 1 or (always_reduce and dist.is_initialized())):   # identical weights / buffers
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=self.group)

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

    def reduce_gradients(self):
        """Average gradients over ranks: one all-reduce of the flat buffer (no-op on 1 rank).
        Parameters that received no gradient (e.g. VectorNonLin.bias under BatchNorm, reference
        nn/nonlin.py:74-77) are skipped -- identically on every rank, since the model is replicated."""
        if self.world == 1 and not (self.always_reduce and dist.is_initialized()):
            return None
        live = [p for p in self.params if p.grad is not None]
        views = self._ensure_flat(live)
        torch._foreach_copy_(views, [p.grad for p in live])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        for p, v in zip(live, views):
            p.grad = v
        return self.flat
