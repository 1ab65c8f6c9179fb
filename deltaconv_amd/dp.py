"""Data-parallel training over the GPUs of one node: one process per GPU, replicated weights,
clouds sharded across ranks (they are independent units: kNN, operators, convolutions and pools
never cross a ``batch`` boundary), ONE flat fp32 gradient all-reduce per step over RCCL/xGMI
(``torch.distributed`` backend "nccl"; "gloo" on CPU in the tests).

The reference has no multi-device code at all; this is the new capability
BASELINE.json asks for.  Design for xGMI: the whole gradient is 5.8-8.9 MB, so a single collective
on a pre-flattened buffer (parameters' .grad are views into it: no flatten/unflatten copies) is
latency- not bandwidth-bound; BatchNorm statistics stay per rank (standard DDP semantics).
"""
import torch
import torch.distributed as dist


class FlatGradDataParallel:
    """Wraps a module: ``zero_grad()`` -> forward/backward as usual -> ``reduce_gradients()``."""

    def __init__(self, module, process_group=None, broadcast=True):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:                      # .grad becomes a view into the flat buffer
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        if broadcast and self.world > 1:           # identical initial weights and BN buffers
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=self.group)

    def __call__(self, *a, **kw):
        return self.module(*a, **kw)

    def zero_grad(self):
        self.flat.zero_()
        for p in self.params:                      # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * self.flat.element_size():
                self._reattach()
                break

    def _reattach(self):
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def reduce_gradients(self):
        """Average gradients over ranks: one all-reduce of the flat buffer (no-op on 1 rank)."""
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world)
        return self.flat
