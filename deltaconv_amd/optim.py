"""Optimizer step on the GPU in one launch.

``SGD`` is ``torch.optim.SGD`` (same constructor, same ``state_dict`` -- ``momentum_buffer`` per parameter -- same
``param_groups``, works with ``torch.optim.lr_scheduler``) whose ``step()`` runs ``dc_sgd_step`` (csrc/optim.hip): every
parameter of a group in one launch, the learning rate read from a device scalar so that a scheduler can move it between
replays of a captured training step (deltaconv_amd/graph_step.py) without a re-capture.  It is the optimizer of the
reference's classification / part-segmentation scripts (experiments/train_modelnet.py:67, train_scanobjectnn.py:77,
train_shapenet.py:95): momentum 0.9, weight decay 1e-4, no dampening, no Nesterov -- anything else (and non-fp32 / CPU /
sparse parameters) goes through torch's own ``step``.
"""
import ctypes

import torch

from ._lib import lib


class SGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False, **kw):
        kw.pop("fused", None)           # this IS the fused form; torch's flag would only select its multi-tensor kernel
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov, **kw)
        self._lr_dev = {}               # id(group) -> [device scalar, the value it holds]

    def _own_kernel(self, group):
        return (group["dampening"] == 0 and not group["nesterov"] and not group.get("maximize", False)
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in group["params"]))

    def _lr_scalar(self, group, device):
        lr = float(group["lr"])
        hit = self._lr_dev.get(id(group))
        if hit is None or hit[0].device != device:
            if torch.cuda.is_current_stream_capturing():      # a captured fill would rewrite the scalar in every replay
                raise RuntimeError("SGD.step(): first step inside a graph capture -- run one eager step (or sync_lr()) before capturing")
            hit = [torch.full((), lr, dtype=torch.float32, device=device), lr]
            self._lr_dev[id(group)] = hit
        elif hit[1] != lr:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("SGD.step(): the learning rate changed inside a graph capture")
            hit[0].fill_(lr)            # a scheduler moved it: one tiny launch, outside any captured graph
            hit[1] = lr
        return hit[0]

    def sync_lr(self):
        """Write the groups' current learning rates to their device scalars (call after scheduler.step() when the
        optimizer step itself only runs inside graph replays)."""
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.is_cuda]
            if ps:
                self._lr_scalar(group, ps[0].device)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        leftover = False
        for group in self.param_groups:
            if not self._own_kernel(group):
                leftover = True
                continue
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            bufs = []
            for p in ps:
                st = self.state[p]
                if st.get("momentum_buffer") is None:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)   # first step: buf = g'
                bufs.append(st["momentum_buffer"])
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            n = len(ps)
            arr = lambda vals: (ctypes.c_int64 * n)(*vals)
            lr = self._lr_scalar(group, ps[0].device)
            rc = lib.raw("dc_sgd_step")(arr(p.data_ptr() for p in ps), arr(g.data_ptr() for g in grads),
                                        arr(b.data_ptr() for b in bufs), arr(p.numel() for p in ps), n, lr.data_ptr(),
                                        float(group["momentum"]), float(group["weight_decay"]),
                                        torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError(f"dc_sgd_step failed (rc={rc}): {lib.last_error()}")
        if leftover:                    # groups the kernel does not cover: torch's step on those groups only
            groups = self.param_groups
            try:
                self.param_groups = [g for g in groups if not self._own_kernel(g)]
                super().step()
            finally:
                self.param_groups = groups
        return loss
