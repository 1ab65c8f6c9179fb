"""Optimizer step on the GPU in one launch.

``SGD`` is ``torch.optim.SGD`` (same constructor, same ``state_dict`` -- ``momentum_buffer`` per parameter -- same
``param_groups``, works with ``torch.optim.lr_scheduler``) whose ``step()`` runs ``dc_sgd_step`` (csrc/optim.hip): every
parameter of a group in one launch, the learning rate read from a device scalar so that a scheduler can move it between
replays of a captured training step (deltaconv_amd/graph_step.py) without a re-capture.  It is the optimizer of the
reference's classification / part-segmentation scripts (experiments/train_modelnet.py:67, train_scanobjectnn.py:77,
train_shapenet.py:95): momentum 0.9, weight decay 1e-4, no dampening, no Nesterov -- anything else (and non-fp32 / CPU /
sparse parameters) goes through torch's own ``step``.

``Adam`` is ``torch.optim.Adam`` (the optimizer of experiments/train_shapeseg.py:82) on ``dc_adam_step`` in the same way: one
launch for all parameters, ``state[p]["step"]`` ONE device scalar shared by the parameters of a group (torch's capturable
layout; ``state_dict`` keys and values as torch's), learning rate from a device scalar.  amsgrad / maximize groups go through
torch's own (capturable) ``step``.
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


def _lr_scalar(cache, group, device, who):
    lr = float(group["lr"])
    hit = cache.get(id(group))
    if hit is None or hit[0].device != device:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"{who}.step(): first step inside a graph capture -- run one eager step (or sync_lr()) before capturing")
        hit = [torch.full((), lr, dtype=torch.float32, device=device), lr]
        cache[id(group)] = hit
    elif hit[1] != lr:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"{who}.step(): the learning rate changed inside a graph capture")
        hit[0].fill_(lr)
        hit[1] = lr
    return hit[0]


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop("fused", None)
        self._capturable_arg = kw.pop("capturable", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._lr_dev = {}
        self._ticket = {}               # device -> int32 zero (the kernel's "last workgroup" counter)
        self._masters = {}              # id(group) -> the step counter its parameters share

    def add_param_group(self, param_group):
        """Groups of device parameters keep their step counters on the device (torch's `capturable` layout: its own step, where
        it runs instead of the kernel, is then graph-capturable too); host parameters stay with torch's defaults."""
        super().add_param_group(param_group)
        group = self.param_groups[-1]
        want = getattr(self, "_capturable_arg", None)
        group["capturable"] = bool(want) if want is not None else all(p.is_cuda for p in group["params"])

    def __setstate__(self, state):
        super().__setstate__(state)             # (load_state_dict lands here with the SAVED groups' flags)
        if getattr(self, "_capturable_arg", None) is None:
            for group in self.param_groups:
                group["capturable"] = all(p.is_cuda for p in group["params"])

    def _own_kernel(self, group):
        return (not group["amsgrad"] and not group.get("maximize", False) and not group.get("differentiable", False)
                and not isinstance(group["lr"], torch.Tensor)
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in group["params"]))

    def sync_lr(self):
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.is_cuda]
            if ps and not isinstance(group["lr"], torch.Tensor):
                _lr_scalar(self._lr_dev, group, ps[0].device, "Adam")

    def state_dict(self):
        """torch's layout with one INDEPENDENT step tensor per parameter: a plain torch.optim.Adam that loads the shared
        counter would advance it once per parameter and step."""
        sd = super().state_dict()
        sd["state"] = {k: (dict(v, step=v["step"].clone()) if isinstance(v, dict) and torch.is_tensor(v.get("step")) else v)
                       for k, v in sd["state"].items()}
        return sd

    def _shared_step(self, group, ps):
        """The ONE device step counter of the group: created with the first state, re-shared after a load_state_dict
        (which hands every parameter its own copy -- equal values, checked once, outside any capture)."""
        live = [self.state[p] for p in group["params"] if "step" in self.state[p]]
        master = self._masters.get(id(group))
        if master is not None and all(st["step"] is master for st in live):
            return master
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("Adam.step(): optimizer state is created / re-shared inside a graph capture -- run one eager step first")
        dev = ps[0].device
        if not live:
            master = torch.zeros((), dtype=torch.float32, device=dev)
        else:
            vals = torch.stack([st["step"].detach().to(device=dev, dtype=torch.float32).reshape(()) for st in live])
            if not bool((vals == vals[0]).all()):
                return None             # parameters at different step counts: torch's per-parameter step
            master = vals[0].clone()
        for st in live:
            st["step"] = master
        self._masters[id(group)] = master
        return master

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        leftover = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not self._own_kernel(group) or any(p.grad.is_sparse for p in ps):
                leftover.append(group)
                continue
            if not ps:
                continue
            master = self._shared_step(group, ps)
            if master is None:
                leftover.append(group)
                continue
            for p in ps:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = master
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            dev = ps[0].device
            if dev not in self._ticket:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("Adam.step(): first step inside a graph capture -- run one eager step before capturing")
                self._ticket[dev] = torch.zeros((), dtype=torch.int32, device=dev)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            n = len(ps)
            arr = lambda vals: (ctypes.c_int64 * n)(*vals)
            lr = _lr_scalar(self._lr_dev, group, dev, "Adam")
            b1, b2 = group["betas"]
            rc = lib.raw("dc_adam_step")(arr(p.data_ptr() for p in ps), arr(g.data_ptr() for g in grads),
                                         arr(self.state[p]["exp_avg"].data_ptr() for p in ps),
                                         arr(self.state[p]["exp_avg_sq"].data_ptr() for p in ps), arr(p.numel() for p in ps), n,
                                         lr.data_ptr(), master.data_ptr(), self._ticket[dev].data_ptr(), float(b1), float(b2),
                                         float(group["eps"]), float(group["weight_decay"]), torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError(f"dc_adam_step failed (rc={rc}): {lib.last_error()}")
        if leftover:
            groups = self.param_groups
            try:
                self.param_groups = leftover
                super().step()
            finally:
                self.param_groups = groups
        return loss
