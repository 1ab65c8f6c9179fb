"""One training step (forward + loss + backward, optionally the optimizer update) captured once in a
HIP graph and replayed with a single host call.

Why: at 32 clouds per GPU a step is ~230 kernel launches of ~17 us each -- the Python / autograd /
ctypes enqueue time (4.3 ms) has caught up with the GPU time (see tools/archive_r01_r04.tar.gz:archive/cpu_overhead.py).  Every entry
point of the C ABI is capture-safe by construction (stream-ordered kernels only: no allocation, no
sync, no memset/memcpy nodes; workspaces come from torch's allocator, which serves captures from a
private pool), so the whole step replays from one `hipGraphLaunch`.  Shapes are static (equal-size
clouds, fixed batch): a new batch is copied into the captured input tensors.

Semantics kept per replay: kNN / MLS operators are rebuilt from the current `pos`, BatchNorm running
statistics and `num_batches_tracked` advance, Dropout draws a fresh mask (torch's graph-safe Philox
offsets), gradients land in the same `.grad` tensors every time (do NOT call zero_grad(set_to_none)
between replays -- the captured backward overwrites them).  The gradient all-reduce (RCCL) stays
outside the graph; the optimizer update is captured only when `optimizer` is given (single-rank case,
constant hyper-parameters: a learning-rate change needs `recapture()`).

ROCm note (measured on ROCm 7.2 / MI355X, tools/graph_bisect*.py history in DESIGN.md): with the
runtime's AQL-packet capture (`DEBUG_CLR_GRAPH_PACKET_CAPTURE`, default on) a replay that follows a
handful of small eager launches reads stale kernel arguments (wrong results, or a memory fault when
the graph holds memset nodes).  The flag must be 0 before the HIP runtime initialises;
`deltaconv_amd/__init__.py` sets that default at import and this class refuses to run without it."""
import os

import torch

from . import _ops
from .nn.fused import invalidate_eval_coeffs, planes_snapshot

_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


class GraphedTrainStep:
    def __init__(self, model, loss_fn, sample_batch, optimizer=None, warmup=3, reducer=None, capture_collectives=None):
        """sample_batch: a deltaconv_amd.Batch on the GPU whose tensors become the static inputs.
        optimizer: captured into the graph when given (its first `warmup` steps are real updates, so
        lazily created state such as momentum buffers exists before the capture).
        reducer: a dp.FlatGradDataParallel whose all-reduce runs between TWO captured graphs -- graph A = forward +
        loss + backward + the gradient pack into the flat buffer, graph B = the 1/world scale + the optimizer update
        on views of that buffer: a data-parallel step is replay -> all-reduce -> replay, three host calls instead of
        the eager pack / div / multi-tensor-update launches behind every replay (1-2 clouds per rank: host-bound).
        capture_collectives: ONE graph holding the whole data-parallel step, collectives included -- the statistics
        all-reduces of synchronised BatchNorm inside forward and backward, the gradient all-reduce, the scale and the
        optimizer update (round 4: RCCL collectives are capturable when the capture runs in `thread_local` error mode --
        the process group's watchdog thread queries events, which a global-mode capture forbids).  Default: on when the
        reducer synchronises BatchNorm statistics (that step cannot be split around its ~40 collectives: eager it is
        host-bound at 9-13 ms for one or two clouds per rank, captured 2.8-4.5 ms: profiles/r04_dp_proxy.txt), off
        otherwise (the two-graph form keeps the one gradient all-reduce an ordinary eager collective)."""
        if os.environ.get(_FLAG, "1") != "0":
            raise RuntimeError(f"GraphedTrainStep needs {_FLAG}=0 in the environment before the HIP runtime "
                               "starts (import deltaconv_amd before the first torch.cuda call, or export it)")
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.reducer = reducer if (reducer is not None and reducer.active()) else None
        if self.reducer is not None:
            assert optimizer is not None, "the data-parallel graph step captures the optimizer update"
        if capture_collectives is None:
            capture_collectives = bool(self.reducer is not None and getattr(self.reducer, "sync_bn", False))
        self.capture_collectives = bool(capture_collectives and self.reducer is not None)
        self.static = sample_batch
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.warmup = warmup
        self._one = None                        # the seed of the backward pass, made once (see _step)
        self.recapture()

    def recapture(self):
        if self.reducer is not None and self.reducer.flat is None and self.warmup == 0:
            self.warmup = 1                     # the flat gradient buffer must exist before the capture
        if hasattr(self.static, "pos") and hasattr(self.static, "batch"):
            from .models.deltanet_base import _ptr_info
            _ptr_info(self.static)              # cloud offsets of the static batch: a host read, never inside the capture
        if self._one is None:                   # made outside the capture (inside, it would be a captured fill again)
            self._one = torch.ones((), dtype=torch.float32, device=self.params[0].device)
            self._one._dc_unit_seed = True      # utils._CELoss.backward: d loss / d logits needs no scaling by this seed
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up outside the capture (allocator, tuning)
            for _ in range(self.warmup):
                self._zero()
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._zero()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_update = None
        if self.capture_collectives:            # the whole data-parallel step, collectives included, in one graph
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss, self.out = self._step(capture=False)
            torch.cuda.synchronize()
            self.flat = self.reducer.flat
            self._planes_keepalive = planes_snapshot()      # raw addresses inside the graph (advisor, round 4)
            invalidate_eval_coeffs()
            self.grads = [(p, p.grad) for p in self.params if p.grad is not None]
            return
        # with a process group alive its watchdog thread may query an event of the last warm-up all-reduce while this thread
        # captures: a global-mode capture would be invalidated by that call (seen with the captured collectives above)
        mode = "thread_local" if self.reducer is not None else "global"
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.loss, self.out = self._step(capture=True)
        torch.cuda.synchronize()
        if self.reducer is not None:            # graph B: scale + update, reading .grad = views of the flat buffer
            # the graphs hold the ADDRESS of the flat buffer that exists now: keep it, and notice when an eager
            # reduce_gradients() / pack() with another parameter set made the reducer allocate a new one
            self.flat = self.reducer.flat
            self.reducer.repoint()
            self.graph_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_update, capture_error_mode=mode):
                self.reducer.scale()
                self.optimizer.step()
            torch.cuda.synchronize()
        # the captured dc_presplit_weights and the plane-fed products hold RAW addresses of the record / chunk tables and of
        # every plane buffer: this graph owns a reference to each, whatever later eager passes do to the cache (advisor, round 4)
        self._planes_keepalive = planes_snapshot()
        invalidate_eval_coeffs()
        self.grads = [(p, p.grad) for p in self.params if p.grad is not None]   # rewritten by every replay

    def _zero(self):
        for p in self.params:
            p.grad = None

    def _step(self, capture=False):
        out = self.model(self.static)
        loss = self.loss_fn(out, self.static.y)
        if loss.dim() == 0 and loss.is_floating_point():
            # loss.backward() seeds the pass with ones_like(loss): a fill launch in every replay (~5 us of launch floor)
            if self._one is None or self._one.dtype != loss.dtype or self._one.device != loss.device:
                self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
                self._one._dc_unit_seed = True
            torch.autograd.backward(loss, grad_tensors=(self._one,))
        else:
            loss.backward()
        if self.reducer is not None:
            self.reducer.pack()                 # captured: one multi-tensor copy into the flat buffer
            if not capture:                     # warm-up steps: the rest of the reduction + the update, eagerly
                self.reducer.all_reduce()
                self.reducer.scale()
                self.reducer.repoint()
                self.optimizer.step()
        elif self.optimizer is not None:
            self.optimizer.step()
        return loss.detach(), out.detach()

    def load(self, batch):
        """Copy a new batch (same shapes) into the captured inputs."""
        s = self.static
        if batch is s:
            return
        pairs = []
        for name in ("pos", "norm", "x", "y", "category"):
            dst, src = getattr(s, name, None), getattr(batch, name, None)
            if dst is not None:
                assert src is not None and src.shape == dst.shape, f"batch.{name}: static shape {tuple(dst.shape)}"
                pairs.append((src, dst))
        # all tensors of the batch in ONE launch of the own copy kernel (csrc/optim.hip: dc_copy_many; host tensors and other
        # layouts fall through to torch's copy).  torch._foreach_copy_ was tried (round 6) and is SLOWER than three plain
        # copies here: mixed dtypes / sizes take its slow path, 38 us of host gap in front of each of its launches
        # (profiles/r06g_step_timeline.txt)
        _ops.copy_many(pairs)

    def __call__(self, batch=None):
        if batch is not None:
            self.load(batch)
        if self.reducer is not None and self.reducer.flat is not self.flat:      # both forms bake the buffer's address in
            raise RuntimeError("GraphedTrainStep: the gradient reducer re-allocated its flat buffer after capture (an eager "
                               "reduce_gradients() with a different set of live parameters); the captured graphs still use "
                               "the old one -- build a new GraphedTrainStep")
        self.graph.replay()
        if self.graph_update is not None:
            self.reducer.all_reduce()           # the one collective of the step, between the two replays
            self.graph_update.replay()
        invalidate_eval_coeffs()      # the replay moved running statistics (and parameters) without a version bump
        for p, g in self.grads:       # a gradient reducer may have re-pointed .grad at its own buffer
            p.grad = g
        return self.loss
