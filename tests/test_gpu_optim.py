"""deltaconv_amd.optim.SGD (csrc/optim.hip: dc_sgd_step) against torch.optim.SGD -- the optimizer of the reference's
training scripts (experiments/train_modelnet.py:67: lr 0.1, momentum 0.9, weight decay 1e-4, cosine schedule).
Tolerance: 1e-6 of the parameter scale per step (fused multiply-adds here, separate roundings in ATen)."""
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(1,), (3,), (5, 7), (4097,), (64, 64), (1024, 512), (40, 256), (256,)]


def _pair(shapes, seed=0, **kw):
    import deltaconv_amd as dc
    g = torch.Generator().manual_seed(seed)
    init = [torch.randn(*s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    return pa, pb, dc.optim.SGD(pa, **kw), torch.optim.SGD(pb, **kw), g


@pytest.mark.parametrize("kw", [dict(lr=0.1, momentum=0.9, weight_decay=1e-4), dict(lr=0.01, momentum=0.0, weight_decay=0.0),
                                dict(lr=0.5, momentum=0.5, weight_decay=1e-2)])
def test_sgd_matches_torch_over_steps(kw):
    pa, pb, oa, ob, g = _pair(SHAPES, **kw)
    sched_a = torch.optim.lr_scheduler.CosineAnnealingLR(oa, 6, eta_min=0.001)
    sched_b = torch.optim.lr_scheduler.CosineAnnealingLR(ob, 6, eta_min=0.001)
    for step in range(6):
        for a, b in zip(pa, pb):
            gr = torch.randn(*a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        sched_a.step(); sched_b.step()                      # the learning rate moves between the steps
        for a, b in zip(pa, pb):
            assert rel_err(a, b) < 2e-6, (step, tuple(a.shape))
    if kw["momentum"] != 0.0:                               # (torch keeps no buffer without momentum)
        for a, b in zip(pa, pb):
            assert rel_err(oa.state[a]["momentum_buffer"], ob.state[b]["momentum_buffer"]) < 2e-6


def test_sgd_many_tensors_unaligned_views_and_missing_grads():
    """> 96 tensors (two launches), parameters that are unaligned views of a larger buffer (scalar path), parameters without a
    gradient (skipped, like torch)."""
    import deltaconv_amd as dc
    g = torch.Generator().manual_seed(1)
    base_a = torch.randn(200 * 37 + 3, generator=g).to(DEV)
    base_b = base_a.clone()
    pa = [torch.nn.Parameter(base_a[1 + i * 37:1 + i * 37 + 35]) for i in range(200)]
    pb = [torch.nn.Parameter(base_b[1 + i * 37:1 + i * 37 + 35]) for i in range(200)]
    oa = dc.optim.SGD(pa, lr=0.1, momentum=0.9, weight_decay=1e-4)
    ob = torch.optim.SGD(pb, lr=0.1, momentum=0.9, weight_decay=1e-4)
    for step in range(3):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i % 7 == 3:
                a.grad = b.grad = None
                continue
            gr = torch.randn(35, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    assert rel_err(base_a, base_b) < 2e-6
    assert torch.equal(base_a[:1], base_b[:1]) and torch.equal(base_a[-2:], base_b[-2:])       # nothing outside the views moved


def test_sgd_state_dict_and_fallback_groups():
    """state_dict round trip with torch.optim.SGD in both directions; a Nesterov group runs torch's own step."""
    import deltaconv_amd as dc
    pa, pb, oa, ob, g = _pair([(33,), (8, 8)], lr=0.1, momentum=0.9, weight_decay=1e-4)
    for a, b in zip(pa, pb):
        gr = torch.randn(*a.shape, generator=g).to(DEV)
        a.grad, b.grad = gr.clone(), gr.clone()
    oa.step(); ob.step()
    import copy
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))      # ours -> torch (deep copy: load_state_dict adopts same-dtype tensors)
    oa.load_state_dict(copy.deepcopy(ob.state_dict()))      # and back
    for a, b in zip(pa, pb):
        gr = torch.randn(*a.shape, generator=g).to(DEV)
        a.grad, b.grad = gr.clone(), gr.clone()
    oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert rel_err(a, b) < 2e-6
    qa, qb = torch.nn.Parameter(torch.ones(5, device=DEV)), torch.nn.Parameter(torch.ones(5, device=DEV))
    na = dc.optim.SGD([qa], lr=0.1, momentum=0.9, nesterov=True)
    nb = torch.optim.SGD([qb], lr=0.1, momentum=0.9, nesterov=True)
    qa.grad, qb.grad = torch.full((5,), 2.0, device=DEV), torch.full((5,), 2.0, device=DEV)
    na.step(); nb.step()
    assert torch.equal(qa, qb)


def test_sgd_learning_rate_moves_between_graph_replays(monkeypatch):
    """The captured step reads the learning rate from a device scalar: a scheduler step between replays takes effect
    without a re-capture."""
    import deltaconv_amd as dc
    p = torch.nn.Parameter(torch.zeros(1000, device=DEV))
    p.grad = torch.ones(1000, device=DEV)
    opt = dc.optim.SGD([p], lr=0.5, momentum=0.0)
    opt.step()                                              # eager first step: state + device scalar exist
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        opt.step()
    before = p.detach().clone()
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(p, before - 0.5)
    opt.param_groups[0]["lr"] = 0.125
    opt.sync_lr()
    before = p.detach().clone()
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(p, before - 0.125)
    q = torch.nn.Parameter(torch.zeros(4, device=DEV))
    q.grad = torch.ones(4, device=DEV)
    fresh = dc.optim.SGD([q], lr=0.1)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)      # (no real capture: nothing to poison)
    with pytest.raises(RuntimeError, match="before capturing"):
        fresh.step()


def test_copy_many_one_launch_for_the_batch_load_and_the_first_layer_blocks():
    """dc_copy_many (csrc/optim.hip): several small, possibly row-strided copies of 4- / 8-byte elements in one launch == torch's
    copies; pairs it does not take (1-byte elements, host tensors) fall through to torch, empty ones are skipped."""
    from deltaconv_amd import _ops
    g = torch.Generator().manual_seed(0)
    n = 5000
    pos, wide = torch.rand(n, 3, generator=g).to(DEV), torch.zeros(n, 12, device=DEV)
    v, vpad = torch.rand(2 * n, 3, generator=g).to(DEV), torch.zeros(2 * n, 72, device=DEV)[:, 2:]
    y, ydst = torch.randint(0, 40, (32,), generator=g).to(DEV), torch.zeros(32, dtype=torch.int64, device=DEV)
    big, bigdst = torch.rand(3000, 130, generator=g).to(DEV)[:, 1:129], torch.zeros(3000, 128, device=DEV)
    i32, i32dst = torch.randint(0, 9, (777,), generator=g, dtype=torch.int32).to(DEV), torch.zeros(777, dtype=torch.int32, device=DEV)
    u8, u8dst = torch.randint(0, 255, (100, 7), generator=g, dtype=torch.uint8).to(DEV), torch.zeros(100, 7, dtype=torch.uint8, device=DEV)
    host, hostdst = torch.rand(10, 3, generator=g), torch.zeros(10, 3, device=DEV)
    empty, emptydst = torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV)
    pairs = [(pos, wide[:, :3]), (v, vpad[:, :3]), (y, ydst), (big, bigdst), (i32, i32dst), (u8, u8dst), (host, hostdst),
             (empty, emptydst)]
    _ops.copy_many(pairs)
    torch.cuda.synchronize()
    for src, dst in pairs:
        assert torch.equal(dst.cpu(), src.cpu())
    assert float(wide[:, 3:].abs().max()) == 0 and float(vpad[:, 3:].abs().max()) == 0      # nothing beyond the blocks
    # more pairs than one table holds (16 per launch)
    many = [(torch.full((50, 5), float(i), device=DEV), torch.zeros(50, 8, device=DEV)[:, :5]) for i in range(40)]
    _ops.copy_many(many)
    assert all(torch.equal(d, s) for s, d in many)
    # capturable: the copies of a captured graph replay with new source contents
    src, dst = torch.zeros(1000, 3, device=DEV), torch.zeros(1000, 4, device=DEV)
    _ops.copy_many([(src, dst[:, :3])])
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        _ops.copy_many([(src, dst[:, :3])])
    src.fill_(3.0)
    gr.replay()
    torch.cuda.synchronize()
    assert float(dst[:, :3].min()) == 3.0 and float(dst[:, 3].max()) == 0.0


# ---- Adam (experiments/train_shapeseg.py:82) -----------------------------------------------------------------------------------
def _adam_pair(shapes, seed=0, **kw):
    import deltaconv_amd as dc
    g = torch.Generator().manual_seed(seed)
    init = [torch.randn(*s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    return pa, pb, dc.optim.Adam(pa, **kw), torch.optim.Adam(pb, foreach=False, **kw), g      # torch's single-tensor form


def _grads(pa, pb, g, scale=1.0):
    for a, b in zip(pa, pb):
        gr = (torch.randn(*a.shape, generator=g) * scale).to(DEV)
        a.grad, b.grad = gr.clone(), gr.clone()


@pytest.mark.parametrize("kw", [dict(lr=5e-3), dict(lr=1e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=1e-2)])
def test_adam_matches_torch_over_steps(kw):
    """dc_adam_step vs torch.optim.Adam (the reference's default, single-tensor form) over 8 steps with a StepLR schedule
    (train_shapeseg.py:83).  Tolerance: 2e-6 of the parameter scale per step for the parameters, 1e-6 for the moments."""
    pa, pb, oa, ob, g = _adam_pair(SHAPES, **kw)
    sa, sb = torch.optim.lr_scheduler.StepLR(oa, 3, gamma=0.1), torch.optim.lr_scheduler.StepLR(ob, 3, gamma=0.1)
    for step in range(8):
        _grads(pa, pb, g, scale=10.0 ** (step % 3 - 1))
        oa.step(); ob.step()
        sa.step(); sb.step()
        for a, b in zip(pa, pb):
            assert rel_err(a, b) < 2e-6 * (step + 1), (step, tuple(a.shape))
    for a, b in zip(pa, pb):
        assert rel_err(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"]) < 1e-6
        assert rel_err(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]) < 1e-6
        assert float(oa.state[a]["step"]) == 8.0 == float(ob.state[b]["step"])
    assert len({id(oa.state[a]["step"]) for a in pa}) == 1                   # ONE device counter for the group


def test_adam_many_tensors_missing_grads_state_dict_and_fallback():
    """> 80 tensors (two launches, the step counter moves once), unaligned views (scalar path), parameters without a gradient;
    state_dict round trip with torch.optim.Adam in both directions (the loaded per-parameter counters are re-shared); an
    amsgrad group runs torch's own step."""
    import copy
    import deltaconv_amd as dc
    g = torch.Generator().manual_seed(1)
    base_a = torch.randn(170 * 37 + 3, generator=g).to(DEV)
    base_b = base_a.clone()
    pa = [torch.nn.Parameter(base_a[1 + i * 37:1 + i * 37 + 35]) for i in range(170)]
    pb = [torch.nn.Parameter(base_b[1 + i * 37:1 + i * 37 + 35]) for i in range(170)]
    oa, ob = dc.optim.Adam(pa, lr=5e-3), torch.optim.Adam(pb, lr=5e-3, foreach=False)
    skip = lambda i: i % 7 == 3
    for step in range(3):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if skip(i):
                a.grad = b.grad = None
                continue
            gr = torch.randn(35, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    assert rel_err(base_a, base_b) < 5e-6
    assert torch.equal(base_a[:1], base_b[:1]) and torch.equal(base_a[-2:], base_b[-2:])
    assert float(oa.state[pa[0]]["step"]) == 3.0
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))       # ours -> torch
    oa.load_state_dict(copy.deepcopy(ob.state_dict()))       # torch -> ours: every parameter comes back with its own counter
    for i, (a, b) in enumerate(zip(pa, pb)):
        if not skip(i):
            gr = torch.randn(35, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
    oa.step(); ob.step()
    assert rel_err(base_a, base_b) < 5e-6
    assert float(oa.state[pa[0]]["step"]) == 4.0 and len({id(oa.state[a]["step"]) for i, a in enumerate(pa) if not skip(i)}) == 1
    qa, qb = torch.nn.Parameter(torch.ones(5, device=DEV)), torch.nn.Parameter(torch.ones(5, device=DEV))
    na, nb = dc.optim.Adam([qa], lr=0.1, amsgrad=True), torch.optim.Adam([qb], lr=0.1, amsgrad=True)
    qa.grad, qb.grad = torch.full((5,), 2.0, device=DEV), torch.full((5,), 2.0, device=DEV)
    na.step(); nb.step()
    assert rel_err(qa, qb) < 1e-6


def test_adam_in_a_captured_step():
    """Replays of a captured step advance the shared device counter (bias corrections move) and read the learning rate from
    its device scalar; the first step must be eager."""
    import deltaconv_amd as dc
    pa, pb, oa, ob, g = _adam_pair([(1000,), (64, 33)], lr=5e-3)
    _grads(pa, pb, g)
    oa.step(); ob.step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        oa.step()
    ob.step()                                               # (the capture itself runs nothing)
    for rep in range(3):
        if rep == 2:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 1e-3
            oa.sync_lr()
        gr.replay()
        if rep:
            ob.step()
        torch.cuda.synchronize()
    # both sides: one eager step, then three more with the same gradients, the last one at the lower learning rate
    assert float(oa.state[pa[0]]["step"]) == 4.0 and float(ob.state[pb[0]]["step"]) == 4.0
    for a, b in zip(pa, pb):
        assert rel_err(a, b) < 1e-5
