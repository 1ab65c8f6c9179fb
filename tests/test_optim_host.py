"""Host logic of deltaconv_amd.optim (no GPU): on CPU parameters both optimizers hand every group to torch's own step -- the
kernels never see a host pointer -- and keep torch's state_dict layout (Adam: one independent `step` per parameter, although the
GPU path shares one device counter per group).  Reference: experiments/train_modelnet.py:67, train_shapeseg.py:82."""
import copy

import torch

import deltaconv_amd as dc


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(5, 3, generator=g)), torch.nn.Parameter(torch.randn(7, generator=g))], g


def _step(ps_a, ps_b, oa, ob, g):
    for a, b in zip(ps_a, ps_b):
        gr = torch.randn(*a.shape, generator=g)
        a.grad, b.grad = gr.clone(), gr.clone()
    oa.step(); ob.step()


def test_sgd_on_cpu_parameters_is_torch_sgd():
    (pa, g), (pb, _) = _params(), _params()
    oa = dc.optim.SGD(pa, lr=0.1, momentum=0.9, weight_decay=1e-4)
    ob = torch.optim.SGD(pb, lr=0.1, momentum=0.9, weight_decay=1e-4)
    for _ in range(3):
        _step(pa, pb, oa, ob, g)
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
    assert oa.state_dict()["param_groups"][0]["momentum"] == 0.9


def test_adam_on_cpu_parameters_is_torch_adam_and_keeps_its_state_dict_layout():
    (pa, g), (pb, _) = _params(1), _params(1)
    oa, ob = dc.optim.Adam(pa, lr=5e-3), torch.optim.Adam(pb, lr=5e-3)
    sched = torch.optim.lr_scheduler.StepLR(oa, step_size=2, gamma=0.1)          # train_shapeseg.py:83
    for i in range(3):
        _step(pa, pb, oa, ob, g)
        sched.step()
        ob.param_groups[0]["lr"] = oa.param_groups[0]["lr"]
    assert all(torch.allclose(a, b, rtol=0, atol=1e-7) for a, b in zip(pa, pb))
    sd = oa.state_dict()
    steps = [st["step"] for st in sd["state"].values()]
    assert len(steps) == 2 and all(float(s) == 3.0 for s in steps)
    assert steps[0].data_ptr() != steps[1].data_ptr()                           # independent counters in the checkpoint
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ob2 = torch.optim.Adam(_params(1)[0], lr=5e-3)
    ob2.load_state_dict(copy.deepcopy(sd))                                      # a plain torch Adam takes the checkpoint
    oa2 = dc.optim.Adam(_params(1)[0], lr=5e-3)
    oa2.load_state_dict(copy.deepcopy(ob.state_dict()))                         # and ours takes torch's
    assert float(next(iter(oa2.state.values()))["step"]) == 3.0


def test_copy_many_and_split_cols_host_logic():
    """Pairs the copy kernel does not take (host tensors here) go through torch's copy; split_cols' backward assembles ONE
    gradient tensor from the two halves (or zeros for a half nobody used)."""
    from deltaconv_amd import _ops
    from deltaconv_amd.nn import fused
    a, b = torch.arange(12.).view(4, 3), torch.zeros(4, 5)
    y, yd = torch.arange(6), torch.zeros(6, dtype=torch.int64)
    _ops.copy_many([(a, b[:, :3]), (y, yd), (torch.zeros(0, 3), torch.zeros(0, 3))])
    assert torch.equal(b[:, :3], a) and float(b[:, 3:].abs().max()) == 0 and torch.equal(yd, y)
    w = torch.randn(6, 10, requires_grad=True)
    wa, wb = fused.split_cols(w, 4)
    assert wa.shape == (6, 4) and wb.shape == (6, 6)
    ga, gb = torch.randn(6, 4), torch.randn(6, 6)
    ((wa * ga).sum() + (wb * gb).sum()).backward()
    assert torch.equal(w.grad, torch.cat([ga, gb], 1))
    w.grad = None
    wa, wb = fused.split_cols(w, 4)
    (wb * gb).sum().backward()
    assert torch.equal(w.grad[:, 4:], gb) and float(w.grad[:, :4].abs().max()) == 0
    with fused.tn_batch():                       # nothing queued: nothing launched
        pass
