"""Seeded probe vectors / gradient summaries shared by the golden generator and the tests."""
import numpy as np
import torch


def probe_vec(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float64)


def param_summaries(model):
    """Two numbers per parameter gradient: L2 norm and dot with a seeded probe vector."""
    names, norms, dots = [], [], []
    for i, (n, p) in enumerate(model.named_parameters()):
        if p.grad is None:
            continue
        names.append(n)
        g = p.grad.detach().double().cpu()
        norms.append(float(g.norm()))
        dots.append(float((g.flatten() * probe_vec(g.numel(), 7000 + i)).sum()))
    return names, norms, dots


def state_checksum(model):
    return np.array([float(t.double().abs().sum()) for t in model.state_dict().values()])
