"""Shared test helpers."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" and a.ndim > 0 else (a.item() if a.ndim == 0 else a)
    return out


def rel_err(a, b):
    """max |a-b| / max |b| -- the scale-relative error every tolerance in tests/ is stated in."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a.detach() - b.detach()).abs().max() / b.abs().max().clamp(min=1e-30))
