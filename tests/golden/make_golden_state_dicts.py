"""Key names + shapes + dtypes of the REAL reference models' ``state_dict()`` (what ``torch.save(model.state_dict())``
at experiments/train_modelnet.py:84 writes), for the strict load test.  Runs only in the build container:

    python tests/golden/make_golden_state_dicts.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tools", "ref_shims"), "/root/reference", ROOT]

from deltaconv.models import DeltaNetClassification, DeltaNetSegmentation   # noqa: E402  (the reference)

CASES = {
    # the four experiment configurations (experiments/train_{modelnet,scanobjectnn,shapenet,shapeseg}.py)
    "modelnet40": ("cls", dict(in_channels=3, num_classes=40, conv_channels=[64, 64, 128, 256], num_neighbors=20,
                               grad_regularizer=1e-3, grad_kernel_width=1)),
    "scanobjectnn": ("cls", dict(in_channels=3, num_classes=15, conv_channels=[64, 64, 64, 128], num_neighbors=20,
                                 grad_regularizer=1e-2, grad_kernel_width=1)),
    "shapenet": ("seg", dict(in_channels=3, num_classes=50, categorical_vector=True, num_neighbors=20)),
    "shapeseg": ("seg", dict(in_channels=3, num_classes=8, conv_channels=[128] * 8, mlp_depth=1, embedding_size=512,
                             num_neighbors=30)),
}

if __name__ == "__main__":
    out = {}
    for name, (kind, kw) in CASES.items():
        torch.manual_seed(1)
        m = (DeltaNetSegmentation if kind == "seg" else DeltaNetClassification)(**kw)
        out[name] = dict(kind=kind, kwargs=kw, repr_lines=len(repr(m).splitlines()),
                         entries=[[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()])
        print(name, len(out[name]["entries"]), "entries")
    with open(os.path.join(HERE, "state_dicts.json"), "w") as f:
        json.dump(out, f, indent=0)
