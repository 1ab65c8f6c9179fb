"""Golden vectors for the host-side transforms: run the REAL reference transforms (pure torch, imported
from /root/reference with the stand-ins of tools/ref_shims) on seeded inputs.  Build container only."""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tools", "ref_shims"), "/root/reference", ROOT]
import deltaconv.transforms as RT   # noqa: E402
from types import SimpleNamespace as NS  # noqa: E402


def mesh(seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.randn(40, 3, generator=g) * torch.tensor([1.0, 2.0, 0.5])
    face = torch.randint(0, 40, (3, 60), generator=g)
    face = face[:, (face[0] != face[1]) & (face[1] != face[2]) & (face[0] != face[2])]
    nrm = torch.nn.functional.normalize(torch.randn(40, 3, generator=g), dim=1)
    return pos, face, nrm


from tests.golden.make_golden_transforms_cases import CASES  # noqa: E402

out = {}
for ci, (name, kw) in enumerate(CASES):
    pos, face, nrm = mesh(100 + ci)
    d = NS(pos=pos.clone(), norm=nrm.clone(), face=(face.t().contiguous() if name == "NormalizeArea" else face.clone()),
           y=torch.arange(40))
    torch.manual_seed(7 + ci)
    random.seed(7 + ci)
    t = getattr(RT, name)(**kw)
    r = t(d)
    out[f"{ci}_pos_in"], out[f"{ci}_face_in"], out[f"{ci}_norm_in"] = pos.numpy(), face.numpy(), nrm.numpy()
    out[f"{ci}_pos"] = r.pos.numpy()
    out[f"{ci}_norm"] = r.norm.numpy()
    out[f"{ci}_repr"] = np.array(repr(t))
np.savez_compressed(os.path.join(HERE, "transforms.npz"), **out)
print("wrote transforms.npz", len(out))
