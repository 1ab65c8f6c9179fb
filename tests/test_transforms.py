"""Host-side transforms vs golden vectors produced by the reference's own transforms
(tests/golden/make_golden_transforms.py): same seed -> same sample.  CPU only."""
import os
import random
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN
from tests.golden.make_golden_transforms_cases import CASES


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_transform_matches_reference(ci):
    import deltaconv_amd.transforms as T
    z = np.load(os.path.join(GOLDEN, "transforms.npz"), allow_pickle=False)
    name, kw = CASES[ci]
    pos, face, nrm = (torch.from_numpy(z[f"{ci}_{k}_in"]) for k in ("pos", "face", "norm"))
    d = NS(pos=pos.clone(), norm=nrm.clone(), face=(face.t().contiguous() if name == "NormalizeArea" else face.clone()),
           y=torch.arange(40))
    torch.manual_seed(7 + ci)
    random.seed(7 + ci)
    t = getattr(T, name)(**kw)
    r = t(d)
    assert repr(t) == str(z[f"{ci}_repr"])
    assert torch.allclose(r.pos, torch.from_numpy(z[f"{ci}_pos"]), rtol=1e-6, atol=1e-6)
    assert torch.allclose(r.norm, torch.from_numpy(z[f"{ci}_norm"]), rtol=1e-6, atol=1e-6)


def test_geodesic_fps_transform_tiles_small_clouds():
    import deltaconv_amd.transforms as T
    torch.manual_seed(0)
    d = NS(pos=torch.rand(50, 3), norm=torch.rand(50, 3), x=None, y=torch.arange(50))
    out = T.GeodesicFPS(120)(d)
    assert out.pos.shape == (120, 3) and out.norm.shape == (120, 3) and out.y.shape == (120,)
    assert out.sample_idx[:50].unique().numel() == 50 and torch.equal(out.sample_idx[50:100], out.sample_idx[:50])
    assert repr(T.GeodesicFPS(8)) == "GeodesicFPS()"
    assert all(hasattr(T, n) for n in ["NormalizeScale", "NormalizeArea", "NormalizeAxes", "RandomScale",
                                       "RandomTranslateGlobal", "RandomRotate", "RandomNormals", "SamplePoints",
                                       "GeodesicFPS"])


def test_batch_level_augmentation_is_per_cloud():
    """RandomScale / RandomTranslateGlobal on a collated Batch (the GPU-side form): every cloud gets its own factor /
    offset, normals follow the inverse scaling and stay unit length, shapes within a cloud stay rigid."""
    import torch
    import deltaconv_amd.transforms as T
    from deltaconv_amd.data import synthetic_batch
    torch.manual_seed(0)
    b = synthetic_batch(4, 64, seed=3)
    pos0, nrm0 = b.pos.clone(), b.norm.clone()
    b = T.RandomScale((0.8, 1.25))(b)
    ratio = (b.pos / pos0).view(4, 64, 3)
    assert torch.allclose(ratio, ratio[:, :1].expand_as(ratio), atol=1e-5)          # one triple per cloud
    assert float(ratio.min()) >= 0.8 - 1e-6 and float(ratio.max()) <= 1.25 + 1e-6
    assert len({round(float(v), 5) for v in ratio[:, 0, 0]}) == 4                    # ... and they differ between clouds
    assert torch.allclose(b.norm.norm(dim=1), torch.ones(256), atol=1e-5)
    expect = nrm0 / ratio.reshape(-1, 3)
    assert torch.allclose(b.norm, expect / expect.norm(dim=1, keepdim=True), atol=1e-5)
    pos1 = b.pos.clone()
    b = T.RandomTranslateGlobal(0.1)(b)
    off = (b.pos - pos1).view(4, 64, 3)
    assert torch.allclose(off, off[:, :1].expand_as(off), atol=1e-6) and float(off.abs().max()) <= 0.1 + 1e-6
    assert len({round(float(v), 6) for v in off[:, 0, 0]}) == 4
