"""Geodesic FPS: the C++ restatement (libdeltaconv_host.so) vs the Python oracle (exact, given the
start point) and the reference's own properties (test/geometry/test_fps.py:8-28).  CPU only."""
import os
import subprocess
import random

import numpy as np
import pytest
import torch

from oracle import fps as ofps
from tests.helpers import ROOT


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "deltaconv_amd", "csrc_host")], check=True)


def test_reference_properties():
    from deltaconv_amd.geometry import geodesic_fps
    n, n_samples = 1024, 512
    rng = np.random.default_rng(0)
    pos = rng.standard_normal((n, 3))
    s1 = geodesic_fps(pos, n)
    assert s1.shape[0] == n and np.unique(s1).shape[0] == n
    s2 = geodesic_fps(pos, n_samples)
    assert s2.shape[0] == n_samples and np.unique(s2).shape[0] == n_samples and s2.dtype == np.int32
    with pytest.raises(ValueError):
        geodesic_fps(torch.rand(n, 3), n)
    with pytest.raises(ValueError):
        geodesic_fps(rng.standard_normal((n, 2, 3)), n)


@pytest.mark.parametrize("n,m,seed", [(200, 50, 0), (333, 333, 1), (64, 10, 2)])
def test_matches_python_restatement(n, m, seed):
    from deltaconv_amd.geometry import geodesic_fps
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    pos = d / np.linalg.norm(d, axis=1, keepdims=True) * (1 + 0.2 * rng.random((n, 1)))
    got = geodesic_fps(pos, m, seed=seed)
    assert np.array_equal(got, geodesic_fps(pos, m, seed=seed))            # reproducible with a seed
    want = ofps.geodesic_fps(pos, m, start=int(got[0]))
    assert np.array_equal(got, want)


def test_farthest_point_property_and_float32_input():
    """Each new sample maximises the graph distance to the set sampled so far (the defining property)."""
    from deltaconv_amd.geometry import geodesic_fps
    rng = np.random.default_rng(3)
    pos = rng.random((150, 3)).astype(np.float32)
    got = geodesic_fps(pos, 20, seed=5)
    want = ofps.geodesic_fps(pos.astype(np.float64), 20, start=int(got[0]))
    assert np.array_equal(got, want)
