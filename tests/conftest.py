import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before HIP starts (deltaconv_amd/graph_step.py)

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Plain `pytest tests` on a box without a HIP device: skip (not fail) everything marked gpu."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


from tests.helpers import load_golden, rel_err  # noqa: E402,F401


@pytest.fixture(scope="session")
def golden():
    return load_golden
