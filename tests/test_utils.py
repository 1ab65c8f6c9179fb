"""Product loss / metric restatement vs the oracle's and vs values produced by the reference
(tests/golden model fixtures hold the reference loss)."""
import numpy as np
import torch

import oracle
from deltaconv_amd.utils import calc_loss, calc_shape_IoU
from tests.helpers import load_golden


def test_calc_loss_matches_oracle_and_golden():
    torch.manual_seed(0)
    pred = torch.randn(37, 40, dtype=torch.float64)
    y = torch.randint(0, 40, (37,))
    for sm in (True, False):
        assert abs(float(calc_loss(pred, y, sm)) - float(oracle.loss.calc_loss(pred, y, sm))) < 1e-12
    g = load_golden("model_cls_B4_N256_k20")
    assert abs(float(calc_loss(g["logits_f64"], g["y"])) - float(g["loss_f64"])) < 1e-10
    g = load_golden("model_seg_B2_N256_k20")
    assert abs(float(calc_loss(g["logits_f64"], g["y"], smoothing=False)) - float(g["loss_f64"])) < 1e-10


def test_shape_iou():
    seg = np.array([[0, 0, 1, 2, 3, 3]])
    pred = np.array([[0, 1, 1, 2, 3, 0]])
    # category 0 owns parts 0..3: IoUs = 1/3, 1/2, 1, 1/2
    assert abs(calc_shape_IoU(pred, seg, np.array([0]), None)[0] - np.mean([1 / 3, 1 / 2, 1.0, 1 / 2])) < 1e-12
    assert calc_shape_IoU(np.array([[4, 4]]), np.array([[4, 4]]), np.array([1]), None)[0] == 1.0   # part 5 empty -> 1


def test_graphed_step_refuses_to_run_with_packet_capture_on(monkeypatch):
    """GraphedTrainStep checks the ROCm runtime flag before touching the GPU (deltaconv_amd/graph_step.py)."""
    import pytest
    from deltaconv_amd.graph_step import GraphedTrainStep
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        GraphedTrainStep(None, None, None)


def test_package_import_defaults_the_runtime_flag():
    import os
    import deltaconv_amd  # noqa: F401
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
