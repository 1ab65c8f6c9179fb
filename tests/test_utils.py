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


def test_evaluate_votes_sums_logits_over_passes():
    """experiments/test_shapenet.py:79-96: predictions of the votes are SUMMED before the arg-max."""
    from deltaconv_amd.utils import evaluate_votes
    from deltaconv_amd.data import synthetic_batch

    class Flaky(torch.nn.Module):
        """class 1 with margin 3 on the first call, class 0 with margin 1 on every later one: the summed vote flips
        to class 0 only from the fourth pass on."""
        def __init__(self):
            super().__init__()
            self.calls = 0

        def forward(self, data):
            self.calls += 1
            out = torch.zeros(data.pos.shape[0], 4)
            if self.calls == 1:
                out[:, 1] = 3.0
            else:
                out[:, 0] = 1.0
            return out

    b = synthetic_batch(2, 16, seed=1, per_point_labels=True, categories=16, num_classes=4)
    b.y = torch.zeros_like(b.y)
    b.category = torch.zeros(2, 16); b.category[:, 0] = 1           # category 0 owns parts 0..3
    r3 = evaluate_votes(Flaky(), [b], num_votes=3)
    assert r3["accuracy"] == 0.0 and r3["pred"].shape == (2, 16)
    r5 = evaluate_votes(Flaky(), [b], num_votes=5)
    assert r5["accuracy"] == 1.0 and r5["balanced_accuracy"] == 1.0 and r5["mean_iou"] == 1.0
    assert list(r5["label"]) == [0, 0]
