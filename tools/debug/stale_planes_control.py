"""Negative control for tests/test_gpu_model.py::test_graph_capture_does_not_reuse_weight_planes_cut_before_it: with the
capture-epoch rule of nn/fused.py defeated (presplit_begin replaced by a stub that declares the planes cut BEFORE the capture
as the capture's own), the same test must fail -- it does see stale planes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deltaconv_amd  # noqa: F401
from deltaconv_amd.nn import fused
from tests import test_gpu_model as T

T.test_graph_capture_does_not_reuse_weight_planes_cut_before_it()
print("with the rule: passes")


def stub():
    fused._PL["captured_epoch"] = fused._PL["epoch"]


fused.presplit_begin = stub
try:
    T.test_graph_capture_does_not_reuse_weight_planes_cut_before_it()
except AssertionError as e:
    print("rule defeated: fails as it must:", str(e)[:200])
else:
    print("rule defeated: STILL PASSES -- the test does not see stale planes")
    sys.exit(1)
