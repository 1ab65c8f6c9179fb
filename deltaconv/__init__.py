"""Import alias so the reference's own scripts resolve unchanged (experiments/train_modelnet.py:12-16:
``from deltaconv.models import DeltaNetClassification``, ``import deltaconv.transforms as T``): every
``deltaconv[.x.y]`` module IS the ``deltaconv_amd[.x.y]`` module.  Put the repository root on PYTHONPATH."""
import importlib
import sys

import deltaconv_amd as _impl

for _name in ("geometry", "geometry.grad_div_mls", "geometry.operators", "geometry.utils", "geometry.fps",
              "nn", "nn.deltaconv", "nn.mlp", "nn.nonlin", "models", "models.deltanet_base",
              "models.deltanet_classification", "models.deltanet_segmentation", "transforms"):
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(f"deltaconv_amd.{_name}")


def __getattr__(name):                      # deltaconv.Batch, deltaconv.models, ... -> deltaconv_amd.<name>
    return getattr(_impl, name)
