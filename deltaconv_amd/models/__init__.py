from .deltanet_base import DeltaNetBase                      # noqa: F401
from .deltanet_classification import DeltaNetClassification  # noqa: F401
from .deltanet_segmentation import DeltaNetSegmentation      # noqa: F401
