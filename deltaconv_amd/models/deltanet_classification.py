"""Classification head (reference: deltaconv/models/deltanet_classification.py:9-51)."""
import torch
from torch.nn import Sequential as Seq, Dropout

from .deltanet_base import DeltaNetBase, _ptr_info
from .pool import embed_and_pool
from ..nn import MLP, fused
from ..nn.mlp import Linear, run_head
from ..nn.layer import cat_outputs


class DeltaNetClassification(torch.nn.Module):
    def __init__(self, in_channels, num_classes, conv_channels=[64, 64, 128, 256], num_neighbors=20,
                 grad_regularizer=1e-3, grad_kernel_width=1):
        super().__init__()
        self.deltanet_base = DeltaNetBase(in_channels, conv_channels, 1, num_neighbors, grad_regularizer,
                                          grad_kernel_width)
        self.lin_embedding = MLP([sum(conv_channels), 1024])
        self.classification_head = Seq(
            MLP([1024 * 2, 512]), Dropout(0.5), MLP([512, 256]), Dropout(0.5),
            Linear(256, num_classes))

    def forward(self, data):
        with fused.defer_counters():
            return self._forward(data)

    def _forward(self, data):
        conv_out = self.deltanet_base(data)
        # lin_embedding -> [global max | global mean] (deltanet_classification.py:42-49), pooling fused in
        x = embed_and_pool(self.lin_embedding, cat_outputs(conv_out), _ptr_info(data), with_mean=True)
        return run_head(self.classification_head, x)      # module by module: Dropout fused into the blocks' kernels
