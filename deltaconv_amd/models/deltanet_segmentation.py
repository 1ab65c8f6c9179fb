"""Segmentation head (reference: deltaconv/models/deltanet_segmentation.py:9-69)."""
import torch
from torch.nn import Sequential as Seq, Dropout, LeakyReLU, Linear

from .deltanet_base import DeltaNetBase, _ptr_info
from .pool import embed_and_pool
from ..nn import MLP, fused


class DeltaNetSegmentation(torch.nn.Module):
    def __init__(self, in_channels, num_classes, conv_channels=[64, 128, 256], mlp_depth=2, embedding_size=1024,
                 categorical_vector=False, num_neighbors=20, grad_regularizer=0.001, grad_kernel_width=1):
        super().__init__()
        self.categorical_vector = categorical_vector
        self.deltanet_base = DeltaNetBase(in_channels, conv_channels, mlp_depth, num_neighbors, grad_regularizer,
                                          grad_kernel_width)
        self.lin_global = MLP([sum(conv_channels), embedding_size])
        extra = 0
        if categorical_vector:
            self.lin_categorical = MLP([16, 64])
            extra = 64
        self.segmentation_head = Seq(
            MLP([embedding_size + sum(conv_channels) + extra, 256]), Dropout(0.5), MLP([256, 256]), Dropout(0.5),
            Linear(256, 128), LeakyReLU(negative_slope=0.2), Linear(128, num_classes))

    def forward(self, data):
        with fused.defer_counters():
            return self._forward(data)

    def _forward(self, data):
        conv_out = self.deltanet_base(data)
        batch = data.batch
        # lin_global -> global max pool -> broadcast back to the points (deltanet_segmentation.py:58-61)
        x_max = embed_and_pool(self.lin_global, torch.cat(conv_out, dim=1), _ptr_info(data), with_mean=False)[batch]
        if self.categorical_vector:
            x_max = torch.cat([x_max, self.lin_categorical(data.category)[batch]], dim=1)
        return self.segmentation_head(torch.cat([x_max] + conv_out, dim=1))
