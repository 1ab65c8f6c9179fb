"""Oracle: DeltaNet backbone + heads on CPU (models/deltanet_base.py:9-87,
deltanet_classification.py:10-51, deltanet_segmentation.py:10-69).  TEST INFRASTRUCTURE ONLY."""
import torch
from torch import nn as tnn

from . import geometry as geo
from .nn import DeltaConv, MLP


class DeltaNetBase(tnn.Module):
    def __init__(self, in_channels, conv_channels, mlp_depth, num_neighbors, grad_regularizer,
                 grad_kernel_width, centralize_first=True):
        super().__init__()
        self.k, self.grad_regularizer, self.grad_kernel_width = num_neighbors, grad_regularizer, grad_kernel_width
        ch = [in_channels] + list(conv_channels)
        self.convs = tnn.ModuleList([
            DeltaConv(ch[i], ch[i + 1], depth=mlp_depth, centralized=(centralize_first and i == 0),
                      vector=(i != len(ch) - 2)) for i in range(len(ch) - 1)])

    def operators(self, data):
        """deltanet_base.py:52-69 -> nbr, grad, div (no autograd: geometry only)."""
        pos = data.pos
        ptr = geo.cloud_ptr(data.batch, pos.shape[0])
        with torch.no_grad():
            nbr = geo.knn(pos, self.k, ptr)
            normal = getattr(data, 'norm', None)
            if normal is not None:
                xb, yb = geo.build_tangent_basis(normal)
            else:
                normal, xb, yb = geo.estimate_basis(pos, geo.knn(pos, 10, ptr), orientation=pos)
            grad, div = geo.build_grad_div(pos, normal, xb, yb, nbr, ptr, self.grad_kernel_width,
                                           self.grad_regularizer)
        return nbr, grad, div

    def forward(self, data):
        nbr, grad, div = self.operators(data)
        x = data.x if getattr(data, 'x', None) is not None else data.pos   # deltanet_base.py:76
        v = grad @ x                                                        # deltanet_base.py:78
        out = []
        for conv in self.convs:
            x, v = conv(x, v, grad, div, nbr)
            out.append(x)
        return out


def _segment(x, ptr, op):
    return torch.stack([op(x[ptr[b]:ptr[b + 1]]) for b in range(len(ptr) - 1)])


class DeltaNetClassification(tnn.Module):
    def __init__(self, in_channels, num_classes, conv_channels=(64, 64, 128, 256), num_neighbors=20,
                 grad_regularizer=1e-3, grad_kernel_width=1):
        super().__init__()
        conv_channels = list(conv_channels)
        self.deltanet_base = DeltaNetBase(in_channels, conv_channels, 1, num_neighbors, grad_regularizer,
                                          grad_kernel_width)
        self.lin_embedding = MLP([sum(conv_channels), 1024])
        self.classification_head = tnn.Sequential(
            MLP([1024 * 2, 512]), tnn.Dropout(0.5), MLP([512, 256]), tnn.Dropout(0.5),
            tnn.Linear(256, num_classes))

    def forward(self, data):
        x = self.lin_embedding(torch.cat(self.deltanet_base(data), 1))
        ptr = geo.cloud_ptr(data.batch, x.shape[0])
        pooled = torch.cat([_segment(x, ptr, lambda t: t.max(0).values),
                            _segment(x, ptr, lambda t: t.mean(0))], 1)
        return self.classification_head(pooled)


class DeltaNetSegmentation(tnn.Module):
    def __init__(self, in_channels, num_classes, conv_channels=(64, 128, 256), mlp_depth=2,
                 embedding_size=1024, categorical_vector=False, num_neighbors=20, grad_regularizer=0.001,
                 grad_kernel_width=1):
        super().__init__()
        conv_channels = list(conv_channels)
        self.categorical_vector = categorical_vector
        self.deltanet_base = DeltaNetBase(in_channels, conv_channels, mlp_depth, num_neighbors,
                                          grad_regularizer, grad_kernel_width)
        self.lin_global = MLP([sum(conv_channels), embedding_size])
        extra = 0
        if categorical_vector:
            self.lin_categorical = MLP([16, 64])
            extra = 64
        self.segmentation_head = tnn.Sequential(
            MLP([embedding_size + sum(conv_channels) + extra, 256]), tnn.Dropout(0.5), MLP([256, 256]),
            tnn.Dropout(0.5), tnn.Linear(256, 128), tnn.LeakyReLU(negative_slope=0.2),
            tnn.Linear(128, num_classes))

    def forward(self, data):
        conv_out = self.deltanet_base(data)
        x = self.lin_global(torch.cat(conv_out, 1))
        ptr = geo.cloud_ptr(data.batch, x.shape[0])
        batch = data.batch if data.batch is not None else torch.zeros(x.shape[0], dtype=torch.long)
        glob = _segment(x, ptr, lambda t: t.max(0).values)[batch]
        if self.categorical_vector:
            glob = torch.cat([glob, self.lin_categorical(data.category)[batch]], 1)
        return self.segmentation_head(torch.cat([glob] + conv_out, 1))
