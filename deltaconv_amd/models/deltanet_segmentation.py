"""Segmentation head (reference: deltaconv/models/deltanet_segmentation.py:9-69)."""
import torch
from torch.nn import Sequential as Seq, Dropout, LeakyReLU

from .deltanet_base import DeltaNetBase, _ptr_info
from .pool import embed_and_pool, broadcast_to_points
from ..nn import MLP, fused
from ..nn.mlp import Linear, run_head
from ..nn.layer import cat_outputs


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
        info, n = _ptr_info(data), data.pos.shape[0]
        conv_cat = cat_outputs(conv_out)
        pooled = embed_and_pool(self.lin_global, conv_cat, info, with_mean=False)
        if self.categorical_vector:
            pooled = torch.cat([pooled, self.lin_categorical(data.category)], dim=1)
        first = self.segmentation_head[0]
        blk = first[0] if len(first) == 1 else None
        _, nc, mx = info
        if (blk is not None and blk[0].bias is None and fused.slope_of(blk[2]) is not None and nc * mx == n
                and conv_cat.is_cuda):
            # Linear([x_max[batch] | conv]) = Linear_a(x_max)[batch] + Linear_b(conv): the per-cloud half of the
            # first head GEMM runs on B rows instead of Nt and the [Nt, E+S] concatenation is never built
            # (deltanet_segmentation.py:66-68; same sum, different association of the fp32 additions).
            # (round 6: the weight halves through fused.split_cols -- one copy launch in backward instead of autograd's fill /
            #  copy / add per slice -- and the join through fused.cloud_bias: in place, backward = own per-cloud column sums)
            wa, wb = fused.split_cols(blk[0].weight, pooled.shape[1])
            h = fused.cloud_bias(fused.linear(conv_cat, wb), fused.linear(pooled, wa), mx)
            y = fused.bn_act(h, blk[1].bn, fused.slope_of(blk[2]))
            return run_head(list(self.segmentation_head)[1:], y)
        x_max = broadcast_to_points(pooled, batch, info, n)
        return run_head(self.segmentation_head, torch.cat([x_max, conv_cat], dim=1))
