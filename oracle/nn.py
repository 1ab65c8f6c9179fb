"""Oracle: scalar/vector MLP stream and the DeltaConv layer on CPU.  TEST INFRASTRUCTURE ONLY.

Module tree / parameter names reproduce the reference's state_dict keys
(nn/mlp.py:7-17, nn/nonlin.py:11-86, nn/deltaconv.py:29-42) so one set of weights drives the
reference, this oracle and the HIP product.
"""
import torch
from torch import nn as tnn
import torch.nn.functional as F

from . import geometry as geo

VEC_EPS = 1e-8  # nn/nonlin.py:8


class BatchNorm1d(tnn.Module):
    """nn/nonlin.py:11-35: batch norm over the rows of an [N,C] tensor (statistics over all N)."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = tnn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)

    def forward(self, x):
        return self.bn(x)  # [N,C] input has the same statistics as the reference's [1,C,N]

    def __repr__(self):
        return f'{self.__class__.__name__}({self.bn.num_features})'


class VectorNonLin(tnn.Module):
    """nn/nonlin.py:38-86: v * relu(bn(|v|)) / max(|v|, 1e-8) per point and channel."""

    def __init__(self, in_channels, nonlin=None, batchnorm=None):
        super().__init__()
        self.bias = tnn.Parameter(torch.zeros(in_channels))
        self.nonlin = tnn.ReLU() if nonlin is None else nonlin
        self.batchnorm = batchnorm

    def forward(self, v):
        n2, c = v.shape
        w = v.view(-1, 2, c)
        mag = w.norm(dim=1)
        shifted = mag + self.bias.view(1, -1) if self.batchnorm is None else self.batchnorm(mag)
        scale = self.nonlin(shifted) / mag.clamp(VEC_EPS)
        return (w * scale[:, None, :]).reshape(n2, c)

    def __repr__(self):
        return f'{self.__class__.__name__}(batchnorm={self.batchnorm.__repr__()})'


def MLP(channels, bias=False, nonlin=None):
    """nn/mlp.py:7-11."""
    return tnn.Sequential(*[
        tnn.Sequential(tnn.Linear(channels[i - 1], channels[i], bias=bias), BatchNorm1d(channels[i]),
                       tnn.LeakyReLU(negative_slope=0.2) if nonlin is None else nonlin)
        for i in range(1, len(channels))])


def VectorMLP(channels, batchnorm=True):
    """nn/mlp.py:13-17."""
    return tnn.Sequential(*[
        tnn.Sequential(tnn.Linear(channels[i - 1], channels[i], bias=False),
                       VectorNonLin(channels[i], batchnorm=BatchNorm1d(channels[i]) if batchnorm else None))
        for i in range(1, len(channels))])


class DeltaConv(tnn.Module):
    """nn/deltaconv.py:29-70.  ``nbr`` replaces ``edge_index`` (same graph, [Nt,k] form)."""

    def __init__(self, in_channels, out_channels, depth=1, centralized=False, vector=True, aggr='max'):
        super().__init__()
        assert aggr in ('max', 'min', 'sum', 'add', 'mean')            # torch_scatter reduce names (deltaconv.py:52,54)
        self.aggr = aggr
        # test hook: [Nt, C] slot per (point, channel) that the max aggregation must take instead of its own arg-max
        # (tests pin both sides of a comparison to the same selection: gradients of a max are piecewise, a near-tie that
        # flips under rounding otherwise dominates every gradient comparison)
        self.pinned_slots = None
        self.in_channels, self.out_channels, self.centralized = in_channels, out_channels, centralized
        self.s_mlp_max = MLP([in_channels] + [out_channels] * depth)
        self.s_mlp = MLP([in_channels * 4] + [out_channels] * depth)
        self.v_mlp = VectorMLP([in_channels * 4 + out_channels * 2] + [out_channels] * depth) if vector else None

    def _reduce(self, h):
        """scatter(..., reduce=aggr) over the k contiguous edges of every centre point: h [Nt, k, C] -> [Nt, C]."""
        if self.aggr == 'max':
            if self.pinned_slots is not None:
                return h.gather(1, self.pinned_slots.long()[:, None, :]).squeeze(1)
            return h.max(dim=1).values
        if self.aggr == 'min':
            return h.min(dim=1).values
        return h.mean(dim=1) if self.aggr == 'mean' else h.sum(dim=1)

    def forward(self, x, v, grad, div, nbr):
        nt, k = nbr.shape
        if self.centralized:                                          # deltaconv.py:50-52
            edge = (x[nbr] - x[:, None, :]).reshape(nt * k, -1)
            x_max = self._reduce(self.s_mlp_max(edge).view(nt, k, -1))
        else:                                                         # deltaconv.py:54
            x_max = self._reduce(self.s_mlp_max(x)[nbr])
        x_cat = torch.cat([x, div @ v, geo.curl(v, div), geo.norm(v)], 1)      # deltaconv.py:57
        x = x_max + self.s_mlp(x_cat)                                 # deltaconv.py:59
        if self.v_mlp is not None:                                    # deltaconv.py:64-68
            v_cat = torch.cat([v, geo.hodge_laplacian(v, grad, div), grad @ x], 1)
            v = self.v_mlp(geo.I_J(v_cat))
        return x, v

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels})'
