"""BatchNorm over rows and the vector non-linearity.  Module tree / parameter names mirror the
reference (deltaconv/nn/nonlin.py:11-86) so its state_dicts load unchanged; the arithmetic runs in
the fused HIP kernels of deltaconv_amd/csrc/nn.hip (see fused.py)."""
import warnings

import torch
from torch import Tensor

from . import fused

EPS = 1e-8  # nonlin.py:8
_WARNED_GENERIC = False


class BatchNorm1d(torch.nn.Module):
    """nonlin.py:11-35: batch norm over the rows of an [N,C] tensor (statistics over all N rows).
    The reference reshapes to [1,C,N]; an [N,C] input has identical statistics."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = torch.nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)
        self.reset_parameters()

    def reset_parameters(self):
        self.bn.reset_parameters()

    def forward(self, x: Tensor) -> Tensor:
        return fused.bn_act(x, self.bn, 1.0)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.bn.num_features})'


class VectorNonLin(torch.nn.Module):
    """nonlin.py:38-86: scale each tangent vector by nonlin(bn(|v|)) / max(|v|, 1e-8)."""

    def __init__(self, in_channels, nonlin=torch.nn.ReLU(), batchnorm=None):
        super().__init__()
        self.bias = torch.nn.Parameter(torch.zeros(in_channels))
        self.nonlin = nonlin
        self.batchnorm = batchnorm
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            self.bias.zero_()
        if self.batchnorm is not None:
            self.batchnorm.reset_parameters()

    def forward(self, x: Tensor, combine: int = 0) -> Tensor:
        """x: [2N,C]; combine=1: x is [2N,2C] = [P | Q]; combine=2: P/Q interleaved (fused.py / mlp.VectorBlock)."""
        if isinstance(self.nonlin, torch.nn.ReLU):
            return fused.vector_nonlin(x, combine, self)
        # any other non-linearity is an arbitrary module the HIP kernels (ReLU) cannot evaluate: the reference's formula
        # (nonlin.py:67-79) composed from torch ops on the device -- said out loud, once, so that nobody times it as the
        # product path (no model of the reference passes anything but the default ReLU)
        global _WARNED_GENERIC
        if not _WARNED_GENERIC:
            _WARNED_GENERIC = True
            warnings.warn(f"VectorNonLin(nonlin={self.nonlin.__class__.__name__}): only ReLU runs on the fused HIP kernels; "
                          "this module is evaluated with composed torch ops")
        assert not combine
        n, c = x.shape
        w = x.view(-1, 2, c)
        mag = w.norm(dim=1)
        shifted = mag + self.bias.view(1, -1) if self.batchnorm is None else self.batchnorm(mag)
        return (w * (self.nonlin(shifted) / mag.clamp(EPS)).unsqueeze(1)).reshape(n, c)

    def __repr__(self):
        return f'{self.__class__.__name__}(batchnorm={self.batchnorm.__repr__()})'
