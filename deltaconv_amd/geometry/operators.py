"""Operator algebra on interleaved vector fields [2N,C] (row 2i = u, row 2i+1 = v component).
API mirror of the reference's deltaconv/geometry/operators.py:4-46; ``grad``/``div`` are SparseOp.
The DeltaConv layer itself uses the fused kernels in deltaconv_amd/_ops.py instead of composing
these (same values, one gather pass)."""
import torch

from .. import _ops


def norm(v):
    """operators.py:4-7"""
    return v.view(-1, 2, v.shape[1]).norm(dim=1)


def J(v):
    """operators.py:9-17: rotate each tangent vector by 90 degrees counter-clockwise."""
    w = v.view(-1, 2, v.shape[1])
    return torch.stack([-w[:, 1], w[:, 0]], 1).reshape(v.shape)


def I_J(v):
    """operators.py:19-21"""
    return torch.cat([v, J(v)], dim=1)


def curl(v, div):
    """operators.py:23-27: curl = -div J v."""
    return -(div @ J(v))


def laplacian(x, grad, div):
    """operators.py:29-33: laplacian = -div grad x."""
    return -(div @ (grad @ x))


def hodge_laplacian(v, grad, div):
    """operators.py:35-46: -(grad div + J grad curl) v, via the fused kernels."""
    dcn = _ops.div_curl_norm(v, div)
    return _ops.hodge_from_dcn(dcn, grad, v.shape[1])
