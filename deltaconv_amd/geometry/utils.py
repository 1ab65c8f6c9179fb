import torch


def batch_dot(a, b):
    """Row-wise dot product, [M,1] (reference: deltaconv/geometry/utils.py:3-4)."""
    return (a * b).sum(dim=1, keepdim=True)
