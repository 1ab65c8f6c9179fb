"""Per-cloud pooling (stands for torch_geometric.nn.global_max_pool / global_mean_pool,
reference: deltaconv/models/deltanet_classification.py:46-47, deltanet_segmentation.py:61)."""
import torch


def _equal(ptr_info, n):
    _, nc, mx = ptr_info
    return nc * mx == n


def global_max_pool(x, ptr_info):
    ptr, nc, mx = ptr_info
    if _equal(ptr_info, x.shape[0]):
        return x.view(nc, mx, x.shape[1]).max(dim=1).values
    p = ptr.tolist()
    return torch.stack([x[p[b]:p[b + 1]].max(dim=0).values for b in range(nc)])


def global_mean_pool(x, ptr_info):
    ptr, nc, mx = ptr_info
    if _equal(ptr_info, x.shape[0]):
        return x.view(nc, mx, x.shape[1]).mean(dim=1)
    p = ptr.tolist()
    return torch.stack([x[p[b]:p[b + 1]].mean(dim=0) for b in range(nc)])
