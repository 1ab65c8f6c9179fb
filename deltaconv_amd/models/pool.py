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


def embed_and_pool(mlp, x, ptr_info, with_mean):
    """``pool(mlp(x))`` for the embedding MLP in front of the global pooling.  With equal-size clouds
    and a single [Linear -> BatchNorm -> piecewise-linear] block the BatchNorm/activation is fused with
    the pooling (the [Nt, E] activation is never written); otherwise the plain composition."""
    from ..nn import fused
    from ..nn.mlp import MLPBlock
    _, nc, mx = ptr_info
    blocks = list(mlp)
    last = blocks[-1]
    slope = fused.slope_of(last[2]) if isinstance(last, MLPBlock) else None
    if _equal(ptr_info, x.shape[0]) and slope is not None and x.is_cuda and fused.sync_group() is None:
        for blk in blocks[:-1]:
            x = blk(x)
        if last[0].bias is None:
            return fused.linear_bn_act_pool(x, last[0], last[1].bn, slope, nc, mx, with_mean)
        h = fused.linear(x, last[0].weight, last[0].bias)
        return fused.bn_act_pool(h, last[1].bn, slope, nc, mx, with_mean)
    x = mlp(x)
    mxp = global_max_pool(x, ptr_info)
    return torch.cat([mxp, global_mean_pool(x, ptr_info)], dim=1) if with_mean else mxp


def broadcast_to_points(pooled, batch, ptr_info, n):
    """``pooled[batch]``: one row per cloud -> one row per point (deltanet_segmentation.py:61,66).  With
    equal-size clouds this is an expand, whose backward is a per-cloud sum (one reduction kernel); the
    advanced-indexing form costs a sort + serialized scatter in backward (4.2 ms of a 16 ms step at
    16 x 2048 points, profiles/r01n_c4_kernel_summary_before.txt)."""
    _, nc, mx = ptr_info
    if _equal(ptr_info, n) and pooled.shape[0] == nc:
        return pooled.unsqueeze(1).expand(nc, mx, pooled.shape[1]).reshape(n, pooled.shape[1])
    return pooled[batch]
