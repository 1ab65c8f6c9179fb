"""Pure-torch stand-in for torch_scatter (generator-side tooling only)."""
import torch


def _dim_size(index, dim_size):
    return int(index.max()) + 1 if dim_size is None else dim_size


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    n = _dim_size(index, dim_size)
    res = src.new_zeros((n,) + tuple(src.shape[1:]))
    return res.index_add(0, index, src)


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    n = _dim_size(index, dim_size)
    s = scatter_add(src, index, 0, dim_size=n)
    cnt = torch.zeros(n, dtype=src.dtype, device=src.device).index_add(
        0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
    return s / cnt.view((-1,) + (1,) * (src.dim() - 1))


def _segments(index):
    # index must be sorted (true for every call site in the reference hot path)
    assert bool((index[1:] >= index[:-1]).all()), "stand-in needs a sorted index"
    counts = torch.bincount(index)
    return counts


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    counts = _segments(index)
    k = int(counts[0])
    if bool((counts == k).all()):
        val, arg = src.view((-1, k) + tuple(src.shape[1:])).max(dim=1)
        base = (torch.arange(val.shape[0], device=src.device) * k)
        arg = arg + base.view((-1,) + (1,) * (arg.dim() - 1))
        return val, arg
    vals, args, start = [], [], 0
    for c in counts.tolist():
        v, a = src[start:start + c].max(dim=0)
        vals.append(v); args.append(a + start); start += c
    return torch.stack(vals), torch.stack(args)


def scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    if reduce in ('sum', 'add'):
        return scatter_add(src, index, dim, dim_size=dim_size)
    if reduce == 'mean':
        return scatter_mean(src, index, dim, dim_size=dim_size)
    if reduce == 'max':
        return scatter_max(src, index, dim, dim_size=dim_size)[0]
    if reduce == 'min':
        return -scatter_max(-src, index, dim, dim_size=dim_size)[0]
    raise NotImplementedError(reduce)
