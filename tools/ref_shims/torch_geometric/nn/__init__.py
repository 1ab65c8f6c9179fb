"""Pure-torch stand-ins for the torch_geometric.nn entry points the reference imports."""
import torch
from . import inits  # noqa: F401


def knn_graph(x, k, batch=None, loop=False, flow='source_to_target'):
    assert loop and flow == 'target_to_source', "only the reference's call form is provided"
    n = x.size(0)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long, device=x.device)
    counts = torch.bincount(batch).tolist()
    cols, start = [], 0
    for c in counts:
        p = x[start:start + c].float()
        d = p[:, None, :] - p[None, :, :]
        dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
        d2 = (dx * dx + dy * dy) + dz * dz
        order = torch.sort(d2, dim=1, stable=True).indices[:, :k]
        cols.append(order + start)
        start += c
    col = torch.cat(cols, 0).reshape(-1)
    row = torch.arange(n, device=x.device).repeat_interleave(k)
    return torch.stack([row, col], 0)


def _ptr(batch):
    counts = torch.bincount(batch)
    return [0] + torch.cumsum(counts, 0).tolist()


def global_max_pool(x, batch):
    p = _ptr(batch)
    return torch.stack([x[p[i]:p[i + 1]].max(dim=0).values for i in range(len(p) - 1)])


def global_mean_pool(x, batch):
    p = _ptr(batch)
    return torch.stack([x[p[i]:p[i + 1]].mean(dim=0) for i in range(len(p) - 1)])
