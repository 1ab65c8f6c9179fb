"""Oracle: geometry of the DeltaConv path (kNN, tangent bases, MLS grad/div assembly, operator
algebra) restated on CPU in a fixed-degree ELL formulation.  TEST INFRASTRUCTURE ONLY.

Layout conventions (identical to the product):
  nbr   [Nt, k]     int64   global neighbour ids, centre-major; edge e = i*k + s
  G, D  [Nt, k, 2]          grad / div coefficients of edge (i, s)
  x     [Nt, C]             scalar features
  v     [2Nt, C]            vector features, row 2i = u-component, row 2i+1 = v-component

Follows /root/reference/deltaconv/geometry/grad_div_mls.py and operators.py; the per-function
docstrings give file:line.  Works in the dtype of its inputs (fp32 = the reference's numerics,
fp64 = "truth" used to set tolerances).
"""
import torch

EPS = 1e-5  # grad_div_mls.py:7


# --------------------------------------------------------------------------------------------
# graph
# --------------------------------------------------------------------------------------------
def cloud_ptr(batch, n=None):
    """Offsets [B+1] of the (sorted, contiguous) clouds of a batch vector."""
    if batch is None:
        return [0, int(n)]
    counts = torch.bincount(batch).tolist()
    ptr = [0]
    for c in counts:
        ptr.append(ptr[-1] + c)
    return ptr


def knn(pos, k, ptr):
    """k nearest neighbours per cloud incl. self -> nbr[Nt,k] (global ids).

    Stands for ``knn_graph(pos, k, batch, loop=True, flow='target_to_source')``
    (models/deltanet_base.py:52,63; third-party torch_cluster, un-pinned).  The order is
    DEFINED here: fp32 ``((dx*dx + dy*dy) + dz*dz)`` ascending, ties by lower index.
    """
    out = []
    for b in range(len(ptr) - 1):
        p = pos[ptr[b]:ptr[b + 1]].to(torch.float32)
        n = p.shape[0]
        assert n >= k, "cloud smaller than k"
        rows = []
        for s in range(0, n, 1024):  # chunk the N x N matrix
            q = p[s:s + 1024]
            dx = q[:, None, 0] - p[None, :, 0]
            dy = q[:, None, 1] - p[None, :, 1]
            dz = q[:, None, 2] - p[None, :, 2]
            d2 = (dx * dx + dy * dy) + dz * dz
            rows.append(torch.sort(d2, dim=1, stable=True).indices[:, :k])
        out.append(torch.cat(rows, 0) + ptr[b])
    return torch.cat(out, 0)


def edge_index_from_nbr(nbr):
    """[2,E] (row = centre, col = neighbour), the reference's edge_index layout."""
    nt, k = nbr.shape
    row = torch.arange(nt, device=nbr.device).repeat_interleave(k)
    return torch.stack([row, nbr.reshape(-1)], 0)


def nbr_from_edge_index(edge_index, k=None):
    row, col = edge_index
    if k is None:
        k = int((row == 0).sum())  # grad_div_mls.py:24,85,224
    return col.view(-1, k)


# --------------------------------------------------------------------------------------------
# tangent bases
# --------------------------------------------------------------------------------------------
def _unit(a):
    return a / a.norm(dim=-1, keepdim=True).clamp(EPS)


def build_tangent_basis(normal):
    """grad_div_mls.py:50-69: t=(1,0,0) unless |n.t|>0.9 -> (0,1,0); x=unit(t x n); y=unit(n x x)."""
    t = torch.zeros_like(normal)
    use_alt = normal[:, 0].abs() > 0.9
    t[:, 0] = (~use_alt).to(normal.dtype)
    t[:, 1] = use_alt.to(normal.dtype)
    x_basis = _unit(torch.linalg.cross(t, normal, dim=1))
    y_basis = _unit(torch.linalg.cross(normal, x_basis, dim=1))
    return x_basis, y_basis


def estimate_basis(pos, nbr, orientation=None):
    """grad_div_mls.py:10-47: SVD of the 3 x k matrix of neighbour offsets; normal = U[:,2]
    (flipped against ``orientation``), x = U[:,0], y = n x x."""
    local = (pos[nbr] - pos[:, None, :]).transpose(1, 2)  # [Nt,3,k]
    U = torch.linalg.svd(local, full_matrices=False).U
    normal = _unit(U[:, :, 2])
    if orientation is not None:
        flip = (normal * orientation).sum(1, keepdim=True) < 0
        normal = torch.where(flip, -normal, normal)
    x_basis = _unit(U[:, :, 0])
    y_basis = _unit(torch.linalg.cross(normal, x_basis, dim=1))
    return normal, x_basis, y_basis


# --------------------------------------------------------------------------------------------
# moving-least-squares assembly
# --------------------------------------------------------------------------------------------
def coords_projected(pos, normal, x_basis, y_basis, nbr):
    """grad_div_mls.py:72-97 -> coords[Nt,k,2]."""
    d = pos[nbr] - pos[:, None, :]
    d = d - normal[:, None, :] * (d * normal[:, None, :]).sum(-1, keepdim=True)
    u = (d * x_basis[:, None, :]).sum(-1)
    v = (d * y_basis[:, None, :]).sum(-1)
    return torch.stack([u, v], -1)


def gaussian_weights(dist, ptr, kernel_width=1.0):
    """grad_div_mls.py:100-116 -> w[Nt,k]; avg = per-cloud mean of the per-point mean distance."""
    per_point = dist.mean(dim=1, keepdim=True)
    avg = torch.empty_like(per_point)
    for b in range(len(ptr) - 1):
        avg[ptr[b]:ptr[b + 1]] = per_point[ptr[b]:ptr[b + 1]].mean()
    w = torch.exp(-dist.pow(2) / (kernel_width * avg).pow(2))
    return w / w.sum(dim=1, keepdim=True).clamp(EPS)


def poly_rows(coords):
    """[1,u,v,u^2,uv,v^2] (grad_div_mls.py:133-137) -> [Nt,k,6]."""
    u, v = coords[..., 0], coords[..., 1]
    return torch.stack([torch.ones_like(u), u, v, u * u, u * v, v * v], -1)


def weighted_least_squares(coords, weights, regularizer):
    """grad_div_mls.py:119-152 -> wls[Nt,k,6] = ((B^T W B + lam I)^-1 B^T W)^T."""
    B = poly_rows(coords)
    BT = (weights[..., None] * B).transpose(1, 2)                     # [Nt,6,k]
    M = BT @ B + regularizer * torch.eye(6, dtype=B.dtype)            # [Nt,6,6]
    return (torch.linalg.inv(M) @ BT).transpose(1, 2).contiguous()


def fit_vector_mapping(pos, normal, x_basis, y_basis, nbr, wls, coords):
    """grad_div_mls.py:155-194 (eq. 15 of the supplement) -> map[Nt,k,2,2]."""
    d = pos[nbr] - pos[:, None, :]
    height = (d * normal[:, None, :]).sum(-1)                         # [Nt,k]
    c = (wls * height[..., None]).sum(1)                              # [Nt,6]
    u, v = coords[..., 0], coords[..., 1]
    h_u = c[:, None, 1] + 2 * c[:, None, 3] * u + c[:, None, 4] * v
    h_v = c[:, None, 2] + c[:, None, 4] * u + 2 * c[:, None, 5] * v
    gam_u = x_basis[:, None, :] + normal[:, None, :] * h_u[..., None]
    gam_v = y_basis[:, None, :] + normal[:, None, :] * h_v[..., None]
    det = 1 + h_u.pow(2) + h_v.pow(2)
    E, F, G = 1 + h_u.pow(2), h_u * h_v, 1 + h_v.pow(2)
    inv_metric = torch.stack([G, -F, -F, E], -1).view(*u.shape, 2, 2) / det[..., None, None]
    xj, yj = x_basis[nbr], y_basis[nbr]
    T = torch.stack([(gam_u * xj).sum(-1), (gam_u * yj).sum(-1),
                     (gam_v * xj).sum(-1), (gam_v * yj).sum(-1)], -1).view(*u.shape, 2, 2)
    return inv_metric @ T


class EllOp:
    """Sparse grad (2Nt x Nt) or div (Nt x 2Nt) held as fixed-degree coefficients."""

    def __init__(self, kind, nbr, coef):
        assert kind in ("grad", "div")
        self.kind, self.nbr, self.coef = kind, nbr, coef

    def size(self, i):
        nt = self.nbr.shape[0]
        return (2 * nt, nt)[i] if self.kind == "grad" else (nt, 2 * nt)[i]

    def __matmul__(self, x):
        nt, k = self.nbr.shape
        c = x.shape[1]
        coef = self.coef.to(x.dtype)
        if self.kind == "grad":                                       # out[2i+a] = sum_s G[i,s,a] x[j]
            return torch.bmm(coef.transpose(1, 2), x[self.nbr]).reshape(2 * nt, c)
        vj = x.view(nt, 2, c)[self.nbr].reshape(nt, 2 * k, c)         # out[i] = sum_s D[i,s,:].v[j]
        return torch.bmm(coef.reshape(nt, 1, 2 * k), vj).reshape(nt, c)

    def to_coo(self):
        nt, k = self.nbr.shape
        i = torch.arange(nt).repeat_interleave(k)
        j = self.nbr.reshape(-1)
        if self.kind == "grad":
            row = torch.stack([2 * i, 2 * i + 1], 1).reshape(-1)
            col = torch.stack([j, j], 1).reshape(-1)
        else:
            row = torch.stack([i, i], 1).reshape(-1)
            col = torch.stack([2 * j, 2 * j + 1], 1).reshape(-1)
        return row, col, self.coef.reshape(-1)


def build_grad_div(pos, normal, x_basis, y_basis, nbr, ptr, kernel_width=1.0, regularizer=1e-3,
                   normalized=True, return_parts=False, shape_regularizer=None):
    """grad_div_mls.py:197-277 -> (grad: EllOp, div: EllOp)."""
    coords = coords_projected(pos, normal, x_basis, y_basis, nbr)
    dist = (pos[nbr] - pos[:, None, :]).norm(dim=-1)
    weights = gaussian_weights(dist, ptr, kernel_width)
    wls = weighted_least_squares(coords, weights, regularizer)
    G = wls[..., 1:3].clone()                                         # grad_div_mls.py:253-255
    if normalized:                                                    # grad_div_mls.py:258-260
        rowsum = G.abs().sum(1).norm(dim=1)                           # [Nt]
        for b in range(len(ptr) - 1):
            m = rowsum[ptr[b]:ptr[b + 1]].max()
            if m > 1e-5:
                G[ptr[b]:ptr[b + 1]] = G[ptr[b]:ptr[b + 1]] / m
    if shape_regularizer is not None:                                 # grad_div_mls.py:241-244,266-267
        wls = weighted_least_squares(coords, weights, shape_regularizer)
    vmap = fit_vector_mapping(pos, normal, x_basis, y_basis, nbr, wls, coords)
    D = (G[..., None, :] @ vmap).squeeze(-2)                          # grad_div_mls.py:271-272
    grad, div = EllOp("grad", nbr, G), EllOp("div", nbr, D.contiguous())
    if return_parts:
        return grad, div, dict(coords=coords, dist=dist, weights=weights, wls=wls, vmap=vmap)
    return grad, div


# --------------------------------------------------------------------------------------------
# operator algebra on interleaved vector fields (operators.py:4-46)
# --------------------------------------------------------------------------------------------
def norm(v):
    return v.view(-1, 2, v.shape[1]).norm(dim=1)


def J(v):
    w = v.view(-1, 2, v.shape[1])
    return torch.stack([-w[:, 1], w[:, 0]], 1).reshape(v.shape)


def I_J(v):
    return torch.cat([v, J(v)], 1)


def curl(v, div):
    return -(div @ J(v))


def laplacian(x, grad, div):
    return -(div @ (grad @ x))


def hodge_laplacian(v, grad, div):
    return -(grad @ (div @ v) + J(grad @ curl(v, div)))


def rotate_around(v, axis, angle):
    """Rodrigues rotation of v about axis (geometry/connection.py:62-76); used by gauge tests."""
    angle = angle.view(-1, 1)
    par = axis * (v * axis).sum(-1, keepdim=True)
    tan = v - par
    return par + torch.cos(angle) * tan + torch.sin(angle) * torch.linalg.cross(axis, tan, dim=1)
