"""Tangent bases and moving-least-squares gradient / divergence operators on the GPU.

API mirror of the reference's deltaconv/geometry/grad_div_mls.py (same names, argument order,
defaults); the work is done by dc_tangent_basis / dc_estimate_basis / dc_mls_assemble
(deltaconv_amd/csrc/{basis,mls}.hip).  ``edge_index`` may be the reference's [2,E] tensor or a
``Graph``; operators come back as ``SparseOp`` (ELL coefficients, supports ``.size(i)`` and ``@``).
"""
import weakref

import torch

from .._lib import lib, require_gpu
from .. import _ops
from .graph import Graph, _ptr_from_batch

EPS = 1e-5


class SparseOp:
    """grad (2Nt x Nt) or div (Nt x 2Nt) as coef[Nt,k,2] over graph.nbr.  Stands where the
    reference holds a torch_sparse.SparseTensor (grad_div_mls.py:263,275)."""

    def __init__(self, kind, graph, coef):
        assert kind in ("grad", "div")
        self.kind, self.graph, self.coef = kind, graph, coef
        self._coefT = None
        self._coefTt = None
        self._coefTt_plan = None
        self._sibling = None

    def coefTt(self):
        """Coefficients in the TILE order of the graph's transposed tile plan (transposed applies from LDS); built once."""
        pt = self.graph.tile_plan_T()
        if self._coefTt is None or self._coefTt_plan is not pt:     # keyed on the plan OBJECT: Graph.tile_plan(force_P=...)
            self._coefTt_plan = pt                                  # rebuilds both plans, the tile order changes with them
            self._coefTt = torch.empty(pt.edges, 2, dtype=torch.float32, device=self.coef.device)
            sib = self._sibling() if self._sibling is not None else None   # grad and div of one build_grad_div call: one launch
            if sib is not None and (sib._coefTt is None or sib._coefTt_plan is not pt) and sib.graph is self.graph:
                sib._coefTt_plan = pt
                sib._coefTt = torch.empty_like(self._coefTt)
                lib.call("dc_tile_plan_T_permute_coef", self.coef, sib.coef, pt.blob, *pt.args, self._coefTt, sib._coefTt)
            else:
                lib.call("dc_tile_plan_T_permute_coef", self.coef, None, pt.blob, *pt.args, self._coefTt, None)
        return self._coefTt

    def coefT(self):
        """Coefficients in CSC order (for the transposed applies of the backward pass); built once."""
        if self._coefT is None:
            _, tedge = self.graph.csc()
            self._coefT = torch.empty_like(self.coef)
            lib.call("dc_csc_permute_coef", self.coef, tedge, tedge.numel(), self._coefT)
        return self._coefT

    def size(self, i):
        n = self.graph.n
        return ((2 * n, n) if self.kind == "grad" else (n, 2 * n))[i]

    def sizes(self):
        return [self.size(0), self.size(1)]

    def __matmul__(self, x):
        return _ops.apply_op(x, self)

    def coo(self):
        """(row, col, value) triplets in the reference's order (grad_div_mls.py:253-255,271-274)."""
        n, k = self.graph.n, self.graph.k
        i = torch.arange(n, device=self.coef.device).repeat_interleave(k)
        j = self.graph.nbr.reshape(-1).long()
        if self.kind == "grad":
            return (torch.stack([2 * i, 2 * i + 1], 1).reshape(-1), torch.stack([j, j], 1).reshape(-1),
                    self.coef.reshape(-1))
        return (torch.stack([i, i], 1).reshape(-1), torch.stack([2 * j, 2 * j + 1], 1).reshape(-1),
                self.coef.reshape(-1))


def _graph_from(edge_index, n, k=None, batch=None):
    if isinstance(edge_index, Graph):
        return edge_index
    from .graph import _GRAPH_OF
    g = _GRAPH_OF.get(id(edge_index))
    if g is not None and g._edge_index is edge_index:
        return g
    return Graph.from_edge_index(edge_index, n, k=k, batch=batch)


def build_tangent_basis(normal):
    """grad_div_mls.py:50-69 -> (x_basis, y_basis)."""
    require_gpu()
    normal = normal.contiguous().float()
    xb, yb = torch.empty_like(normal), torch.empty_like(normal)
    lib.call("dc_tangent_basis", normal, normal.shape[0], xb, yb)
    return xb, yb


def estimate_basis(pos, edge_index, k=None, orientation=None):
    """grad_div_mls.py:10-47 -> (normal, x_basis, y_basis).  x_basis sign: largest-|component|
    positive (the reference inherits LAPACK's arbitrary sign)."""
    require_gpu()
    pos = pos.contiguous().float()
    g = _graph_from(edge_index, pos.shape[0], k)
    normal, xb, yb = torch.empty_like(pos), torch.empty_like(pos), torch.empty_like(pos)
    orient = None if orientation is None else orientation.contiguous().float()
    lib.call("dc_estimate_basis", pos, g.nbr, g.n, g.k, orient, normal, xb, yb)
    return normal, xb, yb


# ---- the stages of build_grad_div on their own: the reference exports and tests them one by one
# (/root/reference/deltaconv/geometry/__init__.py:3, test/geometry/test_grad_div_mls.py:58-275).  Each is one
# entry point of the C ABI running the device functions the fused kernels of dc_mls_assemble call
# (csrc/point_math.h); the product path itself never materialises these tensors.
def _rows_cols(edge_index):
    """(row, col) of an [2, E] tensor, a (row, col) pair or a Graph as contiguous int32 device vectors."""
    if isinstance(edge_index, Graph):
        edge_index = edge_index.edge_index
    row, col = edge_index
    return row.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()


def _k_of(row, k):
    # k = (row == 0).sum() (grad_div_mls.py:24,85,224); generalised to "edges of the first centre" so that the
    # reference's own test graphs, whose centres are not point 0 .. N-1 (test_grad_div_mls.py:247-250), resolve
    return int((row == row[0]).sum()) if k is None else int(k)


def coords_projected(pos, normal, x_basis, y_basis, edge_index, k=None):
    """grad_div_mls.py:72-97 -> coords [E, 2]: neighbours projected onto the tangent plane of their centre."""
    require_gpu()
    row, col = _rows_cols(edge_index)
    if row.numel() == 0:
        return pos.new_zeros(0, 2, dtype=torch.float32)
    k = _k_of(row, k)
    if row.numel() != normal.shape[0] * k:
        raise ValueError(f"coords_projected: {row.numel()} edges for {normal.shape[0]} frames of k = {k} neighbours "
                         "(the reference expands the frames k times by position, grad_div_mls.py:88-90)")
    coords = torch.empty(row.numel(), 2, dtype=torch.float32, device=row.device)
    lib.call("dc_mls_coords", pos.contiguous().float(), normal.contiguous().float(), x_basis.contiguous().float(),
             y_basis.contiguous().float(), row, col, row.numel(), k, coords)
    return coords


def gaussian_weights(dist, k, batch=None, kernel_width=1):
    """grad_div_mls.py:100-116 -> weights [N * k]: Gaussian of the edge length relative to kernel_width x the
    cloud's mean edge length, normalised per neighbourhood.  `batch` must be sorted (clouds contiguous)."""
    require_gpu()
    k = int(k)
    dist = dist.contiguous().float().reshape(-1)
    n = dist.numel() // k
    if dist.numel() != n * k:
        raise ValueError(f"gaussian_weights: {dist.numel()} distances are not a multiple of k = {k}")
    weights = torch.empty_like(dist)
    if n == 0:
        return weights
    if batch is None:
        ptr, num_clouds, max_cloud = torch.tensor([0, n], dtype=torch.int32, device=dist.device), 1, n
    else:
        if batch.numel() != n:
            raise ValueError(f"gaussian_weights: batch holds {batch.numel()} points, dist {n}")
        if bool((batch[1:] < batch[:-1]).any()):
            raise ValueError("gaussian_weights: batch must be sorted (clouds contiguous)")
        info = _ptr_from_batch(batch.to(dist.device), n, dist.device)
        ptr, num_clouds, max_cloud = info[0], info[1], info[2]
    ws = torch.empty(16 * num_clouds, dtype=torch.float64, device=dist.device)   # 16 ordered partial sums per cloud
    lib.call("dc_mls_gaussian_weights", dist, ptr, num_clouds, max_cloud, k, float(kernel_width), weights, ws,
             ws.numel() * 8)
    return weights


def weighted_least_squares(coords, weights, k, regularizer, shape_regularizer=None):
    """grad_div_mls.py:119-152 -> wls [N * k, 6] = ((B^T W B + regularizer I)^-1 B^T W)^T per point; with
    shape_regularizer the pair (wls, wls_shape)."""
    require_gpu()
    k = int(k)
    coords = coords.contiguous().float().reshape(-1, 2)
    weights = weights.contiguous().float().reshape(-1)
    n = weights.numel() // k
    if weights.numel() != n * k or coords.shape[0] != n * k:
        raise ValueError(f"weighted_least_squares: {coords.shape[0]} coords / {weights.numel()} weights for k = {k}")

    def solve(lam):
        wls = torch.empty(n * k, 6, dtype=torch.float32, device=coords.device)
        lib.call("dc_mls_wls", coords, weights, n, k, float(lam), wls)
        return wls

    if shape_regularizer is not None:
        return solve(regularizer), solve(shape_regularizer)
    return solve(regularizer)


def fit_vector_mapping(pos, normal, x_basis, y_basis, edge_index, wls, coords):
    """grad_div_mls.py:155-194 -> [E, 2, 2]: the map between the frame at p_j and the frame of p_i pushed forward to
    p_j (eq. 15 of the supplement).  Edges must be grouped by centre in runs of k (every graph the reference builds)."""
    require_gpu()
    row, col = _rows_cols(edge_index)
    e = row.numel()
    out = torch.empty(e, 2, 2, dtype=torch.float32, device=row.device)
    if e == 0:
        return out
    k = _k_of(row, None)
    if e % k != 0 or bool((row.view(-1, k) != row.view(-1, k)[:, :1]).any()):
        raise ValueError("fit_vector_mapping: edges must come grouped by centre in runs of k = (row == row[0]).sum()")
    lib.call("dc_mls_vector_mapping", pos.contiguous().float(), normal.contiguous().float(),
             x_basis.contiguous().float(), y_basis.contiguous().float(), row, col, e, k,
             wls.contiguous().float().reshape(-1, 6), coords.contiguous().float().reshape(-1, 2), out)
    return out


def build_grad_div(pos, normal, x_basis, y_basis, edge_index, batch=None, kernel_width=1, regularizer=0.001,
                   normalized=True, shape_regularizer=None):
    """grad_div_mls.py:197-277 -> (grad, div) as SparseOp.  x_basis = y_basis = None (this package's models when the data
    carries normals): the frames of build_tangent_basis(normal) are formed inside the assembly (dc_mls_assemble_normals)."""
    require_gpu()
    pos = pos.contiguous().float()
    n = pos.shape[0]
    g = _graph_from(edge_index, n, batch=batch)
    if batch is not None and g.num_clouds == 1 and not isinstance(edge_index, Graph):
        g.ptr, g.num_clouds, g.max_cloud = _ptr_from_batch(batch, n, pos.device)
    if g.pos is None:
        g.pos = pos           # the tile plan of the forward applies orders the points along a Morton curve
    G = torch.empty(n, g.k, 2, dtype=torch.float32, device=pos.device)
    D = torch.empty(n, g.k, 2, dtype=torch.float32, device=pos.device)
    nbytes = lib.raw("dc_mls_workspace_bytes")(g.num_clouds, n)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=pos.device)
    if x_basis is None and y_basis is None and shape_regularizer is None:
        xb = torch.empty(n, 3, dtype=torch.float32, device=pos.device)
        yb = torch.empty(n, 3, dtype=torch.float32, device=pos.device)
        lib.call("dc_mls_assemble_normals", pos, normal.contiguous().float(), g.nbr, g.ptr, g.num_clouds, n, g.max_cloud, g.k,
                 float(kernel_width), float(regularizer), int(bool(normalized)), xb, yb, G, D, ws, ws.numel() * 8)
    elif shape_regularizer is None:
        lib.call("dc_mls_assemble", pos, normal.contiguous().float(), x_basis.contiguous().float(),
                 y_basis.contiguous().float(), g.nbr, g.ptr, g.num_clouds, n, g.max_cloud, g.k, float(kernel_width),
                 float(regularizer), int(bool(normalized)), G, D, ws, ws.numel() * 8)
    else:       # the surface fit with its own regulariser (grad_div_mls.py:241-244,266-267)
        lib.call("dc_mls_assemble_shape", pos, normal.contiguous().float(), x_basis.contiguous().float(),
                 y_basis.contiguous().float(), g.nbr, g.ptr, g.num_clouds, n, g.max_cloud, g.k, float(kernel_width),
                 float(regularizer), float(shape_regularizer), int(bool(normalized)), G, D, ws, ws.numel() * 8)
    grad, div = SparseOp("grad", g, G), SparseOp("div", g, D)
    grad._sibling, div._sibling = weakref.ref(div), weakref.ref(grad)
    return grad, div
