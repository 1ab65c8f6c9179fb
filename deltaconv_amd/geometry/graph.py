"""Fixed-degree kNN graph of a batch of clouds, held the way the HIP kernels want it.

Replaces the ``edge_index`` produced by ``knn_graph(pos, k, batch, loop=True,
flow='target_to_source')`` (reference: deltaconv/models/deltanet_base.py:52,63).  ``Graph``
keeps ``nbr[Nt,k] int32`` (centre-major, so ``edge_index = (repeat_interleave(arange(Nt), k),
nbr.flatten())`` is the reference's tensor) and lazily builds the transposed adjacency used by
every backward op of the step.
"""
import os
import weakref

import torch

from .._lib import lib, require_gpu


class PtrInfo(tuple):
    """(ptr int32 [B+1] on device, num_clouds, max_cloud_size) + ``min_cloud`` (smallest cloud, for the
    k <= cloud size check of the kNN graph)."""
    min_cloud = None

    def __new__(cls, ptr, num_clouds, max_cloud, min_cloud=None):
        self = super().__new__(cls, (ptr, num_clouds, max_cloud))
        self.min_cloud = min_cloud
        return self


def _ptr_from_batch(batch, n, device):
    """PtrInfo of a sorted ``batch`` vector.  One host sync when batch is given."""
    if batch is None:
        return PtrInfo(torch.tensor([0, n], dtype=torch.int32, device=device), 1, n, n)
    counts = torch.bincount(batch)
    ptr = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=device)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    mx_mn = torch.stack([counts.max(), counts.min()]).tolist()       # one sync for both
    return PtrInfo(ptr, int(counts.numel()), int(mx_mn[0]), int(mx_mn[1]))


def _min_cloud(ptr_info):
    mn = getattr(ptr_info, "min_cloud", None)
    if mn is None:
        ptr = ptr_info[0]
        mn = int((ptr[1:] - ptr[:-1]).min())
    return mn


class TilePlan:
    """Per-batch tile plan of a graph (deltaconv_amd/csrc/tile_plan.h): tiles of P points that are consecutive on a
    Morton curve, the unique neighbour rows of each tile and tile-local neighbour indices.  The forward applies and the
    max-aggregation run from it with their neighbour rows in LDS (csrc/ell_tile.h)."""

    def __init__(self, graph, blob, P):
        self.blob, self.P = blob, P
        self.n, self.k, self.num_clouds = graph.n, graph.k, graph.num_clouds
        self.tiles = int(lib.raw("dc_tile_plan_tiles")(graph.n, graph.num_clouds, graph.max_cloud, P))

    @property
    def args(self):
        """(n, num_tiles, k, P): the size arguments every tiled entry point takes after (plan, nbr)."""
        return self.n, self.tiles, self.k, self.P

    def section(self, name):
        """View of one section of the blob (tests / debugging): 'pts' [T,P], 'nu' [T], 'uniq' [T,P*k] int32;
        'loc' [T,P*k], 'self' [T,P] uint16 (as int32 tensors)."""
        T, P, PK = self.tiles, self.P, self.P * self.k
        r4 = lambda w: (w + 3) & ~3
        o_pts = 0
        o_nu = r4(o_pts + T * P)
        o_uniq = r4(o_nu + T)
        o_loc = r4(o_uniq + T * PK)
        o_self = r4(o_loc + (T * PK + 1) // 2)
        if name == "pts":
            return self.blob[o_pts:o_pts + T * P].view(T, P)
        if name == "nu":
            return self.blob[o_nu:o_nu + T]
        if name == "uniq":
            return self.blob[o_uniq:o_uniq + T * PK].view(T, PK)
        if name == "loc":
            return self.blob[o_loc:o_loc + (T * PK + 1) // 2].view(torch.int16)[:T * PK].view(T, PK).to(torch.int32) & 0xffff
        if name == "self":
            return self.blob[o_self:o_self + (T * P + 1) // 2].view(torch.int16)[:T * P].view(T, P).to(torch.int32) & 0xffff
        raise KeyError(name)


class TilePlanT:
    """Transposed tile plan (deltaconv_amd/csrc/tile_plan.h, second half): the tiles of the forward plan seen from the
    target side -- targets ordered by in-degree, unique source rows, the tile's in-edge lists in tile order.  The
    transposed applies and the max-aggregation backward run from it with their source rows in LDS (csrc/ell_tileT.h)."""

    def __init__(self, graph, fwd, blob):
        self.blob, self.P = blob, fwd.P
        self.n, self.k, self.num_clouds, self.tiles = graph.n, graph.k, graph.num_clouds, fwd.tiles
        self.edges = int(lib.raw("dc_tile_plan_T_edges")(self.n, self.num_clouds, self.tiles, self.k))
        self._o_edge = int(lib.raw("dc_tile_plan_T_edge_offset")(self.n, self.num_clouds, self.tiles, self.k, self.P))

    @property
    def args(self):
        """(n, num_clouds, num_tiles, k, P): the size arguments every transposed tiled entry point takes after planT."""
        return self.n, self.num_clouds, self.tiles, self.k, self.P

    @property
    def edge_ids(self):
        """int32 [edges]: global edge ids in tile order = the permutation of an operator's coefficients."""
        return self.blob[self._o_edge:self._o_edge + self.edges]

    def section(self, name):
        """'tg' [T,P,4] (target, offset, degree, 0), 'hdr' [T,4] (U, start, entries, 0), 'uniq' [T,256], 'rec' [edges]
        (local source | slot << 16), 'edge' [edges]."""
        T, P = self.tiles, self.P
        o_tg, o_hdr = 0, 4 * T * P
        o_uniq = o_hdr + 4 * T
        o_rec = o_uniq + 256 * T
        if name == "tg":
            return self.blob[o_tg:o_hdr].view(T, P, 4)
        if name == "hdr":
            return self.blob[o_hdr:o_uniq].view(T, 4)
        if name == "uniq":
            return self.blob[o_uniq:o_rec].view(T, 256)
        if name == "rec":
            return self.blob[o_rec:o_rec + self.edges]
        if name == "edge":
            return self.edge_ids
        raise KeyError(name)


# Forward applies / max-aggregation from a tile plan (True) or through the gather path (False): A/B switch, same results.
USE_TILE_PLAN = [os.environ.get("DC_TILE_PLAN", "1") != "0"]
# Transposed applies / max-aggregation backward from the transposed plan (needs the forward plan): A/B switch, same results.
USE_TILE_PLAN_T = [os.environ.get("DC_TILE_PLAN_T", "1") != "0"]


class Graph:
    def __init__(self, nbr, ptr, num_clouds, max_cloud, pos=None):
        self.nbr = nbr                                    # [Nt,k] int32, global ids
        self.n, self.k = int(nbr.shape[0]), int(nbr.shape[1])
        self.ptr, self.num_clouds, self.max_cloud = ptr, num_clouds, max_cloud
        self.pos = pos                                    # positions the graph was built on (tile plan: Morton order)
        self._csc = None
        self._edge_index = None
        self._tile_plan = None
        self._tile_plan_T = None

    def tile_plan_T(self):
        """TilePlanT of this graph (built once from the forward plan + the CSC: one stream-ordered kernel), or None when
        the graph has no forward plan or the switch is off."""
        if self._tile_plan_T is None:
            self._tile_plan_T = False
            fwd = self.tile_plan()
            if fwd is not None and USE_TILE_PLAN_T[0]:
                tptr, tedge = self.csc()
                words = int(lib.raw("dc_tile_plan_T_words")(self.n, self.num_clouds, fwd.tiles, self.k, fwd.P))
                blob = torch.empty(words, dtype=torch.int32, device=self.nbr.device)
                lib.call("dc_tile_plan_T_build", fwd.blob, tptr, tedge, self.ptr, self.num_clouds, self.n, self.max_cloud,
                         self.k, fwd.P, blob)
                self._tile_plan_T = TilePlanT(self, fwd, blob)
        return self._tile_plan_T or None

    def tile_plan(self, force_P=None):
        """TilePlan of this graph, built once (stream-ordered kernels: capturable), or None when the plan does not
        apply or does not pay: positions unknown, clouds beyond the builder's limit, k > 24, fewer than 8192 points, or
        the switch is off.  force_P = 32 | 64 builds a plan regardless of the pay-off heuristic."""
        if force_P is not None and (self._tile_plan is None or not self._tile_plan or self._tile_plan.P != force_P):
            # (tests / A-B runs) the same applicability checks as the policy path, as errors instead of a silent None
            if self.pos is None or not self.nbr.is_cuda:
                raise ValueError("tile_plan(force_P): the graph has no positions (Morton order) or is not on a HIP device")
            if force_P not in (32, 64) or self.k % 2 or self.k < 2 or force_P * self.k > 2048 or self.n == 0:
                raise ValueError(f"tile_plan(force_P={force_P}): needs P in (32, 64), k even, P * k <= 2048 (k = {self.k})")
            if self.max_cloud > int(lib.raw("dc_tile_plan_max_cloud")()):
                raise ValueError(f"tile_plan(force_P): clouds of more than {int(lib.raw('dc_tile_plan_max_cloud')())} points")
            words = int(lib.raw("dc_tile_plan_words")(int(lib.raw("dc_tile_plan_tiles")(self.n, self.num_clouds, self.max_cloud, force_P)), self.k, force_P))
            blob = torch.empty(words, dtype=torch.int32, device=self.nbr.device)
            lib.call("dc_tile_plan_build", self.pos, self.nbr, self.ptr, self.num_clouds, self.n, self.max_cloud,
                     self.k, force_P, blob)
            self._tile_plan = TilePlan(self, blob, force_P)
            self._tile_plan_T = None
        if self._tile_plan is None:
            self._tile_plan = False
            # Where the plan pays (profiles/r03q_tile_policy.txt): k <= 24 with tiles of 64 points and enough points to
            # amortise the three builder launches.  Tiles of 32 points (k = 30: 64 x 30 rows overflow the LDS capacity
            # too often) do not beat the gather path (C5: 10.28 vs 10.21 ms per step), tiny batches are launch-bound
            # (C1: +40 us on a 1.14 ms step).  DC_TILE_P = 32 | 64 forces a plan (A/B runs, tests).
            forced = int(os.environ.get("DC_TILE_P", 0))
            P = forced or 64
            pays = forced or (self.k <= 24 and self.n >= 8192)
            if (USE_TILE_PLAN[0] and pays and self.pos is not None and self.nbr.is_cuda and self.n > 0 and self.k % 2 == 0
                    and self.max_cloud <= int(lib.raw("dc_tile_plan_max_cloud")()) and P * self.k <= 2048):
                words = int(lib.raw("dc_tile_plan_words")(int(lib.raw("dc_tile_plan_tiles")(self.n, self.num_clouds, self.max_cloud, P)), self.k, P))
                blob = torch.empty(words, dtype=torch.int32, device=self.nbr.device)
                lib.call("dc_tile_plan_build", self.pos, self.nbr, self.ptr, self.num_clouds, self.n, self.max_cloud,
                         self.k, P, blob)
                self._tile_plan = TilePlan(self, blob, P)
        return self._tile_plan or None

    @staticmethod
    def knn(pos, k, batch=None, ptr_info=None, lanes_per_query=0):
        """k nearest neighbours per cloud incl. self (bit-exact order: fp32 ((dx*dx+dy*dy)+dz*dz)
        ascending, ties by lower index)."""
        require_gpu()
        pos = pos.contiguous().float()
        n = pos.shape[0]
        info = ptr_info if ptr_info is not None else _ptr_from_batch(batch, n, pos.device)
        ptr, nc, mx = info
        # a cloud with fewer than k points has no k-nearest-neighbour list (the reference's fixed-k reshape,
        # grad_div_mls.py:24-25,85, fails there too); the kernels index neighbours without bounds checks
        mn = _min_cloud(info)
        if mn < k:
            raise ValueError(f"knn graph: every cloud needs at least k = {k} points, the smallest has {mn}")
        nbr = torch.empty(n, k, dtype=torch.int32, device=pos.device)
        lib.call("dc_knn", pos, ptr, nc, mx, k, lanes_per_query, nbr)
        return Graph(nbr, ptr, nc, mx, pos=pos)

    @staticmethod
    def from_edge_index(edge_index, num_points, k=None, batch=None, ptr_info=None):
        """Adopt a reference-style ``edge_index[2,E]`` (centre-major, fixed k: the layout the
        reference itself relies on, grad_div_mls.py:24-25,85,224)."""
        e = edge_index.shape[1]
        k = e // num_points if k is None else int(k)
        assert k * num_points == e, "edge_index is not a fixed-degree centre-major kNN graph"
        if k > 255:   # the aggregations keep the selected slot in one byte (advisor, round 5: say so here, not as a C-ABI error)
            raise ValueError(f"graphs with k = {k} > 255 neighbours per point are not supported (uint8 slot of the aggregation)")
        nbr = edge_index[1].reshape(num_points, k).to(torch.int32).contiguous()
        ptr, nc, mx = ptr_info if ptr_info is not None else _ptr_from_batch(batch, num_points, edge_index.device)
        g = Graph(nbr, ptr, nc, mx)
        g._edge_index = edge_index
        return g

    @property
    def edge_index(self):
        if self._edge_index is None:
            row = torch.arange(self.n, device=self.nbr.device).repeat_interleave(self.k)
            self._edge_index = torch.stack([row, self.nbr.reshape(-1).long()], 0)
            _GRAPH_OF[id(self._edge_index)] = self
        return self._edge_index

    def csc(self):
        """(tptr[Nt+1], tedge[Nt*k]) int32: in-edges per point, ascending edge id; built once."""
        if self._csc is None:
            dev = self.nbr.device
            tptr = torch.empty(self.n + 1, dtype=torch.int32, device=dev)
            tedge = torch.empty(self.n * self.k, dtype=torch.int32, device=dev)
            ws = torch.empty(self.n * (self.k + 1), dtype=torch.int32, device=dev)
            if self.max_cloud <= 4096:      # count + scan + fill of a cloud in one workgroup (LDS counters)
                lib.call("dc_csc_build_clouds", self.nbr, self.ptr, self.num_clouds, self.n, self.max_cloud, self.k,
                         tptr, tedge, ws, ws.numel() * 4)
            else:
                lib.call("dc_csc_build", self.nbr, self.ptr, self.num_clouds, self.n, self.k, tptr, tedge, ws,
                         ws.numel() * 4)
            self._csc = (tptr, tedge)
        return self._csc


# id(edge_index tensor handed out by Graph.edge_index) -> Graph (cheap round trip; weak: no leak)
_GRAPH_OF = weakref.WeakValueDictionary()


def as_graph(edge_index, like=None):
    """edge_index argument of the reference API -> Graph.  Accepts a Graph, an edge_index tensor
    previously handed out by a Graph (no rebuild), or any centre-major fixed-k edge_index."""
    if isinstance(edge_index, Graph):
        return edge_index
    g = _GRAPH_OF.get(id(edge_index))
    if g is not None and g._edge_index is edge_index:
        return g
    if like is not None and like._edge_index is edge_index:
        return like
    assert like is not None, "need the operator's graph to infer the number of points"
    return Graph.from_edge_index(edge_index, like.n, ptr_info=(like.ptr, like.num_clouds, like.max_cloud))


def knn_graph(x, k, batch=None, loop=True, flow='target_to_source'):
    """Drop-in for the one call form the reference uses (deltanet_base.py:52,63) -> edge_index."""
    assert loop and flow == 'target_to_source', "only the reference's call form is implemented"
    return Graph.knn(x, k, batch).edge_index
