"""The DeltaConv layer (reference: deltaconv/nn/deltaconv.py:29-72), same constructor, same
``forward(x, v, grad, div, edge_index) -> (x, v)``, same parameter names.

What differs underneath: the kNN max-aggregation never materialises the [E,C] gathered tensor,
``div v``, ``curl v`` and ``|v|`` come out of one gather pass, and the Hodge-Laplacian reuses them
instead of recomputing two applies (operators.py:40,43 vs deltaconv.py:57)."""
import torch

from .mlp import MLP, VectorMLP, MLPBlock, VectorBlock, run_mlp
from . import fused
from .layer import DeltaConvLayerFn, LayerCfg
from .. import _ops
from ..geometry.graph import as_graph


class DeltaConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, depth=1, centralized=False, vector=True, aggr='max'):
        super().__init__()
        if aggr not in _ops.AGGREGATIONS:
            raise ValueError(f"aggr must be one of {_ops.AGGREGATIONS} (torch_scatter reduce names), got {aggr!r}")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.centralized = centralized
        self.aggr = aggr
        self.s_mlp_max = MLP([in_channels] + [out_channels] * depth)
        self.s_mlp = MLP([in_channels * 4] + [out_channels] * depth)
        self.v_mlp = VectorMLP([in_channels * 4 + out_channels * 2] + [out_channels] * depth) if vector else None

    fuse_layer = True   # one autograd node per layer (nn/layer.py) when the layer qualifies

    def _fusable(self):
        """MLPs of [Linear(no bias) -> BatchNorm -> piecewise-linear activation] blocks and a ReLU vector
        non-linearity with BatchNorm: -> (slopes of the s_mlp_max blocks, slopes of the s_mlp blocks) or None."""
        out = []
        for mlp in (self.s_mlp_max, self.s_mlp):
            slopes = []
            for blk in mlp:
                if not isinstance(blk, MLPBlock) or blk[0].bias is not None:
                    return None
                sl = fused.slope_of(blk[2])
                if sl is None:
                    return None
                slopes.append(sl)
            if not slopes:
                return None
            out.append(slopes)
        if self.centralized and len(out[0]) == 1 and out[0][0] < 0:
            return None
        if self.v_mlp is not None:
            if len(self.v_mlp) != len(self.s_mlp):
                return None
            for blk in self.v_mlp:
                if (not isinstance(blk, VectorBlock) or blk[1].batchnorm is None
                        or not isinstance(blk[1].nonlin, torch.nn.ReLU)):
                    return None
        return out[0], out[1]

    def forward(self, x, v, grad, div, edge_index, next_layer=None, out_block=None):
        """next_layer (optional, beyond the reference signature): the DeltaConv that consumes this layer's
        outputs next; x' and v' are then produced directly inside its operand buffers (no copies).
        out_block (optional): (buffer [n, W], column offset) of a concatenation buffer; when the layer runs as
        one fused node x' is ALSO written into that column block; the call then returns THREE values
        (x' as that column block, v', x' as the view inside the next layer's operand buffer)."""
        graph = as_graph(edge_index, grad.graph)
        # (the fused node is the max-aggregation layer of every reference model; other aggregations run composed.  Under
        # synchronised BatchNorm (dp.py) the node keeps its GEMM epilogues: they emit this rank's sums, fused.linear_stats
        # all-reduces them; only a centralised edge MLP -- statistics over the edges -- is computed outside, below)
        slopes = self._fusable() if (self.fuse_layer and self.aggr == 'max' and x.is_cuda) else None
        if slopes is None:
            return self.forward_composed(x, v, grad, div, graph)
        slopes_m, slopes_s = slopes
        chain = None
        if (isinstance(next_layer, DeltaConv) and next_layer.fuse_layer and next_layer.in_channels == self.out_channels
                and next_layer._fusable() is not None and self.out_channels % 4 == 0):
            co = self.out_channels
            chain = (4 * co, 2 * co + next_layer.out_channels
                     if (self.v_mlp is not None and next_layer.v_mlp is not None and next_layer.out_channels % 4 == 0)
                     else None)
        x_max = None
        blocks_m = list(self.s_mlp_max)
        if self.centralized and (len(blocks_m) > 1 or fused.sync_group() is not None):
            # edge MLP of depth > 1 (csrc/edge2.hip or the general form), or synchronised statistics over the edges -- computed
            # outside, enters the node as x_max
            x_max = self._centralized_max(x, graph)
            blocks_m = []
        blocks_s = list(self.s_mlp)
        blocks_v = list(self.v_mlp) if self.v_mlp is not None else None
        cfg = LayerCfg(graph, grad, div, [b[1].bn for b in blocks_m] if blocks_m else None, [b[1].bn for b in blocks_s],
                       [b[1].batchnorm.bn for b in blocks_v] if blocks_v is not None else None,
                       self.centralized, slopes_m, slopes_s, chain, out_block)
        params = []
        for b in blocks_m + blocks_s:
            params.extend((b[0].weight, b[1].bn.weight, b[1].bn.bias))
        for b in blocks_v or []:
            params.extend((b[0].weight, b[1].batchnorm.bn.weight, b[1].batchnorm.bn.bias))
        x_new, v_new, x_dup = DeltaConvLayerFn.apply(x, v, x_max, cfg, *params)
        if out_block is not None:         # x_new = what the next layer consumes (lives in its operand buffer)
            return x_dup, (v_new if blocks_v is not None else v), x_new
        return x_new, (v_new if blocks_v is not None else v)

    def forward_composed(self, x, v, grad, div, edge_index):
        """The same layer as a chain of small autograd nodes (any depth / activation)."""
        graph = as_graph(edge_index, grad.graph)
        n, k, ci = graph.n, graph.k, self.in_channels

        # scalar stream: aggregation over the k neighbours, maximum by default (deltaconv.py:50-54)
        if self.centralized:
            x_max = self._centralized_max(x, graph)
        else:
            x_max = _ops.knn_aggregate(self.s_mlp_max(x), graph, self.aggr)

        # [x, div v, curl v, |v|] -> MLP (deltaconv.py:57-59)
        dcn = _ops.div_curl_norm(v, div)
        x = run_mlp(self.s_mlp, torch.cat([x, dcn], dim=1), residual=x_max)

        # vector stream (deltaconv.py:64-68)
        if self.v_mlp is not None:
            v_cat = torch.cat([v, _ops.hodge_from_dcn(dcn, grad, ci), grad @ x], dim=1)
            v = self.v_mlp[0].forward_vcat(v_cat)       # = v_mlp(I_J(v_cat)) without materialising I_J
            for blk in list(self.v_mlp)[1:]:
                v = blk(v)
        return x, v

    def _centralized_max(self, x, graph):
        """max_s s_mlp_max(x_j - x_i)  (deltaconv.py:50-52).  Depth-1 MLP with a monotone piecewise-linear
        activation: analytic form on y = Linear(x), no [E,C] tensor (csrc/edge_math.h).  Otherwise the
        edge tensor is materialised and pushed through the fused blocks."""
        blocks = list(self.s_mlp_max)
        blk = blocks[0]
        slope = fused.slope_of(blk[2]) if isinstance(blk, MLPBlock) else None
        if (self.aggr == 'max' and len(blocks) == 1 and slope is not None and slope >= 0 and blk[0].bias is None
                and fused.sync_group() is None):
            y = fused.linear(x, blk[0].weight)      # own GEMM kernels (csrc/gemm.hip), any K
            return fused.edge_max_bn(y, graph, blk[1].bn, slope)
        if self.aggr == 'max' and len(blocks) == 2 and all(isinstance(b, MLPBlock) for b in blocks):
            s1, s2 = fused.slope_of(blocks[0][2]), fused.slope_of(blocks[1][2])
            if fused.edge_mlp2_ok(x, blocks[0][0], blocks[0][1].bn, blocks[1][0], blocks[1][1].bn, s1, s2):
                # depth 2 (the part-segmentation net): one MFMA pass over the edges, csrc/edge2.hip
                out, slots = fused.edge_mlp2(x, graph, blocks[0][0], blocks[0][1].bn, s1, blocks[1][0], blocks[1][1].bn, s2)
                from . import layer as _layer
                if _layer.SLOT_TAP[0] is not None:          # test hook (layer.py)
                    _layer.SLOT_TAP[0].append(slots.clone())
                return out
        # every other shape (depth >= 3, other widths, other aggregations): the edge tensor is materialised by dc_edge_diff, the
        # MLP blocks run on its E rows, dc_seg_reduce aggregates the k consecutive rows of every point (csrc/edge.hip)
        h = self.s_mlp_max(fused.edge_diff(x, graph))
        out, slots = fused.seg_reduce(h, graph.n, graph.k, self.aggr)
        if self.aggr == 'max':
            from . import layer as _layer
            if _layer.SLOT_TAP[0] is not None:          # test hook (layer.py)
                _layer.SLOT_TAP[0].append(slots.clone())
        return out

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels})'
