"""The DeltaConv layer (reference: deltaconv/nn/deltaconv.py:29-72), same constructor, same
``forward(x, v, grad, div, edge_index) -> (x, v)``, same parameter names.

What differs underneath: the kNN max-aggregation never materialises the [E,C] gathered tensor,
``div v``, ``curl v`` and ``|v|`` come out of one gather pass, and the Hodge-Laplacian reuses them
instead of recomputing two applies (operators.py:40,43 vs deltaconv.py:57)."""
import torch
import torch.nn.functional as F

from .mlp import MLP, VectorMLP, MLPBlock, VectorBlock, run_mlp
from . import fused
from .layer import DeltaConvLayerFn, LayerCfg
from .. import _ops
from ..geometry.graph import as_graph


class DeltaConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, depth=1, centralized=False, vector=True, aggr='max'):
        super().__init__()
        if aggr != 'max':
            raise NotImplementedError("only aggr='max' (the reference default, used by every model)")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.centralized = centralized
        self.aggr = aggr
        self.s_mlp_max = MLP([in_channels] + [out_channels] * depth)
        self.s_mlp = MLP([in_channels * 4] + [out_channels] * depth)
        self.v_mlp = VectorMLP([in_channels * 4 + out_channels * 2] + [out_channels] * depth) if vector else None

    fuse_layer = True   # one autograd node per layer (nn/layer.py) when the layer qualifies
    _chained_x = None

    def _fusable(self):
        """depth-1 MLPs with BatchNorm, piecewise-linear activations, ReLU vector non-linearity, no bias."""
        mlps = [self.s_mlp_max, self.s_mlp] + ([self.v_mlp] if self.v_mlp is not None else [])
        if any(len(m) != 1 for m in mlps):
            return None
        bm, bs = self.s_mlp_max[0], self.s_mlp[0]
        if not (isinstance(bm, MLPBlock) and isinstance(bs, MLPBlock)) or bm[0].bias is not None or bs[0].bias is not None:
            return None
        sm, ss = fused.slope_of(bm[2]), fused.slope_of(bs[2])
        if sm is None or ss is None or (self.centralized and sm < 0):
            return None
        if self.v_mlp is not None:
            bv = self.v_mlp[0]
            if not isinstance(bv, VectorBlock) or bv[1].batchnorm is None or not isinstance(bv[1].nonlin, torch.nn.ReLU):
                return None
        return sm, ss

    def forward(self, x, v, grad, div, edge_index, next_layer=None, out_block=None):
        """next_layer (optional, beyond the reference signature): the DeltaConv that consumes this layer's
        outputs next; x' and v' are then produced directly inside its operand buffers (no copies).
        out_block (optional): (buffer [n, W], column offset) of a concatenation buffer; when the layer runs as
        one fused node x' is ALSO written into that column block, which is returned as x."""
        graph = as_graph(edge_index, grad.graph)
        # synchronised BatchNorm (dp.py) runs through the composed blocks: their statistics kernels have the split form
        slopes = self._fusable() if (self.fuse_layer and fused.sync_group() is None) else None
        if slopes is None:
            return self.forward_composed(x, v, grad, div, graph)
        bm, bs = self.s_mlp_max[0], self.s_mlp[0]
        bv = self.v_mlp[0] if self.v_mlp is not None else None
        chain = None
        if (isinstance(next_layer, DeltaConv) and next_layer.fuse_layer and next_layer.in_channels == self.out_channels
                and next_layer._fusable() is not None and self.out_channels % 4 == 0):
            co = self.out_channels
            chain = (4 * co, 2 * co + next_layer.out_channels
                     if (bv is not None and next_layer.v_mlp is not None and next_layer.out_channels % 4 == 0) else None)
        cfg = LayerCfg(graph, grad, div, bm[1].bn, bs[1].bn, bv[1].batchnorm.bn if bv is not None else None,
                       self.centralized, slopes[0], slopes[1], bv is not None, chain, out_block)
        vb = bv[1].batchnorm.bn if bv is not None else None
        x_new, v_new, x_dup = DeltaConvLayerFn.apply(
            x, v, bm[0].weight, bm[1].bn.weight, bm[1].bn.bias, bs[0].weight, bs[1].bn.weight, bs[1].bn.bias,
            bv[0].weight if bv is not None else None, vb.weight if vb is not None else None,
            vb.bias if vb is not None else None, cfg)
        if out_block is not None:
            self._chained_x = x_new          # what the next layer consumes (lives in its operand buffer)
            return x_dup, (v_new if bv is not None else v)
        return x_new, (v_new if bv is not None else v)

    def forward_composed(self, x, v, grad, div, edge_index):
        """The same layer as a chain of small autograd nodes (any depth / activation)."""
        graph = as_graph(edge_index, grad.graph)
        n, k, ci = graph.n, graph.k, self.in_channels

        # scalar stream: max aggregation over the k neighbours (deltaconv.py:50-54)
        if self.centralized:
            x_max = self._centralized_max(x, graph)
        else:
            x_max = _ops.knn_max(self.s_mlp_max(x), graph)

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
        if len(blocks) == 1 and slope is not None and slope >= 0 and blk[0].bias is None and fused.sync_group() is None:
            y = F.linear(x, blk[0].weight)
            return fused.edge_max_bn(y, graph, blk[1].bn, slope)
        n, k = graph.n, graph.k
        nbr = graph.nbr.long()
        x_edge = (x[nbr] - x.unsqueeze(1)).reshape(n * k, x.shape[1])
        return self.s_mlp_max(x_edge).view(n, k, -1).max(dim=1).values

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels})'
