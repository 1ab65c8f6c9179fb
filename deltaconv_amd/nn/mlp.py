"""Scalar / vector MLP builders (reference: deltaconv/nn/mlp.py:7-46).  Same Sequential nesting
(hence the same state_dict keys: ``0.0.weight``, ``0.1.bn.weight`` ...); each inner block runs as one hand-written
fp32-MFMA product with the BatchNorm statistics in its epilogue + one fused HIP BatchNorm/activation kernel instead of
three ATen ops (nn/fused.py)."""
import torch
from torch.nn import Sequential as Seq, LeakyReLU

from .nonlin import BatchNorm1d, VectorNonLin
from . import fused


class Linear(torch.nn.Linear):
    """torch.nn.Linear (same parameters, state_dict keys and repr) whose weight gradient runs through
    `fused.linear` (own fp32-MFMA kernel for the tall-skinny dW = dY^T X)."""

    def forward(self, x):
        return fused.linear(x, self.weight, self.bias)


Lin = Linear


class MLPBlock(Seq):
    """[Linear -> BatchNorm1d -> nonlin]; forward(x, residual=None) = nonlin(bn(lin(x))) + residual."""

    def forward(self, x, residual=None, dropout=None):
        lin, bn, act = self[0], self[1], self[2]
        slope = fused.slope_of(act)
        if slope is None:                       # exotic activation: BN fused, activation through torch
            out = act(fused.linear_bn_act(x, lin, bn.bn, 1.0))
            return out if residual is None else out + residual
        return fused.linear_bn_act(x, lin, bn.bn, slope, residual, dropout)

    def dropout_ok(self, x):
        """Can a Dropout behind this block run inside the block's kernels (fused.rowblock_dropout_ok)?"""
        return fused.slope_of(self[2]) is not None and fused.rowblock_dropout_ok(x, self[0], self[1].bn)


class VectorBlock(Seq):
    """[Linear(no bias) -> VectorNonLin]."""

    def forward(self, v):
        return self[1](fused.linear(v, self[0].weight))

    def forward_vcat(self, v_cat):
        """Same as forward(I_J(v_cat)) without materialising I_J: with W = [W1 | W2],
        W I_J(a) = (W1 a_u - W2 a_v, W1 a_v + W2 a_u); one GEMM with the weight viewed as [2co, K] (rows
        (c, half), no copy) -> interleaved (P_c, Q_c) columns, combined inside the fused non-linearity kernel."""
        w = self[0].weight
        k = v_cat.shape[1]
        assert w.shape[1] == 2 * k, "first vector block expects I_J(v_cat) (2x the channels of v_cat)"
        if not isinstance(self[1].nonlin, torch.nn.ReLU):
            from ..geometry.operators import I_J
            return self.forward(I_J(v_cat))
        pq = fused.linear(v_cat, w.view(2 * w.shape[0], k))    # [2N, 2*co], columns interleaved (P_c, Q_c)
        return self[1](pq, combine=2)


def MLP(channels, bias=False, nonlin=LeakyReLU(negative_slope=0.2)):
    """mlp.py:7-11: [Linear(no bias) -> BatchNorm over rows -> LeakyReLU(0.2)]*"""
    return Seq(*[
        MLPBlock(Lin(channels[i - 1], channels[i], bias=bias), BatchNorm1d(channels[i]), nonlin)
        for i in range(1, len(channels))])


def VectorMLP(channels, batchnorm=True):
    """mlp.py:13-17: [Linear(no bias) -> VectorNonLin(BN)]*"""
    return Seq(*[
        VectorBlock(Lin(channels[i - 1], channels[i], bias=False),
                    VectorNonLin(channels[i], batchnorm=BatchNorm1d(channels[i]) if batchnorm else None))
        for i in range(1, len(channels))])


def run_mlp(mlp, x, residual=None):
    """Apply an MLP Sequential, adding ``residual`` inside the last block's fused kernel."""
    blocks = list(mlp)
    for blk in blocks[:-1]:
        x = blk(x)
    last = blocks[-1]
    if isinstance(last, MLPBlock):
        return last(x, residual)
    out = last(x)
    return out if residual is None else out + residual


def run_head(modules, x):
    """A head Sequential (or a slice of one) module by module, with `Linear(bias) -> LeakyReLU | ReLU` pairs run as one
    fused call (deltaconv/models/deltanet_segmentation.py:45-51: Linear(256, 128), LeakyReLU(0.2), Linear(128, classes))."""
    mods = []
    for mod in modules:                      # an MLP inside a head is a Sequential of blocks: walk the blocks themselves
        mods.extend(list(mod) if (isinstance(mod, Seq) and len(mod) and all(isinstance(b, MLPBlock) for b in mod)) else [mod])
    i = 0
    salt = 0
    while i < len(mods):
        mod = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(nxt, torch.nn.Dropout):
            salt += 1                        # every Dropout of the head has its own mask stream, fused or not
            # MLP block -> Dropout (deltanet_classification.py:34-36) on a handful of rows: the dropout runs inside the
            # block's own kernels (csrc/rowblock.hip) -- four ATen launches fewer per training step
            if (isinstance(mod, MLPBlock) and nxt.training and 0.0 < nxt.p < 1.0 and not nxt.inplace and torch.is_grad_enabled()
                    and mod.dropout_ok(x)):
                x = mod(x, dropout=(nxt.p, salt))
                i += 2
                continue
        slope = fused.slope_of(nxt) if isinstance(nxt, (LeakyReLU, torch.nn.ReLU)) else None
        if isinstance(mod, torch.nn.Linear) and mod.bias is not None and slope is not None and not getattr(nxt, "inplace", False):
            x = fused.linear_bias_act(x, mod.weight, mod.bias, slope)
            i += 2
        else:
            x = mod(x)
            i += 1
    return x


class ScalarVectorMLP(torch.nn.Module):
    """mlp.py:19-39"""

    def __init__(self, channels, nonlin=True, vector_stream=True):
        super().__init__()
        self.scalar_mlp = MLP(channels, nonlin=LeakyReLU(negative_slope=0.2) if nonlin else torch.nn.Identity())
        self.vector_mlp = VectorMLP(channels) if vector_stream else None

    def forward(self, x):
        assert self.vector_mlp is None or (self.vector_mlp is not None and type(x) is tuple)
        if type(x) is tuple:
            x, v = x
        x = self.scalar_mlp(x)
        if self.vector_mlp is not None:
            x = (x, self.vector_mlp(v))
        return x


class ScalarVectorIdentity(torch.nn.Module):
    """mlp.py:41-46"""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, input):
        return input
