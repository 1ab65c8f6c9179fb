"""Generate golden vectors by running the REAL reference (/root/reference, imported unmodified).

Runs only in the build container (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

The four third-party packages the reference imports are absent here; ``tools/ref_shims`` provides
pure-torch stand-ins (their defined semantics are listed in tools/ref_shims/README.md).  Each
fixture holds inputs + the reference's outputs in fp32 (its native numerics) and, where useful,
in fp64 (same code run on double tensors = "truth" for setting tolerances).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tools", "ref_shims"), "/root/reference", ROOT]

import deltaconv as ref                                   # noqa: E402  (the reference itself)
from deltaconv.geometry import grad_div_mls as R          # noqa: E402
from deltaconv.geometry import operators as RO            # noqa: E402
from torch_geometric.nn import knn_graph                  # noqa: E402  (stand-in, defines kNN order)
from deltaconv_amd.data import synthetic_batch            # noqa: E402
from tests.golden.probes import probe_vec, param_summaries, state_checksum  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def geometry_case(name, data, k, lam, h, use_normals, seed):
    """Every intermediate of SURVEY.md section 8(a) rows a1-a10, fp32 and fp64."""
    out = dict(pos=data.pos, batch=data.batch, k=k, lam=lam, h=h)
    g = torch.Generator().manual_seed(seed)
    nt = data.pos.shape[0]
    xs = torch.randn(nt, 5, generator=g)
    vs = torch.randn(2 * nt, 5, generator=g)
    out.update(x_in=xs, v_in=vs)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        pos = data.pos.to(dt)
        ei = knn_graph(pos, k, data.batch, loop=True, flow='target_to_source')
        if tag == "f32":
            out["edge_index"] = ei
            if not use_normals:
                out["edge_index10"] = knn_graph(pos, 10, data.batch, loop=True, flow='target_to_source')
        if use_normals:
            normal = data.norm.to(dt)
            xb, yb = R.build_tangent_basis(normal)
        else:
            ei10 = knn_graph(pos, 10, data.batch, loop=True, flow='target_to_source')
            normal, xb, yb = R.estimate_basis(pos, ei10, orientation=pos)
        coords = R.coords_projected(pos, normal, xb, yb, ei, k)
        dist = torch.linalg.norm(pos[ei[1]] - pos[ei[0]], dim=1)
        w = R.gaussian_weights(dist, k, data.batch, h)
        wls = R.weighted_least_squares(coords, w, k, lam)
        vmap = R.fit_vector_mapping(pos, normal, xb, yb, (ei[0], ei[1]), wls, coords)
        grad, div = R.build_grad_div(pos, normal, xb, yb, ei.clone(), data.batch, kernel_width=h, regularizer=lam)
        x, v = xs.to(dt), vs.to(dt)
        res = dict(normal=normal, x_basis=xb, y_basis=yb, coords=coords, weights=w, wls=wls, vmap=vmap,
                   grad_row=grad.row, grad_col=grad.col, grad_val=grad.value,
                   div_row=div.row, div_col=div.col, div_val=div.value,
                   grad_x=grad @ x, div_v=div @ v, curl_v=RO.curl(v, div), lap_x=RO.laplacian(x, grad, div),
                   hodge_v=RO.hodge_laplacian(v, grad, div), norm_v=RO.norm(v), IJ_v=RO.I_J(v))
        out.update({f"{key}_{tag}": val for key, val in res.items()})
    save(name, **out)


def shape_reg_case(name, data, k, lam, lam_shape, seed):
    """build_grad_div(..., shape_regularizer=lam_shape): operator values, fp32 and fp64."""
    out = dict(pos=data.pos, normal=data.norm, batch=data.batch, k=k, lam=lam, lam_shape=lam_shape)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        pos, normal = data.pos.to(dt), data.norm.to(dt)
        ei = knn_graph(pos, k, data.batch, loop=True, flow='target_to_source')
        if tag == "f32":
            out["edge_index"] = ei
        xb, yb = R.build_tangent_basis(normal)
        grad, div = R.build_grad_div(pos, normal, xb, yb, ei.clone(), data.batch, regularizer=lam,
                                     shape_regularizer=lam_shape)
        out.update({f"grad_val_{tag}": grad.value, f"div_val_{tag}": div.value})
    save(name, **out)


AGGR_CFGS = {"mean": dict(ci=8, co=16, centralized=False, vector=True, aggr="mean"),
             "min": dict(ci=8, co=8, centralized=False, vector=False, aggr="min"),
             "sumc": dict(ci=3, co=8, centralized=True, vector=True, aggr="sum")}


def nn_case(name, seed, cfgs=None):
    """DeltaConv layers (centralized / plain / scalar-only) fwd + input & parameter grads."""
    from deltaconv.nn import DeltaConv
    data = synthetic_batch(2, 96, seed=seed)
    k, lam = 12, 1e-3
    ei = knn_graph(data.pos, k, data.batch, loop=True, flow='target_to_source')
    out = dict(pos=data.pos, normal=data.norm, batch=data.batch, k=k, lam=lam, edge_index=ei)
    g = torch.Generator().manual_seed(seed)
    cfgs = cfgs or {"cent": dict(ci=3, co=8, centralized=True, vector=True),
                    "plain": dict(ci=8, co=16, centralized=False, vector=True),
                    "last": dict(ci=8, co=8, centralized=False, vector=False)}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        pos, normal = data.pos.to(dt), data.norm.to(dt)
        xb, yb = R.build_tangent_basis(normal)
        grad, div = R.build_grad_div(pos, normal, xb, yb, ei.clone(), data.batch, regularizer=lam)
        for cname, c in cfgs.items():
            torch.manual_seed(100 + seed)
            conv = DeltaConv(c["ci"], c["co"], depth=1, centralized=c["centralized"], vector=c["vector"],
                             aggr=c.get("aggr", "max")).to(dt)
            # non-trivial BN affine so that the test sees gamma/beta
            with torch.no_grad():
                for n_, p_ in conv.named_parameters():
                    if n_.endswith("bn.weight"):
                        p_.copy_(torch.linspace(0.5, 1.5, p_.numel()).to(dt))
                    if n_.endswith("bn.bias"):
                        p_.copy_(torch.linspace(-0.2, 0.2, p_.numel()).to(dt))
            conv.train()
            if tag == "f32":    # weights + pristine BN buffers, captured BEFORE the forward pass
                out.update({f"{cname}_sd_{kk}": vv.clone() for kk, vv in conv.state_dict().items()
                            if "num_batches" not in kk})
            gi = torch.Generator().manual_seed(seed + 17)
            x = torch.randn(pos.shape[0], c["ci"], generator=gi).to(dt).requires_grad_(True)
            v = torch.randn(2 * pos.shape[0], c["ci"], generator=gi).to(dt).requires_grad_(True)
            xo, vo = conv(x, v, grad, div, ei)
            wx = probe_vec(tuple(xo.shape), 11).to(dt)
            wv = probe_vec(tuple(vo.shape), 12).to(dt)
            loss = (xo * wx).sum() + ((vo * wv).sum() if c["vector"] else 0)
            loss.backward()
            if tag == "f32":
                out.update({f"{cname}_x": x.detach(), f"{cname}_v": v.detach()})
            out.update({f"{cname}_xo_{tag}": xo, f"{cname}_vo_{tag}": vo,
                        f"{cname}_dx_{tag}": x.grad, f"{cname}_dv_{tag}": v.grad})
            for n_, p_ in conv.named_parameters():
                if p_.grad is not None:
                    out[f"{cname}_g_{n_}_{tag}"] = p_.grad
            for n_, b_ in conv.named_buffers():
                if "running" in n_:
                    out[f"{cname}_buf_{n_}_{tag}"] = b_
    save(name, **out)


def model_case(name, kind, seed, B, N, k, lam, normals=True, **model_kw):
    """Whole-model step (train-mode BN, dropout off): logits, loss, gradient summaries."""
    sys.path.insert(0, "/root/reference/experiments")
    from utils import calc_loss
    from deltaconv.models import DeltaNetClassification, DeltaNetSegmentation
    seg = kind == "seg"
    data = synthetic_batch(B, N, seed=seed, normals=normals, num_classes=model_kw.get("num_classes", 40),
                           per_point_labels=seg, categories=16 if model_kw.get("categorical_vector") else 0)
    out = dict(pos=data.pos, batch=data.batch, y=data.y, k=k, lam=lam, seed=seed)
    if normals:
        out["normal"] = data.norm
    if data.category is not None:
        out["category"] = data.category
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.manual_seed(1)                                  # train_*.py --seed 1 default init
        cls = DeltaNetSegmentation if seg else DeltaNetClassification
        model = cls(num_neighbors=k, grad_regularizer=lam, **model_kw)
        if tag == "f32":
            out["state_checksum"] = state_checksum(model)
        model = model.to(dt).train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
        d = data
        from types import SimpleNamespace as NS
        d = NS(pos=data.pos.to(dt), batch=data.batch, norm=(data.norm.to(dt) if normals else None), x=None,
               category=(data.category.to(dt) if data.category is not None else None))
        if not normals:
            del d.norm
        logits = model(d)
        loss = calc_loss(logits, data.y, smoothing=not seg)
        loss.backward()
        names, norms, dots = param_summaries(model)
        out.update({f"logits_{tag}": logits, f"loss_{tag}": loss, f"gnorm_{tag}": np.array(norms),
                    f"gdot_{tag}": np.array(dots)})
        out["gnames"] = np.array(names)
        first = dict(model.named_parameters())
        for key in ("deltanet_base.convs.0.s_mlp_max.0.0.weight", "deltanet_base.convs.1.v_mlp.0.0.weight"):
            out[f"g_{key}_{tag}"] = first[key].grad
        bufs = dict(model.named_buffers())
        out[f"rm_embed_{tag}"] = bufs[("lin_global" if seg else "lin_embedding") + ".0.1.bn.running_mean"]
        out[f"rv_embed_{tag}"] = bufs[("lin_global" if seg else "lin_embedding") + ".0.1.bn.running_var"]
    save(name, **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    geometry_case("geom_normals_B2_N128_k20", synthetic_batch(2, 128, seed=3), 20, 1e-3, 1.0, True, 3)
    geometry_case("geom_ragged_dups_k30", synthetic_batch(3, 0, seed=4, sizes=[64, 100, 80], dup_frac=0.05),
                  30, 1e-3, 1.5, True, 4)
    geometry_case("geom_nonormals_N200_k10", synthetic_batch(1, 200, seed=5, normals=False, jitter=0.005),
                  10, 1e-2, 1.0, False, 5)
    nn_case("deltaconv_layers", 6)
    nn_case("deltaconv_layers_aggr", 10, AGGR_CFGS)        # DeltaConv(aggr='mean' | 'min' | 'sum')
    shape_reg_case("geom_shape_regularizer", synthetic_batch(2, 96, seed=11), 16, 1e-3, 5e-2, 11)
    model_case("model_cls_B4_N256_k20", "cls", 7, 4, 256, 20, 1e-3, in_channels=3, num_classes=40)
    model_case("model_seg_B2_N256_k20", "seg", 8, 2, 256, 20, 1e-3, in_channels=3, num_classes=50,
               categorical_vector=True)
    model_case("model_cls_nonormals_B2_N256_k20", "cls", 9, 2, 256, 20, 1e-2, normals=False,
               in_channels=3, num_classes=15, conv_channels=[64, 64, 64, 128])
    # the shapeseg family (train_shapeseg.py:68-78: mlp_depth = 1, k = 30, many equal layers) at reduced width
    model_case("model_seg_depth1_B2_N1024_k30", "seg", 12, 2, 1024, 30, 1e-3, in_channels=3, num_classes=8,
               conv_channels=[32] * 4, mlp_depth=1, embedding_size=128)
