"""Parity report: measured deviations of the HIP path (through the C ABI) from (a) the reference's golden
vectors (fp32 = its native numerics, fp64 = the same code on doubles) and (b) the CPU oracle, per stage.
Runs on the GPU box; the output is committed under profiles/ next to the tolerances the tests assert."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltaconv_amd as dc                                    # noqa: E402
import oracle                                                 # noqa: E402
from oracle import geometry as geo                            # noqa: E402
from tests.helpers import load_golden, rel_err                # noqa: E402
from tests.golden.probes import param_summaries               # noqa: E402
from deltaconv_amd.data import Batch, synthetic_batch         # noqa: E402
from deltaconv_amd.utils import calc_loss                     # noqa: E402

DEV = "cuda"
rows = []


def rec(stage, what, err, tol, note=""):
    rows.append((stage, what, err, tol, note))
    print(f"{stage:34s} {what:30s} {err:10.3e}  (tol {tol:g}) {note}")


print(f"# parity report  torch {torch.__version__}  {torch.cuda.get_device_name(0)}  lib v{dc._lib.lib.load().dc_version()}")
print("# error = max|a-b| / max|b| unless noted; 'bit' rows are exact comparisons (0 = identical)")
for name in ("geom_normals_B2_N128_k20", "geom_ragged_dups_k30", "geom_nonormals_N200_k10"):
    g = load_golden(name)
    k, nt = g["k"], g["pos"].shape[0]
    gr = dc.geometry.Graph.knn(g["pos"].to(DEV), k, g["batch"].to(DEV))
    rec(name, "kNN edge_index (bit)", float((gr.edge_index.cpu() != g["edge_index"]).sum()), 0)
    nbr = geo.nbr_from_edge_index(g["edge_index"], k)
    for tag in ("f64", "f32"):
        fr = [g[f"{n}_{tag}"].float().to(DEV) for n in ("normal", "x_basis", "y_basis")]
        G, D = dc.geometry.build_grad_div(g["pos"].to(DEV), *fr, g["edge_index"].to(DEV), g["batch"].to(DEV),
                                          kernel_width=g["h"], regularizer=g["lam"])
        rec(name, f"grad values vs reference {tag}", rel_err(G.coef.reshape(-1), g[f"grad_val_{tag}"]), 2e-5 if tag == "f64" else 2e-3)
        rec(name, f"div values vs reference {tag}", rel_err(D.coef.reshape(-1), g[f"div_val_{tag}"]), 2e-5 if tag == "f64" else 2e-3)
    rec(name, "reference fp32 vs its own fp64", max(rel_err(g["grad_val_f32"], g["grad_val_f64"]),
                                                    rel_err(g["div_val_f32"], g["div_val_f64"])), float("nan"), "(context)")
    Gd = dc.geometry.SparseOp("grad", gr, g["grad_val_f32"].view(nt, k, 2).contiguous().to(DEV))
    Dd = dc.geometry.SparseOp("div", gr, g["div_val_f32"].view(nt, k, 2).contiguous().to(DEV))
    x, v = g["x_in"].to(DEV), g["v_in"].to(DEV)
    for key, val in dict(grad_x=Gd @ x, div_v=Dd @ v, curl_v=dc.geometry.curl(v, Dd), lap_x=dc.geometry.laplacian(x, Gd, Dd),
                         hodge_v=dc.geometry.hodge_laplacian(v, Gd, Dd)).items():
        rec(name, f"apply {key} vs reference f32", rel_err(val, g[f"{key}_f32"]), 1e-5)

for name, (kind, kw, normals) in {
        "model_cls_B4_N256_k20": ("cls", dict(in_channels=3, num_classes=40), True),
        "model_seg_B2_N256_k20": ("seg", dict(in_channels=3, num_classes=50, categorical_vector=True), True),
        "model_cls_nonormals_B2_N256_k20": ("cls", dict(in_channels=3, num_classes=15, conv_channels=[64, 64, 64, 128]), False)}.items():
    g = load_golden(name)
    torch.manual_seed(1)
    cls = dc.models.DeltaNetSegmentation if kind == "seg" else dc.models.DeltaNetClassification
    model = cls(num_neighbors=g["k"], grad_regularizer=g["lam"], **kw).to(DEV).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    data = Batch(g["pos"], g["batch"], g["normal"] if normals else None, None, g["y"], g["category"] if "category" in g else None).to(DEV)
    logits = model(data)
    loss = calc_loss(logits, data.y, smoothing=(kind != "seg"))
    loss.backward()
    tol = 1e-3 if normals else 5e-3
    rec(name, "logits vs reference f64", rel_err(logits, g["logits_f64"]), tol)
    rec(name, "logits vs reference f32", rel_err(logits, g["logits_f32"]), tol)
    rec(name, "reference f32 vs its own f64", rel_err(g["logits_f32"], g["logits_f64"]), float("nan"), "(context)")
    rec(name, "loss rel. diff vs reference f64", abs(float(loss.detach()) - float(g["loss_f64"])) / abs(float(g["loss_f64"])), tol)
    _, norms, _ = param_summaries(model)
    gn = g["gnorm_f64"].numpy()
    rec(name, "param-grad norms vs reference f64", float(np.max(np.abs(np.array(norms) - gn) / (gn + 1e-3 * gn.max()))), 5 * tol)

b = synthetic_batch(8, 1024, seed=40)
torch.manual_seed(1)
r32 = oracle.models.DeltaNetClassification(3, 40).train()
r64 = oracle.models.DeltaNetClassification(3, 40).double().train()
r64.load_state_dict(r32.state_dict())
model = dc.models.DeltaNetClassification(3, 40)
model.load_state_dict(r32.state_dict())
model = model.to(DEV).train()
for mm in (r32, r64, model):
    for m in mm.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
l32 = r32(b); calc_loss(l32, b.y).backward()
l64 = r64(Batch(b.pos.double(), b.batch, b.norm.double(), None, b.y)); calc_loss(l64, b.y).backward()
bd = b.to(DEV)
ld = model(bd); calc_loss(ld, bd.y).backward()
rec("ModelNet40 B=8 N=1024 k=20", "logits HIP vs oracle f64", rel_err(ld, l64), 3 * rel_err(l32, l64) + 1e-3)
rec("ModelNet40 B=8 N=1024 k=20", "logits oracle f32 vs f64", rel_err(l32, l64), float("nan"), "(context)")
gmax = max(float(p.grad.abs().max()) for p in r64.parameters() if p.grad is not None)
wh = wr = 0.0
for p1, p2, p3 in zip(model.parameters(), r32.parameters(), r64.parameters()):
    if p3.grad is None:
        continue
    sc = max(float(p3.grad.abs().max()), 1e-3 * gmax)
    wh = max(wh, float((p1.grad.cpu().double() - p3.grad).abs().max()) / sc)
    wr = max(wr, float((p2.grad.double() - p3.grad).abs().max()) / sc)
rec("ModelNet40 B=8 N=1024 k=20", "worst param grad HIP vs f64", wh, 3 * wr + 1e-3)
rec("ModelNet40 B=8 N=1024 k=20", "worst param grad oracle f32 vs f64", wr, float("nan"), "(context)")
bad = [r for r in rows if r[3] == r[3] and r[2] > r[3]]
print(f"# {len(rows)} rows, {len(bad)} over tolerance")
sys.exit(1 if bad else 0)
