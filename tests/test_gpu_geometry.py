"""GPU parity: every geometry / operator kernel called through the C ABI (deltaconv_amd._lib ->
libdeltaconv_hip.so) vs the CPU oracle and the reference's golden vectors.

Tolerances are scale-relative (tests/helpers.rel_err = max|a-b| / max|b|):
  * kNN indices, CSC, max-aggregation values/slots: bit-exact.
  * MLS operators vs fp64 truth 2e-5 (fp32 output rounding + fp32-rounded frames), vs the
    reference's native fp32 2e-3 (that is the reference's own fp32 LU error, see test_hostcheck).
  * applies / transposes: 1e-5 (fp32 sums of <= 64 products, different association).
"""
import pytest
import torch

from oracle import geometry as geo
from tests.helpers import load_golden, rel_err
from deltaconv_amd.data import synthetic_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GEOM = ["geom_normals_B2_N128_k20", "geom_ragged_dups_k30", "geom_nonormals_N200_k10"]


def dc():
    import deltaconv_amd
    return deltaconv_amd


def batch_of(g):
    return g["batch"].to(DEV)


# ---------------------------------------------------------------------------------- kNN
@pytest.mark.parametrize("name", GEOM)
@pytest.mark.parametrize("lanes", [0, 1, 8, 64])
def test_knn_golden_bit_exact(name, lanes):
    from deltaconv_amd.geometry import Graph
    g = load_golden(name)
    gr = Graph.knn(g["pos"].to(DEV), g["k"], batch_of(g), lanes_per_query=lanes)
    assert torch.equal(gr.edge_index.cpu(), g["edge_index"])


@pytest.mark.parametrize("B,N,k,kw", [(4, 1024, 20, {}), (1, 4096, 30, {}), (2, 2048, 20, dict(dup_frac=0.03)),
                                      (3, 300, 10, {}), (2, 100, 64, {}), (1, 5000, 16, {}), (2, 40, 40, {})])
def test_knn_vs_oracle(B, N, k, kw):
    from deltaconv_amd.geometry import Graph
    b = synthetic_batch(B, N, seed=31, **kw)
    ref = geo.knn(b.pos, k, geo.cloud_ptr(b.batch))
    for lanes in (0, 1, 8) + ((64,) if N <= 4096 else ()):     # 64 = wave-per-query selection kernel
        gr = Graph.knn(b.pos.to(DEV), k, b.batch.to(DEV), lanes_per_query=lanes)
        got = gr.nbr.cpu().long()
        bad = (got != ref).any(1).nonzero().flatten()
        assert bad.numel() == 0, f"lanes={lanes}: {bad.numel()} rows differ, first {bad[:3].tolist()}: " \
                                 f"{got[bad[:1]].tolist()} vs {ref[bad[:1]].tolist()}"


@pytest.mark.parametrize("N,k,dups", [(1024, 20, 200), (700, 30, 700), (4096, 10, 300), (130, 40, 100)])
def test_knn_massive_ties(N, k, dups):
    """`dups` points coincide (and a second clump shares one distance to them): far more than 64 candidates
    tie at the k-th distance, which sends the wave kernel down its exact radix-select path; ties must come
    out in index order like the oracle's stable sort."""
    from deltaconv_amd.geometry import Graph
    g = torch.Generator().manual_seed(5)
    pos = torch.randn(2 * N, 3, generator=g)
    for c in range(2):
        idx = torch.randperm(N, generator=g)[:dups] + c * N
        pos[idx] = pos[idx[0]].clone()
    batch = torch.arange(2).repeat_interleave(N)
    ref = geo.knn(pos, k, geo.cloud_ptr(batch))
    for lanes in (0, 8, 64):
        got = Graph.knn(pos.to(DEV), k, batch.to(DEV), lanes_per_query=lanes).nbr.cpu().long()
        assert torch.equal(got, ref), f"lanes={lanes}"


def test_knn_ragged_and_api():
    from deltaconv_amd.geometry import knn_graph
    b = synthetic_batch(3, 0, seed=32, sizes=[64, 1000, 257])
    ei = knn_graph(b.pos.to(DEV), 20, b.batch.to(DEV), loop=True, flow='target_to_source')
    ref = geo.edge_index_from_nbr(geo.knn(b.pos, 20, geo.cloud_ptr(b.batch)))
    assert ei.dtype == torch.long and torch.equal(ei.cpu(), ref)
    ei1 = knn_graph(b.pos[:64].to(DEV), 20)                     # batch=None
    assert torch.equal(ei1.cpu(), ref[:, :64 * 20])


# ---------------------------------------------------------------------------------- bases
def test_tangent_basis():
    from deltaconv_amd.geometry import build_tangent_basis
    torch.manual_seed(0)
    n = torch.randn(5000, 3)
    n[:4] = torch.tensor([[1., 0, 0], [-1., 0, 0], [0.9, 0.43589, 0], [0, 0, 1.]])
    n = n / n.norm(dim=1, keepdim=True)
    xb, yb = build_tangent_basis(n.to(DEV))
    xo, yo = geo.build_tangent_basis(n)
    assert rel_err(xb, xo) < 1e-6 and rel_err(yb, yo) < 1e-6


def test_estimate_basis():
    from deltaconv_amd.geometry import estimate_basis, Graph
    g = load_golden("geom_nonormals_N200_k10")
    pos = g["pos"].to(DEV)
    normal, xb, yb = estimate_basis(pos, g["edge_index10"].to(DEV), orientation=pos)
    assert rel_err(normal, g["normal_f64"]) < 1e-5
    sgn = torch.sign((xb.cpu() * g["x_basis_f64"].float()).sum(1, keepdim=True))
    assert rel_err(xb.cpu() * sgn, g["x_basis_f64"]) < 1e-4 and rel_err(yb.cpu() * sgn, g["y_basis_f64"]) < 1e-4
    # bigger, batched, Graph input, no orientation: orthonormal right-handed frames
    b = synthetic_batch(4, 2048, seed=33, normals=False, jitter=0.005)
    p = b.pos.to(DEV)
    gr = Graph.knn(p, 10, b.batch.to(DEV))
    n2, x2, y2 = estimate_basis(p, gr)
    basis = torch.stack([n2, x2, y2], -1)
    eye = torch.eye(3, device=DEV).expand(p.shape[0], 3, 3)
    assert torch.allclose(basis.transpose(1, 2) @ basis, eye, atol=1e-5)
    assert (torch.linalg.cross(x2, y2) * n2).sum(1).min() > 0
    no, xo, yo = geo.estimate_basis(b.pos.double(), gr.nbr.cpu().long())
    assert ((n2.cpu().double() * no).sum(1).abs() - 1).abs().max() < 1e-4


# ---------------------------------------------------------------------------------- MLS operators
def _build(pos, normal, xb, yb, nbr64, batch, h, lam, normalized=True):
    from deltaconv_amd.geometry import build_grad_div
    ei = geo.edge_index_from_nbr(nbr64).to(DEV)
    return build_grad_div(pos.to(DEV), normal.to(DEV), xb.to(DEV), yb.to(DEV), ei, batch.to(DEV),
                          kernel_width=h, regularizer=lam, normalized=normalized)


@pytest.mark.parametrize("name", GEOM)
def test_build_grad_div_golden(name):
    g = load_golden(name)
    k = g["k"]
    nt = g["pos"].shape[0]
    nbr = geo.nbr_from_edge_index(g["edge_index"], k)
    for tag, tol in (("f64", 2e-5), ("f32", 2e-3)):
        fr = [g[f"{n}_{tag}"].float() for n in ("normal", "x_basis", "y_basis")]
        grad, div = _build(g["pos"], *fr, nbr, g["batch"], g["h"], g["lam"])
        assert grad.size(0) == 2 * nt and grad.size(1) == nt and div.size(0) == nt and div.size(1) == 2 * nt
        for op, nm in ((grad, "grad"), (div, "div")):
            row, col, val = op.coo()
            assert torch.equal(row.cpu(), g[f"{nm}_row_f32"]) and torch.equal(col.cpu(), g[f"{nm}_col_f32"])
            assert rel_err(val, g[f"{nm}_val_{tag}"]) < tol, (nm, tag)


@pytest.mark.parametrize("lam,normalized,B,N,k", [(1e-3, True, 8, 1024, 20), (1e-8, False, 2, 512, 20),
                                                  (1e-8, True, 2, 512, 20), (1e-2, True, 2, 2048, 30)])
def test_build_grad_div_vs_oracle_fp64(lam, normalized, B, N, k):
    b = synthetic_batch(B, N, seed=34)
    ptr = geo.cloud_ptr(b.batch)
    nbr = geo.knn(b.pos, k, ptr)
    xb, yb = geo.build_tangent_basis(b.norm)
    grad, div = _build(b.pos, b.norm, xb, yb, nbr, b.batch, 1.0, lam, normalized)
    Go, Do = geo.build_grad_div(b.pos.double(), b.norm.double(), xb.double(), yb.double(), nbr, ptr, 1.0, lam,
                                normalized=normalized)
    tol = 1e-6 if lam >= 1e-4 else 1e-4
    assert rel_err(grad.coef, Go.coef) < tol and rel_err(div.coef, Do.coef) < tol
    assert not torch.isnan(grad.coef).any() and not torch.isnan(div.coef).any()


def test_build_grad_div_shape_regularizer_golden():
    """build_grad_div(shape_regularizer=...) on the GPU against the reference's fp32 / fp64 values."""
    import deltaconv_amd as dc
    g = load_golden("geom_shape_regularizer")
    pos, normal, batch = g["pos"].to(DEV), g["normal"].to(DEV), g["batch"].to(DEV)
    ei = g["edge_index"].to(DEV)
    xb, yb = dc.geometry.build_tangent_basis(normal)
    grad, div = dc.geometry.build_grad_div(pos, normal, xb, yb, ei, batch, regularizer=float(g["lam"]),
                                           shape_regularizer=float(g["lam_shape"]))
    for tag, tol in (("f32", 2e-3), ("f64", 2e-5)):
        assert rel_err(grad.coef.reshape(-1), g[f"grad_val_{tag}"]) < tol
        assert rel_err(div.coef.reshape(-1), g[f"div_val_{tag}"]) < tol
    grad0, div0 = dc.geometry.build_grad_div(pos, normal, xb, yb, ei, batch, regularizer=float(g["lam"]))
    assert torch.equal(grad0.coef, grad.coef) and rel_err(div0.coef.reshape(-1), g["div_val_f64"]) > 1e-3


def test_build_grad_div_with_frames_formed_inside():
    """build_grad_div(pos, normal, None, None, ...) -- the model's path: dc_mls_assemble_normals forms the frames of
    build_tangent_basis in the assembly's first launch -- gives the bits of build_tangent_basis + build_grad_div."""
    import deltaconv_amd as dc
    b = synthetic_batch(3, 0, seed=36, sizes=[700, 1024, 33]).to(DEV)
    gr = dc.geometry.Graph.knn(b.pos, 20, b.batch)
    xb, yb = dc.geometry.build_tangent_basis(b.norm)
    g0, d0 = dc.geometry.build_grad_div(b.pos, b.norm, xb, yb, gr, b.batch, regularizer=1e-3)
    g1, d1 = dc.geometry.build_grad_div(b.pos, b.norm, None, None, gr, b.batch, regularizer=1e-3)
    assert torch.equal(g0.coef, g1.coef) and torch.equal(d0.coef, d1.coef)


# ---------------------------------------------------------------------------------- CSC
def test_csc():
    from deltaconv_amd.geometry import Graph
    b = synthetic_batch(3, 0, seed=35, sizes=[700, 1024, 33], dup_frac=0.05)
    gr = Graph.knn(b.pos.to(DEV), 20, b.batch.to(DEV))
    tptr, tedge = (t.cpu() for t in gr.csc())
    n, k = gr.n, gr.k
    flat = gr.nbr.cpu().reshape(-1).long()
    assert int(tptr[0]) == 0 and int(tptr[-1]) == n * k
    assert torch.equal(torch.bincount(flat, minlength=n).to(torch.int32), tptr[1:] - tptr[:-1])
    assert torch.equal(torch.sort(tedge).values, torch.arange(n * k, dtype=torch.int32))
    assert bool((flat[tedge.long()] == torch.arange(n).repeat_interleave((tptr[1:] - tptr[:-1]).long())).all())
    seg_start = torch.zeros(n * k, dtype=torch.bool); seg_start[tptr[:-1].long().clamp(max=n * k - 1)] = True
    asc = (tedge[1:] > tedge[:-1]) | seg_start[1:]
    assert bool(asc.all()), "columns must be sorted by edge id (deterministic transposed sums)"
    gr2 = Graph.knn(b.pos.to(DEV), 20, b.batch.to(DEV))
    t2 = gr2.csc()
    assert torch.equal(t2[0].cpu(), tptr) and torch.equal(t2[1].cpu(), tedge)      # run-to-run identical


def test_csc_high_in_degree():
    """130 exact copies of one point: the tie rule sends all of them to the lowest-index copies, whose in-degree then
    exceeds a wavefront (the rank kernel walks such a column in chunks of 64)."""
    from deltaconv_amd.geometry import Graph
    torch.manual_seed(5)
    pos = torch.randn(600, 3)
    pos[40:170] = pos[40]
    batch = torch.zeros(600, dtype=torch.long)
    gr = Graph.knn(pos.to(DEV), 20, batch.to(DEV))
    tptr, tedge = (t.cpu() for t in gr.csc())
    n, k = gr.n, gr.k
    deg = tptr[1:] - tptr[:-1]
    assert int(deg.max()) > 64
    flat = gr.nbr.cpu().reshape(-1).long()
    assert torch.equal(torch.bincount(flat, minlength=n).to(torch.int32), deg)
    assert torch.equal(torch.sort(tedge).values, torch.arange(n * k, dtype=torch.int32))
    assert bool((flat[tedge.long()] == torch.arange(n).repeat_interleave(deg.long())).all())
    for j in torch.nonzero(deg > 64).flatten().tolist():
        col = tedge[int(tptr[j]):int(tptr[j + 1])]
        assert bool((col[1:] > col[:-1]).all())


def test_csc_cloud_builder_equals_general_builder():
    """dc_csc_build_clouds (count + scan + fill of a cloud in one workgroup on LDS counters) returns exactly the tptr /
    tedge of dc_csc_build: ragged clouds, duplicate points (hub columns), k = 20 and 30."""
    from deltaconv_amd._lib import lib
    from deltaconv_amd.geometry import Graph
    for k, sizes in ((20, [512, 700, 300, 4096]), (30, [64, 1000, 2048])):
        b = synthetic_batch(len(sizes), 0, seed=21, sizes=sizes, dup_frac=0.05).to(DEV)
        gr = Graph.knn(b.pos, k, b.batch)
        n = gr.n
        outs = []
        for name in ("dc_csc_build", "dc_csc_build_clouds"):
            tptr = torch.full((n + 1,), -1, dtype=torch.int32, device=DEV)
            tedge = torch.full((n * k,), -1, dtype=torch.int32, device=DEV)
            ws = torch.empty(n * (k + 1), dtype=torch.int32, device=DEV)
            if name == "dc_csc_build":
                lib.call(name, gr.nbr, gr.ptr, gr.num_clouds, n, k, tptr, tedge, ws, ws.numel() * 4)
            else:
                lib.call(name, gr.nbr, gr.ptr, gr.num_clouds, n, gr.max_cloud, k, tptr, tedge, ws, ws.numel() * 4)
            outs.append((tptr, tedge))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ---------------------------------------------------------------------------------- applies
@pytest.fixture(scope="module")
def ops():
    """A graph + operators on the device and their oracle twins on the CPU (same coefficient values)."""
    from deltaconv_amd.geometry import Graph, build_grad_div, build_tangent_basis
    b = synthetic_batch(3, 0, seed=36, sizes=[512, 700, 300], dup_frac=0.03)
    bd = b.to(DEV)
    gr = Graph.knn(bd.pos, 20, bd.batch)
    xb, yb = build_tangent_basis(bd.norm)
    grad, div = build_grad_div(bd.pos, bd.norm, xb, yb, gr, bd.batch)
    nbr = gr.nbr.cpu().long()
    return dict(graph=gr, grad=grad, div=div, nbr=nbr, n=gr.n,
                G=geo.EllOp("grad", nbr, grad.coef.cpu()), D=geo.EllOp("div", nbr, div.coef.cpu()))


@pytest.mark.parametrize("C", [1, 3, 5, 8, 64, 128, 256])
def test_apply_forward_backward(ops, C):
    import deltaconv_amd.geometry as dg
    from deltaconv_amd import _ops
    torch.manual_seed(C)
    n, G, D = ops["n"], ops["G"], ops["D"]
    x = torch.randn(n, C, requires_grad=True)
    v = torch.randn(2 * n, C, requires_grad=True)
    xd = x.detach().to(DEV).requires_grad_(True)
    vd = v.detach().to(DEV).requires_grad_(True)
    wy = torch.randn(2 * n, C)
    wz = torch.randn(n, C)

    def check(out_d, out_o, ins_d, ins_o, w, what):
        assert rel_err(out_d, out_o) < 1e-5, what
        gd = torch.autograd.grad(out_d, ins_d, w.to(DEV))
        go = torch.autograd.grad(out_o, ins_o, w)
        for a, b_ in zip(gd, go):
            assert rel_err(a, b_) < 1e-5, what + " backward"

    check(ops["grad"] @ xd, G @ x, [xd], [x], wy, "grad@x")
    check(ops["div"] @ vd, D @ v, [vd], [v], wz, "div@v")
    check(dg.curl(vd, ops["div"]), geo.curl(v, D), [vd], [v], wz, "curl")
    check(dg.laplacian(xd, ops["grad"], ops["div"]), geo.laplacian(x, G, D), [xd], [x], wz, "laplacian")
    check(dg.hodge_laplacian(vd, ops["grad"], ops["div"]), geo.hodge_laplacian(v, G, D), [vd], [v], wy, "hodge")
    check(_ops.div_curl_norm(vd, ops["div"]), torch.cat([D @ v, geo.curl(v, D), geo.norm(v)], 1), [vd], [v],
          torch.randn(n, 3 * C), "div|curl|norm")
    assert rel_err(dg.norm(vd), geo.norm(v)) < 1e-6 and rel_err(dg.I_J(vd), geo.I_J(v)) == 0
    # max aggregation: values and winning slots bit-exact, backward to the first maximal slot
    h = torch.randn(n, C, requires_grad=True)
    hd = h.detach().to(DEV).requires_grad_(True)
    out_d, arg = _ops._KnnMax.apply(hd, ops["graph"])
    out_o, arg_o = h[ops["nbr"]].max(dim=1)
    assert torch.equal(out_d.cpu(), out_o.detach()) and torch.equal(arg.cpu().long(), arg_o)
    (gd,) = torch.autograd.grad(out_d, hd, wz.to(DEV))
    (go,) = torch.autograd.grad(out_o, h, wz)
    assert rel_err(gd, go) < 1e-6


@pytest.mark.parametrize("aggr", ["sum", "add", "mean", "min"])
@pytest.mark.parametrize("C", [3, 64])
def test_other_aggregations(ops, aggr, C):
    """DeltaConv(aggr=...) beyond the default: torch_scatter 'sum' / 'add' / 'mean' / 'min' over the kNN graph, forward and
    backward vs the gather formulation on the CPU."""
    from deltaconv_amd import _ops
    n = ops["n"]
    torch.manual_seed(C)
    h = torch.randn(n, C, requires_grad=True)
    hd = h.detach().to(DEV).requires_grad_(True)
    w = torch.randn(n, C)
    gathered = h[ops["nbr"]]
    ref = {"sum": gathered.sum(1), "add": gathered.sum(1), "mean": gathered.mean(1), "min": gathered.min(1).values}[aggr]
    out = _ops.knn_aggregate(hd, ops["graph"], aggr)
    assert rel_err(out, ref) < (1e-6 if aggr != "min" else 1e-12)
    (gd,) = torch.autograd.grad(out, hd, w.to(DEV))
    (go,) = torch.autograd.grad(ref, h, w)
    assert rel_err(gd, go) < 1e-6
    out2 = _ops.knn_aggregate(hd, ops["graph"], aggr)
    assert torch.equal(out, out2)
    with pytest.raises(ValueError):
        _ops.knn_aggregate(hd, ops["graph"], "mul")


def test_max_ties_first_slot(ops):
    """Duplicate points carry identical features -> ties; the first slot of the k-list wins."""
    from deltaconv_amd import _ops
    n = ops["n"]
    h = torch.zeros(n, 8)
    h[::7] = 1.0
    out, arg = _ops._KnnMax.apply(h.to(DEV), ops["graph"])
    o2, a2 = h[ops["nbr"]].max(dim=1)
    assert torch.equal(out.cpu(), o2) and torch.equal(arg.cpu().long(), a2)


@pytest.mark.parametrize("C", [3, 64, 128])
@pytest.mark.parametrize("two", [False, True])
def test_grad_T_sum_matches_separate_passes(ops, C, two):
    """dc_apply_grad_T_sum (gradient accumulation folded into the transposed apply) == (a + b) + grad^T dy of the
    separate passes, bit for bit (same summation order), with strided operands."""
    from deltaconv_amd._lib import lib
    n, gr, grad = ops["n"], ops["graph"], ops["grad"]
    tptr, tedge = gr.csc()
    torch.manual_seed(C + two)
    k = gr.k
    wide = torch.randn(2 * n, 2 * C + 8, device=DEV)
    dy = wide[:, 8:8 + C]                                   # a column block of a wider tensor (like d v_cat)
    a = torch.randn(n, C, device=DEV)
    b = torch.randn(n, C + 4, device=DEV)[:, :C] if two else None
    ref = (a + b) if two else a.clone()
    lib.call("dc_apply_grad_T", grad.coefT(), tptr, tedge, n, k, dy, C, dy.stride(0), ref, C, 1)
    out = torch.empty(n, C, device=DEV)
    lib.call("dc_apply_grad_T_sum", grad.coefT(), tptr, tedge, n, k, dy, C, dy.stride(0), a, C, b,
             b.stride(0) if two else 0, out, C)
    assert torch.equal(out, ref)


def test_applies_deterministic(ops):
    n = ops["n"]
    x = torch.randn(n, 64, device=DEV, requires_grad=True)
    outs = []
    for _ in range(2):
        y = ops["div"] @ (ops["grad"] @ x)
        (g,) = torch.autograd.grad(y, x, torch.ones_like(y))
        outs.append((y.detach().clone(), g.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_property_suite_on_gpu():
    """The reference's analytic checks of test_build_grad_div (test_grad_div_mls.py:278-401) on the
    HIP operators: de Rham identities, gradient of a height field, gauge equivariance."""
    import deltaconv_amd.geometry as dg
    N, k = 1000, 20
    torch.manual_seed(42)
    coords = torch.rand(N, 2) * 2 - 1
    cc = torch.rand(6)
    x, y = coords.T
    f = (cc[0] + cc[1] * x + cc[2] * y + cc[3] * x * x + cc[4] * x * y + cc[5] * y * y)[:, None]
    pos = torch.cat([coords, f], 1)
    dfdx = torch.stack([torch.ones(N), torch.zeros(N), cc[1] + 2 * cc[3] * x + cc[4] * y], 1)
    dfdy = torch.stack([torch.zeros(N), torch.ones(N), cc[2] + cc[4] * x + 2 * cc[5] * y], 1)
    normal = torch.linalg.cross(dfdx, dfdy)
    normal = normal / normal.norm(dim=1, keepdim=True)
    xb = dfdx / dfdx.norm(dim=1, keepdim=True)
    yb = torch.linalg.cross(normal, xb)
    P, Nn, Xb, Yb = (t.to(DEV) for t in (pos, normal, xb, yb))
    ei = dg.knn_graph(P, k)
    G, D = dg.build_grad_div(P, Nn, Xb, Yb, ei, regularizer=1e-8, normalized=False)
    one = torch.ones(N, 1, device=DEV)
    assert torch.allclose(G @ one, torch.zeros(2 * N, 1, device=DEV), atol=1e-2)
    assert dg.laplacian(one, G, D).abs().mean() < 1e-2
    assert dg.curl(G @ P[:, 0:1], D).pow(2).mean() < 1e-2
    assert (D @ dg.J(G @ P[:, 0:1])).pow(2).mean() < 1e-2
    gx, gy = (G @ f.to(DEV)).view(N, 2).T
    assert torch.allclose(gx, Xb[:, 2], atol=1e-2) and torch.allclose(gy, Yb[:, 2], atol=1e-2)
    H = dg.laplacian(P, G, D)
    assert torch.allclose(-(H * Nn).sum(1, keepdim=True), H.norm(dim=1, keepdim=True), atol=1e-2)
    # the gauge check is marginal at atol=1e-3 and input dependent (the infinity-norm normalisation
    # is not rotation invariant): reproduce the RNG stream of the reference test up to this point
    torch.manual_seed(42)
    _ = torch.rand(N, 2), torch.rand(6), torch.rand(N, 1), torch.rand(2 * N, 1), torch.rand(N, 1)
    ang = torch.rand(N) * 2 * torch.pi
    xr = geo.rotate_around(xb, normal, ang)
    yr = torch.linalg.cross(normal, xr)
    G1, D1 = dg.build_grad_div(P, Nn, Xb, Yb, ei, regularizer=1e-8)
    G2, D2 = dg.build_grad_div(P, Nn, xr.to(DEV), yr.to(DEV), ei, regularizer=1e-8)
    u = torch.rand(N, 1).to(DEV)
    a1, b1 = (G1 @ u).view(-1, 2).T
    a2, b2 = (G2 @ u).view(-1, 2).T
    amb1 = a1[:, None] * Xb + b1[:, None] * Yb
    amb2 = a2[:, None] * xr.to(DEV) + b2[:, None] * yr.to(DEV)
    assert torch.allclose(amb1, amb2, atol=1e-3)
    assert torch.allclose(D1 @ (G1 @ u), D2 @ (G2 @ u), atol=1e-3)
