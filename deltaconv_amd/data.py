"""Minimal batch container + seeded synthetic clouds (no torch_geometric at run time).

``Batch`` carries exactly the attributes the reference's models read from a PyG batch
(models/deltanet_base.py:43-44,59,76; deltanet_segmentation.py:64): ``pos, batch, norm, x, y,
category, num_graphs`` and ``.to(device)``.

``synthetic_batch`` implements the workload generator of SURVEY.md section 8(d): points on a
smooth closed surface r = 1 + 0.25 sin(3 theta) cos(2 phi) with analytic normals, then the
reference's NormalizeScale semantics (transforms/normalize_scale.py:13-19).
"""
import torch


class Batch:
    def __init__(self, pos, batch=None, norm=None, x=None, y=None, category=None, num_graphs=None):
        self.pos, self.norm, self.x, self.y, self.category = pos, norm, x, y, category
        if batch is None:
            batch = torch.zeros(pos.shape[0], dtype=torch.long, device=pos.device)
        self.batch = batch
        self.num_graphs = int(num_graphs) if num_graphs is not None else int(batch.max()) + 1
        self._ptr = None

    @property
    def ptr(self):
        """Cloud offsets [B+1] (int32, on the device of pos); clouds are contiguous and sorted."""
        if self._ptr is None or self._ptr.device != self.pos.device:
            counts = torch.bincount(self.batch, minlength=self.num_graphs)
            ptr = torch.zeros(self.num_graphs + 1, dtype=torch.int32, device=self.pos.device)
            ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
            self._ptr = ptr
        return self._ptr

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        out = Batch(mv(self.pos), mv(self.batch), mv(self.norm), mv(self.x), mv(self.y), mv(self.category),
                    self.num_graphs)
        if self._ptr is not None:
            out._ptr = self._ptr.to(device)
        return out

    def shard(self, rank, world):
        """Clouds [rank*B/world, (rank+1)*B/world) of this batch (data-parallel input sharding)."""
        b = self.num_graphs
        assert b % world == 0, "global batch must divide by world size"
        per = b // world
        ptr = self.ptr.tolist()
        lo, hi = ptr[rank * per], ptr[(rank + 1) * per]
        cut = lambda t: None if t is None else t[lo:hi]
        y = self.y
        if y is not None:
            y = y[rank * per:(rank + 1) * per] if y.shape[0] == b else y[lo:hi]
        cat = None if self.category is None else self.category[rank * per:(rank + 1) * per]
        return Batch(self.pos[lo:hi], self.batch[lo:hi] - rank * per, cut(self.norm), cut(self.x), y, cat, per)


def _surface(n, gen, m=3, l=2, a=0.25):
    """n points on the closed surface r = 1 + a sin(m theta) cos(l phi) with its analytic normals; (m, l, a) = the shape family."""
    d = torch.randn(n, 3, generator=gen, dtype=torch.float64)
    d = d / d.norm(dim=1, keepdim=True)
    theta = torch.acos(d[:, 2].clamp(-1, 1))
    phi = torch.atan2(d[:, 1], d[:, 0])
    r = 1 + a * torch.sin(m * theta) * torch.cos(l * phi)
    p = d * r[:, None]
    # normal of the implicit surface F(p) = |p| - r(theta(p), phi(p)) via autograd
    q = p.clone().requires_grad_(True)
    rad = q.norm(dim=1)
    th = torch.acos((q[:, 2] / rad).clamp(-1 + 1e-12, 1 - 1e-12))
    ph = torch.atan2(q[:, 1], q[:, 0])
    F = rad - (1 + a * torch.sin(m * th) * torch.cos(l * ph))
    (g,) = torch.autograd.grad(F.sum(), q)
    nrm = g / g.norm(dim=1, keepdim=True).clamp(1e-12)
    return p, nrm


def shape_family(cls):
    """(m, l, a) of class `cls` of the learnable synthetic task (examples/train_modelnet_like.py): lobes in theta x lobes in phi."""
    return 1 + cls % 5, 1 + (cls // 5) % 3, 0.25 + 0.05 * ((cls // 15) % 2)


def synthetic_cloud(n, seed, normals=True, dup_frac=0.0, outlier_frac=0.0, jitter=0.0, family=None):
    gen = torch.Generator().manual_seed(int(seed))
    if family is None:
        p, nrm = _surface(n, gen)
    else:       # a member of a shape family, randomly rotated (the label must not be readable off the axes)
        p, nrm = _surface(n, gen, *family)
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=gen, dtype=torch.float64))
        q = q * torch.sign(torch.linalg.det(q))
        p, nrm = p @ q.t(), nrm @ q.t()
    if outlier_frac > 0:
        m = int(n * outlier_frac)
        p[:m] = torch.rand(m, 3, generator=gen, dtype=torch.float64) * 2.5 - 1.25
    if jitter > 0:
        p = p + jitter * torch.randn(n, 3, generator=gen, dtype=torch.float64)
    if dup_frac > 0:   # exact duplicates, as produced by tiling in transforms/geodesic_fps.py:20-23
        m = int(n * dup_frac)
        src = torch.randint(0, n - m, (m,), generator=gen)
        p[n - m:] = p[src]
        nrm[n - m:] = nrm[src]
    lo, hi = p.min(0).values, p.max(0).values          # NormalizeScale: centre by bbox mid ...
    p = p - (lo + hi) / 2
    p = p * (0.999999 / p.norm(dim=1).max())           # ... and scale to max-norm 0.999999
    return p.float(), (nrm.float() if normals else None)


def synthetic_batch(num_clouds, n, seed=0, normals=True, num_classes=40, per_point_labels=False,
                    categories=0, sizes=None, learnable=False, **kw):
    """B clouds x N points (or ragged ``sizes``), labels ``randint``; seed = 1000*seed + cloud.
    learnable: the label of a cloud IS its shape family (``shape_family``), the cloud a randomly rotated member of it -- a task
    a classifier can learn (the default, one surface with random labels, only measures throughput)."""
    sizes = [n] * num_clouds if sizes is None else list(sizes)
    gen = torch.Generator().manual_seed(1000 * seed + 999)
    ycls = torch.randint(0, num_classes, (len(sizes),), generator=gen) if learnable else None
    ps, ns, bs = [], [], []
    for c, m in enumerate(sizes):
        fam = shape_family(int(ycls[c])) if learnable else None
        p, nr = synthetic_cloud(m, 1000 * seed + c, normals, family=fam, **kw)
        ps.append(p); ns.append(nr); bs.append(torch.full((m,), c, dtype=torch.long))
    pos = torch.cat(ps)
    ny = pos.shape[0] if per_point_labels else len(sizes)
    y = ycls if (learnable and not per_point_labels) else torch.randint(0, num_classes, (ny,), generator=gen)
    cat = None
    if categories:
        cat = torch.zeros(len(sizes), categories)
        cat[torch.arange(len(sizes)), torch.randint(0, categories, (len(sizes),), generator=gen)] = 1
    return Batch(pos, torch.cat(bs), torch.cat(ns) if normals else None, None, y, cat, len(sizes))
