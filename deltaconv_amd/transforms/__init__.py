"""Host-side data transforms with the reference's names, arguments and random-number consumption
(reference: deltaconv/transforms/*.py), so ``import deltaconv_amd.transforms as T`` resolves for
experiments/train_*.py:12 and seeded pipelines produce the same samples.  Plain callables on any
object with ``pos`` (+ optional ``norm``/``normal``, ``x``, ``y``, ``face``) attributes -- no
torch_geometric needed.  These run on the CPU once per sample (dataset side), not on the GPU hot path.
"""
import math
import numbers
import random
from math import ceil

import torch

from ..geometry.fps import geodesic_fps

__all__ = ["NormalizeScale", "NormalizeArea", "NormalizeAxes", "RandomScale", "RandomTranslateGlobal",
           "RandomRotate", "RandomNormals", "SamplePoints", "GeodesicFPS"]


class _Transform:
    def _args(self):
        return ""

    def __repr__(self):
        return f"{self.__class__.__name__}({self._args()})"


def _center_bbox(pos):
    return pos - (pos.max(dim=0).values + pos.min(dim=0).values) / 2


def _per_dim(t, dim):
    t = [t] * dim if isinstance(t, numbers.Number) else list(t)
    assert len(t) == dim
    return t


class NormalizeScale(_Transform):
    """normalize_scale.py:5-24: centre on the bounding-box middle, scale so the largest ``norm_ord``
    norm (or ``scaling_factor``) becomes 0.999999."""

    def __init__(self, norm_ord=2, scaling_factor=None):
        self.norm_ord, self.scaling_factor = norm_ord, scaling_factor

    def __call__(self, data):
        pos = _center_bbox(data.pos)
        ref = torch.linalg.norm(pos, ord=self.norm_ord, dim=1).max() if self.scaling_factor is None \
            else self.scaling_factor
        data.pos = pos * ((1 / ref) * 0.999999)
        return data


class NormalizeArea(_Transform):
    """normalize_area.py:5-24: centre, then scale a triangle mesh to unit surface area
    (``data.face`` is [F,3] here, as the reference indexes it)."""

    def __call__(self, data):
        pos = _center_bbox(data.pos)
        f = data.face
        cr = torch.linalg.cross(pos[f[:, 1]] - pos[f[:, 0]], pos[f[:, 2]] - pos[f[:, 0]], dim=-1)
        data.pos = pos * (1 / torch.sqrt(torch.linalg.norm(cr, dim=-1).sum() / 2))
        return data


class NormalizeAxes(_Transform):
    """normalize_axes.py:4-27: order the axes by ascending standard deviation, scale so the largest
    coordinate of the last axis is 0.5."""

    def __init__(self, max_points=-1):
        self.max_points = max_points

    def __call__(self, data):
        pos = data.pos[:, torch.sort(torch.std(data.pos, dim=0)).indices]
        data.pos = pos * (1 / (2 * pos.max(0).values[2]))
        return data


def _is_batch(data):
    """A collated multi-cloud batch (has a per-point cloud index and a cloud count) rather than one shape."""
    return getattr(data, "batch", None) is not None and getattr(data, "num_graphs", None) is not None \
        and data.batch.shape[0] == data.pos.shape[0]


class RandomScale(_Transform):
    """random_scale.py:5-39: an independent factor PER AXIS drawn from ``scales``; normals follow with
    the inverse factors and are re-normalised."""

    def __init__(self, scales):
        assert isinstance(scales, (tuple, list)) and len(scales) == 2
        self.scales = scales

    def _args(self):
        return str(self.scales)

    def __call__(self, data):
        if _is_batch(data):
            # a collated batch (deltaconv_amd.Batch), typically already on the GPU: one factor triple PER CLOUD, drawn
            # and applied on the device of `pos` (two small kernels for the whole batch instead of one host-side
            # transform per shape -- SURVEY.md section 8(f)-2: augmentation must not bottleneck 8 ranks on one host)
            s = data.pos.new_empty(data.num_graphs, 3).uniform_(*self.scales)[data.batch]
        else:
            s = data.pos.new_empty(3).uniform_(*self.scales)
        data.pos = data.pos * s
        if getattr(data, 'norm', None) is not None:
            nrm = data.norm * (1 / s)
            data.norm = nrm / torch.linalg.norm(nrm, dim=1, keepdim=True)
        return data


class RandomTranslateGlobal(_Transform):
    """random_translate_global.py:6-39: one random offset per axis for the whole shape."""

    def __init__(self, translate):
        self.translate = translate

    def _args(self):
        return str(self.translate)

    def __call__(self, data):
        t = _per_dim(self.translate, data.pos.size(1))
        if _is_batch(data):                       # per-cloud offsets on the device of `pos` (see RandomScale)
            lim = data.pos.new_tensor([abs(a) for a in t])
            off = (data.pos.new_empty(data.num_graphs, len(t)).uniform_(-1.0, 1.0) * lim)[data.batch]
            data.pos = data.pos + off
            return data
        off = [data.pos.new_empty(1).uniform_(-abs(a), abs(a)) for a in t]   # one draw per axis, in order
        data.pos = data.pos + torch.stack(off, dim=-1)
        return data


class RandomRotate(_Transform):
    """random_rotate.py:7-50: rotation about one axis by an angle from ``degrees`` (python ``random``)."""

    def __init__(self, degrees, axis=0):
        if isinstance(degrees, numbers.Number):
            degrees = (-abs(degrees), abs(degrees))
        assert isinstance(degrees, (tuple, list)) and len(degrees) == 2
        self.degrees, self.axis = degrees, axis

    def _args(self):
        return f"{self.degrees}, axis={self.axis}"

    def __call__(self, data):
        a = math.pi * random.uniform(*self.degrees) / 180.0
        s, c = math.sin(a), math.cos(a)
        if data.pos.size(-1) == 2:
            m = [[c, s], [-s, c]]
        elif self.axis == 0:
            m = [[1, 0, 0], [0, c, s], [0, -s, c]]
        elif self.axis == 1:
            m = [[c, 0, -s], [0, 1, 0], [s, 0, c]]
        else:
            m = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
        m = torch.tensor(m, dtype=torch.float32)
        data.pos = data.pos @ m.to(data.pos.dtype).to(data.pos.device)
        if getattr(data, 'norm', None) is not None:
            data.norm = data.norm @ m.to(data.norm.dtype).to(data.norm.device)
        return data


class RandomNormals(_Transform):
    """random_normals.py:7-42: per-point, per-axis jitter of the normals, then re-normalisation."""

    def __init__(self, translate):
        self.translate = translate

    def _args(self):
        return str(self.translate)

    def __call__(self, data):
        n, dim = data.pos.size()
        t = _per_dim(self.translate, dim)
        jit = [data.pos.new_empty(n).uniform_(-abs(a), abs(a)) for a in t]
        nrm = data.norm + torch.stack(jit, dim=-1)
        data.norm = nrm / torch.linalg.norm(nrm, dim=-1, keepdim=True).clamp(1e-5)
        return data


class SamplePoints(_Transform):
    """sample_points.py:4-60: ``num`` points on a triangle mesh, faces chosen by area
    (``data.face`` is [3,F] here), uniform barycentric coordinates; optional normals / labels."""

    def __init__(self, num, remove_faces=True, include_normals=False, include_labels=False):
        self.num, self.remove_faces = num, remove_faces
        self.include_normals, self.include_labels = include_normals, include_labels

    def _args(self):
        return str(self.num)

    def __call__(self, data):
        pos, face = data.pos, data.face
        assert pos.size(1) == 3 and face.size(0) == 3
        top = pos.max()
        pos = pos / top
        area = torch.linalg.cross(pos[face[1]] - pos[face[0]], pos[face[2]] - pos[face[0]], dim=1).norm(p=2, dim=1).abs() / 2
        pick = torch.multinomial(area / area.sum(), self.num, replacement=True)
        face = face[:, pick]
        frac = torch.rand(self.num, 2, device=pos.device)
        fold = frac.sum(dim=-1) > 1
        frac[fold] = 1 - frac[fold]
        e1, e2 = pos[face[1]] - pos[face[0]], pos[face[2]] - pos[face[0]]
        if self.include_normals:
            data.norm = torch.nn.functional.normalize(torch.linalg.cross(e1, e2, dim=1), p=2)
        data.pos = (pos[face[0]] + frac[:, :1] * e1 + frac[:, 1:] * e2) * top
        if self.include_labels:
            data.y = data.y[face[0]]
        if self.remove_faces:
            data.face = None
        return data


class GeodesicFPS(_Transform):
    """geodesic_fps.py:5-46: keep ``n_samples`` geodesic-farthest points (tiled if the cloud is smaller)."""

    def __init__(self, n_samples=None, store_original=False):
        self.n_samples, self.store_original = n_samples, store_original

    def __call__(self, data):
        if self.n_samples is None:
            self.n_samples = data.pos.size(0)
        n = data.pos.size(0)
        idx = torch.from_numpy(geodesic_fps(data.pos.cpu().numpy(), self.n_samples)).long().reshape(-1)
        if n < self.n_samples:
            idx = idx[:n].repeat(ceil(self.n_samples / n))
        idx = idx[:self.n_samples]
        assert 0 <= int(idx.min()) and int(idx.max()) <= n
        data.sample_idx = idx
        if self.store_original:
            data.pos_original, data.y_original = data.pos, data.y
        data.pos = data.pos[idx]
        if getattr(data, 'norm', None) is not None:
            data.norm = data.norm[idx]
        if getattr(data, 'normal', None) is not None:
            data.norm = data.normal[idx]
        if getattr(data, 'x', None) is not None:
            data.x = data.x[idx]
        y = getattr(data, 'y', None)
        if y is not None and type(y) is not int and y.size(0) > 1:
            data.y = y[idx]
        return data
