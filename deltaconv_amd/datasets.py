"""Data side of the ModelNet / ShapeNet experiments without torch_geometric:
an OFF mesh reader, a ``Data`` attribute bag the transforms operate on, ``Compose``, a collate into
``deltaconv_amd.Batch`` and in-memory ``ModelNet`` / ``ScanObjectNN`` / ``ShapeNet`` / ``ShapeSeg`` datasets with the constructors and on-disk
layouts of the reference's ``experiments/datasets/modelnet.py:11-114`` (``root/raw/<category>/<train|test>/*.off``
-> ``root/processed/{training,test}.pt`` after ``pre_transform``) and ``shapenet.py:13-200``.

Host-side, one-off work (the hot path starts at the collated batch).  Nothing is downloaded: there is
no network in this environment, the raw folder has to exist.
"""
import copy
import glob
import os
import os.path as osp

import torch

from .data import Batch


class Data:
    """Attribute bag (``pos``, ``face`` [3,F], ``norm``, ``x``, ``y``, ``category`` ...): what the transforms
    in ``deltaconv_amd.transforms`` read and write."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    def clone(self):
        out = Data()
        for k, v in self.__dict__.items():
            setattr(out, k, v.clone() if torch.is_tensor(v) else copy.deepcopy(v))
        return out

    def __repr__(self):
        f = lambda v: list(v.shape) if torch.is_tensor(v) else v
        return "Data(" + ", ".join(f"{k}={f(getattr(self, k))}" for k in self.keys()) + ")"


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data

    def __repr__(self):
        return "Compose([" + ", ".join(repr(t) for t in self.transforms) + "])"


def parse_off(text):
    """OFF text -> Data(pos [V,3] float32, face [3,F] int64).  Accepts the ModelNet files whose counts are
    glued to the magic word (``OFF490 518 0``); polygons with more than three corners are fanned."""
    tok = text.split()
    if not tok or not tok[0].startswith("OFF"):
        raise ValueError("not an OFF file")
    head = tok[0][3:]
    tok = ([head] if head else []) + tok[1:]
    nv, nf = int(tok[0]), int(tok[1])
    p = 3                                              # skip the edge count
    pos = torch.tensor([float(t) for t in tok[p:p + 3 * nv]], dtype=torch.float32).view(nv, 3)
    p += 3 * nv
    tri = []
    for _ in range(nf):
        c = int(tok[p])
        idx = [int(t) for t in tok[p + 1:p + 1 + c]]
        p += 1 + c
        for j in range(1, c - 1):
            tri.append((idx[0], idx[j], idx[j + 1]))
    face = torch.tensor(tri, dtype=torch.long).t().contiguous() if tri else torch.empty(3, 0, dtype=torch.long)
    if face.numel() and (int(face.min()) < 0 or int(face.max()) >= nv):
        raise ValueError("OFF face index out of range")
    return Data(pos=pos, face=face)


def read_off(path):
    with open(path, "r") as fh:
        return parse_off(fh.read())


def collate(data_list):
    """List of per-shape ``Data`` -> one ``Batch`` (pos / norm / x concatenated, ``batch`` vector, labels
    stacked per cloud or concatenated per point, ``category`` stacked)."""
    cat = lambda name: (torch.cat([getattr(d, name) for d in data_list])
                        if all(getattr(d, name, None) is not None for d in data_list) else None)
    pos = cat("pos")
    batch = torch.cat([torch.full((d.pos.shape[0],), i, dtype=torch.long) for i, d in enumerate(data_list)])
    norm = cat("norm")
    if norm is None:
        norm = cat("normal")
    ys = [getattr(d, "y", None) for d in data_list]
    y = None
    if all(v is not None for v in ys):
        ys = [v if torch.is_tensor(v) else torch.tensor([v]) for v in ys]
        y = torch.cat([v.reshape(-1) for v in ys])
    cats = [getattr(d, "category", None) for d in data_list]
    category = torch.stack([c.reshape(-1) for c in cats]) if all(c is not None for c in cats) else None
    return Batch(pos, batch, norm, cat("x"), y, category, len(data_list))


class DataLoader(torch.utils.data.DataLoader):
    """``torch.utils.data.DataLoader`` that collates ``Data`` objects into ``deltaconv_amd.Batch``."""

    def __init__(self, dataset, batch_size=1, shuffle=False, **kw):
        kw.setdefault("collate_fn", collate)
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, **kw)


class ModelNet(torch.utils.data.Dataset):
    """ModelNet10/40 from OFF files (experiments/datasets/modelnet.py:11-114): same arguments, same folder
    layout, labels = index of the category in sorted order; ``pre_transform`` runs once and its result is
    cached under ``root/processed``, ``transform`` runs on a copy at every access."""

    def __init__(self, root, n_per_class=None, name='10', train=True, transform=None, pre_transform=None,
                 pre_filter=None):
        assert name in ['10', '40']
        self.root, self.name, self.n_per_class = root, name, n_per_class
        self.transform, self.pre_transform, self.pre_filter = transform, pre_transform, pre_filter
        self.raw_dir, self.processed_dir = osp.join(root, "raw"), osp.join(root, "processed")
        paths = [osp.join(self.processed_dir, f) for f in ("training.pt", "test.pt")]
        if not all(osp.exists(p) for p in paths):
            if not osp.isdir(self.raw_dir):
                raise FileNotFoundError(f"{self.raw_dir} not found: unpack ModelNet{name}.zip there "
                                        "(<category>/<train|test>/*.off); nothing is downloaded")
            os.makedirs(self.processed_dir, exist_ok=True)
            torch.save(self.process_set("train"), paths[0])
            torch.save(self.process_set("test"), paths[1])
        blob = torch.load(paths[0] if train else paths[1], weights_only=False)
        self.categories, self.items = blob["categories"], [Data(**d) for d in blob["items"]]

    def process_set(self, split):
        categories = sorted(d for d in os.listdir(self.raw_dir) if osp.isdir(osp.join(self.raw_dir, d)))
        items = []
        for target, category in enumerate(categories):
            paths = sorted(glob.glob(osp.join(self.raw_dir, category, split, f"{category}_*.off")))
            for i, path in enumerate(paths):
                if self.n_per_class is not None and i > self.n_per_class:     # modelnet.py:99 keeps n+1 shapes
                    continue
                data = read_off(path)
                data.y = torch.tensor([target])
                items.append(data)
        if self.pre_filter is not None:
            items = [d for d in items if self.pre_filter(d)]
        if self.pre_transform is not None:
            items = [self.pre_transform(d) for d in items]
        return {"categories": categories, "items": [dict(d.__dict__) for d in items]}

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        data = self.items[i].clone()
        return data if self.transform is None else self.transform(data)

    def __repr__(self):
        return '{}{}({})'.format(self.__class__.__name__, self.name, len(self))


class ScanObjectNN(torch.utils.data.Dataset):
    """The pre-processed ScanObjectNN benchmark (experiments/datasets/scanobjectnn.py:12-110): same constructor, same
    folder layout (``root/raw/main_split[_nobg]/<training|test>_objectdataset*.h5`` -> ``root/processed/<bg|nobg>_
    <variant>/{training,test}.pt``), 'data' [N,2048,3] -> ``pos``, 'label' -> ``y``.  The HDF5 files are read by
    ``deltaconv_amd.io_hdf5`` (pure numpy: h5py is not needed); an ``.npz`` with the same two arrays next to (or instead
    of) an ``.h5`` file is accepted as well."""

    url = "https://hkust-vgd.github.io/scanobjectnn/"
    class_names = ['bag', 'bed', 'bin', 'box', 'cabinets', 'chair', 'desk', 'display', 'door', 'pillow', 'shelves',
                   'sink', 'sofa', 'table', 'toilet']
    augmentation_variants = [None, 'PB_T25', 'PB_T25_R', 'PB_T50_R', 'PB_T50_RS']
    raw_file_dict = {
        None: ['training_objectdataset.h5', 'test_objectdataset.h5'],
        'PB_T25': ['training_objectdataset_augmented25_norot.h5', 'test_objectdataset_augmented25_norot.h5'],
        'PB_T25_R': ['training_objectdataset_augmented25rot.h5', 'test_objectdataset_augmented25rot.h5'],
        'PB_T50_R': ['training_objectdataset_augmentedrot.h5', 'test_objectdataset_augmentedrot.h5'],
        'PB_T50_RS': ['training_objectdataset_augmentedrot_scale75.h5', 'test_objectdataset_augmentedrot_scale75.h5'],
    }

    def __init__(self, root, background=False, augmentation=None, train=True, transform=None, pre_transform=None,
                 pre_filter=None):
        assert augmentation in self.augmentation_variants
        self.root, self.background, self.augmentation = root, background, augmentation
        self.transform, self.pre_transform, self.pre_filter = transform, pre_transform, pre_filter
        self.bg_path = 'main_split' if background else 'main_split_nobg'
        self.raw_dir, self.processed_dir = osp.join(root, "raw"), osp.join(root, "processed")
        folder = ('bg' if background else 'nobg') + '_' + (augmentation if augmentation is not None else 'vanilla')
        paths = [osp.join(self.processed_dir, folder, f) for f in ("training.pt", "test.pt")]
        if not all(osp.exists(p) for p in paths):
            raws = [osp.join(self.raw_dir, self.bg_path, f) for f in self.raw_file_dict[augmentation]]
            if not any(osp.exists(r) or osp.exists(r[:-3] + ".npz") for r in raws[:1]):
                raise RuntimeError('Dataset not found, please download the dataset from {} and place the files in {}.'
                                   .format(self.url, self.raw_dir))
            os.makedirs(osp.dirname(paths[0]), exist_ok=True)
            for raw, out in zip(raws, paths):
                torch.save({"items": [dict(d.__dict__) for d in self._process(raw)]}, out)
        blob = torch.load(paths[0] if train else paths[1], weights_only=False)
        self.items = [Data(**d) for d in blob["items"]]

    @staticmethod
    def _arrays(raw):
        import numpy as np
        if osp.exists(raw):
            from .io_hdf5 import File
            f = File(raw)
            return f["data"][...], f["label"][...]
        z = np.load(raw[:-3] + ".npz")
        return z["data"], z["label"]

    def _process(self, raw):
        pos, label = self._arrays(raw)
        items = [Data(pos=torch.from_numpy(pos[i].astype("float32")), y=torch.tensor([int(label[i])]))
                 for i in range(pos.shape[0])]
        if self.pre_filter is not None:
            items = [d for d in items if self.pre_filter(d)]
        if self.pre_transform is not None:
            items = [self.pre_transform(d) for d in items]
        return items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        data = self.items[i].clone()
        return data if self.transform is None else self.transform(data)

    def __repr__(self):
        return '{}({})'.format(self.__class__.__name__, len(self))


def read_txt_array(path):
    """Whitespace-separated numeric table -> float32 tensor [rows, cols]."""
    with open(path, "r") as fh:
        rows = [[float(t) for t in line.split()] for line in fh if line.strip()]
    return torch.tensor(rows, dtype=torch.float32)


class ShapeNet(torch.utils.data.Dataset):
    """ShapeNet part-segmentation benchmark (experiments/datasets/shapenet.py:13-200; used by
    train_shapenet.py:43-44): ``root/raw/<synset>/*.txt`` rows ``x y z nx ny nz part`` and
    ``root/raw/train_test_split/shuffled_{train,val,test}_file_list.json``.  Items carry ``pos``, ``norm``,
    per-point ``y`` (0..49) and the one-hot ``category`` [1,16] indexed within the SELECTED categories, as the
    reference builds it; ``y_mask[c]`` marks the part labels of category c.  Processed splits are cached as
    ``root/processed/<cats>_{train,val,test,trainval}.pt``."""

    _names = ['Airplane', 'Bag', 'Cap', 'Car', 'Chair', 'Earphone', 'Guitar', 'Knife', 'Lamp', 'Laptop', 'Motorbike',
              'Mug', 'Pistol', 'Rocket', 'Skateboard', 'Table']
    _synsets = ['02691156', '02773838', '02954340', '02958343', '03001627', '03261776', '03467517', '03624134',
                '03636649', '03642806', '03790512', '03797390', '03948459', '04099429', '04225987', '04379243']
    _parts = [4, 2, 2, 4, 4, 3, 3, 2, 4, 2, 6, 2, 3, 3, 3, 3]          # part labels per category, consecutive from 0
    category_ids = dict(zip(_names, _synsets))
    seg_classes = None            # filled in below the class body (a class-level comprehension cannot see _parts)
    splits = ['train', 'val', 'test', 'trainval']

    def __init__(self, root, categories=None, n_per_class=None, include_normals=True, split='trainval', transform=None,
                 pre_transform=None, pre_filter=None):
        if categories is None:
            categories = list(self.category_ids.keys())
        if isinstance(categories, str):
            categories = [categories]
        assert all(c in self.category_ids for c in categories)
        if split not in self.splits:
            raise ValueError(f'Split {split} found, but expected either train, val, trainval or test')
        self.root, self.categories, self.n_per_class = root, categories, n_per_class
        self.transform, self.pre_transform, self.pre_filter = transform, pre_transform, pre_filter
        self.include_normals = include_normals
        self.raw_dir, self.processed_dir = osp.join(root, "raw"), osp.join(root, "processed")
        tag = '_'.join(c[:3].lower() for c in categories)
        paths = {s: osp.join(self.processed_dir, f"{tag}_{s}.pt") for s in self.splits}
        if not all(osp.exists(p) for p in paths.values()):
            if not osp.isdir(osp.join(self.raw_dir, "train_test_split")):
                raise FileNotFoundError(f"{self.raw_dir}/train_test_split not found: unpack "
                                        "shapenetcore_partanno_segmentation_benchmark_v0_normal there; nothing is downloaded")
            os.makedirs(self.processed_dir, exist_ok=True)
            self._process(paths)
        self.items = [Data(**d) for d in torch.load(paths[split], weights_only=False)]
        if not include_normals:
            for d in self.items:
                d.norm = None
        self.y_mask = torch.zeros((len(self.seg_classes), 50), dtype=torch.bool)
        for i, labels in enumerate(self.seg_classes.values()):
            self.y_mask[i, labels] = 1

    @property
    def num_classes(self):
        return self.y_mask.size(-1)

    def _process_filenames(self, filenames):
        ids = [self.category_ids[c] for c in self.categories]
        cat_idx = {s: i for i, s in enumerate(ids)}
        left = [self.n_per_class] * len(ids) if self.n_per_class is not None else None
        out = []
        for name in filenames:
            syn = name.split(osp.sep)[0]
            if syn not in cat_idx:
                continue
            if left is not None:
                if left[cat_idx[syn]] <= 0:
                    continue
                left[cat_idx[syn]] -= 1
            tab = read_txt_array(osp.join(self.raw_dir, name))
            onehot = torch.zeros(1, 16)
            onehot[0, cat_idx[syn]] = 1
            data = Data(pos=tab[:, :3].contiguous(), norm=tab[:, 3:6].contiguous(), y=tab[:, -1].long(), category=onehot)
            if self.pre_filter is not None and not self.pre_filter(data):
                continue
            out.append(data if self.pre_transform is None else self.pre_transform(data))
        return out

    def _process(self, paths):
        import json
        trainval = []
        for split in ('train', 'val', 'test'):
            with open(osp.join(self.raw_dir, 'train_test_split', f'shuffled_{split}_file_list.json')) as fh:
                names = [osp.sep.join(n.split('/')[1:]) + '.txt' for n in json.load(fh)]   # drop the leading folder
            items = self._process_filenames(names)
            if split != 'test':
                trainval += items
            torch.save([dict(d.__dict__) for d in items], paths[split])
        torch.save([dict(d.__dict__) for d in trainval], paths['trainval'])

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        data = self.items[i].clone()
        return data if self.transform is None else self.transform(data)

    def __repr__(self):
        return '{}({}, categories={})'.format(self.__class__.__name__, len(self), self.categories)


ShapeNet.seg_classes = {n: list(range(sum(ShapeNet._parts[:i]), sum(ShapeNet._parts[:i + 1])))
                        for i, n in enumerate(ShapeNet._names)}


# ---- ShapeSeg (human-body part segmentation, meshes from Adobe / FAUST / MIT / SCAPE, SHREC for testing) ----
def read_ply(path):
    """PLY (ascii or binary_little_endian) -> Data(pos [V,3] float32, face [3,F] int64).  Reads the x/y/z
    properties of the vertex element and the index list of the face element; polygons are fanned."""
    import struct
    with open(path, "rb") as fh:
        raw = fh.read()
    end = raw.index(b"end_header")
    end = raw.index(b"\n", end) + 1
    header = raw[:end].decode("ascii", "replace").splitlines()
    if not header or header[0].strip() != "ply":
        raise ValueError("not a PLY file")
    fmt, elements, cur = None, [], None
    for line in header[1:]:
        t = line.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            cur = {"name": t[1], "count": int(t[2]), "props": []}
            elements.append(cur)
        elif t[0] == "property" and cur is not None:
            cur["props"].append(t[1:])          # [type, name] or ['list', count_type, item_type, name]
    if fmt not in ("ascii", "binary_little_endian"):
        raise ValueError(f"unsupported PLY format {fmt}")
    codes = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d",
             "int8": "b", "uint8": "B", "int16": "h", "uint16": "H", "int32": "i", "uint32": "I", "float32": "f",
             "float64": "d"}
    pos, tri = None, []
    if fmt == "ascii":
        tok = raw[end:].split()
        p = 0
        for el in elements:
            rows = []
            for _ in range(el["count"]):
                row = {}
                for pr in el["props"]:
                    if pr[0] == "list":
                        c = int(tok[p]); p += 1
                        row[pr[3]] = [int(float(v)) for v in tok[p:p + c]]; p += c
                    else:
                        row[pr[1]] = float(tok[p]); p += 1
                rows.append(row)
            if el["name"] == "vertex":
                pos = torch.tensor([[r["x"], r["y"], r["z"]] for r in rows], dtype=torch.float32)
            elif el["name"] == "face":
                for r in rows:
                    idx = next(v for v in r.values() if isinstance(v, list))
                    tri += [(idx[0], idx[j], idx[j + 1]) for j in range(1, len(idx) - 1)]
    else:
        p = end
        for el in elements:
            rows = []
            for _ in range(el["count"]):
                row = {}
                for pr in el["props"]:
                    if pr[0] == "list":
                        ct, it = "<" + codes[pr[1]], codes[pr[2]]
                        (c,) = struct.unpack_from(ct, raw, p); p += struct.calcsize(ct)
                        row[pr[3]] = list(struct.unpack_from(f"<{c}{it}", raw, p)); p += struct.calcsize(f"<{c}{it}")
                    else:
                        f = "<" + codes[pr[0]]
                        (row[pr[1]],) = struct.unpack_from(f, raw, p); p += struct.calcsize(f)
                rows.append(row)
            if el["name"] == "vertex":
                pos = torch.tensor([[r["x"], r["y"], r["z"]] for r in rows], dtype=torch.float32)
            elif el["name"] == "face":
                for r in rows:
                    idx = next(v for v in r.values() if isinstance(v, list))
                    tri += [(idx[0], idx[j], idx[j + 1]) for j in range(1, len(idx) - 1)]
    if pos is None:
        raise ValueError("PLY without a vertex element")
    face = torch.tensor(tri, dtype=torch.long).t().contiguous() if tri else torch.empty(3, 0, dtype=torch.long)
    return Data(pos=pos, face=face)


def read_obj(path):
    """Wavefront OBJ -> Data(pos, face [3,F]): ``v x y z`` and ``f a b c`` records (``a/b/c`` corner syntax and
    negative indices accepted, polygons fanned).  Stands for the openmesh reader of shape_seg.py:193-199."""
    pos, tri = [], []
    with open(path, "r") as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                pos.append([float(v) for v in t[1:4]])
            elif t[0] == "f":
                idx = [int(c.split("/")[0]) for c in t[1:]]
                idx = [i - 1 if i > 0 else len(pos) + i for i in idx]
                tri += [(idx[0], idx[j], idx[j + 1]) for j in range(1, len(idx) - 1)]
    face = torch.tensor(tri, dtype=torch.long).t().contiguous() if tri else torch.empty(3, 0, dtype=torch.long)
    return Data(pos=torch.tensor(pos, dtype=torch.float32).view(-1, 3), face=face)


def edge_to_vertex_labels(face, labels, n_nodes):
    """Per-edge labels (MeshCNN ``.eseg``, 1-based; edges numbered in order of first appearance while walking the
    faces, corners (0,1), (1,2), (0,2)) -> per-vertex labels, 0-based (shape_seg.py:173-191): every edge writes
    its label to both end points, first all first end points, then all second end points, later edges
    overwriting earlier ones."""
    seen, order = set(), []
    for f in face.t().tolist():
        for a, b in ((f[0], f[1]), (f[1], f[2]), (f[0], f[2])):
            e = (a, b) if a < b else (b, a)
            if e not in seen:
                seen.add(e)
                order.append(e)
    res = [0] * n_nodes
    lab = labels.tolist()
    for end in (0, 1):
        for e, l in zip(order, lab):
            res[e[end]] = int(l)
    return torch.tensor(res, dtype=torch.long) - 1


class ShapeSeg(torch.utils.data.Dataset):
    """Human-body segmentation set of Maron et al. in the MeshCNN remeshing (experiments/datasets/shape_seg.py:
    13-171; used by train_shapeseg.py:44,53): training = Adobe (41 ``.ply`` + ``segs/<i>.pt``), FAUST (100
    ``tr_reg_%03d.ply``, one shared ``faust_seg.pt``), MIT (``.obj`` + per-edge ``.eseg``), SCAPE (71 ``.ply``,
    shared ``scape_seg.pt``); test = SHREC (18 ``.ply`` + ``segs/<i>.pt``).  Expects the archive unpacked as
    ``root/raw/ShapeSeg/<SET>/raw/{meshes,segs}`` (nested ``<set>.zip`` files are extracted when still
    present).  Items: ``pos``, ``face``, per-vertex ``y``; cached as ``root/processed/{training,test}.pt``."""

    def __init__(self, root, train=True, transform=None, pre_transform=None, pre_filter=None):
        self.root, self.transform, self.pre_transform, self.pre_filter = root, transform, pre_transform, pre_filter
        self.raw_dir, self.processed_dir = osp.join(root, "raw"), osp.join(root, "processed")
        paths = [osp.join(self.processed_dir, f) for f in ("training.pt", "test.pt")]
        if not all(osp.exists(p) for p in paths):
            base = osp.join(self.raw_dir, "ShapeSeg")
            if not osp.isdir(base):
                big = osp.join(self.raw_dir, "shapeseg.zip")
                if not osp.exists(big):
                    raise FileNotFoundError(f"{base} (or {big}) not found; nothing is downloaded")
                import zipfile
                with zipfile.ZipFile(big) as z:
                    z.extractall(self.raw_dir)
            os.makedirs(self.processed_dir, exist_ok=True)
            train_items, test_items = self._process(base)
            torch.save([dict(d.__dict__) for d in train_items], paths[0])
            torch.save([dict(d.__dict__) for d in test_items], paths[1])
        self.items = [Data(**d) for d in torch.load(paths[0] if train else paths[1], weights_only=False)]

    def _finish(self, data, out):
        if self.pre_filter is not None and not self.pre_filter(data):
            return
        out.append(data if self.pre_transform is None else self.pre_transform(data))

    @staticmethod
    def _set_dir(base, name):
        d = osp.join(base, name, "raw")
        z = osp.join(d, name.lower() + ".zip")
        if osp.exists(z) and not osp.isdir(osp.join(d, "meshes")):
            import zipfile
            with zipfile.ZipFile(z) as zf:
                zf.extractall(d)
        return d

    def _numbered(self, base, name, pattern, shared_seg, out):
        d = self._set_dir(base, name)
        shared = torch.load(osp.join(d, "segs", shared_seg), weights_only=False) if shared_seg else None
        i = 0
        while osp.exists(osp.join(d, "meshes", pattern.format(i))):
            data = read_ply(osp.join(d, "meshes", pattern.format(i)))
            data.y = shared if shared is not None else torch.load(osp.join(d, "segs", f"{i}.pt"), weights_only=False)
            self._finish(data, out)
            i += 1

    def _process(self, base):
        train, test = [], []
        self._numbered(base, "Adobe", "{}.ply", None, train)
        self._numbered(base, "FAUST", "tr_reg_{0:03d}.ply", "faust_seg.pt", train)
        d = self._set_dir(base, "MIT")
        for fn in os.listdir(osp.join(d, "meshes")):          # directory order, as the reference (shape_seg.py:125)
            data = read_obj(osp.join(d, "meshes", fn))
            with open(osp.join(d, "segs", fn.replace(".obj", ".eseg"))) as fh:
                segs = torch.tensor([int(float(t)) for t in fh.read().split()], dtype=torch.long)
            data.y = edge_to_vertex_labels(data.face, segs, data.pos.shape[0])
            self._finish(data, train)
        self._numbered(base, "SCAPE", "{}.ply", "scape_seg.pt", train)
        self._numbered(base, "SHREC", "{}.ply", None, test)
        return train, test

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        data = self.items[i].clone()
        return data if self.transform is None else self.transform(data)

    def __repr__(self):
        return '{}({})'.format(self.__class__.__name__, len(self))
