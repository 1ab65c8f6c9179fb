"""Data side without torch_geometric (deltaconv_amd/datasets.py): OFF reader, collate, ModelNet layout and
caching (reference: experiments/datasets/modelnet.py:11-114, experiments/train_modelnet.py:29-49)."""
import os

import pytest
import torch

import deltaconv_amd.transforms as T
from deltaconv_amd.datasets import Compose, Data, DataLoader, ModelNet, collate, parse_off, read_off

CUBE = """OFF
8 6 0
0 0 0
1 0 0
1 1 0
0 1 0
0 0 1
1 0 1
1 1 1
0 1 1
4 0 1 2 3
4 4 5 6 7
4 0 1 5 4
4 2 3 7 6
4 1 2 6 5
4 0 3 7 4
"""
TETRA_GLUED = "OFF4 4 0\n0 0 0\n1 0 0\n0 1 0\n0 0 1\n3 0 1 2\n3 0 1 3\n3 0 2 3\n3 1 2 3\n"


def test_parse_off_quads_and_glued_header():
    cube = parse_off(CUBE)
    assert cube.pos.shape == (8, 3) and cube.face.shape == (3, 12) and cube.face.dtype == torch.long   # 6 quads -> 12 triangles
    area = torch.linalg.cross(cube.pos[cube.face[1]] - cube.pos[cube.face[0]],
                              cube.pos[cube.face[2]] - cube.pos[cube.face[0]], dim=1).norm(dim=1).sum() / 2
    assert abs(float(area) - 6.0) < 1e-6
    tet = parse_off(TETRA_GLUED)
    assert tet.pos.shape == (4, 3) and tet.face.shape == (3, 4)
    with pytest.raises(ValueError):
        parse_off("PLY\n1 2 3")
    with pytest.raises(ValueError):
        parse_off("OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n")


def _make_tree(root):
    for cat, text in (("bowl", CUBE), ("airplane", TETRA_GLUED)):
        for split, n in (("train", 3), ("test", 2)):
            d = os.path.join(root, "raw", cat, split)
            os.makedirs(d)
            for i in range(n):
                with open(os.path.join(d, f"{cat}_{i:04d}.off"), "w") as fh:
                    fh.write(text)
            with open(os.path.join(d, "README.txt"), "w") as fh:      # ignored: not <category>_*.off
                fh.write("x")


def test_modelnet_layout_pretransform_cache_and_loader(tmp_path):
    root = str(tmp_path / "ModelNet40")
    with pytest.raises(FileNotFoundError):
        ModelNet(root, None, "40", True)
    _make_tree(root)
    torch.manual_seed(0)
    pre = Compose((T.NormalizeScale(), T.SamplePoints(64, include_normals=True), T.GeodesicFPS(32)))
    train = ModelNet(root, None, "40", True, transform=Compose((T.RandomScale((4 / 5, 5 / 4)),
                                                               T.RandomTranslateGlobal(0.1))), pre_transform=pre)
    test = ModelNet(root, None, "40", False, pre_transform=pre)
    assert len(train) == 6 and len(test) == 4 and repr(train) == "ModelNet40(6)"
    assert train.categories == ["airplane", "bowl"]                    # labels follow the sorted category names
    assert [int(train.items[i].y) for i in range(6)] == [0, 0, 0, 1, 1, 1]
    d = train[4]
    assert d.pos.shape == (32, 3) and d.norm.shape == (32, 3) and d.face is None
    assert torch.allclose(d.norm.norm(dim=1), torch.ones(32), atol=1e-5)
    assert not torch.equal(train[4].pos, train[4].pos)                 # the per-access transform is random ...
    assert torch.equal(test[1].pos, test[1].pos)                       # ... the cached pre-transform is not
    # processed/ is the cache: the raw files are not needed again
    import shutil
    shutil.rmtree(os.path.join(root, "raw"))
    again = ModelNet(root, None, "40", False)
    assert len(again) == 4 and torch.equal(again[1].pos, test[1].pos)
    # loader -> Batch objects the models consume
    loader = DataLoader(train, batch_size=4, shuffle=False, drop_last=True)
    batches = list(loader)
    assert len(batches) == 1
    b = batches[0]
    assert b.pos.shape == (128, 3) and b.norm.shape == (128, 3) and b.num_graphs == 4
    assert b.y.tolist() == [0, 0, 0, 1] and b.batch.tolist() == sum([[i] * 32 for i in range(4)], [])
    assert b.ptr.tolist() == [0, 32, 64, 96, 128]


def test_n_per_class_and_collate_variants(tmp_path):
    root = str(tmp_path / "ModelNet10")
    _make_tree(root)
    ds = ModelNet(root, 0, "10", True)                                 # reference quirk: i > n is skipped -> n+1 kept
    assert len(ds) == 2
    a = Data(pos=torch.zeros(3, 3), x=torch.ones(3, 2), y=torch.tensor([0, 1, 2]), category=torch.tensor([1., 0.]))
    b = Data(pos=torch.ones(2, 3), x=torch.zeros(2, 2), y=torch.tensor([3, 4]), category=torch.tensor([0., 1.]))
    out = collate([a, b])
    assert out.norm is None and out.x.shape == (5, 2) and out.y.tolist() == [0, 1, 2, 3, 4]
    assert out.category.shape == (2, 2) and out.batch.tolist() == [0, 0, 0, 1, 1]
    assert "pos=[3, 3]" in repr(a)


def test_shapenet_part_layout(tmp_path):
    import json
    from deltaconv_amd.datasets import ShapeNet
    from deltaconv_amd.utils import calc_shape_IoU
    root = str(tmp_path / "ShapeNet")
    with pytest.raises(FileNotFoundError):
        ShapeNet(root)
    g = torch.Generator().manual_seed(3)
    files = {"train": [("02691156", "a1"), ("03001627", "c1"), ("02691156", "a2")],
             "val": [("03001627", "c2")], "test": [("02691156", "a3"), ("04379243", "t1")]}
    os.makedirs(os.path.join(root, "raw", "train_test_split"))
    for split, lst in files.items():
        with open(os.path.join(root, "raw", "train_test_split", f"shuffled_{split}_file_list.json"), "w") as fh:
            json.dump([f"shape_data/{syn}/{name}" for syn, name in lst], fh)
        for syn, name in lst:
            os.makedirs(os.path.join(root, "raw", syn), exist_ok=True)
            lo = {"02691156": 0, "03001627": 12, "04379243": 47}[syn]
            n = 20
            tab = torch.cat([torch.randn(n, 3, generator=g), torch.nn.functional.normalize(torch.randn(n, 3, generator=g)),
                             torch.randint(lo, lo + 3, (n, 1), generator=g).float()], 1)
            with open(os.path.join(root, "raw", syn, name + ".txt"), "w") as fh:
                for row in tab.tolist():
                    fh.write(" ".join(f"{v:.6f}" for v in row) + "\n")
    tv = ShapeNet(root, split="trainval")
    te = ShapeNet(root, split="test")
    assert len(tv) == 4 and len(te) == 2 and tv.num_classes == 50 and tv.y_mask.shape == (16, 50)
    assert tv.y_mask[4].nonzero().flatten().tolist() == [12, 13, 14, 15]         # Chair
    d = te[1]                                                                     # the table
    assert d.pos.shape == (20, 3) and d.norm.shape == (20, 3) and d.y.dtype == torch.long
    assert d.category.shape == (1, 16) and int(d.category.argmax()) == 15 and 47 <= int(d.y.min())
    only = ShapeNet(root, categories="Chair", split="trainval")                   # one-hot index WITHIN the selection
    assert len(only) == 2 and int(only[0].category.argmax()) == 0
    assert len(ShapeNet(root, categories=["Airplane", "Chair"], n_per_class=1, split="train")) == 2
    with pytest.raises(ValueError):
        ShapeNet(root, split="training")
    b = next(iter(DataLoader(tv, batch_size=2)))
    assert b.pos.shape == (40, 3) and b.y.shape == (40,) and b.category.shape == (2, 16)
    # the seg tables agree with the IoU metric's tables (experiments/utils.py:27-51)
    ious = calc_shape_IoU(d.y.numpy()[None], d.y.numpy()[None], [15], None)
    assert ious == [1.0]


def _write_ply(path, pos, faces, binary):
    import struct
    head = ["ply", "format " + ("binary_little_endian 1.0" if binary else "ascii 1.0"), "comment test",
            f"element vertex {len(pos)}", "property float x", "property float y", "property float z",
            "property uchar red", f"element face {len(faces)}", "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode())
        if binary:
            for p in pos:
                fh.write(struct.pack("<fffB", *p, 7))
            for f in faces:
                fh.write(struct.pack(f"<B{len(f)}i", len(f), *f))
        else:
            for p in pos:
                fh.write((" ".join(f"{v:.6f}" for v in p) + " 7\n").encode())
            for f in faces:
                fh.write((f"{len(f)} " + " ".join(map(str, f)) + "\n").encode())


def test_ply_obj_readers_and_edge_labels(tmp_path):
    from deltaconv_amd.datasets import edge_to_vertex_labels, read_obj, read_ply
    pos = [(0., 0., 0.), (1., 0., 0.), (1., 1., 0.), (0., 1., 0.), (0.5, 0.5, 1.)]
    faces = [(0, 1, 2, 3), (0, 1, 4), (1, 2, 4)]                      # one quad -> fanned
    for binary in (False, True):
        p = str(tmp_path / f"m{int(binary)}.ply")
        _write_ply(p, pos, faces, binary)
        d = read_ply(p)
        assert d.pos.shape == (5, 3) and torch.allclose(d.pos, torch.tensor(pos))
        assert d.face.t().tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4], [1, 2, 4]]
    with open(tmp_path / "bad.ply", "w") as fh:
        fh.write("plx\nend_header\n")
    with pytest.raises(ValueError):
        read_ply(str(tmp_path / "bad.ply"))
    obj = tmp_path / "m.obj"
    obj.write_text("# c\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4 -3 -1\n")
    o = read_obj(str(obj))
    assert o.pos.shape == (4, 3) and o.face.t().tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 3]]
    # per-edge -> per-vertex labels, against the vectorised formulation of shape_seg.py:173-191
    face = torch.tensor([[0, 1, 2], [0, 2, 3], [0, 1, 4], [1, 2, 4]]).t()
    seen, edges = set(), []
    for f in face.t():
        for e in (f[:2], f[1:], f[::2]):
            key = tuple(sorted(e.tolist()))
            if key not in seen:
                seen.add(key)
                edges.append(key)
    labels = torch.arange(1, len(edges) + 1)
    ei = torch.tensor(edges)
    ref = torch.zeros(5, dtype=torch.long)
    ref[ei[:, 0]] = labels
    ref[ei[:, 1]] = labels
    assert torch.equal(edge_to_vertex_labels(face, labels, 5), ref - 1)


def test_shapeseg_layout(tmp_path):
    from deltaconv_amd.datasets import ShapeSeg
    root = str(tmp_path / "ShapeSeg")
    with pytest.raises(FileNotFoundError):
        ShapeSeg(root)
    pos = [(0., 0., 0.), (1., 0., 0.), (1., 1., 0.), (0., 1., 0.)]
    faces = [(0, 1, 2), (0, 2, 3)]
    base = os.path.join(root, "raw", "ShapeSeg")
    counts = {"Adobe": 2, "FAUST": 3, "SCAPE": 1, "SHREC": 2}
    for name, n in counts.items():
        for sub in ("meshes", "segs"):
            os.makedirs(os.path.join(base, name, "raw", sub))
        for i in range(n):
            fn = f"tr_reg_{i:03d}.ply" if name == "FAUST" else f"{i}.ply"
            _write_ply(os.path.join(base, name, "raw", "meshes", fn), pos, faces, binary=(i % 2 == 0))
            if name in ("Adobe", "SHREC"):
                torch.save(torch.tensor([i, 1, 2, 3]), os.path.join(base, name, "raw", "segs", f"{i}.pt"))
    torch.save(torch.tensor([7, 7, 7, 7]), os.path.join(base, "FAUST", "raw", "segs", "faust_seg.pt"))
    torch.save(torch.tensor([5, 5, 5, 5]), os.path.join(base, "SCAPE", "raw", "segs", "scape_seg.pt"))
    for sub in ("meshes", "segs"):
        os.makedirs(os.path.join(base, "MIT", "raw", sub))
    with open(os.path.join(base, "MIT", "raw", "meshes", "crane_0.obj"), "w") as fh:
        fh.write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3\nf 1 3 4\n")
    with open(os.path.join(base, "MIT", "raw", "segs", "crane_0.eseg"), "w") as fh:
        fh.write("1\n2\n3\n4\n5\n")                                   # 5 unique edges
    tr = ShapeSeg(root, True, pre_transform=T.NormalizeScale())
    te = ShapeSeg(root, False)
    assert len(tr) == 2 + 3 + 1 + 1 and len(te) == 2 and repr(te) == "ShapeSeg(2)"
    assert tr[0].y.tolist() == [0, 1, 2, 3] and tr[2].y.tolist() == [7, 7, 7, 7]      # Adobe first, then FAUST
    mit = tr[5]
    assert mit.pos.shape == (4, 3) and mit.y.shape == (4,) and int(mit.y.min()) >= 0
    assert float(tr[0].pos.norm(dim=1).max()) < 1.0                                   # pre_transform applied
    b = collate([tr[0], tr[1]])
    assert b.pos.shape == (8, 3) and b.y.shape == (8,)


# ---- HDF5 reader + ScanObjectNN (files written by the real libhdf5: tests/golden/make_golden_h5.py) --------------------
import numpy as np                                      # noqa: E402
from tests.helpers import GOLDEN as _GOLDEN             # noqa: E402


@pytest.mark.parametrize("name", ["scanobjectnn_like_contiguous", "scanobjectnn_like_chunked_gzip", "mixed_types",
                                  "nested_groups"])
def test_hdf5_reader_against_libhdf5_files(name):
    from deltaconv_amd.io_hdf5 import File
    f = File(os.path.join(_GOLDEN, "h5", name + ".h5"))
    exp = np.load(os.path.join(_GOLDEN, "h5", name + "_expected.npz"))
    for key in exp.files:
        d = f[key]
        assert d.shape == exp[key].shape and d.dtype == exp[key].dtype
        assert np.array_equal(d[...], exp[key]), key
    if name != "nested_groups":
        assert sorted(f.keys()) == sorted(exp.files)
    else:
        assert f.keys() == ["main_split"] and len(f["main_split/train"].keys()) == 40
    with pytest.raises(KeyError):
        f["no_such_dataset"]


def test_hdf5_reader_rejects_what_it_cannot_read(tmp_path):
    from deltaconv_amd.io_hdf5 import File
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all" * 100)
    with pytest.raises(ValueError):
        File(str(p))
    raw = bytearray(open(os.path.join(_GOLDEN, "h5", "mixed_types.h5"), "rb").read())
    raw[8] = 2                                          # pretend superblock version 2 (libver='latest')
    p.write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError):
        File(str(p))


def test_scanobjectnn_reference_layout(tmp_path):
    """experiments/datasets/scanobjectnn.py: raw/main_split/<...>.h5 -> processed/bg_vanilla/{training,test}.pt."""
    import shutil
    from deltaconv_amd.datasets import ScanObjectNN
    import deltaconv_amd.transforms as T
    root = tmp_path / "ScanObjectNN"
    raw = root / "raw" / "main_split"
    raw.mkdir(parents=True)
    src = os.path.join(_GOLDEN, "h5", "scanobjectnn_like_contiguous.h5")
    shutil.copy(src, raw / "training_objectdataset.h5")
    shutil.copy(src, raw / "test_objectdataset.h5")
    exp = np.load(os.path.join(_GOLDEN, "h5", "scanobjectnn_like_contiguous_expected.npz"))
    ds = ScanObjectNN(str(root), background=True, train=True, pre_transform=T.NormalizeScale())
    assert len(ds) == 5 and repr(ds) == "ScanObjectNN(5)"
    assert os.path.exists(root / "processed" / "bg_vanilla" / "training.pt")
    d = ds[2]
    assert d.pos.shape == (2048, 3) and int(d.y) == int(exp["label"][2])
    assert float(d.pos.norm(dim=1).max()) == pytest.approx(0.999999, abs=1e-5)     # NormalizeScale ran
    test = ScanObjectNN(str(root), background=True, train=False)                   # second open: cached
    assert torch.equal(test[0].pos, ds[0].pos)
    with pytest.raises(RuntimeError, match="Dataset not found"):
        ScanObjectNN(str(tmp_path / "empty"), background=False)
    with pytest.raises(AssertionError):
        ScanObjectNN(str(root), augmentation="nope")
