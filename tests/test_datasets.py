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
