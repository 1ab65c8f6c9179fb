"""Small HDF5 fixtures written by the REAL libhdf5 (h5py 3.3 / libhdf5 1.10.6 of the image's conda python -- h5py is not
importable from the system interpreter), in the layout of the ScanObjectNN files the reference opens
(experiments/datasets/scanobjectnn.py:66-73,90-95: main_split/<train|test>_objectdataset.h5 with 'data' [N,2048,3]
float32, 'label' [N], 'mask' [N,2048]).  The expected arrays are stored next to each file as .npz.

    PYTHONPATH=/opt/conda/lib/python3.9/site-packages /opt/conda/bin/python3.9 tests/golden/make_golden_h5.py
"""
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5")
os.makedirs(HERE, exist_ok=True)
rng = np.random.default_rng(7)


def arrays(n, pts):
    return dict(data=rng.normal(size=(n, pts, 3)).astype(np.float32),
                label=rng.integers(0, 15, size=(n,)).astype(np.int64),
                mask=rng.integers(-1, 2, size=(n, pts)).astype(np.float32))


def write(name, arrs, **kw):
    path = os.path.join(HERE, name)
    with h5py.File(path, "w") as f:
        for k, v in arrs.items():
            f.create_dataset(k, data=v, **kw)
    np.savez_compressed(path.replace(".h5", "_expected.npz"), **arrs)
    print(name, os.path.getsize(path))


write("scanobjectnn_like_contiguous.h5", arrays(5, 2048))                                  # what h5py writes by default
write("scanobjectnn_like_chunked_gzip.h5", arrays(7, 256), chunks=True, compression="gzip", shuffle=True)
a = arrays(3, 64)
a["label"] = a["label"].astype(">i4")                                                      # big-endian, other width
a["small"] = np.arange(6, dtype=np.uint8).reshape(2, 3)
write("mixed_types.h5", a, chunks=(2, 16, 3) if False else None)
with h5py.File(os.path.join(HERE, "nested_groups.h5"), "w") as f:                           # groups inside groups, many links
    g = f.create_group("main_split").create_group("train")
    exp = {}
    for i in range(40):
        v = rng.normal(size=(3, 4)).astype(np.float64)
        g.create_dataset(f"obj_{i:03d}", data=v)
        exp[f"main_split/train/obj_{i:03d}"] = v
    np.savez_compressed(os.path.join(HERE, "nested_groups_expected.npz"), **exp)
