"""The BASELINE.json configurations as data: cloud shape, model, optimizer, synthetic-input recipe
(SURVEY.md section 8 notation and 8(d) generator; reference: experiments/train_modelnet.py:50-68,
train_scanobjectnn.py:60-76,173, train_shapenet.py:77-89,169-192, train_shapeseg.py:68-83,148-163).

Used by bench.py (`--config`), tools/bench_configs.py and the GPU tests of the full-size configurations."""

CONFIGS = {
    "C2": dict(title="ModelNet40 classification", B=32, N=1024, k=20, normals=True, kind="cls",
               model=dict(in_channels=3, num_classes=40), batch={}, optimizer="sgd", C=64, dp_over=None),
    "C3": dict(title="ScanObjectNN (main_split, bg) classification, no normals", B=32, N=2048, k=20, normals=False, kind="cls",
               model=dict(in_channels=3, num_classes=15, conv_channels=[64, 64, 64, 128], grad_regularizer=1e-2),
               batch=dict(outlier_frac=0.05, jitter=0.005, num_classes=15), optimizer="sgd", C=64, dp_over=None),
    "C4": dict(title="ShapeNet part segmentation (all classes)", B=16, N=2048, k=20, normals=True, kind="seg",
               model=dict(in_channels=3, num_classes=50, categorical_vector=True),
               batch=dict(dup_frac=0.03, per_point_labels=True, categories=16, num_classes=50), optimizer="sgd", C=64,
               dp_over=8),
    "C5": dict(title="Human body shape segmentation (shapeseg)", B=8, N=4096, k=30, normals=True, kind="seg",
               model=dict(in_channels=3, num_classes=8, conv_channels=[128] * 8, mlp_depth=1, embedding_size=512),
               batch=dict(per_point_labels=True, num_classes=8), optimizer="adam", C=128, dp_over=8),
}


def build_model(name, package=None, k=None):
    """The model of configuration `name` from `package` (deltaconv_amd.models by default; the oracle's in tests / the CPU leg)."""
    cfg = CONFIGS[name]
    if package is None:
        from . import models as package
    cls = package.DeltaNetSegmentation if cfg["kind"] == "seg" else package.DeltaNetClassification
    return cls(num_neighbors=cfg["k"] if k is None else k, **cfg["model"])


def build_optimizer(name, params):
    """train_modelnet.py:67 (SGD lr 0.1, momentum 0.9, weight decay 1e-4) / train_shapeseg.py:82 (Adam lr 5e-3)."""
    if CONFIGS[name]["optimizer"] == "sgd":
        from .optim import SGD           # torch.optim.SGD with its step in one launch (csrc/optim.hip)
        return SGD(params, lr=0.1, momentum=0.9, weight_decay=1e-4)
    from .optim import Adam              # torch.optim.Adam with its step in one launch (csrc/optim.hip)
    return Adam(params, lr=5e-3)


def make_batch(name, clouds, seed, points=None):
    from .data import synthetic_batch
    cfg = CONFIGS[name]
    return synthetic_batch(clouds, cfg["N"] if points is None else points, seed=seed, normals=cfg["normals"], **cfg["batch"])


def loss_smoothing(name):
    """Label-smoothed cross entropy for the classification nets, plain for the segmentation nets (experiments/utils.py:7-24)."""
    return CONFIGS[name]["kind"] != "seg"
