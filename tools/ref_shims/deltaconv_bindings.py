"""Stub for the reference's un-buildable pybind module (geometry-central/Eigen absent)."""


def geodesicFPS(*a, **k):
    raise RuntimeError("deltaconv_bindings is not buildable in this container")
