"""Geodesic farthest-point sampling (reference: deltaconv/geometry/fps.py:5-17 over the pybind module
of deltaconv/cpp).  Same validation, same return value; the native part is the dependency-free C++
restatement in deltaconv_amd/csrc_host/fps.cpp behind a C ABI (include/deltaconv_host.h)."""
import ctypes
import os
import warnings

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB_PATH = os.path.join(_PKG, "lib", "libdeltaconv_host.so")
_lib = None


def _host():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is not built (make -C deltaconv_amd/csrc_host)")
        lib = ctypes.CDLL(HOST_LIB_PATH)
        lib.dc_geodesic_fps.restype = ctypes.c_int
        lib.dc_geodesic_fps.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
        _lib = lib
    return _lib


def geodesic_fps(points, n_samples, seed=None):
    """points: float [V,3] numpy array -> int32 [n_samples] sample indices.  ``seed=None`` starts from a
    random point like the reference (sampling.cpp:34-40); an integer makes the start reproducible."""
    if n_samples > points.shape[0]:
        warnings.warn("Number of samples is larger than number of points.")
    if type(points) is not np.ndarray:
        raise ValueError("`points` should be a numpy array")
    if (len(points.shape) != 2) or (points.shape[1] != 3):
        raise ValueError("`points` should have shape (V,3), shape is " + str(points.shape))
    pts = np.ascontiguousarray(points, dtype=np.float64)
    out = np.empty(int(n_samples), dtype=np.int32)
    rc = _host().dc_geodesic_fps(pts.ctypes.data, pts.shape[0], int(n_samples), -1 if seed is None else int(seed),
                                 out.ctypes.data)
    if rc != 0:
        raise ValueError("geodesic_fps: bad arguments")
    return out.squeeze()
