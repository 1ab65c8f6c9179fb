"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/*.h declares;
the product refuses to run without a HIP device instead of falling back to anything."""
import ctypes
import os
import subprocess

import pytest
import torch

from tests.helpers import ROOT


@pytest.fixture(scope="module")
def built():
    subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "deltaconv_amd", "csrc")], check=True)
    from deltaconv_amd._lib import lib, LIB_PATH
    return lib, LIB_PATH


def test_every_declared_symbol_is_exported(built):
    lib, path = built
    protos = lib.protos
    assert len(protos) >= 19 and "dc_knn" in protos and "dc_apply_grad_T" in protos
    cdll = ctypes.CDLL(path)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in include/deltaconv_hip.h but not exported"
    exported = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in exported.splitlines() if " T dc_" in l}
    assert exported == set(protos), f"header/library mismatch: {exported ^ set(protos)}"
    assert lib.load().dc_version() >= 100


def test_argument_errors_are_reported(built):
    lib, _ = built
    rc = lib.raw("dc_knn")(None, None, 1, 1, 20, 0, None, None)
    assert rc == -1 and "null" in lib.last_error()
    rc = lib.raw("dc_apply_grad")(None, None, 1, 1, None, 4, 4, None, 4, None)
    assert rc == -1
    assert lib.raw("dc_mls_workspace_bytes")(32, 32768) >= 32768 * 48


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    import deltaconv_amd as dc
    with pytest.raises(RuntimeError, match="no CPU"):
        dc.geometry.build_tangent_basis(torch.randn(4, 3))
    with pytest.raises(RuntimeError):
        dc.geometry.knn_graph(torch.randn(64, 3), 8)


def test_host_library_exports():
    """libdeltaconv_host.so (geodesic FPS, include/deltaconv_host.h)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "deltaconv_amd", "csrc_host")], check=True)
    from deltaconv_amd.geometry.fps import HOST_LIB_PATH
    lib = ctypes.CDLL(HOST_LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "deltaconv_host.h")).read()
    import re
    names = set(re.findall(r"\b(dc_\w+)\s*\(", hdr))
    assert names == {"dc_host_version", "dc_geodesic_fps"}
    for n in names:
        assert hasattr(lib, n)
    assert lib.dc_host_version() >= 100
    assert lib.dc_geodesic_fps(None, 0, 0, 0, None) == -1
