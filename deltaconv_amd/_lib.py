"""ctypes binding of libdeltaconv_hip.so -- the C ABI declared in include/deltaconv_hip.h.

The argtypes are parsed from that header at import, so the Python side cannot drift from the
declared ABI.  There is NO fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
import re

import torch  # must be imported first: the HIP runtime already loaded by torch is reused

_PKG = os.path.dirname(os.path.abspath(__file__))
# DELTACONV_HIP_LIB: another build of the same library (A/B runs of compile-time switches, e.g. -DDC_NT_STORES=0)
LIB_PATH = os.environ.get("DELTACONV_HIP_LIB") or os.path.join(_PKG, "lib", "libdeltaconv_hip.so")
# the declared ABI: include/deltaconv_hip.h of the source tree; `make -C deltaconv_amd/csrc` copies it next to
# the library (deltaconv_amd/lib/), so an installed / copied package without the repository root still imports
_HEADERS = (os.path.join(os.path.dirname(_PKG), "include", "deltaconv_hip.h"),
            os.path.join(_PKG, "lib", "deltaconv_hip.h"))
HEADER_PATH = next((h for h in _HEADERS if os.path.exists(h)), _HEADERS[0])

_CTYPES = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t,
    "float": ctypes.c_float, "double": ctypes.c_double,
}
# element type a pointer parameter is declared with -> the torch dtype a tensor bound to it must have
_PTR_DTYPES = {"float": torch.float32, "double": torch.float64, "int32_t": torch.int32, "int": torch.int32,
               "int64_t": torch.int64, "uint8_t": torch.uint8, "unsigned char": torch.uint8}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames], [pointee dtype or None])} for every dc_* prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(dc_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else _CTYPES[ret.replace("const", "").strip()]
        argtypes, argnames, ptr_dtypes = [], [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split("*")[-1].strip())
                    ptr_dtypes.append(_PTR_DTYPES.get(a.split("*")[0].replace("const", "").strip()))   # void* -> None
                else:
                    ty, nm = a.replace("const ", "").rsplit(" ", 1)
                    argtypes.append(_CTYPES[ty.strip()])
                    argnames.append(nm)
                    ptr_dtypes.append(None)
        protos[name] = (restype, argtypes, argnames, ptr_dtypes)
    return protos


class _Lib:
    def __init__(self):
        self._cdll = None
        self.protos = parse_header()

    def load(self):
        if self._cdll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is not built. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C deltaconv_amd/csrc`). deltaconv_amd has no CPU / eager fallback.")
            cdll = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes, _, _) in self.protos.items():
                fn = getattr(cdll, name)  # AttributeError = header/library mismatch: fail loudly
                fn.restype, fn.argtypes = restype, argtypes
            self._cdll = cdll
            # DC_GEMM_EXACT=1: dense products through the exact fp32 MFMA chain instead of the bf16 split products
            # (option 3 of dc_set_option; A/B runs and bitwise comparisons against an fmaf chain)
            if os.environ.get("DC_GEMM_EXACT", "0") not in ("", "0"):
                cdll.dc_set_option(3, 1)
            # DC_OPTIONS="key=value,...": any experiment switch of dc_set_option at load (same-box A/B runs of bench.py)
            for item in filter(None, os.environ.get("DC_OPTIONS", "").split(",")):
                key, value = item.split("=")
                if cdll.dc_set_option(int(key), int(value)) != 0:
                    raise RuntimeError(f"DC_OPTIONS: dc_set_option({key}, {value}) refused")
        return self._cdll

    def last_error(self):
        return (self.load().dc_last_error() or b"").decode()

    def raw(self, name):
        return getattr(self.load(), name)

    def call(self, name, *args):
        """Call an int-returning entry point on the current torch stream (appended as last arg)."""
        fn = self.raw(name)
        _, _, argnames, ptr_dtypes = self.protos[name]
        conv = []
        dev = None
        for pos, a in enumerate(args):
            if isinstance(a, torch.Tensor):
                # the ABI takes raw pointers: catch what would otherwise be misread silently
                if not a.is_cuda:
                    raise RuntimeError(f"{name}: tensor argument is not on a HIP device (no CPU path exists)")
                want = ptr_dtypes[pos] if pos < len(ptr_dtypes) else None
                if want is not None and a.dtype != want:
                    raise TypeError(f"{name}: argument '{argnames[pos]}' must be {want}, got {a.dtype}")
                if a.dim() and a.stride(-1) != 1 and a.shape[-1] != 1:
                    raise ValueError(f"{name}: argument '{argnames[pos]}' must be contiguous along its last dimension "
                                     f"(strides {tuple(a.stride())}); rows may be strided through the ld argument")
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise RuntimeError(f"{name}: tensors on different devices ({dev} vs {a.device})")
                conv.append(a.data_ptr())
            else:
                conv.append(a)
        if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
            raise RuntimeError(f"{name}: tensors live on {dev}, the current device is cuda:{torch.cuda.current_device()}")
        conv.append(torch.cuda.current_stream().cuda_stream)
        rc = fn(*conv)
        if rc != 0:
            raise RuntimeError(f"{name} failed (rc={rc}): {self.last_error()}")


lib = _Lib()


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("deltaconv_amd needs a HIP device (MI355X); there is no CPU fallback. "
                           "The CPU restatement lives in oracle/ and is test infrastructure only.")
