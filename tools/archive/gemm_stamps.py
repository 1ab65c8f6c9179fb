"""Lab (library built with -DDC_LAB_STAMPS): s_memtime stamps of every workgroup of the embedding forward product (split products,
statistics epilogue): entry / first tile in LDS / K loop done / statistics done / stores issued / stores complete, and the start times
of the workgroups (rounds).   DELTACONV_HIP_LIB=<lab build> python tools/gemm_stamps.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib

dev = "cuda"
M, N, K = 32768, 1024, 448
x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
y, coef = torch.empty(M, N, device=dev), torch.empty(4, N, device=dev)
g = torch.ones(N, device=dev)
nb = lib.raw("dc_linear_stats_workspace_bytes")(M, N, K, 0)
ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
for stagger in (-1, 50):
    lib.raw("dc_set_option")(4, stagger)
    for _ in range(3):
        lib.call("dc_linear_bn_stats_forward", x, K, w, K, M, N, K, y, N, g, g, 1e-5, 0.1, None, None, coef[0], coef[1], coef[2], coef[3], 0, ws, nb)
    torch.cuda.synchronize()
    fn = lib.load().dc_lab_read_stamps
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]
    nblk = (M // 128) * (N // 128)
    buf = np.zeros((8192, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data, 8192 * 8) == 0
    s = buf[:nblk].astype(np.int64)
    t0 = s[:, 0].min()
    ph = {"entry -> first tile in LDS": s[:, 7] - s[:, 0], "K loop": s[:, 2] - s[:, 7], "statistics epilogue": s[:, 3] - s[:, 2],
          "output staging + store issue": s[:, 4] - s[:, 3], "store completion": s[:, 5] - s[:, 4], "whole workgroup": s[:, 5] - s[:, 0]}
    print(f"## phase shift option {stagger}: {nblk} workgroups, kernel span {int(s[:, 5].max() - t0)} ticks")
    for k_, v in ph.items():
        print(f"  {k_:32s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} ticks")
    starts = np.sort(s[:, 0] - t0)
    print("  start times (ticks) at workgroup 0 / 256 / 511 / 512 / 768 / 1024 / 1536 / 2047:", [int(starts[i]) for i in (0, 256, 511, 512, 768, 1024, 1536, 2047)])
