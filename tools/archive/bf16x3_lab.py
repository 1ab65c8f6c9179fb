"""Lab (round-2 verdict item 8): fp32 products emulated on the bf16 MFMA pipe with a 3-term split.

    a = a1 + a2 + a3 (each bf16: 8 + 8 + 8 = 24 mantissa bits, exact), same for b;  a b = sum_ij ai bj.
    Keeping the terms with i + j <= 4 gives 6 products (a1b1, a1b2, a2b1, a1b3, a3b1, a2b2); 8 adds a2b3, a3b2.
Every bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32; the fp32 MFMA rate is 1/16 of bf16.

This script measures, per GEMM shape of the models, (a) the error against an fp64 product of the exact-fp32 kernel
(csrc/gemm.hip), of the vendor fp32 GEMM and of the split products, (b) the time of a bf16 GEMM over the concatenated
planes (K' = 6 K or 8 K) through the vendor library -- the rate a fused split-in-the-loader kernel could approach -- and
of the split pass itself.  Acceptance (verdict): error no larger than the exact-fp32 kernel's on every shape."""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaconv_amd._lib import lib

dev = "cuda"


def split3(a):
    a1 = a.to(torch.bfloat16)
    r1 = a - a1.float()
    a2 = r1.to(torch.bfloat16)
    r2 = r1 - a2.float()
    a3 = r2.to(torch.bfloat16)
    return a1, a2, a3


def planes(x, w, terms):
    x1, x2, x3 = split3(x)
    w1, w2, w3 = split3(w)
    px = {1: x1, 2: x2, 3: x3}
    pw = {1: w1, 2: w2, 3: w3}
    # small terms first in K order (the accumulation then adds small to large inside a row of K tiles)
    order = sorted(terms, key=lambda t: -(t[0] + t[1]))
    return torch.cat([px[i] for i, _ in order], 1).contiguous(), torch.cat([pw[j] for _, j in order], 1).contiguous()


T6 = [(1, 1), (1, 2), (2, 1), (1, 3), (3, 1), (2, 2)]
T8 = T6 + [(2, 3), (3, 2)]


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def mm_bf16_f32(a, b):
    try:
        return torch.mm(a, b.t(), out_dtype=torch.float32)
    except TypeError:
        return None


print(f"# {torch.cuda.get_device_name(0)}, torch {torch.__version__}")
print(f"{'M x N x K':22s} {'err own fp32':>12s} {'err lib fp32':>12s} {'err split6':>11s} {'err split8':>11s} | "
      f"{'own us':>8s} {'lib us':>8s} {'bf16 6K us':>10s} {'bf16 8K us':>10s} {'split us':>9s}")
torch.manual_seed(0)
for (M, N, K) in [(32768, 1024, 512), (32768, 256, 512), (65536, 256, 256), (32768, 128, 256), (65536, 128, 384),
                  (32768, 64, 256), (65536, 64, 140), (32768, 64, 64)]:
    for scale in ("randn", "wide"):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        if scale == "wide":                       # entries over many binades (activations after BN / ReLU look like this)
            x = x * torch.exp(3 * torch.randn(M, K, device=dev))
            x[x.abs() < 0.5] = 0
        ref = (x[:4096].double() @ w.double().t())
        nrm = ref.abs().max()
        y = torch.empty(M, N, device=dev)
        lib.call("dc_linear_forward", x, K, w, K, M, N, K, y, N, 0)
        e_own = float((y[:4096].double() - ref).abs().max() / nrm)
        yl = x @ w.t()
        e_lib = float((yl[:4096].double() - ref).abs().max() / nrm)
        res = {}
        for name, terms in (("6", T6), ("8", T8)):
            xa, wa = planes(x, w, terms)
            ys = mm_bf16_f32(xa, wa)
            if ys is None:
                res[name] = (float("nan"), float("nan"))
                continue
            e = float((ys[:4096].double() - ref).abs().max() / nrm)
            t = timeit(lambda: mm_bf16_f32(xa, wa))
            res[name] = (e, t)
        t_own = timeit(lambda: lib.call("dc_linear_forward", x, K, w, K, M, N, K, y, N, 0))
        t_lib = timeit(lambda: torch.mm(x, w.t()))
        t_split = timeit(lambda: planes(x, w, T6))
        print(f"{M:6d}x{N:5d}x{K:4d} {scale:5s} {e_own:12.2e} {e_lib:12.2e} {res['6'][0]:11.2e} {res['8'][0]:11.2e} | "
              f"{t_own:8.1f} {t_lib:8.1f} {res['6'][1]:10.1f} {res['8'][1]:10.1f} {t_split:9.1f}")
