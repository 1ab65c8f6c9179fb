"""CPU: the arithmetic behind the default dense products (deltaconv_amd/csrc/gemm.hip, split products) restated with torch's
bfloat16 casts (round to nearest even, what v_cvt_pk_bf16_f32 does): an fp32 value is cut into three bf16 planes, a product keeps
the six partial products of weight >= 2^-16.  What must hold for the kernel's error claim (at or below the exact fp32 chain's):
  * the two residuals are exact in fp32 and hi + mid + lo reproduces x to 2^-24 |x| (three planes carry 24+ bits);
  * the six kept partial products reproduce a * b to ~2^-23 |a b| (the dropped ones are mid.lo, lo.mid, lo.lo);
  * every partial product of two bf16 values is exact in fp32 (8 x 8 significant bits), so only the accumulation rounds;
  * a dot product accumulated from the six planes in fp32 is as close to the fp64 result as a plain fp32 dot product."""
import torch


def split3(x):
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    r2 = r1 - mid.float()
    lo = r2.to(torch.bfloat16)
    return hi, mid, lo, r1, r2


def _operands(n, seed, binades):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, generator=g) * torch.exp(binades * torch.randn(n, generator=g))
    x[::17] = 0.0
    return x


def test_planes_reconstruct_the_operand():
    x = _operands(1 << 16, 0, 6.0)
    hi, mid, lo, r1, r2 = split3(x)
    xd = x.double()
    # residuals computed in fp32 are the exact differences
    assert torch.equal(r1.double(), xd - hi.double())
    assert torch.equal(r2.double(), xd - hi.double() - mid.double())
    rec = hi.double() + mid.double() + lo.double()
    nz = x != 0
    assert float(((rec - xd).abs()[nz] / xd.abs()[nz]).max()) <= 2.0 ** -24
    assert torch.equal(rec[~nz], xd[~nz])


def test_six_partial_products_reproduce_the_product():
    a, b = _operands(1 << 16, 1, 4.0), _operands(1 << 16, 2, 4.0)
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    d = lambda t: t.double()
    six = d(al) * d(bh) + d(ah) * d(bl) + d(am) * d(bm) + d(am) * d(bh) + d(ah) * d(bm) + d(ah) * d(bh)
    exact = a.double() * b.double()
    nz = exact != 0
    assert float(((six - exact).abs()[nz] / exact.abs()[nz]).max()) < 2.0 ** -22
    # a product of two bf16 values fits fp32 exactly
    p32 = ah.float() * bh.float()
    assert torch.equal(p32.double(), d(ah) * d(bh))


def test_split_dot_product_is_fp32_grade():
    K = 448
    worst_split = worst_plain = 0.0
    for seed in range(8):
        a, b = _operands(K, 10 + seed, 2.0), _operands(K, 30 + seed, 2.0)
        ref = float((a.double() * b.double()).sum())
        scale = float((a.double() * b.double()).abs().sum())
        ah, am, al, _, _ = split3(a)
        bh, bm, bl, _, _ = split3(b)
        acc = torch.zeros((), dtype=torch.float32)
        for k0 in range(0, K, 16):                          # one MFMA k-step: 16 products summed (exactly, here in fp64), ONE fp32 rounding
            s = slice(k0, k0 + 16)
            for p, q in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
                acc = (acc.double() + (p[s].double() * q[s].double()).sum()).float()
        plain = torch.zeros((), dtype=torch.float32)
        for k in range(K):                                  # the exact chain: one fma per element
            plain = (plain.double() + a[k].double() * b[k].double()).float()
        worst_split = max(worst_split, abs(float(acc) - ref) / scale)
        worst_plain = max(worst_plain, abs(float(plain) - ref) / scale)
    assert worst_split < 2.0 ** -21
    assert worst_split < 2.0 * worst_plain + 2.0 ** -24
