// Element-level formulas of the fused BatchNorm / activation / vector non-linearity kernels
// (nn.hip), shared with the CPU host-check build like point_math.h / ell_math.h.
//
// Reference being restated: /root/reference/deltaconv/nn/mlp.py:7-17 (Linear -> BatchNorm1d ->
// LeakyReLU(0.2); Linear -> VectorNonLin) and nn/nonlin.py:11-86.
//   scalar block : y = act(scale_c * h + shift_c) (+ residual),   act = leaky(slope)
//   vector block : (y_u, y_v) = combine(P,Q);  n = |y|;  s = relu(scale_c * n + shift_c) / max(n, 1e-8)
//                  out = y * s
// with scale = gamma * invstd, shift = beta - mean * scale (train: batch statistics over all rows).
#pragma once
#include "point_math.h"

namespace dcnn {

constexpr float VEC_EPS = 1e-8f;  // nn/nonlin.py:8

DC_HD float act(float z, float slope) { return z > 0.f ? z : slope * z; }
DC_HD float dact(float z, float slope) { return z > 0.f ? 1.f : slope; }  // as ATen leaky_relu_backward

// ---- scalar block backward -----------------------------------------------------------------
// dz = dy * act'(z);  reductions: sum dz, sum dz * xhat  (xhat = (h - mean) * invstd)
DC_HD void bn_bwd_terms(float dy, float h, float scale, float shift, float mean, float invstd, float slope,
                        float& dz, float& dz_xhat) {
    const float z = fmaf(scale, h, shift);
    dz = dy * dact(z, slope);
    dz_xhat = dz * ((h - mean) * invstd);
}
// dh = gamma*invstd * (dz - sum_dz/R - xhat * sum_dz_xhat/R)      (training)
// dh = scale * dz                                                   (eval: running statistics)
DC_HD float bn_bwd_dh(float dy, float h, float scale, float shift, float mean, float invstd, float slope,
                      float gi /*gamma*invstd*/, float m1 /*sum_dz/R*/, float m2 /*sum_dz_xhat/R*/, int training) {
    const float z = fmaf(scale, h, shift);
    const float dz = dy * dact(z, slope);
    if (!training) return scale * dz;
    const float xhat = (h - mean) * invstd;
    return gi * (dz - m1 - xhat * m2);
}

// ---- vector block ------------------------------------------------------------------------------
// With W = [W1 | W2] acting on I_J(a) = [a | J a] (geometry/operators.py:19-21):
//   row u: W1 a_u - W2 a_v,  row v: W1 a_v + W2 a_u.   P = a W1^T, Q = a W2^T  (one GEMM, N = 2*co)
DC_HD void vn_combine(float pu, float qu, float pv, float qv, float& yu, float& yv) {
    yu = pu - qv;
    yv = pv + qu;
}
DC_HD float vn_norm(float yu, float yv) { return sqrtf(fmaf(yu, yu, yv * yv)); }

// forward scale factor s = relu(scale*n + shift) / max(n, eps)                     (nonlin.py:67-79)
DC_HD float vn_scale(float n, float scale, float shift) {
    const float z = fmaf(scale, n, shift);
    return (z > 0.f ? z : 0.f) / fmaxf(n, VEC_EPS);
}

// backward pieces at one (point, channel):  g_s = y . dout;  dz = g_s / max(n,eps) * [z > 0]
DC_HD void vn_bwd_terms(float yu, float yv, float du, float dv, float scale, float shift, float mean, float invstd,
                        float& dz, float& dz_nhat) {
    const float n = vn_norm(yu, yv);
    const float z = fmaf(scale, n, shift);
    const float gs = yu * du + yv * dv;
    dz = z > 0.f ? gs / fmaxf(n, VEC_EPS) : 0.f;
    dz_nhat = dz * ((n - mean) * invstd);
}
// dy = dout * s + dn * y / n,   dn = BN-backward(dz) - g_s * r / nc^2 * [n > eps]
DC_HD void vn_bwd_dy(float yu, float yv, float du, float dv, float scale, float shift, float mean, float invstd,
                     float gi, float m1, float m2, int training, float& dyu, float& dyv) {
    const float n = vn_norm(yu, yv);
    const float nc = fmaxf(n, VEC_EPS);
    const float z = fmaf(scale, n, shift);
    const float r = z > 0.f ? z : 0.f;
    const float s = r / nc;
    const float gs = yu * du + yv * dv;
    const float dz = z > 0.f ? gs / nc : 0.f;
    float dn = training ? gi * (dz - m1 - ((n - mean) * invstd) * m2) : scale * dz;
    if (n > VEC_EPS) dn -= gs * r / (nc * nc);  // through the clamp(EPS) denominator
    const float w = n > 0.f ? dn / n : 0.f;     // d|y|/dy = y/|y|, subgradient 0 at 0 (as torch)
    dyu = fmaf(du, s, w * yu);
    dyv = fmaf(dv, s, w * yv);
}

// ---- dropout inside the row-block kernels (round 6) -------------------------------------------------------------------------
// torch.nn.Dropout(p) between the blocks of the classification head (deltanet_classification.py:34-36): keep with
// probability 1 - p, scale the kept values by 1 / (1 - p).  The draws come from Philox-4x32-10 (Salmon et al., SC'11; the
// generator torch itself uses) keyed by the process seed, counter = (element, layer salt, training-step counter of the
// block's BatchNorm = its num_batches_tracked, read on the device: a new mask in every replay of a captured step).
// Mask streams differ from ATen's (another counter layout) -- no implementation-independent dropout stream exists.
struct U4 {
    unsigned x, y, z, w;
};
DC_HD unsigned mulhi32(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * (unsigned long long)b) >> 32); }
DC_HD U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
DC_HD bool dropout_keep(unsigned seed, long long step, unsigned salt, unsigned element, float p) {
    const U4 r = philox4x32_10(U4{element, salt, (unsigned)step, (unsigned)((unsigned long long)step >> 32)}, seed, 0x64726F70u);
    return (float)(r.x >> 8) * (1.0f / 16777216.0f) >= p;      // 24 uniform bits in [0, 1)
}

}  // namespace dcnn
