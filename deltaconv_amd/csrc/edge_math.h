// Layer-0 ("centralized") edge MLP of DeltaConv without the [E, C] edge tensor.
//
// Reference (deltaconv/nn/deltaconv.py:50-52 with nn/mlp.py:7-11, depth 1):
//     x_edge = x[col] - x[row]                       [E, ci]
//     h      = LeakyReLU(BatchNorm(Linear(x_edge)))  [E, co]   (BN statistics over all E rows)
//     x_max  = scatter(h, row, reduce='max')         [Nt, co]
// With y = x W^T (no bias) the pre-activation of edge e = (i, j) is a_e = y_j - y_i, and because
// z = scale*a + shift followed by a monotone non-decreasing activation is monotone in a,
//     max_s act(z_e) = act(scale * a* + shift),   a* = max_s a_e (scale >= 0)  or  min_s a_e (scale < 0).
// The BN statistics need only sum_e a_e and sum_e a_e^2.  So one gather pass over y gives everything
// (per point: max, min, their slots, sum), and the [E,co] tensors (168 MB at C2 for every one of
// Linear, BN, LeakyReLU, max and their backward) never exist.
//
// Backward.  Only the selected edge of each (i,c) receives dz* = dout * act'(z*); the BN backward
//     da_e = scale_c * (dz_e - m1 - ahat_e * m2),  m1 = sum dz*/E, m2 = sum dz* ahat*/E
// touches every edge through the mean/variance terms, but its sums over the in- and out-edges of a
// point are closed-form in  indeg(j), T_j = sum_{in-edges} y_i  and  s1_j = sum_s a_(j,s):
//     dy_j = scale * [ sum_{in-edges selected} dz*_i  -  dz*_j
//                      - m1 * (indeg_j - k)
//                      - m2 * invstd * ( (indeg_j*y_j - T_j - indeg_j*mu) - (s1_j - k*mu) ) ]
#pragma once
#include "ell_math.h"
#include "nn_math.h"

namespace dcedge {
using dcell::Vec;
using dcell::vload;
using dcell::vstore;

// per (point, channel group): gather pass.  Outputs amax/amin (+ first-extremal slots), s1 = sum_s a,
// and the two statistics contributions sum_s a, sum_s a^2 (fp64: E[a^2]-E[a]^2 cancellation).
template <int V>
DC_HD void edge_gather(long i, int c0, const float* y, long ldy, const int* nbr, int k, float* amax, float* amin,
                       unsigned char* argmax, unsigned char* argmin, float* s1pt, long ldo, long lda,
                       double (&t)[2][V]) {
    const Vec<V> yi = vload<V>(y + i * ldy + c0);
    Vec<V> mx, mn, s1;
    unsigned char smx[V], smn[V];
    double q2[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { smx[q] = 0; smn[q] = 0; q2[q] = 0.0; }
#pragma unroll 4
    for (int s = 0; s < k; ++s) {
        const Vec<V> yj = vload<V>(y + (long)nbr[i * k + s] * ldy + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const float a = yj.v[q] - yi.v[q];
            if (s == 0) { mx.v[q] = a; mn.v[q] = a; s1.v[q] = a; }
            else {
                const bool up = a > mx.v[q], dn = a < mn.v[q];
                mx.v[q] = up ? a : mx.v[q]; smx[q] = up ? (unsigned char)s : smx[q];
                mn.v[q] = dn ? a : mn.v[q]; smn[q] = dn ? (unsigned char)s : smn[q];
                s1.v[q] += a;
            }
            q2[q] += (double)a * (double)a;
        }
    }
    vstore<V>(amax + i * ldo + c0, mx);
    vstore<V>(amin + i * ldo + c0, mn);
    vstore<V>(s1pt + i * ldo + c0, s1);
#pragma unroll
    for (int q = 0; q < V; ++q) {
        argmax[i * lda + c0 + q] = smx[q];
        argmin[i * lda + c0 + q] = smn[q];
        t[0][q] = (double)s1.v[q];
        t[1][q] = q2[q];
    }
}

DC_HD float pick(float scale, float amax, float amin) { return scale >= 0.f ? amax : amin; }

// backward, per element: dz* and dz* * ahat*
DC_HD void edge_bwd_terms(float dout, float amax, float amin, float scale, float shift, float mean, float invstd,
                          float slope, float& dz, float& dz_ahat) {
    const float a = pick(scale, amax, amin);
    const float z = fmaf(scale, a, shift);
    dz = dout * dcnn::dact(z, slope);
    dz_ahat = dz * ((a - mean) * invstd);
}

// backward, CSC pass per (point j, channel group)
template <int V>
DC_HD void edge_bwd_point(long t, int groups, const int* tptr, const int* tedge, int k, const float* y, long ldy,
                          const float* dzs, const float* s1pt, long ldo, const unsigned char* argmax,
                          const unsigned char* argmin, long lda, const float* scale, const float* mean,
                          const float* invstd, const float* m1, const float* m2, int training, float* dy, long lddy) {
    const long j = t / groups;
    const int c0 = (int)(t % groups) * V;
    Vec<V> sel, T;
#pragma unroll
    for (int q = 0; q < V; ++q) { sel.v[q] = 0.f; T.v[q] = 0.f; }
    const int p0 = tptr[j], p1 = tptr[j + 1];
    for (int p = p0; p < p1; ++p) {
        const long e = tedge[p];
        const long i = e / k;
        const unsigned char s = (unsigned char)(e - i * k);
        const Vec<V> yi = vload<V>(y + i * ldy + c0);
        const Vec<V> dz = vload<V>(dzs + i * ldo + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const unsigned char a = scale[c0 + q] >= 0.f ? argmax[i * lda + c0 + q] : argmin[i * lda + c0 + q];
            sel.v[q] += (a == s) ? dz.v[q] : 0.f;
            T.v[q] += yi.v[q];
        }
    }
    const float indeg = (float)(p1 - p0);
    const Vec<V> yj = vload<V>(y + j * ldy + c0), dzj = vload<V>(dzs + j * ldo + c0), s1 = vload<V>(s1pt + j * ldo + c0);
    Vec<V> out;
#pragma unroll
    for (int q = 0; q < V; ++q) {
        const int c = c0 + q;
        float g = sel.v[q] - dzj.v[q];
        if (training) {
            const float col_hat = (indeg * yj.v[q] - T.v[q] - indeg * mean[c]) * invstd[c];
            const float row_hat = (s1.v[q] - (float)k * mean[c]) * invstd[c];
            g -= m1[c] * (indeg - (float)k) + m2[c] * (col_hat - row_hat);
        }
        out.v[q] = scale[c] * g;
    }
    vstore<V>(dy + j * lddy + c0, out);
}

}  // namespace dcedge
