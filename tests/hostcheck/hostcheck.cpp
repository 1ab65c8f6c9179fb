// TEST INFRASTRUCTURE: a g++ (CPU) build of deltaconv_amd/csrc/point_math.h, looping the very
// same per-point / per-edge functions the HIP kernels call, so their arithmetic can be checked
// against the oracle in the GPU-less build container (tests/test_hostcheck.py).  Not shipped,
// not a fallback: nothing in deltaconv_amd/ loads this library.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../deltaconv_amd/csrc/point_math.h"
#include "../../deltaconv_amd/csrc/ell_math.h"
#include "../../deltaconv_amd/csrc/nn_math.h"
#include "../../deltaconv_amd/csrc/edge_math.h"
#include "../../deltaconv_amd/csrc/loss_math.h"

extern "C" {

void hc_tangent_basis(const float* normal, int n, float* xb, float* yb) {
    for (int i = 0; i < n; ++i) dcmath::tangent_basis_point(normal + 3 * i, xb + 3 * i, yb + 3 * i);
}

void hc_estimate_basis(const float* pos, const int* nbr, int n, int k, const float* orient, float* normal, float* xb,
                       float* yb) {
    for (int i = 0; i < n; ++i)
        dcmath::estimate_basis_point(pos, nbr + (long)i * k, i, k, orient, normal + 3 * i, xb + 3 * i, yb + 3 * i);
}

// Same three phases as dc_mls_assemble (mls.hip).
void hc_mls_assemble(const float* pos, const float* normal, const float* xb, const float* yb, const int* nbr,
                     const int* cloud_ptr, int num_clouds, int k, float kernel_width, float regularizer,
                     int normalized, float* G, float* D) {
    for (int c = 0; c < num_clouds; ++c) {
        const int begin = cloud_ptr[c], n = cloud_ptr[c + 1] - begin;
        double acc = 0;
        for (int q = 0; q < n; ++q) {
            const long i = begin + q;
            acc += dcmath::point_dist_sum(pos, nbr + i * k, i, k) / k;
        }
        const double avg = n > 0 ? acc / n : 0.0;
        std::vector<double> coef((size_t)n * 6);
        float inf_norm = 0.f;
        for (int q = 0; q < n; ++q) {
            const long i = begin + q;
            inf_norm = std::max(inf_norm, dcmath::mls_fit_point(pos, normal, xb, yb, nbr + i * k, i, k, avg,
                                                                (double)kernel_width, (double)regularizer,
                                                                G + i * k * 2, coef.data() + (size_t)q * 6));
        }
        for (long le = 0; le < (long)n * k; ++le) {
            const long e = (long)begin * k + le, i = e / k, j = nbr[e];
            const dcmath::Frame fi = dcmath::load_frame(pos, normal, xb, yb, i);
            dcmath::mls_div_edge(fi, coef.data() + (size_t)(i - begin) * 6, dcmath::ld3(pos + 3 * j),
                                 dcmath::ld3(xb + 3 * j), dcmath::ld3(yb + 3 * j), normalized ? inf_norm : 0.f,
                                 G + 2 * e, D + 2 * e);
        }
    }
}

// dc_mls_assemble_shape: the surface fit with its own regulariser (mls_fit_point<true>)
void hc_mls_assemble_shape(const float* pos, const float* normal, const float* xb, const float* yb, const int* nbr,
                           const int* cloud_ptr, int num_clouds, int k, float kernel_width, float regularizer,
                           float shape_regularizer, int normalized, float* G, float* D) {
    for (int c = 0; c < num_clouds; ++c) {
        const int begin = cloud_ptr[c], n = cloud_ptr[c + 1] - begin;
        double acc = 0;
        for (int q = 0; q < n; ++q) {
            const long i = begin + q;
            acc += dcmath::point_dist_sum(pos, nbr + i * k, i, k) / k;
        }
        const double avg = n > 0 ? acc / n : 0.0;
        std::vector<double> coef((size_t)n * 6);
        float inf_norm = 0.f;
        for (int q = 0; q < n; ++q) {
            const long i = begin + q;
            inf_norm = std::max(inf_norm, dcmath::mls_fit_point<true>(pos, normal, xb, yb, nbr + i * k, i, k, avg,
                                                                      (double)kernel_width, (double)regularizer,
                                                                      G + i * k * 2, coef.data() + (size_t)q * 6,
                                                                      (double)shape_regularizer));
        }
        for (long le = 0; le < (long)n * k; ++le) {
            const long e = (long)begin * k + le, i = e / k, j = nbr[e];
            const dcmath::Frame fi = dcmath::load_frame(pos, normal, xb, yb, i);
            dcmath::mls_div_edge(fi, coef.data() + (size_t)(i - begin) * 6, dcmath::ld3(pos + 3 * j),
                                 dcmath::ld3(xb + 3 * j), dcmath::ld3(yb + 3 * j), normalized ? inf_norm : 0.f,
                                 G + 2 * e, D + 2 * e);
        }
    }
}

// ---- the stages of build_grad_div on their own (dc_mls_coords / _gaussian_weights / _wls / _vector_mapping):
// the loops of the four stage kernels of mls.hip around the same dcmath:: functions
void hc_mls_coords(const float* pos, const float* normal, const float* xb, const float* yb, const int* row,
                   const int* col, long num_edges, int k, float* coords) {
    for (long e = 0; e < num_edges; ++e) {
        const long f = e / k;
        dcmath::Frame fr{dcmath::ld3(pos + 3 * (long)row[e]), dcmath::ld3(normal + 3 * f), dcmath::ld3(xb + 3 * f),
                         dcmath::ld3(yb + 3 * f)};
        const dcmath::EdgeGeom g = dcmath::edge_geom(fr, dcmath::ld3(pos + 3 * (long)col[e]));
        coords[2 * e] = (float)g.u;
        coords[2 * e + 1] = (float)g.v;
    }
}

void hc_mls_gaussian_weights(const float* dist, const int* cloud_ptr, int num_clouds, int k, float kernel_width,
                             float* weights) {
    for (int c = 0; c < num_clouds; ++c) {
        const int begin = cloud_ptr[c], n = cloud_ptr[c + 1] - begin;
        double acc = 0;
        for (int q = 0; q < n; ++q) {
            double s = 0;
            for (int e = 0; e < k; ++e) s += (double)dist[(long)(begin + q) * k + e];
            acc += s / k;
        }
        const double avg = n > 0 ? acc / n : 0.0;
        for (int q = 0; q < n; ++q) {
            const long i = begin + q;
            dcmath::gaussian_weights_point(dist + i * k, k, avg, (double)kernel_width, weights + i * k);
        }
    }
}

void hc_mls_wls(const float* coords, const float* weights, int num_points, int k, float regularizer, float* wls) {
    for (long i = 0; i < num_points; ++i)
        dcmath::wls_point(coords + i * k * 2, weights + i * k, k, (double)regularizer, wls + i * k * 6);
}

void hc_mls_vector_mapping(const float* pos, const float* normal, const float* xb, const float* yb, const int* row,
                           const int* col, long num_edges, int k, const float* wls, const float* coords, float* out) {
    for (long e = 0; e < num_edges; ++e) {
        const long g0 = e / k * k, i = row[e], j = col[e];
        const dcmath::Frame fi = dcmath::load_frame(pos, normal, xb, yb, i);
        double c[6] = {0, 0, 0, 0, 0, 0};
        for (int s = 0; s < k; ++s) {
            const dcmath::V3 d = dcmath::sub(dcmath::ld3(pos + 3 * (long)col[g0 + s]), fi.p);
            const double h = dcmath::dot(fi.n, d);
            for (int a = 0; a < 6; ++a) c[a] += (double)wls[(g0 + s) * 6 + a] * h;
        }
        double m[4];
        dcmath::vector_map(fi, c, (double)coords[2 * e], (double)coords[2 * e + 1], dcmath::ld3(xb + 3 * j),
                           dcmath::ld3(yb + 3 * j), m);
        for (int a = 0; a < 4; ++a) out[4 * e + a] = (float)m[a];
    }
}

// ---- ELL applies / aggregation: loop the per-thread bodies of ell_math.h over all threads ----
// CSC build: same result as csc.hip (count -> per-cloud scan -> fill -> sort); the fill walks the
// edges in REVERSE so that sort_column has real work to do.
void hc_csc_build(const int* nbr, int n, int k, int* tptr, int* tedge) {
    std::vector<int> cnt(n, 0);
    for (long e = 0; e < (long)n * k; ++e) cnt[nbr[e]]++;
    int run = 0;
    for (int j = 0; j < n; ++j) { tptr[j] = run; run += cnt[j]; cnt[j] = tptr[j]; }
    tptr[n] = run;
    for (long e = (long)n * k - 1; e >= 0; --e) tedge[cnt[nbr[e]]++] = (int)e;
    for (int j = 0; j < n; ++j) dcell::sort_column(tedge, tptr[j], tptr[j + 1]);
}

// forward: loop all (point, channel group) work items with the rows taken from the global arrays
#define HC_FWD(V, FN)                                                                               \
    for (long i = 0; i < n; ++i)                                                                    \
        for (int c0 = 0; c0 < C; c0 += V) FN<V>(i, c0, C, global_row(coef, nbr, i, k), k, in, ldi, out, ldo)

// op: 0 grad, 1 div, 2 divcurlnorm, 3 hodge
void hc_ell_fwd(int op, int V, const float* coef, const int* nbr, int n, int k, const float* in, int C, long ldi,
                float* out, long ldo) {
    using namespace dcell;
    if (V == 4) {
        if (op == 0) HC_FWD(4, grad_fwd);
        if (op == 1) HC_FWD(4, div_fwd);
        if (op == 2) HC_FWD(4, divcurlnorm_fwd);
        if (op == 3) HC_FWD(4, hodge_fwd);
    } else {
        if (op == 0) HC_FWD(1, grad_fwd);
        if (op == 1) HC_FWD(1, div_fwd);
        if (op == 2) HC_FWD(1, divcurlnorm_fwd);
        if (op == 3) HC_FWD(1, hodge_fwd);
    }
}

#define HC_T(V, OPINIT)                                           \
    for (long j = 0; j < n; ++j)                                  \
        for (int c0 = 0; c0 < C; c0 += V) walk_column(OPINIT, j, c0, coefT.data(), tptr, tedge, k)

// transposed: coefficients are first permuted into CSC order (as dc_csc_permute_coef does)
void hc_ell_T(int op, int V, const float* coef, const int* tptr, const int* tedge, int n, int k, const float* dy,
              int C, long ldy, float* dx, long ldx, int acc, const float* v, long ldv) {
    using namespace dcell;
    std::vector<float> coefT((size_t)n * k * 2);
    for (long t = 0; t < (long)n * k; ++t) { coefT[2 * t] = coef[2 * (long)tedge[t]]; coefT[2 * t + 1] = coef[2 * (long)tedge[t] + 1]; }
    if (V == 4) {
        if (op == 0) HC_T(4, (GradT<4>{dy, ldy, dx, ldx, acc, C}));
        if (op == 1) HC_T(4, (DivT<4>{dy, ldy, dx, ldx, acc, C}));
        if (op == 2) HC_T(4, (DivCurlNormT<4>{dy, ldy, v, ldv, dx, ldx, acc, C}));
        if (op == 3) HC_T(4, (HodgeT<4>{dy, ldy, dx, ldx, acc, C}));
    } else {
        if (op == 0) HC_T(1, (GradT<1>{dy, ldy, dx, ldx, acc, C}));
        if (op == 1) HC_T(1, (DivT<1>{dy, ldy, dx, ldx, acc, C}));
        if (op == 2) HC_T(1, (DivCurlNormT<1>{dy, ldy, v, ldv, dx, ldx, acc, C}));
        if (op == 3) HC_T(1, (HodgeT<1>{dy, ldy, dx, ldx, acc, C}));
    }
}

void hc_knn_max(int V, const int* nbr, int n, int k, const float* h, int C, long ldh, float* out, long ldo,
                unsigned char* arg) {
    using namespace dcell;
    for (long i = 0; i < n; ++i)
        for (int c0 = 0; c0 < C; c0 += V) {
            if (V == 4) knn_max_fwd<4>(i, c0, nbr + i * k, k, h, ldh, out, ldo, arg, C);
            else knn_max_fwd<1>(i, c0, nbr + i * k, k, h, ldh, out, ldo, arg, C);
        }
}

void hc_knn_max_affine(int V, const int* nbr, int n, int k, const float* h, int C, long ldh, const float* scale,
                       const float* shift, float slope, float* out, long ldo, unsigned char* arg) {
    using namespace dcell;
    for (long i = 0; i < n; ++i)
        for (int c0 = 0; c0 < C; c0 += V) {
            if (V == 4) knn_max_affine_fwd<4>(i, c0, nbr + i * k, k, h, ldh, scale, shift, slope, out, ldo, arg, C);
            else knn_max_affine_fwd<1>(i, c0, nbr + i * k, k, h, ldh, scale, shift, slope, out, ldo, arg, C);
        }
}

void hc_knn_max_affine_residual(int V, const int* nbr, int n, int k, const float* h, int C, long ldh, const float* scale,
                                const float* shift, float slope, const float* h2, long ldh2, const float* scale2,
                                const float* shift2, float slope2, float* out, long ldo, float* out2, long ldo2,
                                unsigned char* arg) {
    using namespace dcell;
    for (long i = 0; i < n; ++i)
        for (int c0 = 0; c0 < C; c0 += V) {
            if (V == 4) knn_max_affine_residual_fwd<4>(i, c0, nbr + i * k, k, h, ldh, scale, shift, slope, h2, ldh2, scale2, shift2, slope2, out, ldo, out2, ldo2, arg, C);
            else knn_max_affine_residual_fwd<1>(i, c0, nbr + i * k, k, h, ldh, scale, shift, slope, h2, ldh2, scale2, shift2, slope2, out, ldo, out2, ldo2, arg, C);
        }
}

void hc_knn_max_bwd(int V, const int* tptr, const int* tedge, int n, int k, const unsigned char* arg,
                    const float* dout, int C, long ldo, float* dh, long ldh, int acc) {
    using namespace dcell;
    for (long j = 0; j < n; ++j)
        for (int c0 = 0; c0 < C; c0 += V) {
            if (V == 4) walk_column(KnnMaxT<4>{arg, (long)C, dout, ldo, dh, ldh, acc, C}, j, c0, nullptr, tptr, tedge, k);
            else walk_column(KnnMaxT<1>{arg, (long)C, dout, ldo, dh, ldh, acc, C}, j, c0, nullptr, tptr, tedge, k);
        }
}

// sum / mean aggregation (dc_knn_sum, dc_knn_sum_backward) and the transposed gradient apply with the folded
// accumulation (dc_apply_grad_T_sum)
void hc_knn_sum(int V, const int* nbr, int n, int k, const float* h, int C, long ldh, float scale, float* out, long ldo) {
    using namespace dcell;
    for (long i = 0; i < n; ++i)
        for (int c0 = 0; c0 < C; c0 += V) {
            if (V == 4) knn_sum_fwd<4>(i, c0, nbr + i * k, k, h, ldh, scale, out, ldo);
            else knn_sum_fwd<1>(i, c0, nbr + i * k, k, h, ldh, scale, out, ldo);
        }
}

void hc_knn_sum_bwd(int V, const int* tptr, const int* tedge, int n, int k, const float* dout, int C, long ldo, float scale,
                    float* dh, long ldh, int acc) {
    using namespace dcell;
    for (long j = 0; j < n; ++j)
        for (int c0 = 0; c0 < C; c0 += V) {
            if (V == 4) walk_column(KnnSumT<4>{dout, ldo, dh, ldh, scale, acc, C}, j, c0, nullptr, tptr, tedge, k);
            else walk_column(KnnSumT<1>{dout, ldo, dh, ldh, scale, acc, C}, j, c0, nullptr, tptr, tedge, k);
        }
}

void hc_grad_T_sum(int V, const float* coef, const int* tptr, const int* tedge, int n, int k, const float* dy, int C,
                   long ldy, const float* a, long lda, const float* b, long ldb, float* out, long ldo) {
    using namespace dcell;
    std::vector<float> coefT((size_t)n * k * 2);
    for (long t = 0; t < (long)n * k; ++t) { coefT[2 * t] = coef[2 * (long)tedge[t]]; coefT[2 * t + 1] = coef[2 * (long)tedge[t] + 1]; }
    for (long j = 0; j < n; ++j)
        for (int c0 = 0; c0 < C; c0 += V) {
            if (V == 4) walk_column(GradTSum<4>{dy, ldy, a, lda, b, ldb, out, ldo, C}, j, c0, coefT.data(), tptr, tedge, k);
            else walk_column(GradTSum<1>{dy, ldy, a, lda, b, ldb, out, ldo, C}, j, c0, coefT.data(), tptr, tedge, k);
        }
}

// Philox-4x32-10 of nn_math.h (the dropout draws of the row-block kernels): known-answer vectors + the keep decision
void hc_philox(const unsigned* ctr, unsigned k0, unsigned k1, unsigned* out) {
    const dcnn::U4 r = dcnn::philox4x32_10(dcnn::U4{ctr[0], ctr[1], ctr[2], ctr[3]}, k0, k1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void hc_dropout_keep(unsigned seed, long long step, unsigned salt, int n, float p, unsigned char* keep) {
    for (int i = 0; i < n; ++i) keep[i] = dcnn::dropout_keep(seed, step, salt, (unsigned)i, p) ? 1 : 0;
}

// ---- fused BN / activation / vector non-linearity: serial loops over the formulas of nn_math.h ----
static void hc_coeffs(const std::vector<double>& s1, const std::vector<double>& s2, long R, int C, const float* gamma,
                      const float* beta, float eps, float* mean, float* invstd, float* scale, float* shift) {
    for (int c = 0; c < C; ++c) {
        const double m = s1[c] / R;
        double var = s2[c] / R - m * m;
        if (var < 0) var = 0;
        const double is = 1.0 / sqrt(var + (double)eps);
        mean[c] = (float)m; invstd[c] = (float)is;
        scale[c] = (float)(gamma[c] * is); shift[c] = (float)(beta[c] - m * gamma[c] * is);
    }
}

// y = leaky(bn(h)) + residual ; then backward for a given dy
void hc_bn_act(const float* h, long R, int C, const float* gamma, const float* beta, float eps, float slope,
               const float* residual, int training, const float* run_mean, const float* run_var, float* y,
               const float* dy, float* dh, float* dgamma, float* dbeta) {
    using namespace dcnn;
    std::vector<float> mean(C), invstd(C), scale(C), shift(C);
    if (training) {
        std::vector<double> s1(C, 0.0), s2(C, 0.0);
        for (long r = 0; r < R; ++r)
            for (int c = 0; c < C; ++c) { s1[c] += h[r * C + c]; s2[c] += (double)h[r * C + c] * h[r * C + c]; }
        hc_coeffs(s1, s2, R, C, gamma, beta, eps, mean.data(), invstd.data(), scale.data(), shift.data());
    } else {
        for (int c = 0; c < C; ++c) {
            invstd[c] = 1.f / sqrtf(run_var[c] + eps); mean[c] = run_mean[c];
            scale[c] = gamma[c] * invstd[c]; shift[c] = beta[c] - run_mean[c] * scale[c];
        }
    }
    for (long r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c)
            y[r * C + c] = act(fmaf(scale[c], h[r * C + c], shift[c]), slope) + (residual ? residual[r * C + c] : 0.f);
    std::vector<double> a(C, 0.0), b(C, 0.0);
    for (long r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            float dz, dzx;
            bn_bwd_terms(dy[r * C + c], h[r * C + c], scale[c], shift[c], mean[c], invstd[c], slope, dz, dzx);
            a[c] += dz; b[c] += dzx;
        }
    for (int c = 0; c < C; ++c) { dbeta[c] = (float)a[c]; dgamma[c] = (float)b[c]; }
    for (long r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c)
            dh[r * C + c] = bn_bwd_dh(dy[r * C + c], h[r * C + c], scale[c], shift[c], mean[c], invstd[c], slope,
                                      gamma[c] * invstd[c], (float)(a[c] / R), (float)(b[c] / R), training);
}

// vector block: in = [P|Q] (combine) or y; out; backward din for given dout
void hc_vn(const float* in, long n, int co, int combine, const float* gamma, const float* beta, float eps,
           int training, float* out, const float* dout, float* din, float* dgamma, float* dbeta) {
    using namespace dcnn;
    const long ld = combine ? 2 * co : co;
    std::vector<float> mean(co), invstd(co), scale(co), shift(co);
    auto load = [&](long i, int c, float& yu, float& yv) {
        const float pu = in[(2 * i) * ld + c], pv = in[(2 * i + 1) * ld + c];
        if (combine) vn_combine(pu, in[(2 * i) * ld + co + c], pv, in[(2 * i + 1) * ld + co + c], yu, yv);
        else { yu = pu; yv = pv; }
    };
    if (training) {
        std::vector<double> s1(co, 0.0), s2(co, 0.0);
        for (long i = 0; i < n; ++i)
            for (int c = 0; c < co; ++c) { float yu, yv; load(i, c, yu, yv); const float nn = vn_norm(yu, yv); s1[c] += nn; s2[c] += (double)nn * nn; }
        hc_coeffs(s1, s2, n, co, gamma, beta, eps, mean.data(), invstd.data(), scale.data(), shift.data());
    } else {  // bias mode: scale = 1, shift = beta (VectorNonLin without batchnorm)
        for (int c = 0; c < co; ++c) { mean[c] = 0.f; invstd[c] = 1.f; scale[c] = 1.f; shift[c] = beta[c]; }
    }
    for (long i = 0; i < n; ++i)
        for (int c = 0; c < co; ++c) {
            float yu, yv; load(i, c, yu, yv);
            const float s = vn_scale(vn_norm(yu, yv), scale[c], shift[c]);
            out[(2 * i) * co + c] = yu * s; out[(2 * i + 1) * co + c] = yv * s;
        }
    std::vector<double> a(co, 0.0), b(co, 0.0);
    for (long i = 0; i < n; ++i)
        for (int c = 0; c < co; ++c) {
            float yu, yv, dz, dzn; load(i, c, yu, yv);
            vn_bwd_terms(yu, yv, dout[(2 * i) * co + c], dout[(2 * i + 1) * co + c], scale[c], shift[c], mean[c], invstd[c], dz, dzn);
            a[c] += dz; b[c] += dzn;
        }
    for (int c = 0; c < co; ++c) { dbeta[c] = (float)a[c]; dgamma[c] = (float)b[c]; }
    for (long i = 0; i < n; ++i)
        for (int c = 0; c < co; ++c) {
            float yu, yv, gu, gv; load(i, c, yu, yv);
            vn_bwd_dy(yu, yv, dout[(2 * i) * co + c], dout[(2 * i + 1) * co + c], scale[c], shift[c], mean[c], invstd[c],
                      (training ? gamma[c] : 1.f) * invstd[c], (float)(a[c] / n), (float)(b[c] / n), training, gu, gv);
            din[(2 * i) * ld + c] = gu; din[(2 * i + 1) * ld + c] = gv;
            if (combine) { din[(2 * i) * ld + co + c] = gv; din[(2 * i + 1) * ld + co + c] = -gu; }
        }
}

// ---- layer-0 edge MLP without the [E,C] tensor (edge_math.h): forward + backward, serial ----------
void hc_edge(const float* y, const int* nbr, const int* tptr, const int* tedge, int n, int k, int C,
             const float* gamma, const float* beta, float eps, float slope, int training, const float* run_mean,
             const float* run_var, float* out, unsigned char* arg, const float* dout, float* dy, float* dgamma,
             float* dbeta) {
    using namespace dcedge;
    std::vector<float> amax((size_t)n * C), amin((size_t)n * C), s1((size_t)n * C), dzs((size_t)n * C);
    std::vector<unsigned char> amx((size_t)n * C), amn((size_t)n * C);
    std::vector<double> q1(C, 0.0), q2(C, 0.0);
    for (long i = 0; i < n; ++i)
        for (int c = 0; c < C; ++c) {
            double t[2][1];
            edge_gather<1>(i, c, y, C, nbr, k, amax.data(), amin.data(), amx.data(), amn.data(), s1.data(), C, C, t);
            q1[c] += t[0][0]; q2[c] += t[1][0];
        }
    std::vector<float> mean(C), invstd(C), scale(C), shift(C);
    const long E = (long)n * k;
    if (training) hc_coeffs(q1, q2, E, C, gamma, beta, eps, mean.data(), invstd.data(), scale.data(), shift.data());
    else
        for (int c = 0; c < C; ++c) {
            invstd[c] = 1.f / sqrtf(run_var[c] + eps); mean[c] = run_mean[c];
            scale[c] = gamma[c] * invstd[c]; shift[c] = beta[c] - run_mean[c] * scale[c];
        }
    for (long i = 0; i < n; ++i)
        for (int c = 0; c < C; ++c) {
            out[i * C + c] = dcnn::act(fmaf(scale[c], pick(scale[c], amax[i * C + c], amin[i * C + c]), shift[c]), slope);
            arg[i * C + c] = scale[c] >= 0.f ? amx[i * C + c] : amn[i * C + c];
        }
    std::vector<double> a(C, 0.0), b(C, 0.0);
    for (long i = 0; i < n; ++i)
        for (int c = 0; c < C; ++c) {
            float dz, dza;
            edge_bwd_terms(dout[i * C + c], amax[i * C + c], amin[i * C + c], scale[c], shift[c], mean[c], invstd[c], slope, dz, dza);
            dzs[i * C + c] = dz; a[c] += dz; b[c] += dza;
        }
    std::vector<float> m1(C), m2(C);
    for (int c = 0; c < C; ++c) { dbeta[c] = (float)a[c]; dgamma[c] = (float)b[c]; m1[c] = (float)(a[c] / E); m2[c] = (float)(b[c] / E); }
    for (long t = 0; t < (long)n * C; ++t)
        edge_bwd_point<1>(t, C, tptr, tedge, k, y, C, dzs.data(), s1.data(), C, amx.data(), amn.data(), C, scale.data(),
                          mean.data(), invstd.data(), m1.data(), m2.data(), training, dy, C);
}

// training loss: mean over rows of ce_row, gradient per row (loss.hip runs the same body per thread)
double hc_ce_loss(const float* x, const long* label, long R, int C, float eps, float* dx) {
    double s = 0.0;
    for (long r = 0; r < R; ++r) s += (double)dcloss::ce_row(x + r * C, C, label[r], eps, 1.f / (float)R, dx + r * C);
    return s / (double)R;
}
}
