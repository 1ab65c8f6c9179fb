// Per-point / per-edge arithmetic of the geometry kernels, written once and shared by
//   * the HIP kernels (basis.hip, mls.hip) -- the product, and
//   * tests/hostcheck/hostcheck.cpp -- a g++ build of the SAME functions, looped over points on
//     the CPU so the math can be checked against the oracle in the GPU-less build container.
// Nothing here touches wave intrinsics or LDS.
//
// Reference being restated: /root/reference/deltaconv/geometry/grad_div_mls.py (line numbers in
// the comments).  fp32 in / fp32 out; the moving-least-squares interior runs in fp64 because the
// 6x6 normal equations are ill-conditioned for small lambda (cond ~ 1/lambda) and the work is
// ~4 kflop per point -- free on a GPU whose apply kernels are bandwidth-bound.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define DC_HD __host__ __device__ __forceinline__
#else
#define DC_HD inline
#endif

namespace dcmath {

constexpr float BASIS_EPS = 1e-5f;  // grad_div_mls.py:7 (EPS)

struct V3 {
    double x, y, z;
};
DC_HD V3 ld3(const float* p) { return V3{(double)p[0], (double)p[1], (double)p[2]}; }
DC_HD double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DC_HD V3 sub(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DC_HD V3 axpy(double s, const V3& a, const V3& b) { return V3{s * a.x + b.x, s * a.y + b.y, s * a.z + b.z}; }

// ---- build_tangent_basis (grad_div_mls.py:50-69), fp32 like the reference ------------------
DC_HD void tangent_basis_point(const float* n, float* xb, float* yb) {
    const float nx = n[0], ny = n[1], nz = n[2];
    const bool alt = fabsf(nx) > 0.9f;  // |n . (1,0,0)| > 0.9 -> use (0,1,0)      (:58-60)
    const float tx = alt ? 0.f : 1.f, ty = alt ? 1.f : 0.f, tz = 0.f;
    float x0 = ty * nz - tz * ny, x1 = tz * nx - tx * nz, x2 = tx * ny - ty * nx;  // t x n (:63)
    float inv = 1.f / fmaxf(sqrtf(x0 * x0 + x1 * x1 + x2 * x2), BASIS_EPS);
    x0 *= inv; x1 *= inv; x2 *= inv;
    float y0 = ny * x2 - nz * x1, y1 = nz * x0 - nx * x2, y2 = nx * x1 - ny * x0;  // n x x (:67)
    inv = 1.f / fmaxf(sqrtf(y0 * y0 + y1 * y1 + y2 * y2), BASIS_EPS);
    xb[0] = x0; xb[1] = x1; xb[2] = x2;
    yb[0] = y0 * inv; yb[1] = y1 * inv; yb[2] = y2 * inv;
}

// ---- estimate_basis (grad_div_mls.py:10-47) ---------------------------------------------------
// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (a -> diagonal, columns of v = vectors).
DC_HD void jacobi3(double a[3][3], double v[3][3]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) v[r][c] = (r == c) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-17 * diag || off < 1e-300) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            const double apq = a[p][q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
            const double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int r = 0; r < 3; ++r) {  // A <- A R
                const double arp = a[r][p], arq = a[r][q];
                a[r][p] = c * arp - s * arq;
                a[r][q] = s * arp + c * arq;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {  // A <- R^T A
                const double apr = a[p][r], aqr = a[q][r];
                a[p][r] = c * apr - s * aqr;
                a[q][r] = s * apr + c * aqr;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {  // V <- V R
                const double vrp = v[r][p], vrq = v[r][q];
                v[r][p] = c * vrp - s * vrq;
                v[r][q] = s * vrp + c * vrq;
            }
        }
    }
}

DC_HD void estimate_basis_point(const float* pos, const int* nbr_i, int i, int k, const float* orient,
                                float* normal, float* xb, float* yb) {
    const V3 p = ld3(pos + 3 * (long)i);
    double a[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int s = 0; s < k; ++s) {  // covariance of neighbour offsets = local_pos local_pos^T (:26-29)
        const V3 d = sub(ld3(pos + 3 * (long)nbr_i[s]), p);
        a[0][0] += d.x * d.x; a[0][1] += d.x * d.y; a[0][2] += d.x * d.z;
        a[1][1] += d.y * d.y; a[1][2] += d.y * d.z; a[2][2] += d.z * d.z;
    }
    a[1][0] = a[0][1]; a[2][0] = a[0][2]; a[2][1] = a[1][2];
    double v[3][3];
    jacobi3(a, v);
    // U[:,0] <-> largest, U[:,2] <-> smallest singular value (:32,39)
    int lo = 0, hi = 0;
    for (int c = 1; c < 3; ++c) {
        if (a[c][c] < a[lo][lo]) lo = c;
        if (a[c][c] > a[hi][hi]) hi = c;
    }
    if (lo == hi) { lo = 2; hi = 0; }  // fully degenerate neighbourhood
    double nx = (lo == 0) ? v[0][0] : (lo == 1) ? v[0][1] : v[0][2];
    double ny = (lo == 0) ? v[1][0] : (lo == 1) ? v[1][1] : v[1][2];
    double nz = (lo == 0) ? v[2][0] : (lo == 1) ? v[2][1] : v[2][2];
    double x0 = (hi == 0) ? v[0][0] : (hi == 1) ? v[0][1] : v[0][2];
    double x1 = (hi == 0) ? v[1][0] : (hi == 1) ? v[1][1] : v[1][2];
    double x2 = (hi == 0) ? v[2][0] : (hi == 1) ? v[2][1] : v[2][2];
    double inv = 1.0 / fmax(sqrt(nx * nx + ny * ny + nz * nz), (double)BASIS_EPS);
    nx *= inv; ny *= inv; nz *= inv;
    if (orient) {  // flip against the rough orientation (:35-36)
        const double d = nx * orient[3 * (long)i] + ny * orient[3 * (long)i + 1] + nz * orient[3 * (long)i + 2];
        if (d < 0) { nx = -nx; ny = -ny; nz = -nz; }
    }
    inv = 1.0 / fmax(sqrt(x0 * x0 + x1 * x1 + x2 * x2), (double)BASIS_EPS);
    x0 *= inv; x1 *= inv; x2 *= inv;
    // the sign of a singular vector is a backend convention; ours: largest-|component| positive
    const double ax = fabs(x0), ay = fabs(x1), az = fabs(x2);
    const double lead = (ax >= ay && ax >= az) ? x0 : (ay >= az ? x1 : x2);
    if (lead < 0) { x0 = -x0; x1 = -x1; x2 = -x2; }
    double y0 = ny * x2 - nz * x1, y1 = nz * x0 - nx * x2, y2 = nx * x1 - ny * x0;  // y = n x x (:44)
    inv = 1.0 / fmax(sqrt(y0 * y0 + y1 * y1 + y2 * y2), (double)BASIS_EPS);
    normal[0] = (float)nx; normal[1] = (float)ny; normal[2] = (float)nz;
    xb[0] = (float)x0; xb[1] = (float)x1; xb[2] = (float)x2;
    yb[0] = (float)(y0 * inv); yb[1] = (float)(y1 * inv); yb[2] = (float)(y2 * inv);
}

// ---- moving least squares (grad_div_mls.py:72-277) --------------------------------------------
struct Frame {
    V3 p, n, x, y;
};
DC_HD Frame load_frame(const float* pos, const float* normal, const float* xb, const float* yb, long i) {
    return Frame{ld3(pos + 3 * i), ld3(normal + 3 * i), ld3(xb + 3 * i), ld3(yb + 3 * i)};
}

struct EdgeGeom {
    double u, v, dist2, height;
};
// coords_projected (:72-97) + euclidean distance (:233) + height over the tangent plane (:163)
DC_HD EdgeGeom edge_geom(const Frame& f, const V3& pj) {
    const V3 d = sub(pj, f.p);
    const double h = dot(d, f.n);
    const V3 t = axpy(-h, f.n, d);
    return EdgeGeom{dot(t, f.x), dot(t, f.y), dot(d, d), h};
}

// Sum over the k neighbours of the euclidean edge length (for gaussian_weights' average, :112).
DC_HD double point_dist_sum(const float* pos, const int* nbr_i, long i, int k) {
    const V3 p = ld3(pos + 3 * i);
    double s = 0;
    for (int e = 0; e < k; ++e) {
        const V3 d = sub(ld3(pos + 3 * (long)nbr_i[e]), p);
        s += sqrt(dot(d, d));
    }
    return s;
}

// polynomial basis row of one neighbour: triu of [1,u,v] (x) [1,u,v] (:133-137)
DC_HD void poly_basis(double u, double v, double b[6]) {
    b[0] = 1.0; b[1] = u; b[2] = v; b[3] = u * u; b[4] = u * v; b[5] = v * v;
}

// Cholesky M = L L^T of the symmetric positive definite 6x6 held in the UPPER triangle (diagonal included);
// L lands in the strict lower triangle and on the diagonal (the upper triangle keeps M's off-diagonal entries).
DC_HD void chol6_factor(double M[6][6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = M[j][j];
#pragma unroll
        for (int p = 0; p < j; ++p) s -= M[j][p] * M[j][p];
        const double ljj = sqrt(fmax(s, 1e-300));
        const double inv_l = 1.0 / ljj;
#pragma unroll
        for (int r = j + 1; r < 6; ++r) {
            double t = M[j][r];
#pragma unroll
            for (int p = 0; p < j; ++p) t -= M[r][p] * M[j][p];
            M[r][j] = t * inv_l;
        }
        M[j][j] = ljj;  // diagonal now holds L (M's diagonal is not needed any more)
    }
}

// x <- (L L^T)^-1 x with the factor of chol6_factor: forward substitution L y = x, back substitution L^T x = y
DC_HD void chol6_solve(const double M[6][6], double z[6]) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double t = z[r];
#pragma unroll
        for (int p = 0; p < r; ++p) t -= M[r][p] * z[p];
        z[r] = t * (1.0 / M[r][r]);
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double t = z[r];
#pragma unroll
        for (int p = r + 1; p < 6; ++p) t -= M[p][r] * z[p];
        z[r] = t * (1.0 / M[r][r]);
    }
}

// gaussian_weights for the k edges of one point (:113-114): w = exp(-d^2 / (h avg)^2), normalised by the row sum
// clamped at EPS.  `avg` = the cloud's mean edge length (:112).
DC_HD void gaussian_weights_point(const float* dist_i, int k, double avg_dist, double kernel_width, float* w_out) {
    const double inv_h2 = 1.0 / ((kernel_width * avg_dist) * (kernel_width * avg_dist));
    double wsum = 0;
    for (int e = 0; e < k; ++e) wsum += exp(-((double)dist_i[e] * (double)dist_i[e]) * inv_h2);
    const double inv_w = 1.0 / fmax(wsum, (double)BASIS_EPS);
    for (int e = 0; e < k; ++e) w_out[e] = (float)(exp(-((double)dist_i[e] * (double)dist_i[e]) * inv_h2) * inv_w);
}

// weighted_least_squares for one point (:119-144): wls[e,:] = w_e (B^T W B + lambda I)^-1 b_e -- the k columns of
// (M^-1 B^T W), transposed.  Same normal equations, same factorisation as mls_fit_point (which keeps only rows 1, 2
// of M^-1 and the product with the heights).
DC_HD void wls_point(const float* coords_i, const float* w_i, int k, double lambda, float* wls_out) {
    double M[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) M[a][b] = 0;
    for (int e = 0; e < k; ++e) {
        double b[6];
        poly_basis((double)coords_i[2 * e], (double)coords_i[2 * e + 1], b);
        const double w = (double)w_i[e];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double wb = w * b[a];
#pragma unroll
            for (int c = a; c < 6; ++c) M[a][c] += wb * b[c];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) M[a][a] += lambda;
    chol6_factor(M);
    for (int e = 0; e < k; ++e) {
        double z[6];
        poly_basis((double)coords_i[2 * e], (double)coords_i[2 * e + 1], z);
        chol6_solve(M, z);
        const double w = (double)w_i[e];
#pragma unroll
        for (int a = 0; a < 6; ++a) wls_out[6 * e + a] = (float)(w * z[a]);
    }
}

// fit_vector_mapping for one edge (:168-194, eq. 15 of the supplement): m = g^-1 T, the map from the frame at p_j to
// the frame of p_i pushed forward along the fitted height field (coef = its six coefficients at p_i) to (u, v).
DC_HD void vector_map(const Frame& fi, const double* coef, double u, double v, const V3& xj, const V3& yj,
                      double m[4]) {
    const double hu = coef[1] + 2 * coef[3] * u + coef[4] * v;      // (:168)
    const double hv = coef[2] + coef[4] * u + 2 * coef[5] * v;      // (:169)
    const V3 gam_u = axpy(hu, fi.n, fi.x);                          // (:173)
    const V3 gam_v = axpy(hv, fi.n, fi.y);                          // (:175)
    const double det = 1 + hu * hu + hv * hv;                       // (:179)
    const double E = 1 + hu * hu, F = hu * hv, G = 1 + hv * hv;     // (:180)
    const double t00 = dot(gam_u, xj), t01 = dot(gam_u, yj), t10 = dot(gam_v, xj), t11 = dot(gam_v, yj);
    const double inv_det = 1.0 / det;
    m[0] = (G * t00 - F * t10) * inv_det; m[1] = (G * t01 - F * t11) * inv_det;  // (:181-194)
    m[2] = (-F * t00 + E * t10) * inv_det; m[3] = (-F * t01 + E * t11) * inv_det;
}

// Weighted least squares fit at one point (:100-152,163-165,253-259).
//   g_out[k][2]  un-normalised gradient rows  (wls[e,1], wls[e,2])
//   coef[6]      quadratic height-field coefficients c = sum_e wls[e,:] * height_e
//   returns      ||(sum_e |g_u|, sum_e |g_v|)||_2  (this row's contribution to the infinity norm)
//   SHAPE: the surface fit uses its own regulariser lambda_shape (build_grad_div(shape_regularizer=...), :241-244,
//   266-267): a second factorisation of B^T W B + lambda_shape I for the height-field coefficients only
template <bool SHAPE = false>
DC_HD float mls_fit_point(const float* pos, const float* normal, const float* xb, const float* yb,
                          const int* nbr_i, long i, int k, double avg_dist, double kernel_width, double lambda,
                          float* g_out, double* coef, double lambda_shape = 0.0) {
    const Frame f = load_frame(pos, normal, xb, yb, i);
    const double inv_h2 = 1.0 / ((kernel_width * avg_dist) * (kernel_width * avg_dist));
    double M[6][6];
    double rhs[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        rhs[a] = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) M[a][b] = 0;
    }
    double wsum = 0;
    for (int e = 0; e < k; ++e) {
        const EdgeGeom g = edge_geom(f, ld3(pos + 3 * (long)nbr_i[e]));
        const double w = exp(-g.dist2 * inv_h2);  // (:113)
        const double b[6] = {1.0, g.u, g.v, g.u * g.u, g.u * g.v, g.v * g.v};  // (:133-137)
        wsum += w;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double wb = w * b[a];
            rhs[a] += wb * g.height;
#pragma unroll
            for (int c = a; c < 6; ++c) M[a][c] += wb * b[c];
        }
    }
    const double inv_w = 1.0 / fmax(wsum, (double)BASIS_EPS);  // row-normalised weights (:114)
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        rhs[a] *= inv_w;
#pragma unroll
        for (int c = a; c < 6; ++c) M[a][c] *= inv_w;
    }
    if (SHAPE) {   // surface coefficients from (B^T W B + lambda_shape I) c = B^T W f   (:146-150)
        double S[6][6], c[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            c[a] = rhs[a];
#pragma unroll
            for (int b = a; b < 6; ++b) S[a][b] = M[a][b];
            S[a][a] += lambda_shape;
        }
        chol6_factor(S);
        chol6_solve(S, c);
#pragma unroll
        for (int a = 0; a < 6; ++a) coef[a] = c[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) M[a][a] += lambda;  // B^T W B + lambda I (:141-143)
    chol6_factor(M);
    // three right-hand sides: e1, e2 (rows 1,2 of M^-1 -> gradient) and rhs (surface coefficients)
    double z1[6] = {0, 1, 0, 0, 0, 0}, z2[6] = {0, 0, 1, 0, 0, 0};
    chol6_solve(M, z1);
    chol6_solve(M, z2);
    chol6_solve(M, rhs);
    if (!SHAPE) {
#pragma unroll
        for (int a = 0; a < 6; ++a) coef[a] = rhs[a];
    }
    // second sweep over the neighbours: gradient rows wls[e,1], wls[e,2] = w_e * (z . b_e)
    double au = 0, av = 0;
    for (int e = 0; e < k; ++e) {
        const EdgeGeom g = edge_geom(f, ld3(pos + 3 * (long)nbr_i[e]));
        const double w = exp(-g.dist2 * inv_h2) * inv_w;
        const double b[6] = {1.0, g.u, g.v, g.u * g.u, g.u * g.v, g.v * g.v};
        double gu = 0, gv = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            gu += z1[a] * b[a];
            gv += z2[a] * b[a];
        }
        gu *= w; gv *= w;
        g_out[2 * e] = (float)gu;
        g_out[2 * e + 1] = (float)gv;
        au += fabs((double)(float)gu);
        av += fabs((double)(float)gv);
    }
    return (float)sqrt(au * au + av * av);  // (:259)
}

// Per edge: normalise the gradient row by the cloud's infinity norm (:260) and contract it with
// the vector mapping of fit_vector_mapping (:155-194, eq. 15 of the supplement) -> divergence row.
DC_HD void mls_div_edge(const Frame& fi, const double* coef, const V3& pj, const V3& xj, const V3& yj,
                        float inf_norm, float* g /*in: raw, out: normalised*/, float* d_out) {
    double gu = g[0], gv = g[1];
    if (inf_norm > 1e-5f) {
        gu = (double)(g[0] / inf_norm);
        gv = (double)(g[1] / inf_norm);
    }
    const EdgeGeom e = edge_geom(fi, pj);
    double m[4];
    vector_map(fi, coef, e.u, e.v, xj, yj, m);
    const double m00 = m[0], m01 = m[1], m10 = m[2], m11 = m[3];
    g[0] = (float)gu;
    g[1] = (float)gv;
    d_out[0] = (float)(gu * m00 + gv * m10);  // [g_u g_v] . map (:271-272)
    d_out[1] = (float)(gu * m01 + gv * m11);
}

}  // namespace dcmath
