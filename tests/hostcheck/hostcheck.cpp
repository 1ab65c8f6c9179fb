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

#define HC_LOOP(V, CALL)                                    \
    do {                                                    \
        const int groups = C / V;                           \
        for (long t = 0; t < (long)n * groups; ++t) { CALL; } \
    } while (0)

// op: 0 grad, 1 div, 2 divcurlnorm, 3 hodge
void hc_ell_fwd(int op, int V, const float* coef, const int* nbr, int n, int k, const float* in, int C, long ldi,
                float* out, long ldo) {
    using namespace dcell;
    if (V == 4) {
        if (op == 0) HC_LOOP(4, grad_fwd<4>(t, groups, coef, nbr, k, in, ldi, out, ldo));
        if (op == 1) HC_LOOP(4, div_fwd<4>(t, groups, coef, nbr, k, in, ldi, out, ldo));
        if (op == 2) HC_LOOP(4, divcurlnorm_fwd<4>(t, groups, coef, nbr, k, in, ldi, out, ldo));
        if (op == 3) HC_LOOP(4, hodge_fwd<4>(t, groups, coef, nbr, k, in, ldi, out, ldo));
    } else {
        if (op == 0) HC_LOOP(1, grad_fwd<1>(t, groups, coef, nbr, k, in, ldi, out, ldo));
        if (op == 1) HC_LOOP(1, div_fwd<1>(t, groups, coef, nbr, k, in, ldi, out, ldo));
        if (op == 2) HC_LOOP(1, divcurlnorm_fwd<1>(t, groups, coef, nbr, k, in, ldi, out, ldo));
        if (op == 3) HC_LOOP(1, hodge_fwd<1>(t, groups, coef, nbr, k, in, ldi, out, ldo));
    }
}

void hc_ell_T(int op, int V, const float* coef, const int* tptr, const int* tedge, int n, int k, const float* dy,
              int C, long ldy, float* dx, long ldx, int acc, const float* v, long ldv) {
    using namespace dcell;
    if (V == 4) {
        if (op == 0) HC_LOOP(4, grad_T<4>(t, groups, coef, tptr, tedge, k, dy, ldy, dx, ldx, acc));
        if (op == 1) HC_LOOP(4, div_T<4>(t, groups, coef, tptr, tedge, k, dy, ldy, dx, ldx, acc));
        if (op == 2) HC_LOOP(4, divcurlnorm_T<4>(t, groups, coef, tptr, tedge, k, dy, ldy, v, ldv, dx, ldx, acc));
        if (op == 3) HC_LOOP(4, hodge_T<4>(t, groups, coef, tptr, tedge, k, dy, ldy, dx, ldx, acc));
    } else {
        if (op == 0) HC_LOOP(1, grad_T<1>(t, groups, coef, tptr, tedge, k, dy, ldy, dx, ldx, acc));
        if (op == 1) HC_LOOP(1, div_T<1>(t, groups, coef, tptr, tedge, k, dy, ldy, dx, ldx, acc));
        if (op == 2) HC_LOOP(1, divcurlnorm_T<1>(t, groups, coef, tptr, tedge, k, dy, ldy, v, ldv, dx, ldx, acc));
        if (op == 3) HC_LOOP(1, hodge_T<1>(t, groups, coef, tptr, tedge, k, dy, ldy, dx, ldx, acc));
    }
}

void hc_knn_max(int V, const int* nbr, int n, int k, const float* h, int C, long ldh, float* out, long ldo,
                unsigned char* arg) {
    using namespace dcell;
    if (V == 4) HC_LOOP(4, knn_max_fwd<4>(t, groups, nbr, k, h, ldh, out, ldo, arg, C));
    else HC_LOOP(1, knn_max_fwd<1>(t, groups, nbr, k, h, ldh, out, ldo, arg, C));
}

void hc_knn_max_bwd(int V, const int* tptr, const int* tedge, int n, int k, const unsigned char* arg,
                    const float* dout, int C, long ldo, float* dh, long ldh, int acc) {
    using namespace dcell;
    if (V == 4) HC_LOOP(4, knn_max_bwd<4>(t, groups, tptr, tedge, k, arg, C, dout, ldo, dh, ldh, acc));
    else HC_LOOP(1, knn_max_bwd<1>(t, groups, tptr, tedge, k, arg, C, dout, ldo, dh, ldh, acc));
}
}
