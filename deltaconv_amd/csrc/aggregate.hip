// Max aggregation over the k nearest neighbours + its backward.
// Replaces torch_scatter.scatter(reduce='max') over a gathered [E,C] tensor
// (/root/reference/deltaconv/nn/deltaconv.py:52,54): the [E,C] tensor is never materialised;
// the winning slot is kept as one byte per (point, channel).
// HBM-bound: 4C*Nt in + 4C*Nt out + C*Nt arg + 4E ids per layer (reference: 4*C*E gathered).
#include "common.h"
#include "ell_math.h"

namespace {
using namespace dcell;
constexpr int TPB = 256;

template <int V>
__global__ __launch_bounds__(TPB) void knn_max_fwd_kernel(long total, int groups, int remap, const int* nbr, int k,
                                                          const float* h, long ldh, float* out, long ldo,
                                                          unsigned char* arg, long lda) {
    const long t = dc_xcd_block(remap) * TPB + threadIdx.x;
    if (t >= total) return;
    knn_max_fwd<V>(t, groups, nbr, k, h, ldh, out, ldo, arg, lda);
}

template <int V>
__global__ __launch_bounds__(TPB) void knn_max_bwd_kernel(long total, int groups, int remap, const int* tptr, const int* tedge,
                                                          int k, const unsigned char* arg, long lda,
                                                          const float* dout, long ldo, float* dh, long ldh, int acc) {
    const long t = dc_xcd_block(remap) * TPB + threadIdx.x;
    if (t >= total) return;
    knn_max_bwd<V>(t, groups, tptr, tedge, k, arg, lda, dout, ldo, dh, ldh, acc);
}

inline bool vec_ok(int C, long a, long b, const void* p, const void* q) {
    return C % 4 == 0 && a % 4 == 0 && b % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(q) & 15) == 0;
}
}  // namespace

DC_EXPORT int dc_knn_max(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh, float* out,
                         int64_t ldo, uint8_t* arg, void* stream) {
    DC_REQUIRE(nbr && h && out && arg, "dc_knn_max: null pointer");
    DC_REQUIRE(n >= 0 && k >= 1 && k <= 255 && C >= 0, "dc_knn_max: bad size (k <= 255)");
    DC_REQUIRE(ldh >= C && ldo >= C, "dc_knn_max: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec_ok(C, ldh, ldo, h, out)) {
        const long total = (long)n * (C / 4);
        hipLaunchKernelGGL(knn_max_fwd_kernel<4>, dim3(dc_cdiv(total, TPB)), dim3(TPB), 0, s, total, C / 4, dc_option(DC_OPT_XCD_REMAP), nbr, k, h,
                           (long)ldh, out, (long)ldo, arg, (long)C);
    } else {
        const long total = (long)n * C;
        hipLaunchKernelGGL(knn_max_fwd_kernel<1>, dim3(dc_cdiv(total, TPB)), dim3(TPB), 0, s, total, C, dc_option(DC_OPT_XCD_REMAP), nbr, k, h,
                           (long)ldh, out, (long)ldo, arg, (long)C);
    }
    DC_CHECK_LAUNCH("dc_knn_max");
    return DC_OK;
}

DC_EXPORT int dc_knn_max_backward(const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, const uint8_t* arg,
                                  const float* dout, int32_t C, int64_t ldo, float* dh, int64_t ldh,
                                  int32_t accumulate, void* stream) {
    DC_REQUIRE(tptr && tedge && arg && dout && dh, "dc_knn_max_backward: null pointer");
    DC_REQUIRE(n >= 0 && k >= 1 && k <= 255 && C >= 0, "dc_knn_max_backward: bad size (k <= 255)");
    DC_REQUIRE(ldo >= C && ldh >= C, "dc_knn_max_backward: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec_ok(C, ldo, ldh, dout, dh)) {
        const long total = (long)n * (C / 4);
        hipLaunchKernelGGL(knn_max_bwd_kernel<4>, dim3(dc_cdiv(total, TPB)), dim3(TPB), 0, s, total, C / 4, dc_option(DC_OPT_XCD_REMAP), tptr,
                           tedge, k, arg, (long)C, dout, (long)ldo, dh, (long)ldh, accumulate);
    } else {
        const long total = (long)n * C;
        hipLaunchKernelGGL(knn_max_bwd_kernel<1>, dim3(dc_cdiv(total, TPB)), dim3(TPB), 0, s, total, C, dc_option(DC_OPT_XCD_REMAP), tptr, tedge, k,
                           arg, (long)C, dout, (long)ldo, dh, (long)ldh, accumulate);
    }
    DC_CHECK_LAUNCH("dc_knn_max_backward");
    return DC_OK;
}
