// Tangent frames per point.
//   dc_tangent_basis  <- build_tangent_basis  (/root/reference/deltaconv/geometry/grad_div_mls.py:50-69)
//   dc_estimate_basis <- estimate_basis       (/root/reference/deltaconv/geometry/grad_div_mls.py:10-47)
// One thread per point; the arithmetic lives in point_math.h (shared with the CPU host-check build).
// 36 B in / 24-36 B out per point: launch-latency sized, nowhere near any roofline.
#include "common.h"
#include "point_math.h"

namespace {

__global__ void tangent_basis_kernel(const float* __restrict__ normal, int n, float* __restrict__ xb,
                                     float* __restrict__ yb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dcmath::tangent_basis_point(normal + 3 * (size_t)i, xb + 3 * (size_t)i, yb + 3 * (size_t)i);
}

__global__ void estimate_basis_kernel(const float* __restrict__ pos, const int* __restrict__ nbr, int n, int k,
                                      const float* __restrict__ orient, float* __restrict__ normal,
                                      float* __restrict__ xb, float* __restrict__ yb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dcmath::estimate_basis_point(pos, nbr + (size_t)i * k, i, k, orient, normal + 3 * (size_t)i,
                                 xb + 3 * (size_t)i, yb + 3 * (size_t)i);
}

}  // namespace

DC_EXPORT int dc_tangent_basis(const float* normal, int32_t n, float* x_basis, float* y_basis, void* stream) {
    DC_REQUIRE(normal && x_basis && y_basis, "dc_tangent_basis: null pointer");
    DC_REQUIRE(n >= 0, "dc_tangent_basis: negative size");
    if (n == 0) return DC_OK;
    hipLaunchKernelGGL(tangent_basis_kernel, dim3(dc_cdiv(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       normal, n, x_basis, y_basis);
    DC_CHECK_LAUNCH("dc_tangent_basis");
    return DC_OK;
}

DC_EXPORT int dc_estimate_basis(const float* pos, const int32_t* nbr, int32_t n, int32_t k, const float* orientation,
                                float* normal, float* x_basis, float* y_basis, void* stream) {
    DC_REQUIRE(pos && nbr && normal && x_basis && y_basis, "dc_estimate_basis: null pointer");
    DC_REQUIRE(n >= 0 && k >= 1, "dc_estimate_basis: bad size");
    if (n == 0) return DC_OK;
    hipLaunchKernelGGL(estimate_basis_kernel, dim3(dc_cdiv(n, 128)), dim3(128), 0, static_cast<hipStream_t>(stream),
                       pos, nbr, n, k, orientation, normal, x_basis, y_basis);
    DC_CHECK_LAUNCH("dc_estimate_basis");
    return DC_OK;
}
