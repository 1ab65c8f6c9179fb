// Max aggregation over the k nearest neighbours + its backward.
// Replaces torch_scatter.scatter(reduce='max') over a gathered [E,C] tensor
// (/root/reference/deltaconv/nn/deltaconv.py:52,54): the [E,C] tensor is never materialised;
// the winning slot is kept as one byte per (point, channel).
// HBM-bound: 4C*Nt in + 4C*Nt out + C*Nt arg + 4E ids per layer (reference: 4*C*E gathered).
#include <initializer_list>
#include "common.h"
#include "ell_stage.h"

namespace {
using namespace dcell;
using namespace dcstage;

template <int V>
struct KnnMaxF {
    const float* h; long ldh; float* out; long ldo; unsigned char* arg; long lda;
    __device__ void operator()(long i, int c0, Row r, int k) const {
        knn_max_fwd<V>(i, c0, r.ids, k, h, ldh, out, ldo, arg, lda);
    }
};
template <int V>
struct KnnMaxAffineF {
    const float* h; long ldh; const float *scale, *shift; float slope; float* out; long ldo; unsigned char* arg; long lda;
    __device__ void operator()(long i, int c0, Row r, int k) const {
        knn_max_affine_fwd<V>(i, c0, r.ids, k, h, ldh, scale, shift, slope, out, ldo, arg, lda);
    }
};
}  // namespace

DC_EXPORT int dc_knn_max(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh, float* out,
                         int64_t ldo, uint8_t* arg, void* stream) {
    DC_REQUIRE(nbr && h && out && arg, "dc_knn_max: null pointer");
    DC_REQUIRE(n >= 0 && k >= 1 && k <= 255 && C >= 0, "dc_knn_max: bad size (k <= 255)");
    DC_REQUIRE(ldh >= C && ldo >= C, "dc_knn_max: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pick_v(C, {(long)ldh, (long)ldo}, {h, out}) == 4)
        launch_fwd<4>(n, C, nullptr, nbr, k, KnnMaxF<4>{h, (long)ldh, out, (long)ldo, arg, (long)C}, s);
    else
        launch_fwd<1>(n, C, nullptr, nbr, k, KnnMaxF<1>{h, (long)ldh, out, (long)ldo, arg, (long)C}, s);
    DC_CHECK_LAUNCH("dc_knn_max");
    return DC_OK;
}

// out[i,c] = max_s leaky_slope(scale_c * h[nbr[i,s],c] + shift_c): dc_bn_act + dc_knn_max in one gather pass.
DC_EXPORT int dc_knn_max_affine(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh,
                                const float* scale, const float* shift, float slope, float* out, int64_t ldo,
                                uint8_t* arg, void* stream) {
    DC_REQUIRE(nbr && h && scale && shift && out && arg, "dc_knn_max_affine: null pointer");
    DC_REQUIRE(n >= 0 && k >= 1 && k <= 255 && C >= 0, "dc_knn_max_affine: bad size (k <= 255)");
    DC_REQUIRE(ldh >= C && ldo >= C, "dc_knn_max_affine: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pick_v(C, {(long)ldh, (long)ldo}, {h, out}) == 4)
        launch_fwd<4>(n, C, nullptr, nbr, k,
                      KnnMaxAffineF<4>{h, (long)ldh, scale, shift, slope, out, (long)ldo, arg, (long)C}, s);
    else
        launch_fwd<1>(n, C, nullptr, nbr, k,
                      KnnMaxAffineF<1>{h, (long)ldh, scale, shift, slope, out, (long)ldo, arg, (long)C}, s);
    DC_CHECK_LAUNCH("dc_knn_max_affine");
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
    if (pick_v(C, {(long)ldo, (long)ldh}, {dout, dh}) == 4)
        launch_T<4>(n, C, nullptr, tptr, tedge, k, KnnMaxT<4>{arg, (long)C, dout, (long)ldo, dh, (long)ldh, accumulate, C}, s);
    else
        launch_T<1>(n, C, nullptr, tptr, tedge, k, KnnMaxT<1>{arg, (long)C, dout, (long)ldo, dh, (long)ldh, accumulate, C}, s);
    DC_CHECK_LAUNCH("dc_knn_max_backward");
    return DC_OK;
}
