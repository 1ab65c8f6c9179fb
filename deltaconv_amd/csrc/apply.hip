// Sparse operator applies in fixed-degree (ELL) form + their transposes (backward).
// Replaces every `SparseTensor @ dense` on the hot path (third-party torch_sparse spmm; call sites
// /root/reference/deltaconv/models/deltanet_base.py:78, nn/deltaconv.py:57,66,
// geometry/operators.py:27,33,40,43) and the operator algebra built from them
// (geometry/operators.py:23-46: curl, hodge_laplacian), fused so v / (div v, curl v) are gathered once.
//
// HBM-bound: algorithmic bytes per apply = 12*C*Nt + 12*E (read input once, write output once,
// read ids + coefficients once); 4*E*C flop -> ~5 flop/B at C=64,k=20, far below the fp32 ridge.
// Thread bodies: ell_math.h.  One thread per (point, 4-channel group); 256-thread blocks.
#include "common.h"
#include "ell_math.h"

namespace {

using namespace dcell;
constexpr int TPB = 256;

#define DC_ELL_KERNEL(NAME, BODY, PARAMS, ARGS)                                   \
    template <int V>                                                              \
    __global__ __launch_bounds__(TPB) void NAME##_kernel(long total, int groups, int remap, PARAMS) { \
        const long t = dc_xcd_block(remap) * TPB + threadIdx.x;                   \
        if (t >= total) return;                                                   \
        BODY<V>(t, groups, ARGS);                                                 \
    }

#define P_FWD const float *coef, const int *nbr, int k, const float *in, long ldi, float *out, long ldo
#define A_FWD coef, nbr, k, in, ldi, out, ldo
// forward kernels: V = vector width, U = neighbour rows requested per batch (gathers in flight)
#define DC_ELL_FWD_KERNEL(NAME)                                                                        \
    template <int V, int U>                                                                            \
    __global__ __launch_bounds__(TPB) void NAME##_kernel(long total, int groups, int remap, P_FWD) {   \
        const long t = dc_xcd_block(remap) * TPB + threadIdx.x;                                        \
        if (t >= total) return;                                                                        \
        NAME<V, U>(t, groups, A_FWD);                                                                  \
    }
DC_ELL_FWD_KERNEL(grad_fwd)
DC_ELL_FWD_KERNEL(div_fwd)
DC_ELL_FWD_KERNEL(divcurlnorm_fwd)
DC_ELL_FWD_KERNEL(hodge_fwd)

#define DC_LAUNCH_FWD(NAME, V, U, n, C, stream, ...)                                                        \
    do {                                                                                                    \
        const int groups_ = (C) / (V);                                                                      \
        const long total_ = (long)(n) * groups_;                                                            \
        hipLaunchKernelGGL((NAME##_kernel<V, U>), dim3(dc_cdiv(total_, TPB)), dim3(TPB), 0, stream, total_, \
                           groups_, dc_option(DC_OPT_XCD_REMAP), __VA_ARGS__);                              \
    } while (0)

// DC_OPT_GATHER_BATCH: 0 (default) -> 10 rows in flight, 1 -> 4, 2 -> 20   (A/B switch)
#define DC_DISPATCH_FWD(NAME, v, n, C, stream, ...)                                              \
    do {                                                                                         \
        const int ub_ = dc_option(DC_OPT_GATHER_BATCH);                                          \
        if ((v) == 1)                                                                            \
            DC_LAUNCH_FWD(NAME, 1, 4, n, C, stream, __VA_ARGS__);                                \
        else if (ub_ == 1)                                                                       \
            DC_LAUNCH_FWD(NAME, 4, 4, n, C, stream, __VA_ARGS__);                                \
        else if (ub_ == 2)                                                                       \
            DC_LAUNCH_FWD(NAME, 4, 20, n, C, stream, __VA_ARGS__);                               \
        else                                                                                     \
            DC_LAUNCH_FWD(NAME, 4, 10, n, C, stream, __VA_ARGS__);                               \
    } while (0)

#define P_T const float *coef, const int *tptr, const int *tedge, int k, const float *dy, long ldy, float *dx, long ldx, int acc
#define A_T coef, tptr, tedge, k, dy, ldy, dx, ldx, acc
DC_ELL_KERNEL(grad_T, grad_T, P_T, A_T)
DC_ELL_KERNEL(div_T, div_T, P_T, A_T)
DC_ELL_KERNEL(hodge_T, hodge_T, P_T, A_T)

#define P_DCT const float *coef, const int *tptr, const int *tedge, int k, const float *dout, long ldo, const float *v, long ldv, float *dv, long lddv, int acc
#define A_DCT coef, tptr, tedge, k, dout, ldo, v, ldv, dv, lddv, acc
DC_ELL_KERNEL(divcurlnorm_T, divcurlnorm_T, P_DCT, A_DCT)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// vector width: 16-byte path when channels, strides and bases allow it
inline int pick_v(int C, std::initializer_list<long> lds, std::initializer_list<const void*> ptrs) {
    if (C % 4) return 1;
    for (long l : lds)
        if (l % 4) return 1;
    for (const void* p : ptrs)
        if (!aligned16(p)) return 1;
    return 4;
}

#define DC_LAUNCH_ELL(NAME, V, n, C, stream, ...)                                                         \
    do {                                                                                                  \
        const int groups_ = (C) / (V);                                                                    \
        const long total_ = (long)(n) * groups_;                                                          \
        hipLaunchKernelGGL((NAME##_kernel<V>), dim3(dc_cdiv(total_, TPB)), dim3(TPB), 0, stream, total_,  \
                           groups_, dc_option(DC_OPT_XCD_REMAP), __VA_ARGS__);                                                         \
    } while (0)

#define DC_DISPATCH_V(NAME, v, n, C, stream, ...)                      \
    do {                                                               \
        if ((v) == 4)                                                  \
            DC_LAUNCH_ELL(NAME, 4, n, C, stream, __VA_ARGS__);         \
        else                                                           \
            DC_LAUNCH_ELL(NAME, 1, n, C, stream, __VA_ARGS__);         \
    } while (0)

int check_common(const char* name, const void* a, const void* b, const void* c, const void* d, int n, int k, int C) {
    if (!a || !b || !c || !d) {
        dc_set_error("%s: null pointer", name);
        return DC_ERR_ARG;
    }
    if (n < 0 || k < 1 || C < 0) {
        dc_set_error("%s: bad size n=%d k=%d C=%d", name, n, k, C);
        return DC_ERR_ARG;
    }
    return DC_OK;
}

}  // namespace

// ---- forward ---------------------------------------------------------------------------------
#define DC_FWD_ENTRY(FN, KERNEL, MINLDI, MINLDO)                                                                  \
    DC_EXPORT int FN(const float* coef, const int32_t* nbr, int32_t n, int32_t k, const float* in, int32_t C,      \
                     int64_t ldi, float* out, int64_t ldo, void* stream) {                                        \
        if (int rc = check_common(#FN, coef, nbr, in, out, n, k, C)) return rc;                                   \
        DC_REQUIRE(ldi >= (MINLDI) && ldo >= (MINLDO), #FN ": leading dimension smaller than the row");           \
        if (n == 0 || C == 0) return DC_OK;                                                                       \
        const int v = pick_v(C, {(long)ldi, (long)ldo}, {in, out});                                               \
        DC_DISPATCH_FWD(KERNEL, v, n, C, static_cast<hipStream_t>(stream), coef, nbr, k, in, (long)ldi, out,      \
                        (long)ldo);                                                                               \
        DC_CHECK_LAUNCH(#FN);                                                                                     \
        return DC_OK;                                                                                             \
    }

DC_FWD_ENTRY(dc_apply_grad, grad_fwd, C, C)
DC_FWD_ENTRY(dc_apply_div, div_fwd, C, C)
DC_FWD_ENTRY(dc_apply_div_curl_norm, divcurlnorm_fwd, C, 3 * C)
DC_FWD_ENTRY(dc_apply_hodge, hodge_fwd, 2 * C, C)

// ---- transposed ------------------------------------------------------------------------------
#define DC_T_ENTRY(FN, KERNEL, MINLDY, MINLDX)                                                                    \
    DC_EXPORT int FN(const float* coef, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k,           \
                     const float* dy, int32_t C, int64_t ldy, float* dx, int64_t ldx, int32_t accumulate,          \
                     void* stream) {                                                                              \
        if (int rc = check_common(#FN, coef, tptr, dy, dx, n, k, C)) return rc;                                   \
        DC_REQUIRE(tedge, #FN ": null pointer");                                                                  \
        DC_REQUIRE(ldy >= (MINLDY) && ldx >= (MINLDX), #FN ": leading dimension smaller than the row");           \
        if (n == 0 || C == 0) return DC_OK;                                                                       \
        const int v = pick_v(C, {(long)ldy, (long)ldx}, {dy, dx});                                                \
        DC_DISPATCH_V(KERNEL, v, n, C, static_cast<hipStream_t>(stream), coef, tptr, tedge, k, dy, (long)ldy, dx, \
                      (long)ldx, accumulate);                                                                     \
        DC_CHECK_LAUNCH(#FN);                                                                                     \
        return DC_OK;                                                                                             \
    }

DC_T_ENTRY(dc_apply_grad_T, grad_T, C, C)
DC_T_ENTRY(dc_apply_div_T, div_T, C, C)
DC_T_ENTRY(dc_apply_hodge_T, hodge_T, C, 2 * C)

DC_EXPORT int dc_apply_div_curl_norm_T(const float* D, const int32_t* tptr, const int32_t* tedge, int32_t n,
                                       int32_t k, const float* dout, int32_t C, int64_t ldo, const float* v,
                                       int64_t ldv, float* dv, int64_t lddv, int32_t accumulate, void* stream) {
    if (int rc = check_common("dc_apply_div_curl_norm_T", D, tptr, dout, dv, n, k, C)) return rc;
    DC_REQUIRE(tedge && v, "dc_apply_div_curl_norm_T: null pointer");
    DC_REQUIRE(ldo >= 3 * C && ldv >= C && lddv >= C, "dc_apply_div_curl_norm_T: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    const int vw = pick_v(C, {(long)ldo, (long)ldv, (long)lddv}, {dout, v, dv});
    DC_DISPATCH_V(divcurlnorm_T, vw, n, C, static_cast<hipStream_t>(stream), D, tptr, tedge, k, dout, (long)ldo, v,
                  (long)ldv, dv, (long)lddv, accumulate);
    DC_CHECK_LAUNCH("dc_apply_div_curl_norm_T");
    return DC_OK;
}
