// Sparse operator applies in fixed-degree (ELL) form + their transposes (backward).
// Replaces every `SparseTensor @ dense` on the hot path (third-party torch_sparse spmm; call sites
// /root/reference/deltaconv/models/deltanet_base.py:78, nn/deltaconv.py:57,66,
// geometry/operators.py:27,33,40,43) and the operator algebra built from them
// (geometry/operators.py:23-46: curl, hodge_laplacian), fused so v / (div v, curl v) are gathered once.
//
// HBM-bound by algorithmic bytes (12*C*Nt + 12*E per plain apply: input once, output once, ids +
// coefficients once; 4*E*C flop -> ~5 flop/B at C=64,k=20, far below the fp32 ridge); PMC shows the
// HBM traffic equals that figure, and the on-chip limiter to be the 64 B/clk/CU vector-memory
// (texture addresser / L1) path that serves the 16-byte neighbour-row gathers.
// Thread bodies: ell_math.h.  Launch skeletons (LDS staging of ids / coefficients): ell_stage.h.
#include <initializer_list>
#include "common.h"
#include "ell_stage.h"
#include "ell_tile.h"
#include "ell_tileT.h"

namespace {
using namespace dcell;
using namespace dcstage;

#define DC_FWD_BODY(NAME, FN)                                                                  \
    template <int V>                                                                           \
    struct NAME {                                                                              \
        const float* in; long ldi; float* out; long ldo; int C;                                \
        __device__ void operator()(long i, int c0, Row r, int k) const {                       \
            FN<V>(i, c0, C, r, k, in, ldi, out, ldo);                                          \
        }                                                                                      \
    };
DC_FWD_BODY(GradF, grad_fwd)
DC_FWD_BODY(DivF, div_fwd)
DC_FWD_BODY(DivCurlNormF, divcurlnorm_fwd)
DC_FWD_BODY(HodgeF, hodge_fwd)

int check_common(const char* name, const void* a, const void* b, const void* c, const void* d, int n, int k, int C) {
    if (!a || !b || !c || !d) {
        dc_set_error("%s: null pointer", name);
        return DC_ERR_ARG;
    }
    if (n < 0 || k < 1 || k > 255 || C < 0) {
        dc_set_error("%s: bad size n=%d k=%d C=%d (k <= 255)", name, n, k, C);
        return DC_ERR_ARG;
    }
    return DC_OK;
}
}  // namespace

// ---- forward ---------------------------------------------------------------------------------
#define DC_FWD_ENTRY(FN, BODY, MINLDI, MINLDO)                                                                    \
    DC_EXPORT int FN(const float* coef, const int32_t* nbr, int32_t n, int32_t k, const float* in, int32_t C,      \
                     int64_t ldi, float* out, int64_t ldo, void* stream) {                                        \
        if (int rc = check_common(#FN, coef, nbr, in, out, n, k, C)) return rc;                                   \
        DC_REQUIRE(ldi >= (MINLDI) && ldo >= (MINLDO), #FN ": leading dimension smaller than the row");           \
        if (n == 0 || C == 0) return DC_OK;                                                                       \
        hipStream_t s = static_cast<hipStream_t>(stream);                                                         \
        const int vw = pick_v(C, {(long)ldi, (long)ldo}, {in, out});                                              \
        if (vw == 4)                                                                                              \
            launch_fwd<4>(n, C, coef, nbr, k, BODY<4>{in, (long)ldi, out, (long)ldo, C}, s);                      \
        else if (vw == 2)                                                                                         \
            launch_fwd<2>(n, C, coef, nbr, k, BODY<2>{in, (long)ldi, out, (long)ldo, C}, s);                      \
        else                                                                                                      \
            launch_fwd<1>(n, C, coef, nbr, k, BODY<1>{in, (long)ldi, out, (long)ldo, C}, s);                      \
        DC_CHECK_LAUNCH(#FN);                                                                                     \
        return DC_OK;                                                                                             \
    }

DC_FWD_ENTRY(dc_apply_grad, GradF, C, C)
DC_FWD_ENTRY(dc_apply_div, DivF, C, C)
DC_FWD_ENTRY(dc_apply_div_curl_norm, DivCurlNormF, C, 3 * C)
DC_FWD_ENTRY(dc_apply_hodge, HodgeF, 2 * C, C)

// ---- transposed ------------------------------------------------------------------------------
// coefT = the operator's coefficients permuted into CSC order (dc_csc_permute_coef)
#define DC_T_ENTRY(FN, OP, MINLDY, MINLDX)                                                                        \
    DC_EXPORT int FN(const float* coefT, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k,          \
                     const float* dy, int32_t C, int64_t ldy, float* dx, int64_t ldx, int32_t accumulate,          \
                     void* stream) {                                                                              \
        if (int rc = check_common(#FN, coefT, tptr, dy, dx, n, k, C)) return rc;                                  \
        DC_REQUIRE(tedge, #FN ": null pointer");                                                                  \
        DC_REQUIRE(ldy >= (MINLDY) && ldx >= (MINLDX), #FN ": leading dimension smaller than the row");           \
        if (n == 0 || C == 0) return DC_OK;                                                                       \
        hipStream_t s = static_cast<hipStream_t>(stream);                                                         \
        const int vw = pick_v(C, {(long)ldy, (long)ldx}, {dy, dx});                                               \
        if (vw == 4)                                                                                              \
            launch_T<4>(n, C, coefT, tptr, tedge, k, OP<4>{dy, (long)ldy, dx, (long)ldx, accumulate, C}, s);      \
        else if (vw == 2)                                                                                         \
            launch_T<2>(n, C, coefT, tptr, tedge, k, OP<2>{dy, (long)ldy, dx, (long)ldx, accumulate, C}, s);      \
        else                                                                                                      \
            launch_T<1>(n, C, coefT, tptr, tedge, k, OP<1>{dy, (long)ldy, dx, (long)ldx, accumulate, C}, s);      \
        DC_CHECK_LAUNCH(#FN);                                                                                     \
        return DC_OK;                                                                                             \
    }

DC_T_ENTRY(dc_apply_grad_T, GradT, C, C)
DC_T_ENTRY(dc_apply_div_T, DivT, C, C)
DC_T_ENTRY(dc_apply_hodge_T, HodgeT, C, 2 * C)

// out[n, C] = a (+ b) + grad^T dy: the transposed gradient apply with the accumulation of d x' folded in
DC_EXPORT int dc_apply_grad_T_sum(const float* GT, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k,
                                  const float* dy, int32_t C, int64_t ldy, const float* a, int64_t lda, const float* b,
                                  int64_t ldb, float* out, int64_t ldo, void* stream) {
    if (int rc = check_common("dc_apply_grad_T_sum", GT, tptr, dy, out, n, k, C)) return rc;
    DC_REQUIRE(tedge && a, "dc_apply_grad_T_sum: null pointer");
    DC_REQUIRE(ldy >= C && lda >= C && ldo >= C && (!b || ldb >= C), "dc_apply_grad_T_sum: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long ldb_ = b ? (long)ldb : (long)lda;
    const int vw = pick_v(C, {(long)ldy, (long)lda, ldb_, (long)ldo}, {dy, a, b ? b : a, out});
    if (vw == 4)
        launch_T<4>(n, C, GT, tptr, tedge, k, GradTSum<4>{dy, (long)ldy, a, (long)lda, b, (long)ldb, out, (long)ldo, C}, s);
    else if (vw == 2)
        launch_T<2>(n, C, GT, tptr, tedge, k, GradTSum<2>{dy, (long)ldy, a, (long)lda, b, (long)ldb, out, (long)ldo, C}, s);
    else
        launch_T<1>(n, C, GT, tptr, tedge, k, GradTSum<1>{dy, (long)ldy, a, (long)lda, b, (long)ldb, out, (long)ldo, C}, s);
    DC_CHECK_LAUNCH("dc_apply_grad_T_sum");
    return DC_OK;
}

DC_EXPORT int dc_apply_div_curl_norm_T(const float* DT, const int32_t* tptr, const int32_t* tedge, int32_t n,
                                       int32_t k, const float* dout, int32_t C, int64_t ldo, const float* v,
                                       int64_t ldv, float* dv, int64_t lddv, int32_t accumulate, void* stream) {
    if (int rc = check_common("dc_apply_div_curl_norm_T", DT, tptr, dout, dv, n, k, C)) return rc;
    DC_REQUIRE(tedge && v, "dc_apply_div_curl_norm_T: null pointer");
    DC_REQUIRE(ldo >= 3 * C && ldv >= C && lddv >= C, "dc_apply_div_curl_norm_T: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int vw = pick_v(C, {(long)ldo, (long)ldv, (long)lddv}, {dout, v, dv});
    if (vw == 4)
        launch_T<4>(n, C, DT, tptr, tedge, k,
                    DivCurlNormT<4>{dout, (long)ldo, v, (long)ldv, dv, (long)lddv, accumulate, C}, s);
    else if (vw == 2)
        launch_T<2>(n, C, DT, tptr, tedge, k,
                    DivCurlNormT<2>{dout, (long)ldo, v, (long)ldv, dv, (long)lddv, accumulate, C}, s);
    else
        launch_T<1>(n, C, DT, tptr, tedge, k,
                    DivCurlNormT<1>{dout, (long)ldo, v, (long)ldv, dv, (long)lddv, accumulate, C}, s);
    DC_CHECK_LAUNCH("dc_apply_div_curl_norm_T");
    return DC_OK;
}

// ---- forward, from a tile plan (tile_plan.h, ell_tile.h) ---------------------------------------------------------
// Same results bit for bit as the entry points above (same FMAs, same slot order); the neighbour rows come from LDS.
// coef = the operator as for the plain entry points; nbr is read only by tiles whose unique rows exceed the LDS capacity.  C must be a multiple of 64 and rows 16-byte aligned (otherwise DC_ERR_ARG: use the entry
// points above).
namespace {
int check_tiled(const char* name, const void* a, const void* b, const void* c, const void* d, const void* e, int n, int nc,
                int k, int P, int C, bool ok16) {
    if (!a || !b || !c || !d || !e) {
        dc_set_error("%s: null pointer", name);
        return DC_ERR_ARG;
    }
    if (n < 0 || nc < 0 || k < 2 || k % 2 || k > 64 || (P != 32 && P != 64) || P * k > 2048) {
        dc_set_error("%s: bad size n=%d num_tiles=%d k=%d P=%d", name, n, nc, k, P);
        return DC_ERR_ARG;
    }
    if (!ok16) {
        dc_set_error("%s: needs C %% 64 == 0 and 16-byte aligned rows (C=%d)", name, C);
        return DC_ERR_ARG;
    }
    return DC_OK;
}
}  // namespace

#define DC_TILED_ENTRY(FN, BODY, R, LDJ, HS, MINLDI, MINLDO, ...)                                                     \
    DC_EXPORT int FN(const float* coef, const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles,     \
                     int32_t k, int32_t P, const float* in, int32_t C, int64_t ldi, float* out, int64_t ldo,          \
                     void* stream) {                                                                                 \
        if (int rc = check_tiled(#FN, coef, plan, nbr, in, out, n, num_tiles, k, P, C,                              \
                                 dctile::eligible(C, {(long)ldi, (long)ldo}, {in, out, coef})))                      \
            return rc;                                                                                               \
        DC_REQUIRE(ldi >= (MINLDI) && ldo >= (MINLDO), #FN ": leading dimension smaller than the row");              \
        if (n == 0) return DC_OK;                                                                                    \
        const DcTilePlan L = dc_tile_plan_layout(num_tiles, k, P);                                               \
        dctile::launch<R>(L, plan, coef, nbr, C, dctile::BODY{in, (long)(LDJ), (long)(HS), out, (long)ldo __VA_ARGS__}, \
                          static_cast<hipStream_t>(stream));                                                         \
        DC_CHECK_LAUNCH(#FN);                                                                                        \
        return DC_OK;                                                                                                \
    }

DC_TILED_ENTRY(dc_apply_grad_tiled, GradB, 1, ldi, 0, C, C)
DC_TILED_ENTRY(dc_apply_div_tiled, DivB, 2, 2 * ldi, ldi, C, C)
DC_TILED_ENTRY(dc_apply_div_curl_norm_tiled, DivCurlNormB, 2, 2 * ldi, ldi, C, 3 * C, , C)
DC_TILED_ENTRY(dc_apply_hodge_tiled, HodgeB, 2, ldi, C, 2 * C, C)

// ---- transposed, from the transposed tile plan (tile_plan.h second half, ell_tileT.h) ------------------------------------
// Same results bit for bit as the entry points above (same FMAs, ascending edge id per target); the source rows come from
// LDS.  coefTt = the operator's coefficients in TILE order (dc_tile_plan_T_permute_coef); planT from
// dc_tile_plan_T_build.  C must be a multiple of 64 and all rows 16-byte aligned (otherwise DC_ERR_ARG: use the entry points above).
namespace {
int check_tiledT(const char* name, std::initializer_list<const void*> ptrs, int n, int nc, int nt, int k, int P, bool ok16, int C) {
    for (const void* p : ptrs)
        if (!p) {
            dc_set_error("%s: null pointer", name);
            return DC_ERR_ARG;
        }
    if (n < 0 || nc < 0 || nt < 0 || k < 2 || k % 2 || k > 64 || (P != 32 && P != 64) || P * k > 2048) {
        dc_set_error("%s: bad size n=%d num_clouds=%d num_tiles=%d k=%d P=%d", name, n, nc, nt, k, P);
        return DC_ERR_ARG;
    }
    if (!ok16) {
        dc_set_error("%s: needs C %% 64 == 0 and 16-byte aligned rows (C=%d)", name, C);
        return DC_ERR_ARG;
    }
    return DC_OK;
}
}  // namespace

DC_EXPORT int dc_apply_grad_T_tiled(const float* GTt, const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles,
                                    int32_t k, int32_t P, const float* dy, int32_t C, int64_t ldy, float* dx, int64_t ldx,
                                    int32_t accumulate, void* stream) {
    if (int rc = check_tiledT("dc_apply_grad_T_tiled", {GTt, planT, dy, dx}, n, num_clouds, num_tiles, k, P,
                              dctile::eligible(C, {(long)ldy, (long)ldx}, {dy, dx, GTt}), C))
        return rc;
    DC_REQUIRE(ldy >= C && ldx >= C, "dc_apply_grad_T_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlanT L = dc_tile_plan_T_layout(n, num_clouds, num_tiles, k, P);
    dctileT::launch<2>(L, planT, GTt, C,
                       dctileT::GradTB<false>{dy, 2 * (long)ldy, (long)ldy, nullptr, 0, nullptr, 0, nullptr, 0, dx, (long)ldx, accumulate},
                       static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_apply_grad_T_tiled");
    return DC_OK;
}

DC_EXPORT int dc_apply_grad_T_sum_tiled(const float* GTt, const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles,
                                        int32_t k, int32_t P, const float* dy, int32_t C, int64_t ldy, const float* a, int64_t lda,
                                        const float* b, int64_t ldb, float* out, int64_t ldo, void* stream) {
    const long ldb_ = b ? (long)ldb : (long)lda;
    if (int rc = check_tiledT("dc_apply_grad_T_sum_tiled", {GTt, planT, dy, a, out}, n, num_clouds, num_tiles, k, P,
                              dctile::eligible(C, {(long)ldy, (long)lda, ldb_, (long)ldo}, {dy, a, b ? b : a, out, GTt}), C))
        return rc;
    DC_REQUIRE(ldy >= C && lda >= C && ldo >= C && (!b || ldb >= C), "dc_apply_grad_T_sum_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlanT L = dc_tile_plan_T_layout(n, num_clouds, num_tiles, k, P);
    dctileT::launch<2>(L, planT, GTt, C,
                       dctileT::GradTB<true>{dy, 2 * (long)ldy, (long)ldy, nullptr, 0, a, (long)lda, b, (long)ldb, out, (long)ldo, 0},
                       static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_apply_grad_T_sum_tiled");
    return DC_OK;
}

DC_EXPORT int dc_apply_div_T_tiled(const float* DTt, const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles,
                                   int32_t k, int32_t P, const float* dy, int32_t C, int64_t ldy, float* dv, int64_t ldv,
                                   int32_t accumulate, void* stream) {
    if (int rc = check_tiledT("dc_apply_div_T_tiled", {DTt, planT, dy, dv}, n, num_clouds, num_tiles, k, P,
                              dctile::eligible(C, {(long)ldy, (long)ldv}, {dy, dv, DTt}), C))
        return rc;
    DC_REQUIRE(ldy >= C && ldv >= C, "dc_apply_div_T_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlanT L = dc_tile_plan_T_layout(n, num_clouds, num_tiles, k, P);
    dctileT::launch<1>(L, planT, DTt, C, dctileT::DivTB{dy, (long)ldy, 0, nullptr, 0, dv, (long)ldv, accumulate},
                       static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_apply_div_T_tiled");
    return DC_OK;
}

DC_EXPORT int dc_apply_hodge_T_tiled(const float* GTt, const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles,
                                     int32_t k, int32_t P, const float* dh, int32_t C, int64_t ldh, float* ddc, int64_t ldd,
                                     int32_t accumulate, void* stream) {
    if (int rc = check_tiledT("dc_apply_hodge_T_tiled", {GTt, planT, dh, ddc}, n, num_clouds, num_tiles, k, P,
                              dctile::eligible(C, {(long)ldh, (long)ldd}, {dh, ddc, GTt}), C))
        return rc;
    DC_REQUIRE(ldh >= C && ldd >= 2 * C, "dc_apply_hodge_T_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlanT L = dc_tile_plan_T_layout(n, num_clouds, num_tiles, k, P);
    dctileT::launch<2>(L, planT, GTt, C, dctileT::HodgeTB{dh, 2 * (long)ldh, (long)ldh, nullptr, 0, ddc, (long)ldd, accumulate, C},
                       static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_apply_hodge_T_tiled");
    return DC_OK;
}

DC_EXPORT int dc_apply_div_curl_norm_T_tiled(const float* DTt, const int32_t* planT, int32_t n, int32_t num_clouds,
                                             int32_t num_tiles, int32_t k, int32_t P, const float* dout, int32_t C, int64_t ldo,
                                             const float* v, int64_t ldv, float* dv, int64_t lddv, int32_t accumulate,
                                             void* stream) {
    if (int rc = check_tiledT("dc_apply_div_curl_norm_T_tiled", {DTt, planT, dout, v, dv}, n, num_clouds, num_tiles, k, P,
                              dctile::eligible(C, {(long)ldo, (long)ldv, (long)lddv}, {dout, v, dv, DTt}), C))
        return rc;
    DC_REQUIRE(ldo >= 3 * C && ldv >= C && lddv >= C, "dc_apply_div_curl_norm_T_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlanT L = dc_tile_plan_T_layout(n, num_clouds, num_tiles, k, P);
    dctileT::launch<2>(L, planT, DTt, C,
                       dctileT::DivCurlNormTB{dout, (long)ldo, (long)C, nullptr, 0, v, (long)ldv, dv, (long)lddv, accumulate, C},
                       static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_apply_div_curl_norm_T_tiled");
    return DC_OK;
}
