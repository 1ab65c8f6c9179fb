// Max aggregation over the k nearest neighbours + its backward.
// Replaces torch_scatter.scatter(reduce='max') over a gathered [E,C] tensor
// (/root/reference/deltaconv/nn/deltaconv.py:52,54): the [E,C] tensor is never materialised;
// the winning slot is kept as one byte per (point, channel).
// HBM-bound: 4C*Nt in + 4C*Nt out + C*Nt arg + 4E ids per layer (reference: 4*C*E gathered).
#include <initializer_list>
#include "common.h"
#include "ell_stage.h"
#include "ell_tile.h"
#include "ell_tileT.h"

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

template <int V>
struct KnnMaxAffineResF {
    const float* h; long ldh; const float *scale, *shift; float slope; const float* h2; long ldh2; const float *scale2, *shift2;
    float slope2; float* out; long ldo; float* out2; long ldo2; unsigned char* arg; long lda;
    __device__ void operator()(long i, int c0, Row r, int k) const {
        knn_max_affine_residual_fwd<V>(i, c0, r.ids, k, h, ldh, scale, shift, slope, h2, ldh2, scale2, shift2, slope2, out, ldo, out2,
                                       ldo2, arg, lda);
    }
};

template <int V>
struct KnnSumF {
    const float* h; long ldh; float scale; float* out; long ldo;
    __device__ void operator()(long i, int c0, Row r, int k) const { knn_sum_fwd<V>(i, c0, r.ids, k, h, ldh, scale, out, ldo); }
};

// ---- backward: dh[j,c] (+)= sum over in-edges (i,s) of j with arg[i,c] == s of dout[i,c] -----------------------------
// Same sums in the same (ascending edge) order as the generic transposed skeleton, but the loop over the in-edge list
// runs in batches of KB edges: all KB slot words arg[i, c0:c0+4] are loaded first (independent 4-byte loads), then the
// dout rows of the batch's hits (exec-masked 16-byte loads, issued together).  The edge-at-a-time form has two
// DEPENDENT global loads per in-edge (the test needs the slot word, the value load needs the test): ~40 serialized L2
// round trips per thread, 0.13 of the HBM roofline in r01 (profiles/r01p_kernels.log); batched it is 2 per KB edges.
constexpr int KB = 8;

template <int V>
__global__ __launch_bounds__(256) void knn_max_bwd_kernel(long total, int groups, int remap, const int* __restrict__ tptr,
                                                          const int* __restrict__ tedge, int k,
                                                          const unsigned char* __restrict__ arg, long lda,
                                                          const float* __restrict__ dout, long ldo,
                                                          float* __restrict__ dh, long ldh, int accumulate) {
    __shared__ int src[T_CHUNK];
    __shared__ unsigned char slot[T_CHUNK];
    const int tpb = blockDim.x;
    const long t0 = dc_xcd_block(remap) * tpb;
    if (t0 >= total) return;  // block-uniform
    const long tl = min(t0 + (long)tpb, total) - 1;
    const long pf = t0 / groups, pl = tl / groups;
    const int e_begin = tptr[pf], e_end = tptr[pl + 1];
    const long t = t0 + threadIdx.x;
    const bool active = t < total;
    const long j = active ? t / groups : pf;
    const int c0 = active ? (int)(t - j * groups) * V : 0;
    const int cb = active ? tptr[j] : 0, ce = active ? tptr[j + 1] : 0;
    Vec<V> acc = vzero<V>();
    for (int base = e_begin; base < e_end; base += T_CHUNK) {
        const int cnt = min(T_CHUNK, e_end - base);
        __syncthreads();
        for (int q = threadIdx.x; q < cnt; q += tpb) {
            const int e = tedge[base + q];
            const int i = e / k;
            src[q] = i;
            slot[q] = (unsigned char)(e - i * k);
        }
        __syncthreads();
        const int lo = max(cb, base) - base, hi = min(ce, base + cnt) - base;
        for (int p0 = lo; p0 < hi; p0 += KB) {
            unsigned hit[KB];       // bit q: channel c0 + q of edge p0 + u is the arg-max of its source point
            long row[KB];
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                const int p = p0 + u;
                const bool ok = p < hi;
                row[u] = ok ? (long)src[p] : 0;
                const unsigned s = ok ? (unsigned)slot[p] : 0x1ffu;      // 0x1ff never equals a slot byte
                unsigned m = 0;
                if (V == 4) {   // the four slot bytes in one 32-bit load (c0 and lda are multiples of 4)
                    const unsigned w = *reinterpret_cast<const unsigned*>(arg + row[u] * lda + c0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) m |= (((w >> (8 * q)) & 0xffu) == s ? 1u : 0u) << q;
                } else {
#pragma unroll
                    for (int q = 0; q < V; ++q) m |= ((unsigned)arg[row[u] * lda + c0 + q] == s ? 1u : 0u) << q;
                }
                hit[u] = m;
            }
            Vec<V> g[KB];
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                g[u] = vzero<V>();
                if (hit[u]) g[u] = vload<V>(dout + row[u] * ldo + c0);
            }
#pragma unroll
            for (int u = 0; u < KB; ++u)
#pragma unroll
                for (int q = 0; q < V; ++q) acc.v[q] += ((hit[u] >> q) & 1u) ? g[u].v[q] : 0.f;
        }
    }
    if (active) vout<V>(dh + j * ldh + c0, acc, accumulate);
}

template <int V>
void launch_knn_max_bwd(long n, int C, const int* tptr, const int* tedge, int k, const unsigned char* arg,
                        const float* dout, long ldo, float* dh, long ldh, int accumulate, hipStream_t s) {
    const int groups = C / V;
    const long total = n * groups;
    hipLaunchKernelGGL((knn_max_bwd_kernel<V>), dim3(dc_cdiv(total, 256)), dim3(256), 0, s, total, groups,
                       dc_option(DC_OPT_XCD_REMAP), tptr, tedge, k, arg, (long)C, dout, ldo, dh, ldh, accumulate);
}
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

// out[i,c] = act2(scale2_c h2[i,c] + shift2_c) + max_s act(scale_c h[nbr[i,s],c] + shift_c) (+ second copy out2): the gather-path twin
// of dc_knn_max_affine_residual_tiled (any C / alignment / k)
DC_EXPORT int dc_knn_max_affine_residual(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh,
                                         const float* scale, const float* shift, float slope, const float* h2, int64_t ldh2,
                                         const float* scale2, const float* shift2, float slope2, float* out, int64_t ldo,
                                         float* out2, int64_t ldo2, uint8_t* arg, void* stream) {
    DC_REQUIRE(nbr && h && scale && shift && h2 && scale2 && shift2 && out && arg, "dc_knn_max_affine_residual: null pointer");
    DC_REQUIRE(n >= 0 && k >= 1 && k <= 255 && C >= 0, "dc_knn_max_affine_residual: bad size (k <= 255)");
    DC_REQUIRE(ldh >= C && ldo >= C && ldh2 >= C && (!out2 || ldo2 >= C), "dc_knn_max_affine_residual: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pick_v(C, {(long)ldh, (long)ldo, (long)ldh2, (long)(out2 ? ldo2 : 4)}, {h, out, h2, out2}) == 4)
        launch_fwd<4>(n, C, nullptr, nbr, k, KnnMaxAffineResF<4>{h, (long)ldh, scale, shift, slope, h2, (long)ldh2, scale2, shift2, slope2,
                                                                 out, (long)ldo, out2, (long)ldo2, arg, (long)C}, s);
    else
        launch_fwd<1>(n, C, nullptr, nbr, k, KnnMaxAffineResF<1>{h, (long)ldh, scale, shift, slope, h2, (long)ldh2, scale2, shift2, slope2,
                                                                 out, (long)ldo, out2, (long)ldo2, arg, (long)C}, s);
    DC_CHECK_LAUNCH("dc_knn_max_affine_residual");
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
    if (dc_option(DC_OPT_GATHER_BATCH)) {       // A/B switch: the edge-at-a-time generic skeleton
        if (pick_v(C, {(long)ldo, (long)ldh}, {dout, dh}) == 4)
            launch_T<4>(n, C, nullptr, tptr, tedge, k, KnnMaxT<4>{arg, (long)C, dout, (long)ldo, dh, (long)ldh, accumulate, C}, s);
        else
            launch_T<1>(n, C, nullptr, tptr, tedge, k, KnnMaxT<1>{arg, (long)C, dout, (long)ldo, dh, (long)ldh, accumulate, C}, s);
    } else if (pick_v(C, {(long)ldo, (long)ldh}, {dout, dh}) == 4) {
        launch_knn_max_bwd<4>(n, C, tptr, tedge, k, arg, dout, (long)ldo, dh, (long)ldh, accumulate, s);
    } else {
        launch_knn_max_bwd<1>(n, C, tptr, tedge, k, arg, dout, (long)ldo, dh, (long)ldh, accumulate, s);
    }
    DC_CHECK_LAUNCH("dc_knn_max_backward");
    return DC_OK;
}

// out[i,c] = scale * sum_s h[nbr[i,s],c]  (aggr = 'sum' / 'add': scale 1; 'mean': 1/k -- every point has exactly k
// neighbours incl. itself): torch_scatter.scatter(reduce=...) at nn/deltaconv.py:52,54.  Fixed slot order.
DC_EXPORT int dc_knn_sum(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh, float scale,
                         float* out, int64_t ldo, void* stream) {
    DC_REQUIRE(nbr && h && out, "dc_knn_sum: null pointer");
    DC_REQUIRE(n >= 0 && k >= 1 && k <= 255 && C >= 0, "dc_knn_sum: bad size (k <= 255)");
    DC_REQUIRE(ldh >= C && ldo >= C, "dc_knn_sum: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pick_v(C, {(long)ldh, (long)ldo}, {h, out}) == 4)
        launch_fwd<4>(n, C, nullptr, nbr, k, KnnSumF<4>{h, (long)ldh, scale, out, (long)ldo}, s);
    else
        launch_fwd<1>(n, C, nullptr, nbr, k, KnnSumF<1>{h, (long)ldh, scale, out, (long)ldo}, s);
    DC_CHECK_LAUNCH("dc_knn_sum");
    return DC_OK;
}

// dh[j,c] (+)= scale * sum over the in-edges (i,s) of j of dout[i,c], ascending edge id (CSC of dc_csc_build)
DC_EXPORT int dc_knn_sum_backward(const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, const float* dout,
                                  int32_t C, int64_t ldo, float scale, float* dh, int64_t ldh, int32_t accumulate,
                                  void* stream) {
    DC_REQUIRE(tptr && tedge && dout && dh, "dc_knn_sum_backward: null pointer");
    DC_REQUIRE(n >= 0 && k >= 1 && k <= 255 && C >= 0, "dc_knn_sum_backward: bad size (k <= 255)");
    DC_REQUIRE(ldo >= C && ldh >= C, "dc_knn_sum_backward: leading dimension smaller than the row");
    if (n == 0 || C == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pick_v(C, {(long)ldo, (long)ldh}, {dout, dh}) == 4)
        launch_T<4>(n, C, nullptr, tptr, tedge, k, KnnSumT<4>{dout, (long)ldo, dh, (long)ldh, scale, accumulate, C}, s);
    else
        launch_T<1>(n, C, nullptr, tptr, tedge, k, KnnSumT<1>{dout, (long)ldo, dh, (long)ldh, scale, accumulate, C}, s);
    DC_CHECK_LAUNCH("dc_knn_sum_backward");
    return DC_OK;
}

// ---- max aggregation from a tile plan (tile_plan.h, ell_tile.h): same values and slots, rows from LDS ---------------
DC_EXPORT int dc_knn_max_tiled(const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles, int32_t k, int32_t P,
                               const float* h, int32_t C, int64_t ldh, float* out, int64_t ldo, uint8_t* arg, void* stream) {
    DC_REQUIRE(plan && nbr && h && out && arg, "dc_knn_max_tiled: null pointer");
    DC_REQUIRE(n >= 0 && num_tiles >= 0 && k >= 2 && k % 2 == 0 && k <= 64 && (P == 32 || P == 64) && P * k <= 2048,
               "dc_knn_max_tiled: bad size");
    DC_REQUIRE(dctile::eligible(C, {(long)ldh, (long)ldo}, {h, out, arg}), "dc_knn_max_tiled: needs C %% 64 == 0 and 16-byte aligned rows");
    DC_REQUIRE(ldh >= C && ldo >= C, "dc_knn_max_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlan L = dc_tile_plan_layout(num_tiles, k, P);
    dctile::launch<1>(L, plan, nullptr, nbr, C,
                      dctile::KnnMaxB<false>{h, (long)ldh, 0, nullptr, nullptr, 0.f, out, (long)ldo, arg, (long)C},
                      static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_knn_max_tiled");
    return DC_OK;
}

DC_EXPORT int dc_knn_max_affine_tiled(const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles, int32_t k,
                                      int32_t P, const float* h, int32_t C, int64_t ldh, const float* scale,
                                      const float* shift, float slope, float* out, int64_t ldo, uint8_t* arg, void* stream) {
    DC_REQUIRE(plan && nbr && h && scale && shift && out && arg, "dc_knn_max_affine_tiled: null pointer");
    DC_REQUIRE(n >= 0 && num_tiles >= 0 && k >= 2 && k % 2 == 0 && k <= 64 && (P == 32 || P == 64) && P * k <= 2048,
               "dc_knn_max_affine_tiled: bad size");
    DC_REQUIRE(dctile::eligible(C, {(long)ldh, (long)ldo}, {h, out, arg}), "dc_knn_max_affine_tiled: needs C %% 64 == 0 and 16-byte aligned rows");
    DC_REQUIRE(ldh >= C && ldo >= C, "dc_knn_max_affine_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlan L = dc_tile_plan_layout(num_tiles, k, P);
    dctile::launch<1>(L, plan, nullptr, nbr, C,
                      dctile::KnnMaxB<true>{h, (long)ldh, 0, scale, shift, slope, out, (long)ldo, arg, (long)C},
                      static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_knn_max_affine_tiled");
    return DC_OK;
}

// max aggregation with the layer's last s_mlp block in its epilogue (ell_tile.h: KnnMaxB::h2): out = act2(bn2(h2)) + max_j act(bn(h_j)),
// i.e. `x = s_mlp(...) + x_max` of nn/deltaconv.py:59 with both BatchNorm + activation pairs folded in; out2 (may be NULL) = second copy
DC_EXPORT int dc_knn_max_affine_residual_tiled(const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles, int32_t k,
                                               int32_t P, const float* h, int32_t C, int64_t ldh, const float* scale,
                                               const float* shift, float slope, const float* h2, int64_t ldh2,
                                               const float* scale2, const float* shift2, float slope2, float* out, int64_t ldo,
                                               float* out2, int64_t ldo2, uint8_t* arg, void* stream) {
    DC_REQUIRE(plan && nbr && h && scale && shift && h2 && scale2 && shift2 && out && arg, "dc_knn_max_affine_residual_tiled: null pointer");
    DC_REQUIRE(n >= 0 && num_tiles >= 0 && k >= 2 && k % 2 == 0 && k <= 64 && (P == 32 || P == 64) && P * k <= 2048,
               "dc_knn_max_affine_residual_tiled: bad size");
    DC_REQUIRE(dctile::eligible(C, {(long)ldh, (long)ldo, (long)ldh2, (long)(out2 ? ldo2 : 4)}, {h, out, arg, h2, scale2, shift2, out2}),
               "dc_knn_max_affine_residual_tiled: needs C %% 64 == 0 and 16-byte aligned rows / coefficient vectors");
    DC_REQUIRE(ldh >= C && ldo >= C && ldh2 >= C && (!out2 || ldo2 >= C), "dc_knn_max_affine_residual_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlan L = dc_tile_plan_layout(num_tiles, k, P);
    dctile::KnnMaxB<true> body{h, (long)ldh, 0, scale, shift, slope, out, (long)ldo, arg, (long)C};
    body.h2 = h2; body.ldh2 = ldh2; body.scale2 = scale2; body.shift2 = shift2; body.slope2 = slope2; body.out2 = out2; body.ldo2 = ldo2;
    dctile::launch<1>(L, plan, nullptr, nbr, C, body, static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_knn_max_affine_residual_tiled");
    return DC_OK;
}

// ---- max-aggregation backward from the transposed tile plan (ell_tileT.h): same sums in the same order, rows + slot words in LDS
DC_EXPORT int dc_knn_max_backward_tiled(const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles, int32_t k,
                                        int32_t P, const uint8_t* arg, const float* dout, int32_t C, int64_t ldo, float* dh,
                                        int64_t ldh, int32_t accumulate, void* stream) {
    DC_REQUIRE(planT && arg && dout && dh, "dc_knn_max_backward_tiled: null pointer");
    DC_REQUIRE(n >= 0 && num_clouds >= 0 && num_tiles >= 0 && k >= 2 && k % 2 == 0 && k <= 64 && (P == 32 || P == 64) && P * k <= 2048,
               "dc_knn_max_backward_tiled: bad size");
    DC_REQUIRE(dctile::eligible(C, {(long)ldo, (long)ldh}, {dout, dh, arg}), "dc_knn_max_backward_tiled: needs C %% 64 == 0 and 16-byte aligned rows");
    DC_REQUIRE(ldo >= C && ldh >= C, "dc_knn_max_backward_tiled: leading dimension smaller than the row");
    if (n == 0) return DC_OK;
    const DcTilePlanT L = dc_tile_plan_T_layout(n, num_clouds, num_tiles, k, P);
    dctileT::launch<1>(L, planT, nullptr, C, dctileT::KnnMaxTB{dout, (long)ldo, 0, arg, (long)C, dh, (long)ldh, accumulate},
                       static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_knn_max_backward_tiled");
    return DC_OK;
}
