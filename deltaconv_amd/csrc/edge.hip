// Layer-0 edge MLP + max aggregation without the [E,C] tensor (math and derivation: edge_math.h).
// Replaces, for a depth-1 `s_mlp_max` with centralized=True,
//   scatter(s_mlp_max(x[col] - x[row]), row, reduce='max')   (/root/reference/deltaconv/nn/deltaconv.py:50-52)
// i.e. index_select + addmm + native_batch_norm + leaky_relu + scatter_max over E = Nt*k rows.
// HBM-bound gather passes: fwd 4C*Nt in + 13C*Nt out + 4E ids; bwd CSC pass ~9C*E gathered bytes.
#include "common.h"
#include "colreduce.h"
#include "edge_math.h"
#include "ell_tileT.h"

namespace {
using namespace dccol;
using namespace dcedge;

template <int V>
struct EdgeStatsF {
    const float* y; long ldy; const int* nbr; int k;
    float *amax, *amin; unsigned char *argmax, *argmin; float* s1pt; long ldo, lda;
    __device__ void operator()(long i, int c0, double (&t)[2][V]) const {
        edge_gather<V>(i, c0, y, ldy, nbr, k, amax, amin, argmax, argmin, s1pt, ldo, lda, t);
    }
};

template <int V>
struct EdgeBwdF {   // writes dz* and returns (dz*, dz* ahat*)
    const float *dout, *amax, *amin, *scale, *shift, *mean, *invstd; long lddo, ldo; float slope; float* dzs;
    __device__ void operator()(long i, int c0, double (&t)[2][V]) const {
        const FV<V> g = ldv<V>(dout + i * lddo + c0), mx = ldv<V>(amax + i * ldo + c0), mn = ldv<V>(amin + i * ldo + c0);
        FV<V> dz;
#pragma unroll
        for (int q = 0; q < V; ++q) {
            float b;
            edge_bwd_terms(g.v[q], mx.v[q], mn.v[q], scale[c0 + q], shift[c0 + q], mean[c0 + q], invstd[c0 + q], slope,
                           dz.v[q], b);
            t[0][q] = dz.v[q];
            t[1][q] = b;
        }
        stv<V>(dzs + i * ldo + c0, dz);
    }
};

template <int V>
__global__ __launch_bounds__(256) void edge_apply_kernel(const float* __restrict__ amax, const float* __restrict__ amin,
                                                         const unsigned char* __restrict__ argmax,
                                                         const unsigned char* __restrict__ argmin, long n, int groups,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float slope,
                                                         float* __restrict__ out, long ldo,
                                                         unsigned char* __restrict__ arg,
                                                         // optional epilogue (round 6): the layer's last s_mlp block rides along,
                                                         // out = act2(scale2 h2 + shift2) + x_max (deltaconv.py:59), copy in out2
                                                         const float* __restrict__ h2, long ldh2,
                                                         const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                         float slope2, float* __restrict__ out2, long ldo2) {
    const long total = n * groups;
    const int C = groups * V;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long i = t / groups;
        const int c0 = (int)(t % groups) * V;
        const FV<V> mx = ldv<V>(amax + i * C + c0), mn = ldv<V>(amin + i * C + c0);
        FV<V> o;
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const int c = c0 + q;
            o.v[q] = dcnn::act(fmaf(scale[c], pick(scale[c], mx.v[q], mn.v[q]), shift[c]), slope);
            if (arg) arg[i * C + c] = scale[c] >= 0.f ? argmax[i * C + c] : argmin[i * C + c];
        }
        if (h2) {
            const FV<V> x = ldv<V>(h2 + i * ldh2 + c0);
#pragma unroll
            for (int q = 0; q < V; ++q) o.v[q] = dcnn::act(fmaf(scale2[c0 + q], x.v[q], shift2[c0 + q]), slope2) + o.v[q];
            if (out2) stv<V>(out2 + i * ldo2 + c0, o);
        }
        stv<V>(out + i * ldo + c0, o);
    }
}

template <int V>
__global__ __launch_bounds__(256) void edge_bwd_kernel(long total, int groups, int remap, const int* tptr,
                                                       const int* tedge, int k, const float* y, long ldy,
                                                       const float* dzs, const float* s1pt, long ldo,
                                                       const unsigned char* argmax, const unsigned char* argmin,
                                                       long lda, const float* scale, const float* mean,
                                                       const float* invstd, const float* m1, const float* m2,
                                                       int training, float* dy, long lddy) {
    const long t = dc_xcd_block(remap) * 256 + threadIdx.x;
    if (t >= total) return;
    edge_bwd_point<V>(t, groups, tptr, tedge, k, y, ldy, dzs, s1pt, ldo, argmax, argmin, lda, scale, mean, invstd, m1, m2,
                      training, dy, lddy);
}

// CSC pass of the backward from the transposed tile plan (ell_tileT.h): rows of the in-edges' sources = (y_i, dz*_i) as the
// two pieces, the selected slot bytes (dc_edge_max_apply's `arg`) as the staged slot words.  Same sums in the same
// (ascending edge id) order and the same closing expression as edge_bwd_point: bit-identical.
struct EdgeTB {
    static constexpr int TAG = 16;
    static constexpr int CAPT = 248, ECAPT = 2560, WGS = 1;
    static constexpr bool COEF = false, ARG = true;
    static constexpr int NST = 1;
    const float* in; long ldj, hs; const unsigned char* arg; long lda;      // in = y, in + hs = dzs (same row stride)
    const float* s1pt; const float *scale, *mean, *invstd, *m1, *m2; int k, training; float* dy; long lddy;
    dcell::Vec<4> sel, T; float indeg;
    __device__ void init() { sel = dcell::vzero<4>(); T = dcell::vzero<4>(); indeg = 0.f; }
    __device__ void step(int s, dcell::G2, const dcell::Vec<4>& yi, const dcell::Vec<4>& dz, unsigned w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sel.v[q] += ((w >> (8 * q)) & 0xffu) == (unsigned)s ? dz.v[q] : 0.f;
            T.v[q] += yi.v[q];
        }
        indeg += 1.f;
    }
    __device__ void finish(long j, int c0) {
        const dcell::Vec<4> yj = dcell::vload<4>(in + j * ldj + c0), dzj = dcell::vload<4>(in + hs + j * ldj + c0),
                            s1 = dcell::vload<4>(s1pt + j * ldj + c0);
        dcell::Vec<4> out;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q;
            float g = sel.v[q] - dzj.v[q];
            if (training) {
                const float col_hat = (indeg * yj.v[q] - T.v[q] - indeg * mean[c]) * invstd[c];
                const float row_hat = (s1.v[q] - (float)k * mean[c]) * invstd[c];
                g -= m1[c] * (indeg - (float)k) + m2[c] * (col_hat - row_hat);
            }
            out.v[q] = scale[c] * g;
        }
        dctileT::st16<dctileT::ST>(dy + j * lddy + c0, out);
    }
};
}  // namespace

// Gather pass over y[Nt,C] (= Linear(x), no bias): per point amax/amin of a_e = y_j - y_i with
// first-extremal slots, s1pt = sum_s a_e; when compute_stats != 0 also the BatchNorm statistics over
// all E edges -> mean, invstd, scale, shift (+ running statistics).  amax/amin/s1pt are [Nt,C]
// contiguous, argmax/argmin uint8 [Nt,C].  Workspace: dc_bn_workspace_bytes(Nt, C).
DC_EXPORT int dc_edge_gather_stats(const float* y, int64_t ldy, const int32_t* nbr, int32_t n, int32_t k, int32_t C,
                                   int32_t compute_stats, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, float* amax, float* amin,
                                   uint8_t* argmax, uint8_t* argmin, float* s1pt, float* mean, float* invstd,
                                   float* scale, float* shift, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(y && nbr && amax && amin && argmax && argmin && s1pt, "dc_edge_gather_stats: null pointer");
    DC_REQUIRE(n >= 1 && k >= 1 && k <= 255 && C >= 1 && ldy >= C, "dc_edge_gather_stats: bad size");
    DC_REQUIRE(!compute_stats || (mean && invstd && scale && shift), "dc_edge_gather_stats: null pointer");
    if (!workspace || workspace_bytes < ws_need(n, C)) {
        dc_set_error("dc_edge_gather_stats: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, n, C);
    const bool v4 = C % 4 == 0 && ldy % 4 == 0 && al16(y);
    const EdgeStatsF<4> f4{y, (long)ldy, nbr, k, amax, amin, argmax, argmin, s1pt, (long)C, (long)C};
    const EdgeStatsF<1> f1{y, (long)ldy, nbr, k, amax, amin, argmax, argmin, s1pt, (long)C, (long)C};
    if (compute_stats) {   // statistics over all E = n*k edges
        const BnFin fin{(long)n * k, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
        if (v4) run_colreduce<4>(f4, n, C, w, s, fin, COLRED_BLOCKS_MAX); else run_colreduce<1>(f1, n, C, w, s, fin, COLRED_BLOCKS_MAX);
    } else {
        if (v4) run_colreduce<4>(f4, n, C, w, s, NoFin{}, COLRED_BLOCKS_MAX); else run_colreduce<1>(f1, n, C, w, s, NoFin{}, COLRED_BLOCKS_MAX);
    }
    DC_CHECK_LAUNCH("dc_edge_gather_stats");
    return DC_OK;
}

// out[i,c] = leaky_slope(scale_c * (scale_c >= 0 ? amax : amin) + shift_c); arg (may be NULL) = slot
static int edge_max_apply(const char* name, const float* amax, const float* amin, const uint8_t* argmax, const uint8_t* argmin,
                          int32_t n, int32_t C, const float* scale, const float* shift, float slope, float* out, int64_t ldo,
                          uint8_t* arg, const float* h2, int64_t ldh2, const float* scale2, const float* shift2, float slope2,
                          float* out2, int64_t ldo2, void* stream) {
    if (!(amax && amin && argmax && argmin && scale && shift && out)) {
        dc_set_error("%s: null pointer", name);
        return DC_ERR_ARG;
    }
    if (!(n >= 0 && C >= 1 && ldo >= C && slope >= 0.f && (!h2 || (ldh2 >= C && (!out2 || ldo2 >= C))))) {
        dc_set_error("%s: bad size / negative slope", name);
        return DC_ERR_ARG;
    }
    if (n == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (C % 4 == 0 && ldo % 4 == 0 && al16(out) && al16(amax) && al16(amin) &&
        (!h2 || (ldh2 % 4 == 0 && al16(h2) && (!out2 || (ldo2 % 4 == 0 && al16(out2))))))
        hipLaunchKernelGGL(edge_apply_kernel<4>, dim3(stream_grid((long)n * (C / 4))), dim3(256), 0, s, amax, amin, argmax,
                           argmin, (long)n, C / 4, scale, shift, slope, out, (long)ldo, arg, h2, (long)ldh2, scale2, shift2, slope2,
                           out2, (long)ldo2);
    else
        hipLaunchKernelGGL(edge_apply_kernel<1>, dim3(stream_grid((long)n * C)), dim3(256), 0, s, amax, amin, argmax,
                           argmin, (long)n, C, scale, shift, slope, out, (long)ldo, arg, h2, (long)ldh2, scale2, shift2, slope2, out2,
                           (long)ldo2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dc_set_error("%s: %s", name, hipGetErrorString(e));
        return DC_ERR_LAUNCH;
    }
    return DC_OK;
}

DC_EXPORT int dc_edge_max_apply(const float* amax, const float* amin, const uint8_t* argmax, const uint8_t* argmin,
                                int32_t n, int32_t C, const float* scale, const float* shift, float slope, float* out,
                                int64_t ldo, uint8_t* arg, void* stream) {
    return edge_max_apply("dc_edge_max_apply", amax, amin, argmax, argmin, n, C, scale, shift, slope, out, ldo, arg, nullptr, 0,
                          nullptr, nullptr, 0.f, nullptr, 0, stream);
}

// The same with the layer's last s_mlp block in its epilogue (round 6, as dc_knn_max_affine_residual for the other layers):
// out = act2(scale2 h2 + shift2) + x_max, second copy in out2 (may be NULL).  Same bits as dc_edge_max_apply + dc_bn_act2(residual).
DC_EXPORT int dc_edge_max_apply_residual(const float* amax, const float* amin, const uint8_t* argmax, const uint8_t* argmin,
                                         int32_t n, int32_t C, const float* scale, const float* shift, float slope,
                                         const float* h2, int64_t ldh2, const float* scale2, const float* shift2, float slope2,
                                         float* out, int64_t ldo, float* out2, int64_t ldo2, uint8_t* arg, void* stream) {
    if (!(h2 && scale2 && shift2)) {
        dc_set_error("dc_edge_max_apply_residual: null pointer");
        return DC_ERR_ARG;
    }
    return edge_max_apply("dc_edge_max_apply_residual", amax, amin, argmax, argmin, n, C, scale, shift, slope, out, ldo, arg, h2, ldh2,
                          scale2, shift2, slope2, out2, ldo2, stream);
}

// Backward of the pair above: dy[Nt,C] (gradient w.r.t. y = Linear(x)), dgamma, dbeta.
// dzs is an [Nt,C] scratch tensor.  Workspace: dc_bn_workspace_bytes(Nt, C).
DC_EXPORT int dc_edge_max_backward(const float* dout, int64_t lddo, const float* y, int64_t ldy, const int32_t* tptr,
                                   const int32_t* tedge, int32_t n, int32_t k, int32_t C, const float* amax,
                                   const float* amin, const uint8_t* argmax, const uint8_t* argmin, const float* s1pt,
                                   const float* scale, const float* shift, const float* mean, const float* invstd,
                                   float slope, int32_t training, float* dzs, float* dy, int64_t lddy, float* dgamma,
                                   float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(dout && y && tptr && tedge && amax && amin && argmax && argmin && s1pt && scale && shift && mean &&
                   invstd && dzs && dy, "dc_edge_max_backward: null pointer");
    DC_REQUIRE(n >= 1 && k >= 1 && k <= 255 && C >= 1 && lddo >= C && ldy >= C && lddy >= C, "dc_edge_max_backward: bad size");
    if (!workspace || workspace_bytes < ws_need(n, C)) {
        dc_set_error("dc_edge_max_backward: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, n, C);
    const bool v4 = C % 4 == 0 && lddo % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && al16(dout) && al16(y) && al16(dy);
    const BwdFin fin{(long)n * k, dgamma, dbeta, w.m1, w.m2};   // m1, m2 are means over all E edges
    if (v4)
        run_colreduce<4>(EdgeBwdF<4>{dout, amax, amin, scale, shift, mean, invstd, (long)lddo, (long)C, slope, dzs}, n, C, w, s, fin);
    else
        run_colreduce<1>(EdgeBwdF<1>{dout, amax, amin, scale, shift, mean, invstd, (long)lddo, (long)C, slope, dzs}, n, C, w, s, fin);
    const int remap = dc_option(DC_OPT_XCD_REMAP);
    if (v4) {
        const long total = (long)n * (C / 4);
        hipLaunchKernelGGL(edge_bwd_kernel<4>, dim3(dc_cdiv(total, 256)), dim3(256), 0, s, total, C / 4, remap, tptr, tedge, k,
                           y, (long)ldy, dzs, s1pt, (long)C, argmax, argmin, (long)C, scale, mean, invstd, w.m1, w.m2,
                           training, dy, (long)lddy);
    } else {
        const long total = (long)n * C;
        hipLaunchKernelGGL(edge_bwd_kernel<1>, dim3(dc_cdiv(total, 256)), dim3(256), 0, s, total, C, remap, tptr, tedge, k, y,
                           (long)ldy, dzs, s1pt, (long)C, argmax, argmin, (long)C, scale, mean, invstd, w.m1, w.m2,
                           training, dy, (long)lddy);
    }
    DC_CHECK_LAUNCH("dc_edge_max_backward");
    return DC_OK;
}

// dc_edge_max_backward with the CSC pass running from the transposed tile plan (tile_plan.h second half, ell_tileT.h).
// argsel = the selected slot per (point, channel) as written by dc_edge_max_apply(arg != NULL); y, dzs contiguous [Nt, C]
// (row stride C), C % 64 == 0, 16-byte aligned.  Same results bit for bit.
DC_EXPORT int dc_edge_max_backward_tiled(const float* dout, int64_t lddo, const float* y, const int32_t* planT, int32_t n,
                                         int32_t num_clouds, int32_t num_tiles, int32_t k, int32_t P, int32_t C,
                                         const float* amax, const float* amin, const uint8_t* argsel, const float* s1pt,
                                         const float* scale, const float* shift, const float* mean, const float* invstd,
                                         float slope, int32_t training, float* dzs, float* dy, int64_t lddy, float* dgamma,
                                         float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(dout && y && planT && amax && amin && argsel && s1pt && scale && shift && mean && invstd && dzs && dy,
               "dc_edge_max_backward_tiled: null pointer");
    DC_REQUIRE(n >= 1 && num_clouds >= 1 && num_tiles >= 1 && k >= 2 && k % 2 == 0 && k <= 64 && (P == 32 || P == 64) && P * k <= 2048 &&
                   lddo >= C && lddy >= C, "dc_edge_max_backward_tiled: bad size");
    DC_REQUIRE(dctile::eligible(C, {(long)lddo, (long)lddy}, {dout, y, dzs, dy, amax, amin, s1pt, argsel}) &&
                   ((dzs - y) % 4 == 0), "dc_edge_max_backward_tiled: needs C %% 64 == 0 and 16-byte aligned rows");
    if (!workspace || workspace_bytes < ws_need(n, C)) {
        dc_set_error("dc_edge_max_backward_tiled: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, n, C);
    const BwdFin fin{(long)n * k, dgamma, dbeta, w.m1, w.m2};   // m1, m2 are means over all E edges
    run_colreduce<4>(EdgeBwdF<4>{dout, amax, amin, scale, shift, mean, invstd, (long)lddo, (long)C, slope, dzs}, n, C, w, s, fin);
    const DcTilePlanT L = dc_tile_plan_T_layout(n, num_clouds, num_tiles, k, P);
    dctileT::launch<2>(L, planT, nullptr, C,
                       EdgeTB{y, (long)C, (long)(dzs - y), argsel, (long)C, s1pt, scale, mean, invstd, w.m1, w.m2, k, training, dy, (long)lddy},
                       s);
    DC_CHECK_LAUNCH("dc_edge_max_backward_tiled");
    return DC_OK;
}

// ---- general form of the centralised edge MLP (any depth / width / aggregation): the edge tensor is materialised -------------
// x_edge = x[col] - x[row] (deltaconv/nn/deltaconv.py:50), the MLP runs on its E = n k rows through the ordinary block kernels,
// scatter(h, row, reduce=aggr) (deltaconv.py:52) is a reduction over the k CONSECUTIVE rows of a point (edges are centre-major).
// The two reference models that use a centralised first layer never get here (depth 1: the analytic form above; depth 2 x 64
// channels: edge2.hip); this is the path of every other shape, on hand-written kernels as well.
namespace {
template <int V>
__global__ __launch_bounds__(256) void edge_diff_kernel(const float* __restrict__ x, long ldx, const int* __restrict__ nbr, long E,
                                                        int k, int groups, float* __restrict__ out) {
    const long C = (long)groups * V;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < E * groups; t += (long)gridDim.x * 256) {
        const long e = t / groups;
        const int c0 = (int)(t % groups) * V;
        const long i = e / k;
        const FV<V> a = ldv<V>(x + (long)nbr[e] * ldx + c0), b = ldv<V>(x + i * ldx + c0);
        FV<V> o;
#pragma unroll
        for (int q = 0; q < V; ++q) o.v[q] = a.v[q] - b.v[q];
        *reinterpret_cast<FV<V>*>(out + e * C + c0) = o;
    }
}
// dx[j] = sum over in-edges (ascending edge id) dE[e] - sum_s dE[(j, s)]
template <int V>
__global__ __launch_bounds__(256) void edge_diff_bwd_kernel(const float* __restrict__ dE, const int* __restrict__ tptr,
                                                            const int* __restrict__ tedge, long n, int k, int groups,
                                                            float* __restrict__ dx, long lddx) {
    const long C = (long)groups * V;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * groups) return;
    const long j = t / groups;
    const int c0 = (int)(t % groups) * V;
    FV<V> acc;
#pragma unroll
    for (int q = 0; q < V; ++q) acc.v[q] = 0.f;
    for (int p = tptr[j]; p < tptr[j + 1]; ++p) {
        const FV<V> d = ldv<V>(dE + (long)tedge[p] * C + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) acc.v[q] += d.v[q];
    }
    for (int s = 0; s < k; ++s) {
        const FV<V> d = ldv<V>(dE + (j * k + s) * C + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) acc.v[q] -= d.v[q];
    }
    *reinterpret_cast<FV<V>*>(dx + j * lddx + c0) = acc;
}
// mode 0 max, 1 min (first extremal slot), 2 sum, 3 mean over the k consecutive rows of a point
template <int V>
__global__ __launch_bounds__(256) void seg_reduce_kernel(const float* __restrict__ h, long n, int k, int groups, int mode,
                                                         float* __restrict__ out, unsigned char* __restrict__ arg) {
    const long C = (long)groups * V;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n * groups; t += (long)gridDim.x * 256) {
        const long i = t / groups;
        const int c0 = (int)(t % groups) * V;
        FV<V> acc = ldv<V>(h + (i * k) * C + c0);
        unsigned char sl[V];
#pragma unroll
        for (int q = 0; q < V; ++q) sl[q] = 0;
        for (int s = 1; s < k; ++s) {
            const FV<V> v = ldv<V>(h + (i * k + s) * C + c0);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                if (mode >= 2) acc.v[q] += v.v[q];
                else {
                    const bool take = mode == 0 ? v.v[q] > acc.v[q] : v.v[q] < acc.v[q];
                    acc.v[q] = take ? v.v[q] : acc.v[q];
                    sl[q] = take ? (unsigned char)s : sl[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            if (mode == 3) acc.v[q] /= (float)k;
            if (arg && mode < 2) arg[i * C + c0 + q] = sl[q];
        }
        *reinterpret_cast<FV<V>*>(out + i * C + c0) = acc;
    }
}
template <int V>
__global__ __launch_bounds__(256) void seg_reduce_bwd_kernel(const float* __restrict__ dout, long lddo,
                                                             const unsigned char* __restrict__ arg, long E, int k, int groups,
                                                             int mode, float* __restrict__ dh) {
    const long C = (long)groups * V;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < E * groups; t += (long)gridDim.x * 256) {
        const long e = t / groups;
        const int c0 = (int)(t % groups) * V;
        const long i = e / k;
        const int s = (int)(e - i * k);
        const FV<V> g = ldv<V>(dout + i * lddo + c0);
        FV<V> o;
#pragma unroll
        for (int q = 0; q < V; ++q)
            o.v[q] = mode == 2 ? g.v[q] : mode == 3 ? g.v[q] / (float)k : (arg[i * C + c0 + q] == (unsigned char)s ? g.v[q] : 0.f);
        *reinterpret_cast<FV<V>*>(dh + e * C + c0) = o;
    }
}
}  // namespace

// out[e, :] = x[nbr[e], :] - x[e / k, :]   ([n k, C] contiguous; x row stride ldx): the edge tensor of deltaconv.py:50
DC_EXPORT int dc_edge_diff(const float* x, int64_t ldx, const int32_t* nbr, int32_t n, int32_t k, int32_t C, float* out,
                           void* stream) {
    DC_REQUIRE(x && nbr && out, "dc_edge_diff: null pointer");
    DC_REQUIRE(n >= 1 && k >= 1 && C >= 1 && ldx >= C, "dc_edge_diff: bad size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long E = (long)n * k;
    if (C % 4 == 0 && ldx % 4 == 0 && al16(x) && al16(out))
        hipLaunchKernelGGL(edge_diff_kernel<4>, dim3(stream_grid(E * (C / 4))), dim3(256), 0, s, x, (long)ldx, nbr, E, k, C / 4, out);
    else
        hipLaunchKernelGGL(edge_diff_kernel<1>, dim3(stream_grid(E * C)), dim3(256), 0, s, x, (long)ldx, nbr, E, k, C, out);
    DC_CHECK_LAUNCH("dc_edge_diff");
    return DC_OK;
}
// its transpose: dx[j] = sum_{in-edges of j, ascending edge id} dE[e] - sum_s dE[j k + s]   (CSC from dc_csc_build)
DC_EXPORT int dc_edge_diff_backward(const float* dE, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, int32_t C,
                                    float* dx, int64_t lddx, void* stream) {
    DC_REQUIRE(dE && tptr && tedge && dx, "dc_edge_diff_backward: null pointer");
    DC_REQUIRE(n >= 1 && k >= 1 && C >= 1 && lddx >= C, "dc_edge_diff_backward: bad size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (C % 4 == 0 && lddx % 4 == 0 && al16(dE) && al16(dx))
        hipLaunchKernelGGL(edge_diff_bwd_kernel<4>, dim3(dc_cdiv((long)n * (C / 4), 256)), dim3(256), 0, s, dE, tptr, tedge, (long)n, k,
                           C / 4, dx, (long)lddx);
    else
        hipLaunchKernelGGL(edge_diff_bwd_kernel<1>, dim3(dc_cdiv((long)n * C, 256)), dim3(256), 0, s, dE, tptr, tedge, (long)n, k, C, dx,
                           (long)lddx);
    DC_CHECK_LAUNCH("dc_edge_diff_backward");
    return DC_OK;
}
// scatter(h, row, reduce=...) for centre-major edges (deltaconv.py:52): reduction over the k consecutive rows of every point.
// mode 0 = max, 1 = min (arg = first extremal slot, uint8 [n, C], may be NULL), 2 = sum / add, 3 = mean.  h [n k, C], out [n, C].
DC_EXPORT int dc_seg_reduce(const float* h, int32_t n, int32_t k, int32_t C, int32_t mode, float* out, uint8_t* arg, void* stream) {
    DC_REQUIRE(h && out, "dc_seg_reduce: null pointer");
    DC_REQUIRE(n >= 1 && k >= 1 && k <= 255 && C >= 1 && mode >= 0 && mode <= 3, "dc_seg_reduce: bad size / mode");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (C % 4 == 0 && al16(h) && al16(out))
        hipLaunchKernelGGL(seg_reduce_kernel<4>, dim3(stream_grid((long)n * (C / 4))), dim3(256), 0, s, h, (long)n, k, C / 4, mode, out, arg);
    else
        hipLaunchKernelGGL(seg_reduce_kernel<1>, dim3(stream_grid((long)n * C)), dim3(256), 0, s, h, (long)n, k, C, mode, out, arg);
    DC_CHECK_LAUNCH("dc_seg_reduce");
    return DC_OK;
}
DC_EXPORT int dc_seg_reduce_backward(const float* dout, int64_t lddo, const uint8_t* arg, int32_t n, int32_t k, int32_t C,
                                     int32_t mode, float* dh, void* stream) {
    DC_REQUIRE(dout && dh && (mode >= 2 || arg), "dc_seg_reduce_backward: null pointer");
    DC_REQUIRE(n >= 1 && k >= 1 && k <= 255 && C >= 1 && lddo >= C && mode >= 0 && mode <= 3, "dc_seg_reduce_backward: bad size / mode");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long E = (long)n * k;
    if (C % 4 == 0 && lddo % 4 == 0 && al16(dout) && al16(dh))
        hipLaunchKernelGGL(seg_reduce_bwd_kernel<4>, dim3(stream_grid(E * (C / 4))), dim3(256), 0, s, dout, (long)lddo, arg, E, k, C / 4,
                           mode, dh);
    else
        hipLaunchKernelGGL(seg_reduce_bwd_kernel<1>, dim3(stream_grid(E * C)), dim3(256), 0, s, dout, (long)lddo, arg, E, k, C, mode, dh);
    DC_CHECK_LAUNCH("dc_seg_reduce_backward");
    return DC_OK;
}
