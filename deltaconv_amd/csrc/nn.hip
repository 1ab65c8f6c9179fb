// Fused BatchNorm(+LeakyReLU, +residual) and vector non-linearity kernels for the scalar / vector
// MLP stream.  Replace the ATen chain the reference runs per block
//   Linear -> BatchNorm1d -> LeakyReLU(0.2)         (/root/reference/deltaconv/nn/mlp.py:7-11,
//                                                    nn/nonlin.py:11-35)
//   Linear -> VectorNonLin(BatchNorm1d)             (nn/mlp.py:13-17, nn/nonlin.py:38-86)
// around the dense GEMM (which stays a library GEMM for now).  Everything here is HBM-bound
// streaming over [rows, C] matrices:
//   stats   : read h once                       -> 4*C*R bytes
//   apply   : read h (+residual), write y       -> 8..12*C*R bytes
//   bwd     : reduce (read dy, h) + apply (read dy, h, write dh) -> 20*C*R bytes
// Column reductions are two-stage and ordered (per-block partials in double, then a fixed-order
// sum): bit-reproducible, no fp atomics.  Element formulas: nn_math.h.
#include "common.h"
#include "nn_math.h"
#include "colreduce.h"

namespace {

using namespace dcnn;

using namespace dccol;

// ---- functors ---------------------------------------------------------------------------------
template <int V>
struct StatsF {  // x, x^2
    const float* h; long ld;
    __device__ void operator()(long r, int c0, double (&t)[2][V]) const {
        const FV<V> x = ldv<V>(h + r * ld + c0);
#pragma unroll
        for (int j = 0; j < V; ++j) { t[0][j] = (double)x.v[j]; t[1][j] = (double)x.v[j] * (double)x.v[j]; }
    }
};
template <int V>
struct BnBwdF {  // dz, dz*xhat
    const float *dy, *h, *scale, *shift, *mean, *invstd; long lddy, ldh; float slope;
    __device__ void operator()(long r, int c0, double (&t)[2][V]) const {
        const FV<V> g = ldv<V>(dy + r * lddy + c0), x = ldv<V>(h + r * ldh + c0);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float a, b;
            bn_bwd_terms(g.v[j], x.v[j], scale[c0 + j], shift[c0 + j], mean[c0 + j], invstd[c0 + j], slope, a, b);
            t[0][j] = a; t[1][j] = b;
        }
    }
};
// vector block: "row" = point i; input rows 2i, 2i+1 of pq.  combine 1: [P | Q] blocked (2*co columns);
// combine 2: P and Q interleaved (column 2c = P_c, 2c+1 = Q_c) -- what `v_cat @ W.view(2co, K)^T` produces
// from the reference's [co, 2K] weight without re-stacking it.
template <int V>
__device__ __forceinline__ void vn_load_y(const float* in, long ld, long i, int c0, int co, int combine, FV<V>& yu,
                                          FV<V>& yv) {
    if (combine == 2) {
        const float* ru = in + (2 * i) * ld + 2 * c0;
        const float* rv = in + (2 * i + 1) * ld + 2 * c0;
        const FV<V> u0 = ldv<V>(ru), u1 = ldv<V>(ru + V), v0 = ldv<V>(rv), v1 = ldv<V>(rv + V);
        float eu[2 * V], ev[2 * V];
#pragma unroll
        for (int j = 0; j < V; ++j) { eu[j] = u0.v[j]; eu[V + j] = u1.v[j]; ev[j] = v0.v[j]; ev[V + j] = v1.v[j]; }
#pragma unroll
        for (int j = 0; j < V; ++j) vn_combine(eu[2 * j], eu[2 * j + 1], ev[2 * j], ev[2 * j + 1], yu.v[j], yv.v[j]);
        return;
    }
    const FV<V> pu = ldv<V>(in + (2 * i) * ld + c0), pv = ldv<V>(in + (2 * i + 1) * ld + c0);
    if (combine) {
        const FV<V> qu = ldv<V>(in + (2 * i) * ld + co + c0), qv = ldv<V>(in + (2 * i + 1) * ld + co + c0);
#pragma unroll
        for (int j = 0; j < V; ++j) vn_combine(pu.v[j], qu.v[j], pv.v[j], qv.v[j], yu.v[j], yv.v[j]);
    } else {
        yu = pu;
        yv = pv;
    }
}
template <int V>
struct VnStatsF {  // n, n^2
    const float* in; long ld; int co, combine;
    __device__ void operator()(long i, int c0, double (&t)[2][V]) const {
        FV<V> yu, yv;
        vn_load_y<V>(in, ld, i, c0, co, combine, yu, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) { const double n = vn_norm(yu.v[j], yv.v[j]); t[0][j] = n; t[1][j] = n * n; }
    }
};
template <int V>
struct VnBwdF {  // dz, dz*nhat
    const float *in, *dout, *scale, *shift, *mean, *invstd; long ld, lddo; int co, combine;
    __device__ void operator()(long i, int c0, double (&t)[2][V]) const {
        FV<V> yu, yv;
        vn_load_y<V>(in, ld, i, c0, co, combine, yu, yv);
        const FV<V> du = ldv<V>(dout + (2 * i) * lddo + c0), dv = ldv<V>(dout + (2 * i + 1) * lddo + c0);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float a, b;
            vn_bwd_terms(yu.v[j], yv.v[j], du.v[j], dv.v[j], scale[c0 + j], shift[c0 + j], mean[c0 + j],
                         invstd[c0 + j], a, b);
            t[0][j] = a; t[1][j] = b;
        }
    }
};

// ---- streaming applies ---------------------------------------------------------------------
// 2-D tiles like the reductions: a thread owns V fixed columns (its per-channel coefficients live in
// registers, loaded once) and walks rows rl, rl + RT, ... of its row chunk.  (A flat grid-stride
// version re-loaded 2-7 coefficient vectors per element: 3.5x the load instructions of the data.)
template <int V, class BODY>
__global__ __launch_bounds__(TPB) void tile_kernel(long R, int C, int rpc, BODY body) {
    const int cgl = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c0 = (blockIdx.y * CT + cgl) * V;
    if (c0 >= C) return;
    body.init(c0);
    const long r0 = (long)blockIdx.x * rpc;
    const long r1 = min(r0 + rpc, R);
#ifndef DC_TILE_UNROLL
#define DC_TILE_UNROLL 2
#endif
#pragma unroll DC_TILE_UNROLL
    for (long r = r0 + rl; r < r1; r += RT) body.row(r, c0);
}

template <int V>
struct BnActBody {
    const float *h, *scale, *shift, *res; float* y; long ldh, ldr, ldy; float slope;
    float* y2; long ldy2;   // optional second destination (the layer output also lands in the concat buffer)
    float sc[V], sh[V];
    __device__ void init(int c0) {
#pragma unroll
        for (int j = 0; j < V; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
    }
    __device__ void row(long r, int c0) {
        const FV<V> x = ldv<V>(h + r * ldh + c0);
        FV<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = act(fmaf(sc[j], x.v[j], sh[j]), slope);
        if (res) {
            const FV<V> rr = ldv<V>(res + r * ldr + c0);
#pragma unroll
            for (int j = 0; j < V; ++j) o.v[j] += rr.v[j];
        }
        stv<V>(y + r * ldy + c0, o);
        if (y2) stv<V>(y2 + r * ldy2 + c0, o);
    }
};

template <int V>
struct BnActBwdBody {
    const float *dy, *h, *scale, *shift, *mean, *invstd, *gamma, *m1, *m2; float* dh;
    long lddy, ldh, lddh; float slope; int training;
    float sc[V], sh[V], mu[V], is[V], gi[V], a1[V], a2[V];
    __device__ void init(int c0) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = c0 + j;
            sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
            gi[j] = (gamma ? gamma[c] : 1.f) * invstd[c]; a1[j] = m1[c]; a2[j] = m2[c];
        }
    }
    __device__ void row(long r, int c0) {
        const FV<V> g = ldv<V>(dy + r * lddy + c0), x = ldv<V>(h + r * ldh + c0);
        FV<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j)
            o.v[j] = bn_bwd_dh(g.v[j], x.v[j], sc[j], sh[j], mu[j], is[j], slope, gi[j], a1[j], a2[j], training);
        stv<V>(dh + r * lddh + c0, o);
    }
};

template <int V>
struct VnApplyBody {
    const float *in, *scale, *shift; float* out; long ld, ldo; int co, combine;
    float sc[V], sh[V];
    __device__ void init(int c0) {
#pragma unroll
        for (int j = 0; j < V; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
    }
    __device__ void row(long i, int c0) {
        FV<V> yu, yv;
        vn_load_y<V>(in, ld, i, c0, co, combine, yu, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float s = vn_scale(vn_norm(yu.v[j], yv.v[j]), sc[j], sh[j]);
            yu.v[j] *= s;
            yv.v[j] *= s;
        }
        stv<V>(out + (2 * i) * ldo + c0, yu);
        stv<V>(out + (2 * i + 1) * ldo + c0, yv);
    }
};

template <int V>
struct VnBwdBody {
    const float *in, *dout, *scale, *shift, *mean, *invstd, *gamma, *m1, *m2; float* din;
    long ld, lddo, lddi; int co, combine, training;
    float sc[V], sh[V], mu[V], is[V], gi[V], a1[V], a2[V];
    __device__ void init(int c0) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = c0 + j;
            sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
            gi[j] = (gamma ? gamma[c] : 1.f) * invstd[c]; a1[j] = m1[c]; a2[j] = m2[c];
        }
    }
    __device__ void row(long i, int c0) {
        FV<V> yu, yv, gu, gv;
        vn_load_y<V>(in, ld, i, c0, co, combine, yu, yv);
        const FV<V> du = ldv<V>(dout + (2 * i) * lddo + c0), dv = ldv<V>(dout + (2 * i + 1) * lddo + c0);
#pragma unroll
        for (int j = 0; j < V; ++j)
            vn_bwd_dy(yu.v[j], yv.v[j], du.v[j], dv.v[j], sc[j], sh[j], mu[j], is[j], gi[j], a1[j], a2[j], training,
                      gu.v[j], gv.v[j]);
        // d[P|Q]: row u = [dy_u | dy_v], row v = [dy_v | -dy_u]   (transpose of vn_combine)
        if (combine == 2) {                      // interleaved pairs (dP_c, dQ_c)
            float eu[2 * V], ev[2 * V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                eu[2 * j] = gu.v[j]; eu[2 * j + 1] = gv.v[j];
                ev[2 * j] = gv.v[j]; ev[2 * j + 1] = -gu.v[j];
            }
            FV<V> a, b, c, d;
#pragma unroll
            for (int j = 0; j < V; ++j) { a.v[j] = eu[j]; b.v[j] = eu[V + j]; c.v[j] = ev[j]; d.v[j] = ev[V + j]; }
            float* ru = din + (2 * i) * lddi + 2 * c0;
            float* rv = din + (2 * i + 1) * lddi + 2 * c0;
            stv<V>(ru, a); stv<V>(ru + V, b); stv<V>(rv, c); stv<V>(rv + V, d);
            return;
        }
        stv<V>(din + (2 * i) * lddi + c0, gu);
        stv<V>(din + (2 * i + 1) * lddi + c0, gv);
        if (combine) {
            FV<V> ngu;
#pragma unroll
            for (int j = 0; j < V; ++j) ngu.v[j] = -gu.v[j];
            stv<V>(din + (2 * i) * lddi + co + c0, gv);
            stv<V>(din + (2 * i + 1) * lddi + co + c0, ngu);
        }
    }
};

// ---- embedding head: BatchNorm + LeakyReLU fused with the per-cloud max / mean pooling -------------
// (reference: MLP([sum c, 1024]) -> global_max_pool | global_mean_pool, models/deltanet_classification.py:
// 42-49; -> global_max_pool, deltanet_segmentation.py:58-61).  The [Nt,1024] activation is never written:
// forward reads h once and emits pooled[B, 2C] (+ the arg-max row), backward rebuilds
// dy[i,c] = dmax[b,c]*[i == arg[b,c]] + dmean[b,c]/N on the fly inside the BN-backward passes.
// Equal-size clouds (N rows each).
template <int V>
__global__ __launch_bounds__(TPB) void pool_fwd_kernel(const float* __restrict__ h, long ldh, int N, int C,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float slope, int with_mean,
                                                       float* __restrict__ pooled, long ldp, int* __restrict__ argmax) {
    __shared__ float smx[RT][CT * V];
    __shared__ float ssm[RT][CT * V];
    __shared__ int sam[RT][CT * V];
    const int cloud = blockIdx.x;
    const int cgl = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c0 = (blockIdx.y * CT + cgl) * V;
    float mx[V], sm[V], sc[V], sh[V];
    int am[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        mx[j] = -INFINITY; sm[j] = 0.f; am[j] = 0;
        sc[j] = c0 < C ? scale[c0 + j] : 0.f; sh[j] = c0 < C ? shift[c0 + j] : 0.f;
    }
    if (c0 < C) {
        const float* base = h + (long)cloud * N * ldh + c0;
        // 64 rows per thread at 1024 points, two workgroups per CU: without the unroll one or two 16-byte loads per thread are in
        // flight (3.5 TB/s); eight of them hoisted: 31.6 -> 22.7 us at [32768, 1024] (same order of the sums and comparisons)
#pragma unroll 8
        for (int r = rl; r < N; r += RT) {
            const FV<V> x = ldv<V>(base + (long)r * ldh);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float y = act(fmaf(sc[j], x.v[j], sh[j]), slope);
                sm[j] += y;
                if (y > mx[j]) { mx[j] = y; am[j] = r; }   // rows ascend: first maximal row of this lane
            }
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { smx[rl][cgl * V + j] = mx[j]; ssm[rl][cgl * V + j] = sm[j]; sam[rl][cgl * V + j] = am[j]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < CT * V; idx += TPB) {
        const int col = blockIdx.y * CT * V + idx;
        if (col >= C) continue;
        float m = smx[0][idx], t = ssm[0][idx];
        int a = sam[0][idx];
#pragma unroll
        for (int rr = 1; rr < RT; ++rr) {   // fixed order; ties -> lowest row
            const float v = smx[rr][idx];
            const int ar = sam[rr][idx];
            if (v > m || (v == m && ar < a)) { m = v; a = ar; }
            t += ssm[rr][idx];
        }
        pooled[(long)cloud * ldp + col] = m;
        if (with_mean) pooled[(long)cloud * ldp + C + col] = t / (float)N;
        argmax[(long)cloud * C + col] = a;
    }
}

// dy rebuilt from the pooled gradients (dp[b, 0:C] = d max, dp[b, C:2C] = d mean).  A thread walks
// ascending rows of fixed columns, so the per-cloud values (d max, d mean / N, arg-max row) are cached
// in registers and reloaded only when the row crosses into the next cloud.
template <int V>
struct PoolDy {
    const float* dp; const int* argmax; long ldp; int N, C, with_mean;
    int cur_b; float dmx[V], dme[V]; int ar[V];
    __device__ void reset() { cur_b = -1; }
    __device__ FV<V> get(long r, int c0) {
        const int b = (int)((unsigned)r / (unsigned)N);   // r < 2^31 (checked on the host)
        if (b != cur_b) {
            cur_b = b;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                dmx[j] = dp[(long)b * ldp + c0 + j];
                dme[j] = with_mean ? dp[(long)b * ldp + C + c0 + j] / (float)N : 0.f;
                ar[j] = argmax[(long)b * C + c0 + j];
            }
        }
        const int row = (int)r - b * N;
        FV<V> g;
#pragma unroll
        for (int j = 0; j < V; ++j) g.v[j] = (ar[j] == row ? dmx[j] : 0.f) + dme[j];
        return g;
    }
};
template <int V>
struct PoolBwdF {  // reduction functor: dz, dz*xhat
    PoolDy<V> dy; const float *h, *scale, *shift, *mean, *invstd; long ldh; float slope; int primed;
    __device__ void operator()(long r, int c0, double (&t)[2][V]) {
        if (!primed) { dy.reset(); primed = 1; }
        const FV<V> g = dy.get(r, c0), x = ldv<V>(h + r * ldh + c0);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float a, b;
            bn_bwd_terms(g.v[j], x.v[j], scale[c0 + j], shift[c0 + j], mean[c0 + j], invstd[c0 + j], slope, a, b);
            t[0][j] = a; t[1][j] = b;
        }
    }
};
template <int V>
struct PoolBwdBody {
    PoolDy<V> dy; const float *h, *scale, *shift, *mean, *invstd, *gamma, *m1, *m2; float* dh;
    long ldh, lddh; float slope; int training;
    float sc[V], sh[V], mu[V], is[V], gi[V], a1[V], a2[V];
    __device__ void init(int c0) {
        dy.reset();
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = c0 + j;
            sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
            gi[j] = (gamma ? gamma[c] : 1.f) * invstd[c]; a1[j] = m1[c]; a2[j] = m2[c];
        }
    }
    __device__ void row(long r, int c0) {
        const FV<V> g = dy.get(r, c0), x = ldv<V>(h + r * ldh + c0);
        FV<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j)
            o.v[j] = bn_bwd_dh(g.v[j], x.v[j], sc[j], sh[j], mu[j], is[j], slope, gi[j], a1[j], a2[j], training);
        stv<V>(dh + r * lddh + c0, o);
    }
};

// y[i] = h[i] + g[cloud of i]: the per-cloud half of the segmentation head's first Linear, computed on B rows, joins the per-point
// half (models/deltanet_segmentation.py; reference: x_max[batch] concatenated in front of the features, deltanet_segmentation.py:59-64)
template <int V>
struct CloudBiasBody {
    const float *h, *g; float* y; long ldh, ldg, ldy, mx;
    __device__ void init(int) {}
    __device__ void row(long r, int c0) {
        const FV<V> x = ldv<V>(h + r * ldh + c0), b = ldv<V>(g + (r / mx) * ldg + c0);
        FV<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = x.v[j] + b.v[j];
        stv<V>(y + r * ldy + c0, o);
    }
};

// Its backward: per-cloud column sums of d h (the gradient of `x_max[batch]`: index_select backward = index_add over the cloud's
// points).  Ordered two-stage reduction like colreduce.h: fp64 partial per (cloud, row chunk, column), then one wave per
// (cloud, column) over the chunks -- bit-reproducible.  grid = (chunks, column tiles, clouds).
template <int V>
__global__ __launch_bounds__(TPB) void cloud_colsum_kernel(const float* __restrict__ x, long ldx, long mx, int C, int chunks,
                                                           int rpc, double* __restrict__ partial) {
    __shared__ double sm[RT][CT * V];
    const int cgl = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c0 = (blockIdx.y * CT + cgl) * V;
    const long cloud = blockIdx.z;
    const long r0 = cloud * mx + (long)blockIdx.x * rpc, r1 = min(r0 + rpc, (cloud + 1) * mx);
    double acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.0;
    if (c0 < C) {
#pragma unroll 2
        for (long r = r0 + rl; r < r1; r += RT) {
            const FV<V> v = ldv<V>(x + r * ldx + c0);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += (double)v.v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) sm[rl][cgl * V + j] = acc[j];
    __syncthreads();
    for (int cl = threadIdx.x; cl < CT * V; cl += TPB) {
        const int col = blockIdx.y * CT * V + cl;
        if (col < C) {
            double t = 0;
#pragma unroll
            for (int rr = 0; rr < RT; ++rr) t += sm[rr][cl];
            partial[(cloud * C + col) * chunks + blockIdx.x] = t;
        }
    }
}
__global__ __launch_bounds__(64) void cloud_colsum_final_kernel(const double* __restrict__ partial, int chunks, int C,
                                                                float* __restrict__ out, long ldo) {
    const int col = blockIdx.x;
    const long cloud = blockIdx.y;
    const double* p = partial + (cloud * C + col) * chunks;
    double t = 0;
    for (int ch = threadIdx.x; ch < chunks; ch += 64) t += p[ch];
    t = dc_wave_sum(t);
    if (threadIdx.x == 0) out[cloud * ldo + col] = (float)t;
}
inline int cloud_colsum_chunks(long num_clouds, long mx, int C) {       // ~768 workgroups, >= RT rows each
    const long coltiles = dc_cdiv(C, CT * 4);
    const long want = std::max<long>(1, 768 / std::max<long>(1, num_clouds * coltiles));
    return (int)std::max<long>(1, std::min<long>(want, dc_cdiv(mx, RT)));
}

template <int V, class BODY>
void run_tile(BODY body, long R, int C, hipStream_t s) {
    const long coltiles = dc_cdiv(C, CT * V);
#ifndef DC_TILE_BLOCKS
#define DC_TILE_BLOCKS 1024     // round 6, same-box A/B of the step (profiles/r06_labs.txt item 6): 1024 beats 2048 / 4096 / 512
#endif
    long rpc = (R * coltiles / DC_TILE_BLOCKS + RT - 1) / RT * RT;      // ~1024 blocks, >= 1 row per thread
    rpc = std::min<long>(std::max<long>(rpc, RT), 1024);
    dim3 grid(dc_cdiv(R, rpc), (unsigned)coltiles);
    hipLaunchKernelGGL((tile_kernel<V, BODY>), grid, dim3(TPB), 0, s, R, C, (int)rpc, body);
}

}  // namespace

DC_EXPORT size_t dc_bn_workspace_bytes(int64_t rows, int32_t C) { return ws_need(rows, C); }

#define DC_WS_CHECK(name, R, C)                                                   \
    if (!workspace || workspace_bytes < ws_need(R, C)) {                          \
        dc_set_error(name ": workspace too small (%zu < %zu)", workspace_bytes, ws_need(R, C)); \
        return DC_ERR_WORKSPACE;                                                  \
    }

// Batch statistics of h[R,C] -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale;
// running_mean / running_var (may be NULL) updated with `momentum` (unbiased variance).
DC_EXPORT int dc_bn_stats(const float* h, int64_t R, int32_t C, int64_t ldh, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, float* mean,
                          float* invstd, float* scale, float* shift, void* workspace, size_t workspace_bytes,
                          void* stream) {
    DC_REQUIRE(h && mean && invstd && scale && shift, "dc_bn_stats: null pointer");
    DC_REQUIRE(R >= 1 && C >= 1 && ldh >= C, "dc_bn_stats: bad size");
    DC_WS_CHECK("dc_bn_stats", R, C)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, R, C);
    const BnFin fin{(long)R, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
    if (C % 4 == 0 && ldh % 4 == 0 && al16(h))
        run_colreduce<4>(StatsF<4>{h, (long)ldh}, R, C, w, s, fin);
    else
        run_colreduce<1>(StatsF<1>{h, (long)ldh}, R, C, w, s, fin);
    DC_CHECK_LAUNCH("dc_bn_stats");
    return DC_OK;
}

// Inference coefficients from the running statistics.
DC_EXPORT int dc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, int32_t C, float* mean, float* invstd,
                                float* scale, float* shift, void* stream) {
    DC_REQUIRE(running_mean && running_var && mean && invstd && scale && shift, "dc_bn_eval_coeffs: null pointer");
    DC_REQUIRE(C >= 1, "dc_bn_eval_coeffs: bad size");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(dc_cdiv(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), gamma,
                       beta, running_mean, running_var, eps, C, mean, invstd, scale, shift);
    DC_CHECK_LAUNCH("dc_bn_eval_coeffs");
    return DC_OK;
}

// y = leaky_slope(scale*h + shift) (+ residual).  slope = 1 -> identity, 0 -> ReLU.
static int bn_act_impl(const float* h, int64_t R, int32_t C, int64_t ldh, const float* scale, const float* shift,
                       float slope, const float* residual, int64_t ldr, float* y, int64_t ldy, float* y2, int64_t ldy2,
                       void* stream) {
    DC_REQUIRE(h && scale && shift && y, "dc_bn_act: null pointer");
    DC_REQUIRE(R >= 0 && C >= 1 && ldh >= C && ldy >= C && (!residual || ldr >= C) && (!y2 || ldy2 >= C),
               "dc_bn_act: bad size");
    if (R == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = C % 4 == 0 && ldh % 4 == 0 && ldy % 4 == 0 && al16(h) && al16(y) &&
                    (!residual || (ldr % 4 == 0 && al16(residual))) && (!y2 || (ldy2 % 4 == 0 && al16(y2)));
    if (v4)
        run_tile<4>(BnActBody<4>{h, scale, shift, residual, y, (long)ldh, (long)ldr, (long)ldy, slope, y2, (long)ldy2},
                    R, C, s);
    else
        run_tile<1>(BnActBody<1>{h, scale, shift, residual, y, (long)ldh, (long)ldr, (long)ldy, slope, y2, (long)ldy2},
                    R, C, s);
    DC_CHECK_LAUNCH("dc_bn_act");
    return DC_OK;
}

DC_EXPORT int dc_bn_act(const float* h, int64_t R, int32_t C, int64_t ldh, const float* scale, const float* shift,
                        float slope, const float* residual, int64_t ldr, float* y, int64_t ldy, void* stream) {
    return bn_act_impl(h, R, C, ldh, scale, shift, slope, residual, ldr, y, ldy, nullptr, 0, stream);
}

// Same, writing the result to two destinations (y2 may be NULL).
DC_EXPORT int dc_bn_act2(const float* h, int64_t R, int32_t C, int64_t ldh, const float* scale, const float* shift,
                         float slope, const float* residual, int64_t ldr, float* y, int64_t ldy, float* y2,
                         int64_t ldy2, void* stream) {
    return bn_act_impl(h, R, C, ldh, scale, shift, slope, residual, ldr, y, ldy, y2, ldy2, stream);
}

// y[n, C] (ldy) = h[n, C] (ldh) + g[i / mx, C] (ldg): equal-size clouds of mx points, y may be h.
DC_EXPORT int dc_cloud_bias_add(const float* h, int64_t ldh, const float* g, int64_t ldg, int64_t n, int32_t C, int64_t mx,
                                float* y, int64_t ldy, void* stream) {
    DC_REQUIRE(h && g && y, "dc_cloud_bias_add: null pointer");
    DC_REQUIRE(n >= 0 && C >= 1 && mx >= 1 && ldh >= C && ldg >= C && ldy >= C, "dc_cloud_bias_add: bad size");
    if (n == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool v4 = C % 4 == 0 && ldh % 4 == 0 && ldg % 4 == 0 && ldy % 4 == 0 && al16(h) && al16(g) && al16(y);
    if (v4) run_tile<4>(CloudBiasBody<4>{h, g, y, (long)ldh, (long)ldg, (long)ldy, (long)mx}, n, C, s);
    else run_tile<1>(CloudBiasBody<1>{h, g, y, (long)ldh, (long)ldg, (long)ldy, (long)mx}, n, C, s);
    DC_CHECK_LAUNCH("dc_cloud_bias_add");
    return DC_OK;
}

DC_EXPORT size_t dc_cloud_colsum_workspace_bytes(int32_t num_clouds, int64_t mx, int32_t C) {
    return (size_t)std::max(num_clouds, 0) * (size_t)std::max(C, 0) * cloud_colsum_chunks(num_clouds, mx, C) * sizeof(double);
}

// out[num_clouds, C] (ldo) = column sums of x[num_clouds * mx, C] (ldx) over each cloud's mx rows.
DC_EXPORT int dc_cloud_colsum(const float* x, int64_t ldx, int32_t num_clouds, int64_t mx, int32_t C, float* out, int64_t ldo,
                              void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(x && out, "dc_cloud_colsum: null pointer");
    DC_REQUIRE(num_clouds >= 0 && mx >= 1 && C >= 1 && ldx >= C && ldo >= C, "dc_cloud_colsum: bad size");
    if (num_clouds == 0) return DC_OK;
    if (!workspace || workspace_bytes < dc_cloud_colsum_workspace_bytes(num_clouds, mx, C)) {
        dc_set_error("dc_cloud_colsum: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int chunks = cloud_colsum_chunks(num_clouds, mx, C);
    const int rpc = (int)(dc_cdiv(dc_cdiv(mx, chunks), RT) * RT);
    const int used = dc_cdiv(mx, rpc);                                    // <= chunks
    double* partial = static_cast<double*>(workspace);
    const bool v4 = C % 4 == 0 && ldx % 4 == 0 && al16(x);
    if (v4)
        hipLaunchKernelGGL(cloud_colsum_kernel<4>, dim3(used, dc_cdiv(C, CT * 4), num_clouds), dim3(TPB), 0, s, x, (long)ldx,
                           (long)mx, C, used, rpc, partial);
    else
        hipLaunchKernelGGL(cloud_colsum_kernel<1>, dim3(used, dc_cdiv(C, CT), num_clouds), dim3(TPB), 0, s, x, (long)ldx, (long)mx,
                           C, used, rpc, partial);
    hipLaunchKernelGGL(cloud_colsum_final_kernel, dim3(C, num_clouds), dim3(64), 0, s, partial, used, C, out, (long)ldo);
    DC_CHECK_LAUNCH("dc_cloud_colsum");
    return DC_OK;
}

// Backward of y = leaky(scale*h + shift): dh (through the batch statistics when training != 0),
// dgamma[C], dbeta[C] (either may be NULL).
DC_EXPORT int dc_bn_act_backward(const float* dy, int64_t lddy, const float* h, int64_t ldh, int64_t R, int32_t C,
                                 const float* scale, const float* shift, const float* mean, const float* invstd,
                                 const float* gamma, float slope, int32_t training, float* dh, int64_t lddh,
                                 float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(dy && h && scale && shift && mean && invstd && dh, "dc_bn_act_backward: null pointer");
    DC_REQUIRE(R >= 1 && C >= 1 && lddy >= C && ldh >= C && lddh >= C, "dc_bn_act_backward: bad size");
    DC_WS_CHECK("dc_bn_act_backward", R, C)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, R, C);
    const bool v4 = C % 4 == 0 && lddy % 4 == 0 && ldh % 4 == 0 && lddh % 4 == 0 && al16(dy) && al16(h) && al16(dh);
    const BwdFin fin{(long)R, dgamma, dbeta, w.m1, w.m2};
    if (v4)
        run_colreduce<4>(BnBwdF<4>{dy, h, scale, shift, mean, invstd, (long)lddy, (long)ldh, slope}, R, C, w, s, fin);
    else
        run_colreduce<1>(BnBwdF<1>{dy, h, scale, shift, mean, invstd, (long)lddy, (long)ldh, slope}, R, C, w, s, fin);
    if (v4)
        run_tile<4>(BnActBwdBody<4>{dy, h, scale, shift, mean, invstd, gamma, w.m1, w.m2, dh, (long)lddy, (long)ldh,
                                    (long)lddh, slope, training}, R, C, s);
    else
        run_tile<1>(BnActBwdBody<1>{dy, h, scale, shift, mean, invstd, gamma, w.m1, w.m2, dh, (long)lddy, (long)ldh,
                                    (long)lddh, slope, training}, R, C, s);
    DC_CHECK_LAUNCH("dc_bn_act_backward");
    return DC_OK;
}

// ---- vector block -----------------------------------------------------------------------------
// in: combine != 0 -> [2n, 2*co] = [P | Q] (Linear applied to v_cat with weights [W1^T | W2^T]);
//     combine == 0 -> [2n, co] = y.   Statistics of |y| over the n points.
DC_EXPORT int dc_vn_stats(const float* in, int64_t n, int32_t co, int64_t ld, int32_t combine, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                          float* mean, float* invstd, float* scale, float* shift, void* workspace,
                          size_t workspace_bytes, void* stream) {
    DC_REQUIRE(in && mean && invstd && scale && shift, "dc_vn_stats: null pointer");
    DC_REQUIRE(n >= 1 && co >= 1 && ld >= (combine ? 2 * co : co), "dc_vn_stats: bad size");
    DC_WS_CHECK("dc_vn_stats", n, co)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, n, co);
    const BnFin fin{(long)n, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
    if (co % 4 == 0 && ld % 4 == 0 && al16(in))
        run_colreduce<4>(VnStatsF<4>{in, (long)ld, co, combine}, n, co, w, s, fin);
    else
        run_colreduce<1>(VnStatsF<1>{in, (long)ld, co, combine}, n, co, w, s, fin);
    DC_CHECK_LAUNCH("dc_vn_stats");
    return DC_OK;
}

// out[2n, co] = y * relu(scale*|y| + shift) / max(|y|, 1e-8)      (nn/nonlin.py:63-82)
DC_EXPORT int dc_vn_apply(const float* in, int64_t n, int32_t co, int64_t ld, int32_t combine, const float* scale,
                          const float* shift, float* out, int64_t ldo, void* stream) {
    DC_REQUIRE(in && scale && shift && out, "dc_vn_apply: null pointer");
    DC_REQUIRE(n >= 0 && co >= 1 && ld >= (combine ? 2 * co : co) && ldo >= co, "dc_vn_apply: bad size");
    if (n == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (co % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && al16(in) && al16(out))
        run_tile<4>(VnApplyBody<4>{in, scale, shift, out, (long)ld, (long)ldo, co, combine}, n, co, s);
    else
        run_tile<1>(VnApplyBody<1>{in, scale, shift, out, (long)ld, (long)ldo, co, combine}, n, co, s);
    DC_CHECK_LAUNCH("dc_vn_apply");
    return DC_OK;
}

// Backward of dc_vn_apply (+ its statistics when training != 0): din has the shape of `in`.
DC_EXPORT int dc_vn_backward(const float* dout, int64_t lddo, const float* in, int64_t ld, int32_t combine, int64_t n,
                             int32_t co, const float* scale, const float* shift, const float* mean,
                             const float* invstd, const float* gamma, int32_t training, float* din, int64_t lddi,
                             float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(dout && in && scale && shift && mean && invstd && din, "dc_vn_backward: null pointer");
    DC_REQUIRE(n >= 1 && co >= 1 && lddo >= co && ld >= (combine ? 2 * co : co) && lddi >= (combine ? 2 * co : co),
               "dc_vn_backward: bad size");
    DC_WS_CHECK("dc_vn_backward", n, co)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, n, co);
    const bool v4 = co % 4 == 0 && ld % 4 == 0 && lddo % 4 == 0 && lddi % 4 == 0 && al16(in) && al16(dout) && al16(din);
    const BwdFin fin{(long)n, dgamma, dbeta, w.m1, w.m2};
    if (v4)
        run_colreduce<4>(VnBwdF<4>{in, dout, scale, shift, mean, invstd, (long)ld, (long)lddo, co, combine}, n, co, w, s, fin);
    else
        run_colreduce<1>(VnBwdF<1>{in, dout, scale, shift, mean, invstd, (long)ld, (long)lddo, co, combine}, n, co, w, s, fin);
    if (v4)
        run_tile<4>(VnBwdBody<4>{in, dout, scale, shift, mean, invstd, gamma, w.m1, w.m2, din, (long)ld, (long)lddo,
                                 (long)lddi, co, combine, training}, n, co, s);
    else
        run_tile<1>(VnBwdBody<1>{in, dout, scale, shift, mean, invstd, gamma, w.m1, w.m2, din, (long)ld, (long)lddo,
                                 (long)lddi, co, combine, training}, n, co, s);
    DC_CHECK_LAUNCH("dc_vn_backward");
    return DC_OK;
}

// Reduction half of dc_bn_act_backward for the fused GEMM prologue: dgamma, dbeta and the packed coefficients
// coefs[5*C] (c_sc | c_sh | c_g | c_a | c_b) from which dc_linear_bn_backward_input / _weight rebuild
// dh = c_g * dy * act'(c_sc h + c_sh) + c_a h + c_b on the fly -- the [R,C] gradient dh is never materialised.
DC_EXPORT int dc_bn_act_backward_reduce(const float* dy, int64_t lddy, const float* h, int64_t ldh, int64_t R, int32_t C,
                                        const float* scale, const float* shift, const float* mean, const float* invstd,
                                        const float* gamma, float slope, int32_t training, float* dgamma, float* dbeta,
                                        float* coefs, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(dy && h && scale && shift && mean && invstd && coefs, "dc_bn_act_backward_reduce: null pointer");
    DC_REQUIRE(R >= 1 && C >= 1 && lddy >= C && ldh >= C, "dc_bn_act_backward_reduce: bad size");
    DC_WS_CHECK("dc_bn_act_backward_reduce", R, C)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, R, C);
    const BwdCoefFin fin{(long)R, gamma, scale, shift, mean, invstd, training, dgamma, dbeta, coefs, C};
    if (C % 4 == 0 && lddy % 4 == 0 && ldh % 4 == 0 && al16(dy) && al16(h)) {
        const BnBwdF<4> f{dy, h, scale, shift, mean, invstd, (long)lddy, (long)ldh, slope};
        static_assert(sizeof(f) <= DC_FIN_BLOB, "row functor larger than the queue's blob");
        if (dc_fin_take_request()) {             // inside a batch, announced: BOTH stages wait for dc_finalisers_end
            dc_fin_push(DC_FIN_BWD_COEF, w.partial, chunks_of(R, C), C, &fin, sizeof(fin), &f, sizeof(f), (long)R, rows_per_chunk(R, C));
            return DC_OK;
        }
        run_colreduce<4>(f, R, C, w, s, fin);
    }
    else
        run_colreduce<1>(BnBwdF<1>{dy, h, scale, shift, mean, invstd, (long)lddy, (long)ldh, slope}, R, C, w, s, fin, DC_COLRED_BLOCKS,
                         DC_FIN_BWD_COEF);
    DC_CHECK_LAUNCH("dc_bn_act_backward_reduce");
    return DC_OK;
}

// Closes the batch opened by dc_finalisers_begin: ONE launch for every finaliser queued since (those of dc_linear_bn_stats_forward
// and dc_bn_act_backward_reduce calls announced by dc_finaliser_defer_next); discard != 0 drops the queue instead (error paths).
// The coefficient outputs of the queued calls are valid behind this launch; their workspaces must stay alive until then.
DC_EXPORT int dc_finalisers_end(int32_t discard, void* stream) {
    if (discard) {
        dc_gemm_discard();
        dc_fin_clear();
        return DC_OK;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    dc_gemm_flush(stream);                    // queued dense products first: their statistics partials feed the finalisers below
    // queued FIRST stages (BatchNorm-backward reductions): two of them as one launch, a single one as it would have run
    DcFinPending* q;
    const int n = dc_fin_pending(&q);
    int idx[DC_FIN_MAX], m = 0;
    for (int i = 0; i < n; ++i)
        if (q[i].stage1) idx[m++] = i;
    for (int j = 0; j < m; j += 2) {
        BnBwdF<4> f0, f1;
        DcFinPending& a = q[idx[j]];
        memcpy(&f0, a.functor, sizeof(f0));
        if (j + 1 < m) {
            DcFinPending& b = q[idx[j + 1]];
            memcpy(&f1, b.functor, sizeof(f1));
            const dim3 grid(std::max(a.chunks, b.chunks), dc_cdiv(std::max(a.C, b.C), CT * 4), 2);
            hipLaunchKernelGGL((colreduce_pair_kernel<4, 2, BnBwdF<4>>), grid, dim3(TPB), 0, s, f0, f1, a.R, b.R, a.C, b.C, a.chunks,
                               b.chunks, a.rpc, b.rpc, const_cast<double*>(a.partial), const_cast<double*>(b.partial));
            b.stage1 = 0;
        } else {
            hipLaunchKernelGGL((colreduce_kernel<4, 2, BnBwdF<4>>), dim3(a.chunks, dc_cdiv(a.C, CT * 4)), dim3(TPB), 0, s, f0, a.R, a.C,
                               a.chunks, a.rpc, const_cast<double*>(a.partial));
        }
        a.stage1 = 0;
    }
    flush_finalisers(s);
    DC_CHECK_LAUNCH("dc_finalisers_end");
    return DC_OK;
}

// ---- split forms for synchronised BatchNorm (data parallel) -------------------
// The fused entry points above reduce and finalise in one call.  With the batch sharded over ranks the
// statistics of nn/nonlin.py:24-35 belong to the GLOBAL batch: each rank reduces its rows to fp64 column sums
// (dc_bn_sums / dc_vn_sums / dc_*_backward_sums), the sums are all-reduced by the host (2C doubles per layer,
// deltaconv_amd/dp.py), and dc_bn_coeffs_from_sums / dc_*_backward_apply continue from the global sums.
DC_EXPORT int dc_bn_sums(const float* h, int64_t R, int32_t C, int64_t ldh, double* sums, void* workspace,
                         size_t workspace_bytes, void* stream) {
    DC_REQUIRE(h && sums, "dc_bn_sums: null pointer");
    DC_REQUIRE(R >= 1 && C >= 1 && ldh >= C, "dc_bn_sums: bad size");
    DC_WS_CHECK("dc_bn_sums", R, C)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, R, C);
    const SumsFin fin{sums, C, (double)R};
    if (C % 4 == 0 && ldh % 4 == 0 && al16(h))
        run_colreduce<4>(StatsF<4>{h, (long)ldh}, R, C, w, s, fin);
    else
        run_colreduce<1>(StatsF<1>{h, (long)ldh}, R, C, w, s, fin);
    DC_CHECK_LAUNCH("dc_bn_sums");
    return DC_OK;
}

DC_EXPORT int dc_vn_sums(const float* in, int64_t n, int32_t co, int64_t ld, int32_t combine, double* sums,
                         void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(in && sums, "dc_vn_sums: null pointer");
    DC_REQUIRE(n >= 1 && co >= 1 && ld >= (combine ? 2 * co : co), "dc_vn_sums: bad size");
    DC_WS_CHECK("dc_vn_sums", n, co)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, n, co);
    const SumsFin fin{sums, co, (double)n};
    if (co % 4 == 0 && ld % 4 == 0 && al16(in))
        run_colreduce<4>(VnStatsF<4>{in, (long)ld, co, combine}, n, co, w, s, fin);
    else
        run_colreduce<1>(VnStatsF<1>{in, (long)ld, co, combine}, n, co, w, s, fin);
    DC_CHECK_LAUNCH("dc_vn_sums");
    return DC_OK;
}

// sums[2C] = (sum x, sum x^2) over `count` rows (all ranks) -> mean / invstd / scale / shift (+ running statistics).
// count <= 0: the row count is read from the device, sums[2C] (a third block of one double, all-reduced with the sums).
DC_EXPORT int dc_bn_coeffs_from_sums(const double* sums, int64_t count, int32_t C, const float* gamma,
                                     const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                     void* stream) {
    DC_REQUIRE(sums && mean && invstd && scale && shift, "dc_bn_coeffs_from_sums: null pointer");
    DC_REQUIRE(C >= 1, "dc_bn_coeffs_from_sums: bad size");
    const BnFin fin{(long)count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
    hipLaunchKernelGGL(bn_coeffs_from_sums_kernel, dim3(dc_cdiv(C, 128)), dim3(128), 0, static_cast<hipStream_t>(stream),
                       sums, (long)count, C, fin);
    DC_CHECK_LAUNCH("dc_bn_coeffs_from_sums");
    return DC_OK;
}

// backward, step 1: sums[2C] = (sum dz, sum dz * xhat) over this rank's rows
DC_EXPORT int dc_bn_act_backward_sums(const float* dy, int64_t lddy, const float* h, int64_t ldh, int64_t R, int32_t C,
                                      const float* scale, const float* shift, const float* mean, const float* invstd,
                                      float slope, double* sums, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    DC_REQUIRE(dy && h && scale && shift && mean && invstd && sums, "dc_bn_act_backward_sums: null pointer");
    DC_REQUIRE(R >= 1 && C >= 1 && lddy >= C && ldh >= C, "dc_bn_act_backward_sums: bad size");
    DC_WS_CHECK("dc_bn_act_backward_sums", R, C)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, R, C);
    const SumsFin fin{sums, C, (double)R};
    if (C % 4 == 0 && lddy % 4 == 0 && ldh % 4 == 0 && al16(dy) && al16(h))
        run_colreduce<4>(BnBwdF<4>{dy, h, scale, shift, mean, invstd, (long)lddy, (long)ldh, slope}, R, C, w, s, fin);
    else
        run_colreduce<1>(BnBwdF<1>{dy, h, scale, shift, mean, invstd, (long)lddy, (long)ldh, slope}, R, C, w, s, fin);
    DC_CHECK_LAUNCH("dc_bn_act_backward_sums");
    return DC_OK;
}

// backward, between the two steps: the global means and this rank's parameter gradients from the two records of `sums`
// (first record all-reduced, second local): m1 = sum_0 / rows, m2 = sum_1 / rows, dbeta = local sum_0, dgamma = local sum_1.
namespace {
__global__ void sync_means_kernel(const double* __restrict__ sums, int C, float* m1, float* m2, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double* loc = sums + 2 * C + 1;
    const double cnt = sums[2 * C];
    m1[c] = (float)(sums[c] / cnt);
    m2[c] = (float)(sums[C + c] / cnt);
    if (dbeta) dbeta[c] = (float)loc[c];
    if (dgamma) dgamma[c] = (float)loc[C + c];
}
}  // namespace
DC_EXPORT int dc_sync_means(const double* sums, int32_t C, float* m1, float* m2, float* dgamma, float* dbeta, void* stream) {
    DC_REQUIRE(sums && m1 && m2, "dc_sync_means: null pointer");
    DC_REQUIRE(C >= 1, "dc_sync_means: bad size");
    hipLaunchKernelGGL(sync_means_kernel, dim3(dc_cdiv(C, 128)), dim3(128), 0, static_cast<hipStream_t>(stream), sums, C, m1, m2,
                       dgamma, dbeta);
    DC_CHECK_LAUNCH("dc_sync_means");
    return DC_OK;
}

// backward, step 2 for the FUSED form (dc_linear_bn_backward_input / _weight rebuild dh in their operand loaders): the five
// per-column coefficient rows of the GEMM prologue from the GLOBAL sums (count <= 0: the row count is on the device at
// global_sums[2C]), dgamma / dbeta from this rank's own sums (they are averaged with every other gradient afterwards).
namespace {
__global__ void bn_bwd_coefs_from_sums_kernel(const double* __restrict__ gs, long count, const double* __restrict__ ls, int C,
                                              BwdCoefFin fin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    fin.R = count > 0 ? count : (long)(gs[2 * C] + 0.5);
    float* dg = fin.dgamma; float* db = fin.dbeta;
    fin.dgamma = nullptr; fin.dbeta = nullptr;
    fin(c, gs[c], gs[C + c]);
    if (db) db[c] = (float)ls[c];
    if (dg) dg[c] = (float)ls[C + c];
}
}  // namespace
DC_EXPORT int dc_bn_backward_coefs_from_sums(const double* global_sums, int64_t count, const double* local_sums, int32_t C,
                                             const float* gamma, const float* scale, const float* shift, const float* mean,
                                             const float* invstd, int32_t training, float* dgamma, float* dbeta, float* coefs,
                                             void* stream) {
    DC_REQUIRE(global_sums && local_sums && scale && shift && mean && invstd && coefs, "dc_bn_backward_coefs_from_sums: null pointer");
    DC_REQUIRE(C >= 1, "dc_bn_backward_coefs_from_sums: bad size");
    const BwdCoefFin fin{(long)count, gamma, scale, shift, mean, invstd, training, dgamma, dbeta, coefs, C};
    hipLaunchKernelGGL(bn_bwd_coefs_from_sums_kernel, dim3(dc_cdiv(C, 128)), dim3(128), 0, static_cast<hipStream_t>(stream),
                       global_sums, (long)count, local_sums, C, fin);
    DC_CHECK_LAUNCH("dc_bn_backward_coefs_from_sums");
    return DC_OK;
}

// backward, step 2: dh from the GLOBAL means m1 = sum dz / count, m2 = sum dz * xhat / count (fp32 [C] each)
DC_EXPORT int dc_bn_act_backward_apply(const float* dy, int64_t lddy, const float* h, int64_t ldh, int64_t R, int32_t C,
                                       const float* scale, const float* shift, const float* mean, const float* invstd,
                                       const float* gamma, float slope, int32_t training, const float* m1,
                                       const float* m2, float* dh, int64_t lddh, void* stream) {
    DC_REQUIRE(dy && h && scale && shift && mean && invstd && m1 && m2 && dh, "dc_bn_act_backward_apply: null pointer");
    DC_REQUIRE(R >= 1 && C >= 1 && lddy >= C && ldh >= C && lddh >= C, "dc_bn_act_backward_apply: bad size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (C % 4 == 0 && lddy % 4 == 0 && ldh % 4 == 0 && lddh % 4 == 0 && al16(dy) && al16(h) && al16(dh))
        run_tile<4>(BnActBwdBody<4>{dy, h, scale, shift, mean, invstd, gamma, m1, m2, dh, (long)lddy, (long)ldh,
                                    (long)lddh, slope, training}, R, C, s);
    else
        run_tile<1>(BnActBwdBody<1>{dy, h, scale, shift, mean, invstd, gamma, m1, m2, dh, (long)lddy, (long)ldh,
                                    (long)lddh, slope, training}, R, C, s);
    DC_CHECK_LAUNCH("dc_bn_act_backward_apply");
    return DC_OK;
}

DC_EXPORT int dc_vn_backward_sums(const float* dout, int64_t lddo, const float* in, int64_t ld, int32_t combine,
                                  int64_t n, int32_t co, const float* scale, const float* shift, const float* mean,
                                  const float* invstd, double* sums, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    DC_REQUIRE(dout && in && scale && shift && mean && invstd && sums, "dc_vn_backward_sums: null pointer");
    DC_REQUIRE(n >= 1 && co >= 1 && lddo >= co && ld >= (combine ? 2 * co : co), "dc_vn_backward_sums: bad size");
    DC_WS_CHECK("dc_vn_backward_sums", n, co)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, n, co);
    const SumsFin fin{sums, co, (double)n};
    if (co % 4 == 0 && ld % 4 == 0 && lddo % 4 == 0 && al16(in) && al16(dout))
        run_colreduce<4>(VnBwdF<4>{in, dout, scale, shift, mean, invstd, (long)ld, (long)lddo, co, combine}, n, co, w, s, fin);
    else
        run_colreduce<1>(VnBwdF<1>{in, dout, scale, shift, mean, invstd, (long)ld, (long)lddo, co, combine}, n, co, w, s, fin);
    DC_CHECK_LAUNCH("dc_vn_backward_sums");
    return DC_OK;
}

DC_EXPORT int dc_vn_backward_apply(const float* dout, int64_t lddo, const float* in, int64_t ld, int32_t combine,
                                   int64_t n, int32_t co, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, const float* gamma, int32_t training, const float* m1,
                                   const float* m2, float* din, int64_t lddi, void* stream) {
    DC_REQUIRE(dout && in && scale && shift && mean && invstd && m1 && m2 && din, "dc_vn_backward_apply: null pointer");
    DC_REQUIRE(n >= 1 && co >= 1 && lddo >= co && ld >= (combine ? 2 * co : co) && lddi >= (combine ? 2 * co : co),
               "dc_vn_backward_apply: bad size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (co % 4 == 0 && ld % 4 == 0 && lddo % 4 == 0 && lddi % 4 == 0 && al16(in) && al16(dout) && al16(din))
        run_tile<4>(VnBwdBody<4>{in, dout, scale, shift, mean, invstd, gamma, m1, m2, din, (long)ld, (long)lddo,
                                 (long)lddi, co, combine, training}, n, co, s);
    else
        run_tile<1>(VnBwdBody<1>{in, dout, scale, shift, mean, invstd, gamma, m1, m2, din, (long)ld, (long)lddo,
                                 (long)lddi, co, combine, training}, n, co, s);
    DC_CHECK_LAUNCH("dc_vn_backward_apply");
    return DC_OK;
}

// ---- embedding head fused with per-cloud pooling ---------------------------------------------------
// pooled[B, ldp] = [max_i y | mean_i y] over the N rows of each cloud, y = leaky(scale*h + shift);
// argmax[B, C] = first maximal row within the cloud.  with_mean == 0 writes only the max block.
DC_EXPORT int dc_bn_act_pool(const float* h, int64_t ldh, int32_t num_clouds, int32_t N, int32_t C, const float* scale,
                             const float* shift, float slope, int32_t with_mean, float* pooled, int64_t ldp,
                             int32_t* argmax, void* stream) {
    DC_REQUIRE(h && scale && shift && pooled && argmax, "dc_bn_act_pool: null pointer");
    DC_REQUIRE(num_clouds >= 1 && N >= 1 && C >= 1 && ldh >= C && ldp >= (with_mean ? 2 * C : C), "dc_bn_act_pool: bad size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (C % 4 == 0 && ldh % 4 == 0 && al16(h))
        hipLaunchKernelGGL(pool_fwd_kernel<4>, dim3(num_clouds, dc_cdiv(C, CT * 4)), dim3(TPB), 0, s, h, (long)ldh, N, C, scale,
                           shift, slope, with_mean, pooled, (long)ldp, argmax);
    else
        hipLaunchKernelGGL(pool_fwd_kernel<1>, dim3(num_clouds, dc_cdiv(C, CT)), dim3(TPB), 0, s, h, (long)ldh, N, C, scale,
                           shift, slope, with_mean, pooled, (long)ldp, argmax);
    DC_CHECK_LAUNCH("dc_bn_act_pool");
    return DC_OK;
}

// Backward of dc_bn_act_pool through the BatchNorm: dh[B*N, C], dgamma, dbeta from dpooled[B, ldp].
DC_EXPORT int dc_bn_act_pool_backward(const float* dpooled, int64_t ldp, const int32_t* argmax, const float* h,
                                      int64_t ldh, int32_t num_clouds, int32_t N, int32_t C, const float* scale,
                                      const float* shift, const float* mean, const float* invstd, const float* gamma,
                                      float slope, int32_t with_mean, int32_t training, float* dh, int64_t lddh,
                                      float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    DC_REQUIRE(dpooled && argmax && h && scale && shift && mean && invstd && dh, "dc_bn_act_pool_backward: null pointer");
    DC_REQUIRE(num_clouds >= 1 && N >= 1 && C >= 1 && ldh >= C && lddh >= C && ldp >= (with_mean ? 2 * C : C),
               "dc_bn_act_pool_backward: bad size");
    const long R = (long)num_clouds * N;
    DC_WS_CHECK("dc_bn_act_pool_backward", R, C)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = carve(workspace, R, C);
    const BwdFin fin{R, dgamma, dbeta, w.m1, w.m2};
    const bool v4 = C % 4 == 0 && ldh % 4 == 0 && lddh % 4 == 0 && al16(h) && al16(dh);
    DC_REQUIRE(R < 2147483647L, "dc_bn_act_pool_backward: too many rows");
    if (v4) {
        PoolBwdF<4> f{};
        f.dy.dp = dpooled; f.dy.argmax = argmax; f.dy.ldp = ldp; f.dy.N = N; f.dy.C = C; f.dy.with_mean = with_mean;
        f.h = h; f.scale = scale; f.shift = shift; f.mean = mean; f.invstd = invstd; f.ldh = ldh; f.slope = slope; f.primed = 0;
        run_colreduce<4>(f, R, C, w, s, fin);
        PoolBwdBody<4> b{};
        b.dy = f.dy; b.h = h; b.scale = scale; b.shift = shift; b.mean = mean; b.invstd = invstd; b.gamma = gamma;
        b.m1 = w.m1; b.m2 = w.m2; b.dh = dh; b.ldh = ldh; b.lddh = lddh; b.slope = slope; b.training = training;
        run_tile<4>(b, R, C, s);
    } else {
        PoolBwdF<1> f{};
        f.dy.dp = dpooled; f.dy.argmax = argmax; f.dy.ldp = ldp; f.dy.N = N; f.dy.C = C; f.dy.with_mean = with_mean;
        f.h = h; f.scale = scale; f.shift = shift; f.mean = mean; f.invstd = invstd; f.ldh = ldh; f.slope = slope; f.primed = 0;
        run_colreduce<1>(f, R, C, w, s, fin);
        PoolBwdBody<1> b{};
        b.dy = f.dy; b.h = h; b.scale = scale; b.shift = shift; b.mean = mean; b.invstd = invstd; b.gamma = gamma;
        b.m1 = w.m1; b.m2 = w.m2; b.dh = dh; b.ldh = ldh; b.lddh = lddh; b.slope = slope; b.training = training;
        run_tile<1>(b, R, C, s);
    }
    DC_CHECK_LAUNCH("dc_bn_act_pool_backward");
    return DC_OK;
}
