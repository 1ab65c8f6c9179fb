// Ordered two-stage column reduction over row-major [R, C] data + BatchNorm finalisation kernels,
// shared by nn.hip (MLP stream) and edge.hip (layer-0 edge MLP).  Bit-reproducible: per-block
// partials in fp64, then one wave per (quantity, column) with a fixed butterfly.
#pragma once
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <cstring>

namespace dccol {
namespace {   // internal linkage: this header is included by several translation units

constexpr int RT = 16;          // row lanes per block
constexpr int CT = 16;          // column groups per block
constexpr int TPB = RT * CT;    // 256

template <int V>
struct alignas(4 * V) FV {
    float v[V];
};
template <int V>
__device__ __forceinline__ FV<V> ldv(const float* p) { return *reinterpret_cast<const FV<V>*>(p); }
template <int V>
__device__ __forceinline__ void stv(float* p, const FV<V>& a) {
    if constexpr (V == 4) dc_store16<DC_ST_NN>(p, *reinterpret_cast<const dc_f32x4*>(&a));   // streamed out (common.h)
    else *reinterpret_cast<FV<V>*>(p) = a;
}

// ---- generic ordered column reduction: NQ quantities per element ------------------------------
// grid = (row_chunks, col_tiles); partial[(chunk*NQ + q)*C + col] (double)
template <int V, int NQ, class F>
__device__ __forceinline__ void colreduce_body(F& f, long R, int C, int chunks, int rpc, double* __restrict__ partial, int bx,
                                               int by) {
    __shared__ double sm[NQ][RT][CT * V];
    const int cgl = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c0 = (by * CT + cgl) * V;
    const long r0 = (long)bx * rpc;
    const long r1 = min(r0 + rpc, R);
    double acc[NQ][V];   // fp64: sum x^2 - (sum x)^2 / R must survive cancellation (R can be 2)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[q][j] = 0.0;
#ifndef DC_COLRED_UNROLL
#define DC_COLRED_UNROLL 1
#endif
    if (c0 < C) {
#pragma unroll DC_COLRED_UNROLL
        for (long r = r0 + rl; r < r1; r += RT) {
            double t[NQ][V];
            f(r, c0, t);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < V; ++j) acc[q][j] += t[q][j];
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < V; ++j) sm[q][rl][cgl * V + j] = acc[q][j];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NQ * CT * V; idx += TPB) {
        const int q = idx / (CT * V), cl = idx % (CT * V);
        const int col = by * CT * V + cl;
        if (col < C) {
            double s = 0;
#pragma unroll
            for (int rr = 0; rr < RT; ++rr) s += sm[q][rr][cl];
            partial[((long)q * C + col) * chunks + bx] = s;   // [q][col][chunk]
        }
    }
}
template <int V, int NQ, class F>
__global__ __launch_bounds__(TPB) void colreduce_kernel(F f, long R, int C, int chunks, int rpc,
                                                        double* __restrict__ partial) {
    colreduce_body<V, NQ, F>(f, R, C, chunks, rpc, partial, blockIdx.x, blockIdx.y);
}
// two reductions of the same kind in one launch (deferred first stages, common.h): blockIdx.z picks the parameter set
template <int V, int NQ, class F>
__global__ __launch_bounds__(TPB) void colreduce_pair_kernel(F f0, F f1, long R0, long R1, int C0, int C1, int chunks0, int chunks1,
                                                             int rpc0, int rpc1, double* __restrict__ partial0,
                                                             double* __restrict__ partial1) {
    const bool second = blockIdx.z != 0;
    const int chunks = second ? chunks1 : chunks0, C = second ? C1 : C0;
    if ((int)blockIdx.x >= chunks || (int)blockIdx.y * CT * V >= C) return;      // (grid = the larger of the two)
    if (second) colreduce_body<V, NQ, F>(f1, R1, C1, chunks1, rpc1, partial1, blockIdx.x, blockIdx.y);
    else colreduce_body<V, NQ, F>(f0, R0, C0, chunks0, rpc0, partial0, blockIdx.x, blockIdx.y);
}

// Final stage: one wave per column, lanes stride the chunks of both quantities, butterfly reduce
// (association fixed by the structure -> bit-reproducible), then lane 0 runs the finaliser FIN
// (statistics -> scale/shift + running statistics, or backward sums -> dgamma/dbeta/means) --
// no separate finalize launch.
template <class FIN>
__global__ __launch_bounds__(64) void colreduce_final_kernel(const double* __restrict__ partial, int chunks, int C,
                                                             FIN fin) {
    const int col = blockIdx.x;
    const double* p0 = partial + (long)col * chunks;
    const double* p1 = partial + ((long)C + col) * chunks;
    double s0 = 0, s1 = 0;
    for (int ch = threadIdx.x; ch < chunks; ch += 64) {
        s0 += p0[ch];
        s1 += p1[ch];
    }
    s0 = dc_wave_sum(s0);
    s1 = dc_wave_sum(s1);
    if (threadIdx.x == 0) fin(col, s0, s1);
}

// batch statistics -> mean/invstd/scale/shift (+ running statistics, unbiased variance)
struct BnFin {
    long R; const float *gamma, *beta; float eps, momentum; float *running_mean, *running_var;
    float *mean, *invstd, *scale, *shift;
    __device__ void operator()(int c, double s0, double s1) const {
        const double m = s0 / (double)R;
        double var = s1 / (double)R - m * m;  // biased (normalisation)
        if (var < 0) var = 0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        mean[c] = (float)m;
        invstd[c] = (float)is;
        scale[c] = (float)(g * is);
        shift[c] = (float)(b - m * g * is);
        if (running_mean) {
            const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
};
// backward sums: dgamma = sum dz*xhat, dbeta = sum dz; per-column means for the apply pass
struct BwdFin {
    long R; float *dgamma, *dbeta, *m1, *m2;
    __device__ void operator()(int c, double s0, double s1) const {
        if (dbeta) dbeta[c] = (float)s0;
        if (dgamma) dgamma[c] = (float)s1;
        m1[c] = (float)(s0 / (double)R);
        m2[c] = (float)(s1 / (double)R);
    }
};
// backward sums -> dgamma / dbeta and the five per-column coefficients of the GEMM prologue (gemm.hip):
//   dz = dy * act'(c_sc h + c_sh);  dh = c_g dz + c_a h + c_b  ==  gamma invstd (dz - m1 - xhat m2)   (training)
//                                                              ==  scale dz                            (eval)
struct BwdCoefFin {
    long R; const float *gamma, *scale, *shift, *mean, *invstd; int training; float *dgamma, *dbeta, *coefs; int C;
    __device__ void operator()(int c, double s0, double s1) const {
        if (dbeta) dbeta[c] = (float)s0;
        if (dgamma) dgamma[c] = (float)s1;
        const double gi = (double)(gamma ? gamma[c] : 1.f) * (double)invstd[c];
        coefs[c] = scale[c];
        coefs[C + c] = shift[c];
        if (training) {
            const double m1 = s0 / (double)R, m2 = s1 / (double)R, is = invstd[c], mu = mean[c];
            coefs[2 * C + c] = (float)gi;
            coefs[3 * C + c] = (float)(-gi * m2 * is);
            coefs[4 * C + c] = (float)(-gi * m1 + gi * m2 * is * mu);
        } else {
            coefs[2 * C + c] = scale[c];
            coefs[3 * C + c] = 0.f;
            coefs[4 * C + c] = 0.f;
        }
    }
};
struct NoFin {
    __device__ void operator()(int, double, double) const {}
};
// the two column sums themselves (fp64): the data-parallel SyncBN path all-reduces them across ranks.  Layout of `sums`:
// double [2][2 C + 1] = two identical records (sum_0[C] | sum_1[C] | rows): the host all-reduces the FIRST record in place
// and keeps the second as this rank's own sums (dgamma / dbeta are local quantities) -- no fill, no clone launch.
struct SumsFin {
    double* sums; int C; double rows;
    __device__ void operator()(int c, double s0, double s1) const {
        double* dup = sums + 2 * C + 1;
        sums[c] = s0;
        sums[C + c] = s1;
        dup[c] = s0;
        dup[C + c] = s1;
        if (c == 0) { sums[2 * C] = rows; dup[2 * C] = rows; }
    }
};

// Several finalisers in ONE launch (deferred finalisers, common.h): block -> (entry, column); the same lane-strided sums and
// butterfly as colreduce_final_kernel, hence the same bits.
struct FinTable {
    int count;
    int first[DC_FIN_MAX + 1];                 // prefix sums of the entries' column counts
    int kind[DC_FIN_MAX], chunks[DC_FIN_MAX], C[DC_FIN_MAX];
    const double* partial[DC_FIN_MAX];
    BnFin bn[DC_FIN_MAX];
    BwdCoefFin bc[DC_FIN_MAX];
};
__global__ __launch_bounds__(64) void colreduce_final_many_kernel(FinTable t) {
    int e = 0;
    while (e + 1 < t.count && (int)blockIdx.x >= t.first[e + 1]) ++e;
    const int col = blockIdx.x - t.first[e], chunks = t.chunks[e], C = t.C[e];
    const double* p0 = t.partial[e] + (long)col * chunks;
    const double* p1 = t.partial[e] + ((long)C + col) * chunks;
    double s0 = 0, s1 = 0;
    for (int ch = threadIdx.x; ch < chunks; ch += 64) {
        s0 += p0[ch];
        s1 += p1[ch];
    }
    s0 = dc_wave_sum(s0);
    s1 = dc_wave_sum(s1);
    if (threadIdx.x == 0) {
        if (t.kind[e] == DC_FIN_BN) t.bn[e](col, s0, s1);
        else t.bc[e](col, s0, s1);
    }
}
// launch (or queue, when the caller asked for it inside an open batch) the finaliser of a reduction whose partials are in place
template <class FIN>
inline void finalise_or_defer(int kind, const double* partial, int chunks, int C, const FIN& fin, hipStream_t s) {
    static_assert(sizeof(FIN) <= DC_FIN_BLOB, "finaliser object larger than the queue's blob");
    if (dc_fin_take_request()) {
        dc_fin_push(kind, partial, chunks, C, &fin, sizeof(FIN));
        return;
    }
    hipLaunchKernelGGL((colreduce_final_kernel<FIN>), dim3(C), dim3(64), 0, s, partial, chunks, C, fin);
}
inline void flush_finalisers(hipStream_t s) {
    DcFinPending* q;
    const int n = dc_fin_pending(&q);
    if (n > 0) {
        FinTable t;
        t.count = n;
        int cols = 0;
        for (int i = 0; i < n; ++i) {
            t.first[i] = cols;
            t.kind[i] = q[i].kind; t.chunks[i] = q[i].chunks; t.C[i] = q[i].C; t.partial[i] = q[i].partial;
            if (q[i].kind == DC_FIN_BN) memcpy(&t.bn[i], q[i].blob, sizeof(BnFin));
            else memcpy(&t.bc[i], q[i].blob, sizeof(BwdCoefFin));
            cols += q[i].C;
        }
        t.first[n] = cols;
        hipLaunchKernelGGL(colreduce_final_many_kernel, dim3(cols), dim3(64), 0, s, t);
    }
    dc_fin_clear();
}

// coefficients from (possibly all-reduced) column sums over `count` rows; one thread per column
// count <= 0: the row count is itself on the device, sums[2C] (all-reduced together with the sums: no host sync)
__global__ void bn_coeffs_from_sums_kernel(const double* __restrict__ sums, long count, int C, BnFin fin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    fin.R = count > 0 ? count : (long)(sums[2 * C] + 0.5);
    if (c < C) fin(c, sums[c], sums[C + c]);
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* mean, float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean[c] = rm[c];
    invstd[c] = is;
    scale[c] = g * is;
    shift[c] = b - rm[c] * g * is;
}

// ---- host helpers --------------------------------------------------------------------------
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// rows per block: enough blocks (~768: three per CU) to fill 256 CUs even for narrow matrices; multiple of RT.  The GATHERING
// reduction of the layer-0 edge statistics (edge.hip: EdgeStatsF, k row gathers per element) wants twice as many, shorter
// workgroups (30 vs 35 us at C2): `blocks` is a parameter of the launch, the workspace is sized for the larger count.
#ifndef DC_COLRED_BLOCKS
#define DC_COLRED_BLOCKS 768      // round 6, same-box A/B of the step (profiles/r06_labs.txt item 6): 768 beats 1536 / 3072 / 384
#endif
constexpr int COLRED_BLOCKS_MAX = 2 * DC_COLRED_BLOCKS;
inline int rows_per_chunk(long R, int C, int blocks = DC_COLRED_BLOCKS) {
    const long coltiles = (C + CT * 4 - 1) / (CT * 4);
    long rpc = (R * coltiles / blocks + RT - 1) / RT * RT;
    return (int)std::min<long>(std::max<long>(rpc, RT), 512);
}
inline int chunks_of(long R, int C, int blocks = DC_COLRED_BLOCKS) { const int rpc = rows_per_chunk(R, C, blocks); return (int)((R + rpc - 1) / rpc); }
inline size_t ws_need(long R, int C) { return ((size_t)chunks_of(R, C, COLRED_BLOCKS_MAX) * 2 * C + 2 * (size_t)C) * 8 + 2 * (size_t)C * 4; }
inline int stream_grid(long total) { return (int)std::min<long>((total + 255) / 256, 256L * 16); }

struct Ws {
    double* partial; double* sums; float* m1; float* m2;
};
inline Ws carve(void* ws, long R, int C) {
    Ws w;
    w.partial = static_cast<double*>(ws);
    w.sums = w.partial + (size_t)chunks_of(R, C, COLRED_BLOCKS_MAX) * 2 * C;
    w.m1 = reinterpret_cast<float*>(w.sums + 2 * (size_t)C);
    w.m2 = w.m1 + C;
    return w;
}

template <int V, class F, class FIN>
void run_colreduce(F f, long R, int C, const Ws& w, hipStream_t s, FIN fin, int blocks = DC_COLRED_BLOCKS, int defer_kind = -1) {
    const int rpc = rows_per_chunk(R, C, blocks), chunks = chunks_of(R, C, blocks);
    dim3 grid(chunks, dc_cdiv(C, CT * V));
    hipLaunchKernelGGL((colreduce_kernel<V, 2, F>), grid, dim3(TPB), 0, s, f, R, C, chunks, rpc, w.partial);
    if constexpr (std::is_same<FIN, BnFin>::value || std::is_same<FIN, BwdCoefFin>::value) {
        if (defer_kind >= 0) {                  // (deferrable kinds only: the queue's kernel knows these two finalisers)
            finalise_or_defer(defer_kind, w.partial, chunks, C, fin, s);
            return;
        }
    }
    hipLaunchKernelGGL((colreduce_final_kernel<FIN>), dim3(C), dim3(64), 0, s, w.partial, chunks, C, fin);
}


}  // namespace
}  // namespace dccol
