// Generic launch skeletons of the ELL kernels: LDS staging of a block's k-lists (forward) or
// in-edge lists (transposed), then the per-thread bodies of ell_math.h.
//
// Why staging: a thread = (point, 4-channel group), so the 16 lanes of a point (C = 64) would each
// fetch the same neighbour id + coefficient pair per slot through the vector-memory path; PMC showed
// those broadcast loads to be 2/3 of all VMEM instructions and the texture addresser busy for 2/3 of
// the kernel.  The rows of a block's points are contiguous in memory: one coalesced copy into LDS,
// then LDS broadcast reads (which do not touch the TA) feed the 16-byte feature gathers.
#pragma once
#include "common.h"
#include "ell_math.h"

namespace dcstage {
using dcell::G2;
using dcell::Row;

constexpr int T_CHUNK = 2048;  // in-edge entries staged per pass (transposed kernels)

inline int fwd_block_threads(int groups) { return groups >= 4 ? 256 : 64 * groups; }
inline size_t fwd_lds_bytes(int groups, int k) {
    const int tpb = fwd_block_threads(groups);
    const size_t npts = (size_t)(tpb + groups - 1) / groups + 1;
    return npts * k * 12 + 16;
}

// BODY: __device__ void operator()(long i, int c0, Row r, int k) const
template <int V, class BODY>
__global__ __launch_bounds__(256) void ell_fwd_kernel(long total, int groups, int remap, const float* __restrict__ coef,
                                                      const int* __restrict__ nbr, int k, BODY body) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tpb = blockDim.x;
    const long t0 = dc_xcd_block(remap) * tpb;
    if (t0 >= total) return;  // block-uniform
    const long tl = min(t0 + (long)tpb, total) - 1;
    const long pf = t0 / groups, pl = tl / groups;
    const int nent = (int)(pl - pf + 1) * k;
    int* ids = reinterpret_cast<int*>(smem);
    G2* cf = reinterpret_cast<G2*>(smem + ((size_t)nent * 4 + 15) / 16 * 16);
    const int* gn = nbr + pf * k;
    for (int q = threadIdx.x; q < nent; q += tpb) ids[q] = gn[q];
    if (coef) {
        const G2* gc = reinterpret_cast<const G2*>(coef) + pf * k;
        for (int q = threadIdx.x; q < nent; q += tpb) cf[q] = gc[q];
    }
    __syncthreads();
    const long t = t0 + threadIdx.x;
    if (t >= total) return;
    const long i = t / groups;
    const int c0 = (int)(t - i * groups) * V;
    const int off = (int)(i - pf) * k;
    body(i, c0, Row{ids + off, cf + off}, k);
}

// OP: accumulator functor of ell_math.h (init / step / finish)
template <int V, class OP>
__global__ __launch_bounds__(256) void ell_T_kernel(long total, int groups, int remap, const float* __restrict__ coefT,
                                                    const int* __restrict__ tptr, const int* __restrict__ tedge, int k,
                                                    OP op) {
    __shared__ int src[T_CHUNK];
    __shared__ G2 cf[T_CHUNK];
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
    op.init();
    for (int base = e_begin; base < e_end; base += T_CHUNK) {
        const int cnt = min(T_CHUNK, e_end - base);
        __syncthreads();
        for (int q = threadIdx.x; q < cnt; q += tpb) {
            const int e = tedge[base + q];
            const int i = e / k;
            src[q] = i;
            slot[q] = (unsigned char)(e - i * k);
            if (coefT) cf[q] = reinterpret_cast<const G2*>(coefT)[base + q];
        }
        __syncthreads();
        const int lo = max(cb, base) - base, hi = min(ce, base + cnt) - base;
#pragma unroll 4
        for (int p = lo; p < hi; ++p) op.step(src[p], slot[p], cf[p], c0);
    }
    if (active) op.finish(j, c0);
}

template <int V, class BODY>
inline void launch_fwd(long n, int C, const float* coef, const int* nbr, int k, BODY body, hipStream_t s) {
    const int groups = C / V;
    const long total = n * groups;
    const int tpb = fwd_block_threads(groups);
    hipLaunchKernelGGL((ell_fwd_kernel<V, BODY>), dim3(dc_cdiv(total, tpb)), dim3(tpb), fwd_lds_bytes(groups, k), s, total,
                       groups, dc_option(DC_OPT_XCD_REMAP), coef, nbr, k, body);
}

template <int V, class OP>
inline void launch_T(long n, int C, const float* coefT, const int* tptr, const int* tedge, int k, OP op,
                     hipStream_t s) {
    const int groups = C / V;
    const long total = n * groups;
    hipLaunchKernelGGL((ell_T_kernel<V, OP>), dim3(dc_cdiv(total, 256)), dim3(256), 0, s, total, groups,
                       dc_option(DC_OPT_XCD_REMAP), coefT, tptr, tedge, k, op);
}

inline bool aligned_to(const void* p, int bytes) { return (reinterpret_cast<uintptr_t>(p) & (bytes - 1)) == 0; }
// vector width: 16-byte path when channels, strides and bases allow it; 8-byte path for even channel counts / strides /
// bases (the layer-0 blocks: grad x' sits at column 6 of a 70-float row -- 8-byte loads need half the texture-addresser
// cycles of dword loads: 38 -> 22 us for that transposed apply); else dword
inline int pick_v(int C, std::initializer_list<long> lds, std::initializer_list<const void*> ptrs) {
    int v = C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 1);
    for (long l : lds) {
        if (l % 4 && v == 4) v = 2;
        if (l % 2) return 1;
    }
    for (const void* p : ptrs) {
        if (!aligned_to(p, 16) && v == 4) v = 2;
        if (!aligned_to(p, 8)) return 1;
    }
    return v;
}

}  // namespace dcstage
