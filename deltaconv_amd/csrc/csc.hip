// Transposed adjacency ("who lists me as a neighbour") of the kNN graph, built once per batch and
// shared by every transposed apply / max-aggregation backward of the step.  The reference gets the
// same effect from torch_sparse's autograd (spmm with A^T) and torch_scatter's arg-indexed backward.
//   tptr[Nt+1], tedge[Nt*k]: in-edges of point j are tedge[tptr[j] .. tptr[j+1]), edge id e = i*k + s,
//   sorted ascending so every transposed sum has a fixed order (bit-reproducible, no fp atomics).
// Neighbours never leave their cloud, so the in-edges of cloud b total exactly N_b*k and the scan
// base of cloud b is cloud_ptr[b]*k: the scan is local to a cloud (one block per cloud).
#include <algorithm>
#include "common.h"
#include "ell_math.h"

namespace {
constexpr int TPB = 256;

__global__ void csc_count_kernel(const int* __restrict__ nbr, long ne, int* __restrict__ cnt) {
    const long e = (long)blockIdx.x * TPB + threadIdx.x;
    if (e < ne) atomicAdd(cnt + nbr[e], 1);
}

// exclusive scan of cnt over one cloud; writes tptr and the fill cursors (cursor aliases cnt)
__global__ __launch_bounds__(TPB) void csc_scan_kernel(const int* __restrict__ cloud_ptr, int k, int num_clouds,
                                                       int* __restrict__ cnt, int* __restrict__ tptr) {
    __shared__ int part[TPB];
    const int cloud = blockIdx.x;
    const int begin = cloud_ptr[cloud], n = cloud_ptr[cloud + 1] - begin;
    const int per = (n + TPB - 1) / TPB;
    const int lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int q = lo; q < hi; ++q) s += cnt[begin + q];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = begin * k;  // in-edges of earlier clouds
        for (int w = 0; w < TPB; ++w) {
            const int c = part[w];
            part[w] = run;
            run += c;
        }
        if (cloud == num_clouds - 1) tptr[begin + n] = run;  // = Nt*k
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int q = lo; q < hi; ++q) {
        const int c = cnt[begin + q];
        tptr[begin + q] = run;
        cnt[begin + q] = run;  // becomes the fill cursor
        run += c;
    }
}

__global__ void csc_fill_kernel(const int* __restrict__ nbr, long ne, int* __restrict__ cursor,
                                int* __restrict__ tedge) {
    const long e = (long)blockIdx.x * TPB + threadIdx.x;
    if (e < ne) tedge[atomicAdd(cursor + nbr[e], 1)] = (int)e;
}

// Count, scan and fill of ONE cloud in one workgroup (clouds of at most CLOUD_MAX points): the in-degree counters and
// the fill cursors live in LDS (integer LDS atomics instead of 2 x E global atomics: csc_count + csc_scan + csc_fill +
// the zero fill took 4.7 + 20 + 5 + 23 us at 32 x 1024 points, k = 20).  The columns come out unordered (as from
// csc_fill_kernel) and are ordered by csc_rank_kernel afterwards: the result is the same bit for bit.
constexpr int CLOUD_MAX = 4096, CLOUD_TPB = 1024;
__global__ __launch_bounds__(CLOUD_TPB) void csc_cloud_kernel(const int* __restrict__ nbr, const int* __restrict__ cloud_ptr,
                                                              int k, int num_clouds, int* __restrict__ tptr,
                                                              int* __restrict__ unordered) {
    __shared__ int cnt[CLOUD_MAX];
    __shared__ int part[CLOUD_TPB];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const int begin = cloud_ptr[cloud], n = cloud_ptr[cloud + 1] - begin;
    for (int q = tid; q < n; q += CLOUD_TPB) cnt[q] = 0;
    __syncthreads();
    const long e0 = (long)begin * k, ne = (long)n * k;
    // (eight independent loads in flight per thread: the one-load-one-atomic loop ran 20 dependent global round trips per
    //  pass and made this kernel 25 us at 32 x 1024 points, k = 20, with one workgroup per cloud)
    for (long eb = 0; eb < ne; eb += 8 * CLOUD_TPB) {
        int col[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long e = eb + u * CLOUD_TPB + tid;
            const int c = nbr[e0 + min(e, ne - 1)];        // unconditional (clamped) load: see csc_range_kernel
            col[u] = e < ne ? c - begin : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (col[u] >= 0) atomicAdd(&cnt[col[u]], 1);
    }
    __syncthreads();
    // exclusive scan of cnt over the cloud (same partition as csc_scan_kernel: tptr is identical)
    const int per = (n + CLOUD_TPB - 1) / CLOUD_TPB;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int q = lo; q < hi; ++q) s += cnt[q];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < CLOUD_TPB; off <<= 1) {        // inclusive Hillis-Steele scan of the partials
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = (int)e0 + (tid ? part[tid - 1] : 0);          // in-edges of earlier clouds + earlier partitions
    if (cloud == num_clouds - 1 && tid == CLOUD_TPB - 1) tptr[begin + n] = (int)e0 + part[tid];   // = Nt * k
    for (int q = lo; q < hi; ++q) {
        const int c = cnt[q];
        tptr[begin + q] = run;
        cnt[q] = run;                                       // becomes the fill cursor
        run += c;
    }
    __syncthreads();
    for (long eb = 0; eb < ne; eb += 8 * CLOUD_TPB) {       // (nbr is re-read from L2: the loads of a batch fly together)
        int col[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long e = eb + u * CLOUD_TPB + tid;
            const int c = nbr[e0 + min(e, ne - 1)];        // unconditional (clamped) load: see csc_range_kernel
            col[u] = e < ne ? c - begin : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (col[u] >= 0) unordered[atomicAdd(&cnt[col[u]], 1)] = (int)(e0 + eb + u * CLOUD_TPB + tid);
    }
}

// Count, scan and fill of ONE COLUMN RANGE of a cloud per workgroup: RANGES workgroups per cloud instead of one.  The LDS
// atomics of csc_cloud_kernel (2 x N * k per cloud, all in one CU) bound it: 25 us at 32 x 1024 points, k = 20 on 32 of 256
// CUs.  Every workgroup scans ALL edges of its cloud (coalesced, L2-resident) but counts / fills only the targets of its
// range; the targets BELOW the range are counted in registers on the way (no atomics), which gives the range its base
// offset without any hand-off between workgroups.  tptr is identical to csc_cloud_kernel's, the columns come out unordered
// and are ordered by csc_rank_kernel as before.
constexpr int RANGES = 8, RANGE_MAX = CLOUD_MAX / RANGES;
constexpr int RB = 20;   // loads in flight per thread and batch (the kernel is bound by the latency of its global loads: 20 = one batch at 1024 points, k = 20)
__global__ __launch_bounds__(CLOUD_TPB) void csc_range_kernel(const int* __restrict__ nbr, const int* __restrict__ cloud_ptr,
                                                              int k, int num_clouds, int* __restrict__ tptr,
                                                              int* __restrict__ unordered) {
    __shared__ int cnt[RANGE_MAX];
    __shared__ int red[CLOUD_TPB / 64];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const int begin = cloud_ptr[cloud], n = cloud_ptr[cloud + 1] - begin;
    const int rs = (n + RANGES - 1) / RANGES;               // columns per range
    const int r0 = blockIdx.y * rs, r1 = min(r0 + rs, n);
    if (r0 >= n) return;                                    // block-uniform
    for (int q = tid; q < rs; q += CLOUD_TPB) cnt[q] = 0;
    __syncthreads();
    const long e0 = (long)begin * k, ne = (long)n * k;
    int below = 0;
    for (long eb = 0; eb < ne; eb += RB * CLOUD_TPB) {
        int col[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const long e = eb + u * CLOUD_TPB + tid;
            const int c = nbr[e0 + min(e, ne - 1)];        // unconditional (clamped): a conditional load compiles to a branch
            col[u] = e < ne ? c - begin : n;               // with its own vmcnt(0), i.e. twenty serial round trips
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            below += col[u] < r0;
            if (col[u] >= r0 && col[u] < r1) atomicAdd(&cnt[col[u] - r0], 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = below;
    __syncthreads();
    int base = (int)e0;
#pragma unroll
    for (int w = 0; w < CLOUD_TPB / 64; ++w) base += red[w];
    // exclusive scan of the range's counts by the first wavefront (rs <= 512: eight columns per lane)
    if (tid < 64) {
        const int per = (rs + 63) / 64;
        const int lo = min(tid * per, r1 - r0), hi = min(lo + per, r1 - r0);
        int s = 0;
        for (int q = lo; q < hi; ++q) s += cnt[q];
        int incl = s;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (tid >= o) incl += u;
        }
        int run = base + incl - s;
        for (int q = lo; q < hi; ++q) {
            const int c = cnt[q];
            tptr[begin + r0 + q] = run;
            cnt[q] = run;                                   // becomes the fill cursor
            run += c;
        }
        if (tid == 63 && cloud == num_clouds - 1 && r1 == n) tptr[begin + n] = base + incl;   // = Nt * k
    }
    __syncthreads();
    for (long eb = 0; eb < ne; eb += RB * CLOUD_TPB) {
        int col[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const long e = eb + u * CLOUD_TPB + tid;
            const int c = nbr[e0 + min(e, ne - 1)];        // unconditional (clamped): a conditional load compiles to a branch
            col[u] = e < ne ? c - begin : n;               // with its own vmcnt(0), i.e. twenty serial round trips
        }
#pragma unroll
        for (int u = 0; u < RB; ++u)
            if (col[u] >= r0 && col[u] < r1) unordered[atomicAdd(&cnt[col[u] - r0], 1)] = (int)(e0 + eb + u * CLOUD_TPB + tid);
    }
}

// Order every column by edge id without a serial sort: the rank of an entry = the number of smaller entries of its
// column.  One wavefront per column: the lanes hold the column (64 entries at a time), every entry is broadcast
// once (v_readlane) and compared by all lanes -- the column is read from memory once, not once per entry.
__global__ __launch_bounds__(TPB) void csc_rank_kernel(int num_points, const int* __restrict__ tptr,
                                                       const int* __restrict__ unordered, int* __restrict__ tedge) {
    const int lane = threadIdx.x & 63;
    // (readfirstlane: the wave id IS wave-uniform; this lets the compiler keep the column bounds in scalar registers)
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (long)TPB + threadIdx.x) >> 6));
    const int nwaves = (int)((gridDim.x * (long)TPB) >> 6);
    for (int j = wave; j < num_points; j += nwaves) {
        const int lo = tptr[j], deg = tptr[j + 1] - lo;
        for (int ba = 0; ba < deg; ba += 64) {
            const bool valid = ba + lane < deg;
            const int mine = valid ? unordered[lo + ba + lane] : 0x7fffffff;
            int rank = 0;
            for (int bb = 0; bb < deg; bb += 64) {
                const int other = bb == ba ? mine : (bb + lane < deg ? unordered[lo + bb + lane] : 0x7fffffff);
                const int cnt = min(64, deg - bb);
                for (int q = 0; q < cnt; ++q) rank += __builtin_amdgcn_readlane(other, q) < mine;
            }
            if (valid) tedge[lo + rank] = mine;
        }
    }
}
__global__ void csc_permute_kernel(const float2* __restrict__ coef, const int* __restrict__ tedge, long ne,
                                   float2* __restrict__ coefT) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    if (t < ne) coefT[t] = coef[tedge[t]];
}
}  // namespace

// coefT[t] = coef[tedge[t]]: operator coefficients in CSC order, so the transposed applies stream
// them instead of gathering 8 bytes per in-edge.  Once per batch and operator.
DC_EXPORT int dc_csc_permute_coef(const float* coef, const int32_t* tedge, int64_t num_edges, float* coefT,
                                  void* stream) {
    DC_REQUIRE(coef && tedge && coefT, "dc_csc_permute_coef: null pointer");
    DC_REQUIRE(num_edges >= 0, "dc_csc_permute_coef: bad size");
    if (num_edges == 0) return DC_OK;
    hipLaunchKernelGGL(csc_permute_kernel, dim3(dc_cdiv(num_edges, TPB)), dim3(TPB), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(coef), tedge, (long)num_edges, reinterpret_cast<float2*>(coefT));
    DC_CHECK_LAUNCH("dc_csc_permute_coef");
    return DC_OK;
}

// workspace: fill cursors [Nt] + the unordered fill [Nt*k]  (k <= 255)
DC_EXPORT size_t dc_csc_workspace_bytes(int32_t num_points) { return (size_t)num_points * 4 * 256; }

DC_EXPORT int dc_csc_build(const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points,
                           int32_t k, int32_t* tptr, int32_t* tedge, void* workspace, size_t workspace_bytes,
                           void* stream) {
    DC_REQUIRE(nbr && cloud_ptr && tptr && tedge, "dc_csc_build: null pointer");
    DC_REQUIRE(num_clouds >= 0 && num_points >= 0 && k >= 1, "dc_csc_build: bad size");
    DC_REQUIRE((long long)num_points * k < 2147483647LL, "dc_csc_build: edge ids overflow int32");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (num_points == 0 || num_clouds == 0) return DC_OK;
    if (!workspace || workspace_bytes < (size_t)num_points * 4 * ((size_t)k + 1)) {
        dc_set_error("dc_csc_build: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    int* cnt = static_cast<int*>(workspace);
    const long ne = (long)num_points * k;
    dc_zero_words(cnt, num_points, s);
    hipLaunchKernelGGL(csc_count_kernel, dim3(dc_cdiv(ne, TPB)), dim3(TPB), 0, s, nbr, ne, cnt);
    hipLaunchKernelGGL(csc_scan_kernel, dim3(num_clouds), dim3(TPB), 0, s, cloud_ptr, k, num_clouds, cnt, tptr);
    int* unordered = cnt + num_points;
    hipLaunchKernelGGL(csc_fill_kernel, dim3(dc_cdiv(ne, TPB)), dim3(TPB), 0, s, nbr, ne, cnt, unordered);
    hipLaunchKernelGGL(csc_rank_kernel, dim3(std::min<long>(dc_cdiv((long)num_points * 64, TPB), 256 * 16)), dim3(TPB), 0, s,
                       num_points, tptr, unordered, tedge);
    DC_CHECK_LAUNCH("dc_csc_build");
    return DC_OK;
}

// dc_csc_build for clouds of at most 4096 points (max_cloud from the host): count + scan + fill of a cloud in ONE
// workgroup with LDS counters, then the same column ranking: identical tptr / tedge.  Larger clouds: dc_csc_build.
// workspace: the unordered fill [Nt*k] ints.
DC_EXPORT int dc_csc_build_clouds(const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points,
                                  int32_t max_cloud, int32_t k, int32_t* tptr, int32_t* tedge, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    DC_REQUIRE(nbr && cloud_ptr && tptr && tedge, "dc_csc_build_clouds: null pointer");
    DC_REQUIRE(num_clouds >= 0 && num_points >= 0 && k >= 1 && max_cloud >= 0, "dc_csc_build_clouds: bad size");
    DC_REQUIRE(max_cloud <= CLOUD_MAX, "dc_csc_build_clouds: clouds of more than 4096 points: use dc_csc_build");
    DC_REQUIRE((long long)num_points * k < 2147483647LL, "dc_csc_build_clouds: edge ids overflow int32");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (num_points == 0 || num_clouds == 0) return DC_OK;
    if (!workspace || workspace_bytes < (size_t)num_points * 4 * (size_t)k) {
        dc_set_error("dc_csc_build_clouds: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    int* unordered = static_cast<int*>(workspace);
    if (dc_option(DC_OPT_CSC_ONE_WG))                       // A/B switch: the one-workgroup-per-cloud kernel of round 3
        hipLaunchKernelGGL(csc_cloud_kernel, dim3(num_clouds), dim3(CLOUD_TPB), 0, s, nbr, cloud_ptr, k, num_clouds, tptr, unordered);
    else
        hipLaunchKernelGGL(csc_range_kernel, dim3(num_clouds, RANGES), dim3(CLOUD_TPB), 0, s, nbr, cloud_ptr, k, num_clouds, tptr, unordered);
    hipLaunchKernelGGL(csc_rank_kernel, dim3(std::min<long>(dc_cdiv((long)num_points * 64, TPB), 256 * 16)), dim3(TPB), 0, s,
                       num_points, tptr, unordered, tedge);
    DC_CHECK_LAUNCH("dc_csc_build_clouds");
    return DC_OK;
}
