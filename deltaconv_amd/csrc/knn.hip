// Batched brute-force k-nearest-neighbour graph (self included) -- replaces the reference's
// knn_graph(pos, k, batch, loop=True, flow='target_to_source') call sites
// (/root/reference/deltaconv/models/deltanet_base.py:52,63; third-party torch_cluster).
//
// Order (bit-exact contract shared with oracle/geometry.py:knn): fp32 squared distance
// ((dx*dx + dy*dy) + dz*dz) evaluated WITHOUT fma contraction, ascending, ties by lower index.
//
// Mapping: one cloud per blockIdx.y; the cloud's points are staged through LDS in SoA tiles and
// read as broadcasts.  P lanes share one query: lane r scans candidates r, r+P, ... keeping a
// sorted top-K list in registers (fully unrolled insertion); the P lists are then merged
// through LDS with K rounds of a lexicographic (distance, index) min over the P-lane group.
// Compute-bound (N^2 distance evaluations per cloud), not HBM-bound: 12 B/point in, 4k B out.
#include "common.h"
#include <math.h>

namespace {

constexpr int KNN_THREADS = 256;
constexpr int KNN_TILE = 2048;  // candidates staged per LDS tile: 3 x 2048 x 4 B = 24 KiB

template <int K>
struct TopK {
    float d[K];
    int id[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            d[s] = INFINITY;
            id[s] = 0x7fffffff;
        }
    }
    // Insert (nd, nid) AFTER every entry with distance <= nd.  Candidates arrive in ascending
    // index order, so equal distances stay ordered by index.
    __device__ __forceinline__ void push(float nd, int nid) {
        if (nd < d[K - 1]) {
#pragma unroll
            for (int s = K - 1; s > 0; --s) {
                const bool shift = nd < d[s - 1];
                const bool place = nd < d[s];
                const float vd = shift ? d[s - 1] : nd;
                const int vi = shift ? id[s - 1] : nid;
                d[s] = place ? vd : d[s];
                id[s] = place ? vi : id[s];
            }
            if (nd < d[0]) {
                d[0] = nd;
                id[0] = nid;
            }
        }
    }
};

template <int K, int P>
__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const float* __restrict__ pos,
                                                          const int* __restrict__ cloud_ptr, int k,
                                                          int* __restrict__ nbr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Q = KNN_THREADS / P;
    float* tx = reinterpret_cast<float*>(smem);
    float* ty = tx + KNN_TILE;
    float* tz = ty + KNN_TILE;
    float* ld = tz + KNN_TILE;                               // [K][KNN_THREADS] (P > 1 only)
    int* li = reinterpret_cast<int*>(ld + KNN_THREADS * K);  // [K][KNN_THREADS]

    const int cloud = blockIdx.y;
    const int begin = cloud_ptr[cloud];
    const int n = cloud_ptr[cloud + 1] - begin;
    const int q0 = blockIdx.x * Q;
    if (q0 >= n) return;  // block-uniform
    const int tid = threadIdx.x;
    const int r = tid % P;
    const int q = q0 + tid / P;
    const bool active = q < n;

    float px = 0.f, py = 0.f, pz = 0.f;
    if (active) {
        const float* p = pos + 3 * (size_t)(begin + q);
        px = p[0];
        py = p[1];
        pz = p[2];
    }
    TopK<K> best;
    best.init();

    for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
        const int tn = min(KNN_TILE, n - t0);
        __syncthreads();
        for (int c = tid; c < tn; c += KNN_THREADS) {
            const float* p = pos + 3 * (size_t)(begin + t0 + c);
            tx[c] = p[0];
            ty[c] = p[1];
            tz[c] = p[2];
        }
        __syncthreads();
        if (active) {
            for (int c = r; c < tn; c += P) {
                const float dx = __fsub_rn(px, tx[c]);
                const float dy = __fsub_rn(py, ty[c]);
                const float dz = __fsub_rn(pz, tz[c]);
                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                best.push(d2, t0 + c);
            }
        }
    }

    if (P == 1) {
        if (active) {
            int* out = nbr + (size_t)(begin + q) * k;
#pragma unroll
            for (int s = 0; s < K; ++s)
                if (s < k) out[s] = begin + best.id[s];
        }
        return;
    }

    // ---- merge the P sorted lists of each query (all threads of the block take part) ----
#pragma unroll
    for (int s = 0; s < K; ++s) {
        ld[s * KNN_THREADS + tid] = best.d[s];
        li[s * KNN_THREADS + tid] = best.id[s];
    }
    __syncthreads();
    int h = 0;
    for (int s = 0; s < k; ++s) {
        const float hd = (h < K) ? ld[h * KNN_THREADS + tid] : INFINITY;
        const int hi = (h < K) ? li[h * KNN_THREADS + tid] : 0x7fffffff;
        float bd = hd;
        int bi = hi;
#pragma unroll
        for (int o = 1; o < P; o <<= 1) {
            const float od = __shfl_xor(bd, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            const bool take = (od < bd) || (od == bd && oi < bi);
            bd = take ? od : bd;
            bi = take ? oi : bi;
        }
        if (hi == bi && hd == bd) ++h;  // the (unique) winning lane pops its head
        if (r == 0 && active) nbr[(size_t)(begin + q) * k + s] = begin + bi;
    }
}

template <int K, int P>
int launch_knn(const float* pos, const int* cloud_ptr, int num_clouds, int max_cloud, int k, int* nbr,
               hipStream_t stream) {
    constexpr int Q = KNN_THREADS / P;
    size_t lds = 3 * KNN_TILE * sizeof(float);
    if (P > 1) lds += (size_t)KNN_THREADS * K * 8;
    static bool attr_done = false;
    if (!attr_done) {  // > 64 KiB of dynamic LDS needs an explicit opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_kernel<K, P>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid(dc_cdiv(max_cloud, Q), num_clouds);
    hipLaunchKernelGGL((knn_kernel<K, P>), grid, dim3(KNN_THREADS), lds, stream, pos, cloud_ptr, k, nbr);
    DC_CHECK_LAUNCH("dc_knn");
    return DC_OK;
}

template <int P>
int dispatch_k(const float* pos, const int* cloud_ptr, int num_clouds, int max_cloud, int k, int* nbr,
               hipStream_t stream) {
    if (k <= 10) return launch_knn<10, P>(pos, cloud_ptr, num_clouds, max_cloud, k, nbr, stream);
    if (k <= 16) return launch_knn<16, P>(pos, cloud_ptr, num_clouds, max_cloud, k, nbr, stream);
    if (k <= 20) return launch_knn<20, P>(pos, cloud_ptr, num_clouds, max_cloud, k, nbr, stream);
    if (k <= 30) return launch_knn<30, P>(pos, cloud_ptr, num_clouds, max_cloud, k, nbr, stream);
    if (k <= 40) return launch_knn<40, P>(pos, cloud_ptr, num_clouds, max_cloud, k, nbr, stream);
    return launch_knn<64, P>(pos, cloud_ptr, num_clouds, max_cloud, k, nbr, stream);
}

}  // namespace

DC_EXPORT int dc_knn(const float* pos, const int32_t* cloud_ptr, int32_t num_clouds, int32_t max_cloud_size,
                     int32_t k, int32_t lanes_per_query, int32_t* nbr, void* stream) {
    DC_REQUIRE(pos && cloud_ptr && nbr, "dc_knn: null pointer");
    DC_REQUIRE(k >= 1 && k <= 64, "dc_knn: k=%d outside [1,64]", k);
    DC_REQUIRE(num_clouds >= 0 && max_cloud_size >= 0, "dc_knn: negative size");
    DC_REQUIRE(lanes_per_query == 0 || lanes_per_query == 1 || lanes_per_query == 8,
               "dc_knn: lanes_per_query must be 0 (auto), 1 or 8");
    if (num_clouds == 0 || max_cloud_size == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lanes_per_query == 1) return dispatch_k<1>(pos, cloud_ptr, num_clouds, max_cloud_size, k, nbr, s);
    return dispatch_k<8>(pos, cloud_ptr, num_clouds, max_cloud_size, k, nbr, s);
}
