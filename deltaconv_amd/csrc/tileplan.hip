// Tile plan builder: Morton order of every cloud -> tiles of P points -> unique neighbour rows + tile-local indices.
// Layout and motivation: tile_plan.h.  Built once per batch (positions + kNN graph only), shared by every forward
// apply / max-aggregation of the step, like the CSC of the transposed applies (csc.hip).  Stream-ordered kernels
// only (capturable), deterministic (no float atomics; integer LDS atomics only to set bits of a bitmap).
//
// The reference has no counterpart: torch_sparse / torch_scatter gather every neighbour row from global memory
// (call sites /root/reference/deltaconv/nn/deltaconv.py:52-57,66, geometry/operators.py:27-43).
#include "common.h"
#include "tile_plan.h"

namespace {
constexpr int MAX_CLOUD = 4096;      // points of one cloud held in LDS by the ordering kernel / bits of the tile bitmap
constexpr int MAX_PK = 2048;         // P * k

__device__ __forceinline__ unsigned spread10(unsigned x) {   // 10 bits -> every third bit
    x &= 0x3ff;
    x = (x | (x << 16)) & 0x30000ff;
    x = (x | (x << 8)) & 0x300f00f;
    x = (x | (x << 4)) & 0x30c30c3;
    x = (x | (x << 2)) & 0x9249249;
    return x;
}

__global__ void tile_clear_kernel(int* __restrict__ plan, DcTilePlan L) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)L.T * L.P) plan[L.o_pts + i] = -1;
    if (i < L.T) plan[L.o_nu + i] = 0;
}

// one workgroup per cloud: bounding box -> 30-bit Morton keys -> bitonic sort of (key, local index) in LDS -> the
// cloud's tiles (point ids in Morton order, -1 padding).  Ties in the key are broken by the index: deterministic.
__global__ __launch_bounds__(1024) void tile_order_kernel(const float* __restrict__ pos, const int* __restrict__ cloud_ptr,
                                                          int* __restrict__ plan, DcTilePlan L) {
    __shared__ unsigned long long key[MAX_CLOUD];
    __shared__ float red[6][16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int begin = cloud_ptr[b], N = cloud_ptr[b + 1] - begin;
    if (N <= 0) return;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = tid; i < N; i += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pos[(long)(begin + i) * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = -dc_wave_max(-lo[a]);
        hi[a] = dc_wave_max(hi[a]);
    }
    if ((tid & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[a][tid >> 6] = lo[a];
            red[3 + a][tid >> 6] = hi[a];
        }
    __syncthreads();
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a][0], h = red[3 + a][0];
        for (int w = 1; w < 16; ++w) {
            l = fminf(l, red[a][w]);
            h = fmaxf(h, red[3 + a][w]);
        }
        lo[a] = l;
        inv[a] = h > l ? 1024.f / (h - l) : 0.f;
    }
    int M = 2;
    while (M < N) M <<= 1;
    for (int i = tid; i < M; i += 1024) {
        unsigned long long kv = ~0ull;
        if (i < N) {
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float f = (pos[(long)(begin + i) * 3 + a] - lo[a]) * inv[a];
                q[a] = (unsigned)min(1023, max(0, (int)f));
            }
            const unsigned m = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
            kv = ((unsigned long long)m << 32) | (unsigned)i;
        }
        key[i] = kv;
    }
    for (int size = 2; size <= M; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (M >> 1); t += 1024) {
                const int i = ((t / stride) * stride << 1) + (t % stride), j = i + stride;
                const unsigned long long a = key[i], c = key[j];
                const bool up = (i & size) == 0;
                if ((a > c) == up) {
                    key[i] = c;
                    key[j] = a;
                }
            }
        }
    __syncthreads();
    const int tile0 = dc_tile_base(begin, b, L.P);
    const int ntile = (N + L.P - 1) / L.P;
    int* pts = plan + L.o_pts + (long)tile0 * L.P;
    for (int i = tid; i < ntile * L.P; i += 1024) pts[i] = i < N ? begin + (int)(unsigned)(key[i] & 0xffffffffu) : -1;
}

// one workgroup per tile: bitmap of the tile's rows over the cloud's local ids -> prefix popcounts -> unique list
// (ascending id) and the tile-local index of every (point, slot) and of every point itself.
__global__ __launch_bounds__(256) void tile_unique_kernel(const int* __restrict__ nbr, const int* __restrict__ cloud_ptr,
                                                          int num_clouds, int* __restrict__ plan, DcTilePlan L) {
    __shared__ int pts[64];
    __shared__ unsigned bm[MAX_CLOUD / 32];
    __shared__ int pre[MAX_CLOUD / 32 + 1];
    __shared__ int uq[MAX_PK];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int P = L.P, k = L.k, PK = L.PK;
    const int* pts_g = plan + L.o_pts + (long)t * P;
    const int first = pts_g[0];
    if (first < 0) return;                                  // unused tile id (stays empty: nu = 0)
    int lo = 0, hi = num_clouds;                            // cloud of the tile: cloud_ptr[lo] <= first < cloud_ptr[lo + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cloud_ptr[mid] <= first) lo = mid; else hi = mid;
    }
    const int base = cloud_ptr[lo], N = cloud_ptr[lo + 1] - base;
    const int W = (N + 31) >> 5;
    if (tid < P) pts[tid] = pts_g[tid];
    for (int w = tid; w < W; w += 256) bm[w] = 0u;
    __syncthreads();
    if (tid < P && pts[tid] >= 0) {
        const int j = pts[tid] - base;
        atomicOr(&bm[j >> 5], 1u << (j & 31));
    }
    for (int q = tid; q < PK; q += 256) {
        const int p = q / k, pt = pts[p];
        if (pt >= 0) {
            const int j = nbr[(long)pt * k + (q - p * k)] - base;
            atomicOr(&bm[j >> 5], 1u << (j & 31));
        }
    }
    __syncthreads();
    if (tid == 0) pre[0] = 0;
    if (tid < 128) {                                        // inclusive scan of the word popcounts (W <= 128)
        int v = tid < W ? __popc(bm[tid]) : 0;
        pre[tid + 1] = v;
    }
    __syncthreads();
    for (int off = 1; off < 128; off <<= 1) {
        int v = 0;
        if (tid < 128 && tid >= off) v = pre[tid + 1 - off];
        __syncthreads();
        if (tid < 128 && tid >= off) pre[tid + 1] += v;
        __syncthreads();
    }
    const int U = pre[W];
    for (int w = tid; w < W; w += 256) {
        unsigned bits = bm[w];
        int o = pre[w];
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            if (o < MAX_PK) uq[o] = base + (w << 5) + bit;
            ++o;
        }
    }
    __syncthreads();
    // (U <= P + P*k can exceed the uniq section by at most P entries when a point is not its own neighbour, i.e.
    //  more than k coincident points; the section holds P*k: such a tile keeps its first P*k rows in the list and the
    //  kernels fetch rows whose local index is not below min(U, P*k, capacity) from global memory by id -- see
    //  ell_tile.h.  nu is stored clamped.)
    const int Uc = min(U, PK);
    if (tid == 0) plan[L.o_nu + t] = Uc;
    int* uq_g = plan + L.o_uniq + (long)t * PK;
    for (int q = tid; q < PK; q += 256) uq_g[q] = uq[min(q, Uc - 1)];
    unsigned short* loc_g = reinterpret_cast<unsigned short*>(plan + L.o_loc) + (long)t * PK;
    for (int q = tid; q < PK; q += 256) {
        const int p = q / k, pt = pts[p];
        int l = 0;
        if (pt >= 0) {
            const int j = nbr[(long)pt * k + (q - p * k)] - base;
            l = pre[j >> 5] + __popc(bm[j >> 5] & ((1u << (j & 31)) - 1u));
        }
        loc_g[q] = (unsigned short)l;
    }
    unsigned short* self_g = reinterpret_cast<unsigned short*>(plan + L.o_self) + (long)t * P;
    if (tid < P) {
        int l = 0;
        if (pts[tid] >= 0) {
            const int j = pts[tid] - base;
            l = pre[j >> 5] + __popc(bm[j >> 5] & ((1u << (j & 31)) - 1u));
        }
        self_g[tid] = (unsigned short)l;
    }
}

__global__ void tile_permute_kernel(const float2* __restrict__ coef, const int* __restrict__ plan, DcTilePlan L,
                                    float2* __restrict__ coefP) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)L.T * L.PK) return;
    const long t = i / L.PK;
    const int q = (int)(i - t * L.PK), p = q / L.k;
    const int pt = plan[L.o_pts + t * L.P + p];
    coefP[i] = pt >= 0 ? coef[(long)pt * L.k + (q - p * L.k)] : make_float2(0.f, 0.f);
}

int check_plan_args(const char* name, int num_points, int num_clouds, int k, int P) {
    if (num_points < 0 || num_clouds < 0 || k < 1 || (P != 32 && P != 64) || P * k > MAX_PK || (P * k) % 8) {
        dc_set_error("%s: bad size (num_points=%d num_clouds=%d k=%d P=%d; P in {32, 64}, P*k <= %d)", name, num_points,
                     num_clouds, k, P, MAX_PK);
        return DC_ERR_ARG;
    }
    return DC_OK;
}
}  // namespace

DC_EXPORT int32_t dc_tile_plan_tiles(int32_t num_points, int32_t num_clouds, int32_t P) {
    return (num_points + P - 1) / P + num_clouds;
}

DC_EXPORT size_t dc_tile_plan_words(int32_t num_points, int32_t num_clouds, int32_t k, int32_t P) {
    return (size_t)dc_tile_plan_layout(num_points, num_clouds, k, P).words;
}

DC_EXPORT int32_t dc_tile_plan_max_cloud(void) { return MAX_CLOUD; }

DC_EXPORT int dc_tile_plan_build(const float* pos, const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds,
                                 int32_t num_points, int32_t max_cloud, int32_t k, int32_t P, int32_t* plan, void* stream) {
    DC_REQUIRE(pos && nbr && cloud_ptr && plan, "dc_tile_plan_build: null pointer");
    if (int rc = check_plan_args("dc_tile_plan_build", num_points, num_clouds, k, P)) return rc;
    DC_REQUIRE(max_cloud <= MAX_CLOUD, "dc_tile_plan_build: clouds of more than %d points are not supported (max_cloud=%d)",
               MAX_CLOUD, max_cloud);
    if (num_points == 0 || num_clouds == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DcTilePlan L = dc_tile_plan_layout(num_points, num_clouds, k, P);
    hipLaunchKernelGGL(tile_clear_kernel, dim3(dc_cdiv((long)L.T * L.P, 256)), dim3(256), 0, s, plan, L);
    hipLaunchKernelGGL(tile_order_kernel, dim3(num_clouds), dim3(1024), 0, s, pos, cloud_ptr, plan, L);
    hipLaunchKernelGGL(tile_unique_kernel, dim3(L.T), dim3(256), 0, s, nbr, cloud_ptr, num_clouds, plan, L);
    DC_CHECK_LAUNCH("dc_tile_plan_build");
    return DC_OK;
}

// coefP[T][P*k][2]: an operator's coefficients in tile order (zeros for padding), so a tile's coefficients are one
// contiguous LDS-DMA copy.  Once per batch and operator.
DC_EXPORT int dc_tile_permute_coef(const float* coef, const int32_t* plan, int32_t num_points, int32_t num_clouds, int32_t k,
                                   int32_t P, float* coefP, void* stream) {
    DC_REQUIRE(coef && plan && coefP, "dc_tile_permute_coef: null pointer");
    if (int rc = check_plan_args("dc_tile_permute_coef", num_points, num_clouds, k, P)) return rc;
    if (num_points == 0 || num_clouds == 0) return DC_OK;
    const DcTilePlan L = dc_tile_plan_layout(num_points, num_clouds, k, P);
    hipLaunchKernelGGL(tile_permute_kernel, dim3(dc_cdiv((long)L.T * L.PK, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(coef), plan, L, reinterpret_cast<float2*>(coefP));
    DC_CHECK_LAUNCH("dc_tile_permute_coef");
    return DC_OK;
}
