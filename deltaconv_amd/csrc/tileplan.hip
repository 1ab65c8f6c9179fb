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

// Morton order of every cloud by RANKING.
// Sort keys: an 18-bit Morton code (6 bits per axis of the cloud's bounding box: 262 144 cells for <= 4096 points) above the
//   12-bit index inside the cloud -- one 32-bit word per point, unique, so the order is total and deterministic.
// tile_rank_kernel: workgroup (cloud, 64 points) forms the cloud's keys in LDS (chunk 0 also resets the cloud's tile range: -1
//   padding behind the last point, zero unique-row counts: unused tile ids stay empty); four lanes per point each count the keys
//   below their point's key in a quarter of the list (broadcast 16-byte LDS reads, one compare + add per candidate),
//   the four counts are added and the point's id is written at its rank.  No sort passes, no barriers in the loop,
//   N / 64 workgroups per cloud (a bitonic sort in one workgroup per cloud took 24 us at 32 x 1024 points and left 7/8
//   of the chip idle at 8 x 4096; one thread per point over the whole list 111 us there).
__device__ __forceinline__ int dc_wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// First tile of cloud b = the tiles of the clouds before it, sum of ceil(N_c / P): the occupied tile ids are a dense prefix
// [0, T_used) and the unused ids of the host-side bound T trail behind it -- a launch over T workgroups then meets its
// empty tiles last, in the tail, instead of spending resident-workgroup slots on them in the middle of the grid (32 x 1024
// points, P = 64: 512 occupied tiles = exactly two rounds of 256 CUs; with one spare id per cloud in between, 544).
// Every workgroup of the two kernels below recomputes the sum (<= a few thousand cached words): no scan kernel.
__device__ __forceinline__ int block_tile_base(const int* __restrict__ cloud_ptr, int b, int P, int* red /* [4] LDS */) {
    int s = 0;
    for (int c = threadIdx.x; c < b; c += 256) s += (cloud_ptr[c + 1] - cloud_ptr[c] + P - 1) / P;
    s = dc_wave_sum_i(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const int total = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return total;
}

// cloud of a point id: cloud_ptr[b] <= first < cloud_ptr[b + 1].  Every thread tests its own clouds (one memory round trip
// for the workgroup; a binary search is log2(B) DEPENDENT round trips, a fifth of these latency-bound builder kernels).
__device__ __forceinline__ int block_find_cloud(const int* __restrict__ cloud_ptr, int num_clouds, int first, int* slot /* LDS */) {
    for (int c = threadIdx.x; c < num_clouds; c += blockDim.x)
        if (cloud_ptr[c] <= first && first < cloud_ptr[c + 1]) *slot = c;
    __syncthreads();
    const int b = *slot;
    __syncthreads();
    return b;
}

// (round 6: ONE kernel.  The keys were a launch of their own -- workgroup (cloud, 256 points) -- that wrote one word per point for
//  the ranking workgroups to read back; every ranking workgroup now forms the cloud's keys itself, straight into LDS: the bounding
//  box and the Morton codes of N points are N x 24 bytes of cached reads and a few dozen operations per point, less than the
//  launch they replace.  min / max are exact, so the keys -- and the order -- are the same bits.)
__global__ __launch_bounds__(256) void tile_rank_kernel(const float* __restrict__ pos, const int* __restrict__ cloud_ptr,
                                                        int num_clouds, int* __restrict__ plan, DcTilePlan L) {
    __shared__ __attribute__((aligned(16))) unsigned key[MAX_CLOUD];
    __shared__ float red[6][4];
    __shared__ int tred[4];
    const int b = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int begin = cloud_ptr[b], N = cloud_ptr[b + 1] - begin;
    if (chunk * 64 >= N && chunk != 0) return;             // block-uniform
    const int tile0 = block_tile_base(cloud_ptr, b, L.P, tred);
    if (chunk == 0) {                                      // the cloud's tile range: -1 padding behind the last point, zero unique-row counts
        const int tile1 = b + 1 < num_clouds ? tile0 + (max(N, 0) + L.P - 1) / L.P : L.T;   // the last cloud also clears the unused ids
        int* pts = plan + L.o_pts + (long)tile0 * L.P;
        for (int i = max(N, 0) + tid; i < (tile1 - tile0) * L.P; i += 256) pts[i] = -1;
        for (int t = tile0 + tid; t < tile1; t += 256) plan[L.o_nu + t] = 0;
        if (N <= 0) return;
    }
    // bounding box of the cloud
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = tid; i < N; i += 256)
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
    float l3[3], inv3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a][0], h = red[3 + a][0];
        for (int w = 1; w < 4; ++w) {
            l = fminf(l, red[a][w]);
            h = fmaxf(h, red[3 + a][w]);
        }
        l3[a] = l;
        inv3[a] = h > l ? 64.f / (h - l) : 0.f;
    }
    // the sort keys of the whole cloud: 18-bit Morton code above the 12-bit index inside the cloud (unique: a total, deterministic order)
    const int N16 = (N + 15) & ~15;
    for (int i = tid; i < N16; i += 256) {
        unsigned kv = 0xffffffffu;                          // padding sorts last
        if (i < N) {
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                q[a] = (unsigned)min(63, max(0, (int)((pos[(long)(begin + i) * 3 + a] - l3[a]) * inv3[a])));
            kv = ((spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2)) << 12) | (unsigned)i;
        }
        key[i] = kv;
    }
    __syncthreads();
    const int i = chunk * 64 + (tid >> 2), sub = tid & 3;
    const unsigned mine = key[min(i, N - 1)];
    int rank = 0;
#pragma unroll 4
    for (int j = sub * 4; j < N16; j += 16) {              // lanes of a point interleave 16-byte pieces of the key list
        const uint4 kq = *reinterpret_cast<const uint4*>(key + j);
        rank += (kq.x < mine) + (kq.y < mine) + (kq.z < mine) + (kq.w < mine);
    }
    rank += __shfl_xor(rank, 1, 64);
    rank += __shfl_xor(rank, 2, 64);
    if (sub == 0 && i < N) plan[L.o_pts + (long)tile0 * L.P + rank] = begin + i;
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
    __shared__ int cslot;
    const int lo = block_find_cloud(cloud_ptr, num_clouds, first, &cslot);   // cloud of the tile
    const int base = cloud_ptr[lo], N = cloud_ptr[lo + 1] - base;
    const int W = (N + 31) >> 5;
    if (tid < P) pts[tid] = pts_g[tid];
    for (int w = tid; w < W; w += 256) bm[w] = 0u;
    __syncthreads();
    if (tid < P && pts[tid] >= 0) {
        const int j = pts[tid] - base;
        atomicOr(&bm[j >> 5], 1u << (j & 31));
    }
    // (neighbour ids by unconditional, clamped loads, eight in flight: a conditional load compiles to a branch with its own
    //  wait -- the five loads of a thread were five serial round trips, twice)
    int jv[8];
    for (int q0 = 0; q0 < PK; q0 += 8 * 256) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u * 256 + tid, qq = min(q, PK - 1);
            const int p = qq / k, pt = pts[p];
            const int j = nbr[(long)max(pt, 0) * k + (qq - p * k)];
            jv[u] = (q < PK && pt >= 0) ? j - base : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (jv[u] >= 0) atomicOr(&bm[jv[u] >> 5], 1u << (jv[u] & 31));
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
    //  ell_tile.h.  nu holds the unclamped count: a tile with U > min(P*k, capacity) takes the kernels' by-id path.)
    const int Uc = min(U, PK);
    if (tid == 0) plan[L.o_nu + t] = U;                     // the TRUE count: the kernels compare it with what they hold
    int* uq_g = plan + L.o_uniq + (long)t * PK;
    for (int q = tid; q < PK; q += 256) uq_g[q] = uq[min(q, Uc - 1)];
    unsigned short* loc_g = reinterpret_cast<unsigned short*>(plan + L.o_loc) + (long)t * PK;
    for (int q0 = 0; q0 < PK; q0 += 8 * 256) {
        if (PK > 8 * 256 || q0 > 0) {                       // (one batch covers P * k <= 2048: the ids are still in registers)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * 256 + tid, qq = min(q, PK - 1);
                const int p = qq / k, pt = pts[p];
                const int j = nbr[(long)max(pt, 0) * k + (qq - p * k)];
                jv[u] = (q < PK && pt >= 0) ? j - base : -1;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u * 256 + tid;
            const int j = max(jv[u], 0);
            const int l = jv[u] >= 0 ? pre[j >> 5] + __popc(bm[j >> 5] & ((1u << (j & 31)) - 1u)) : 0;
            if (q < PK) loc_g[q] = (unsigned short)l;
        }
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

// ---- transposed plan (tile_plan.h, second half): one workgroup per tile ---------------------------------------------
// Inputs: the forward plan's tiles (pts) and the CSC of the graph (tptr, tedge: in-edges per point, ascending edge id).
//   1. in-degrees of the tile's targets; rank by (degree descending, position ascending) -> lane group of every target;
//   2. start of the tile's range: the cloud's base + the rounded lengths of the cloud's earlier tiles (recomputed by every
//      workgroup from pts + tptr: <= 4096 cached words, no scan kernel, no atomics);
//   3. bitmap of the sources over the cloud's local ids -> prefix popcounts -> unique list + the tile-local index of every
//      in-edge's source (as tile_unique_kernel does for the neighbours);
//   4. records + edge ids in (lane group, ascending edge id) order.
__global__ __launch_bounds__(1024) void tileT_build_kernel(const int* __restrict__ plan, DcTilePlan L, const int* __restrict__ tptr,
                                                           const int* __restrict__ tedge, const int* __restrict__ cloud_ptr,
                                                           int num_clouds, int* __restrict__ planT, DcTilePlanT LT) {
    __shared__ int pts[64], deg[64], spt[64], sdeg[64], sbeg[64], scol[64];
    __shared__ unsigned bm[MAX_CLOUD / 32];
    __shared__ int pre[MAX_CLOUD / 32 + 1];
    __shared__ int red[16];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int P = L.P, k = L.k;
    int4* tg_g = reinterpret_cast<int4*>(planT + LT.o_tg) + (long)t * P;
    int4* hdr_g = reinterpret_cast<int4*>(planT + LT.o_hdr) + t;
    const int* pts_g = plan + L.o_pts + (long)t * P;
    const int first = pts_g[0];
    if (first < 0) {                                        // unused tile id: empty (block-uniform)
        if (tid < P) tg_g[tid] = make_int4(-1, 0, 0, 0);
        if (tid == 0) *hdr_g = make_int4(0, 0, 0, 0);
        return;
    }
    __shared__ int cslot;
    const int b = block_find_cloud(cloud_ptr, num_clouds, first, &cslot);   // cloud of the tile
    const int base = cloud_ptr[b], N = cloud_ptr[b + 1] - base;
    // first tile of the cloud (block_tile_base over 1024 threads)
    int tsum = 0;
    for (int c = tid; c < b; c += 1024) tsum += (cloud_ptr[c + 1] - cloud_ptr[c] + P - 1) / P;
    tsum = dc_wave_sum_i(tsum);
    if ((tid & 63) == 0) red[tid >> 6] = tsum;
    if (tid < 64) {
        const int j = tid < P ? pts_g[tid] : -1;
        pts[tid] = j;
        deg[tid] = j >= 0 ? tptr[j + 1] - tptr[j] : 0;
    }
    const int W = (N + 31) >> 5;
    for (int w = tid; w < W; w += 1024) bm[w] = 0u;
    __syncthreads();
    int tile0 = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tile0 += red[w];
    const int tc = t - tile0;                               // tile index inside the cloud
    // 2. rounded lengths of the cloud's earlier tiles (segments of P lanes = one tile)
    int part = 0;
    for (int q0 = 0; q0 < tc * P; q0 += 1024) {
        const int q = q0 + tid;
        // (unconditional, clamped loads: a conditional load compiles to a branch with its own wait)
        const int j = plan[L.o_pts + (long)tile0 * P + min(q, tc * P - 1)];
        const int jj = max(j, 0);
        int d = tptr[jj + 1] - tptr[jj];
        d = (q < tc * P && j >= 0) ? d : 0;
        for (int o = P >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if ((tid & (P - 1)) == 0) part += (d + 3) & ~3;
    }
    part = dc_wave_sum_i(part);
    __syncthreads();                                        // (every thread has read red: it is free again)
    if ((tid & 63) == 0) red[tid >> 6] = part;
    // 1. degree order (64 x 64 compares)
    if (tid < 64) {
        const int d = deg[tid];
        int r = 0;
        for (int q = 0; q < 64; ++q) r += (deg[q] > d) || (deg[q] == d && q < tid);
        spt[r] = pts[tid];
        sdeg[r] = d;
        scol[r] = pts[tid] >= 0 ? tptr[pts[tid]] : 0;
    }
    __syncthreads();
    long toff = (((long)base * k + 3) & ~3L) + 4L * (tile0 + b);
#pragma unroll
    for (int w = 0; w < 16; ++w) toff += red[w];
    if (tid < 64) {                                         // exclusive scan of the sorted degrees (one wavefront)
        int v = sdeg[tid];
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(v, o, 64);
            if (tid >= o) v += u;
        }
        sbeg[tid] = v - sdeg[tid];
    }
    // 3. bitmap of the sources: sixteen lanes per target walk its list (one or two entries per lane at k = 20)
    const int g = tid >> 4, sub = tid & 15;
    const int dgt = sdeg[g], col = scol[g];
    // the first 64 entries of the list by unconditional, clamped loads (four in flight per lane; kept for the record pass)
    int ev[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ev[u] = tedge[dgt > 0 ? col + min(sub + 16 * u, dgt - 1) : 0];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (sub + 16 * u < dgt) {
            const int j = ev[u] / k - base;
            atomicOr(&bm[j >> 5], 1u << (j & 31));
        }
    for (int r = 64 + sub; r < dgt; r += 16) {
        const int j = tedge[col + r] / k - base;
        atomicOr(&bm[j >> 5], 1u << (j & 31));
    }
    __syncthreads();
    const int Et = sbeg[63] + sdeg[63];
    if (tid == 0) pre[0] = 0;
    if (tid < 128) pre[tid + 1] = tid < W ? __popc(bm[tid]) : 0;
    __syncthreads();
    for (int off = 1; off < 128; off <<= 1) {
        int v = 0;
        if (tid < 128 && tid >= off) v = pre[tid + 1 - off];
        __syncthreads();
        if (tid < 128 && tid >= off) pre[tid + 1] += v;
        __syncthreads();
    }
    const int U = pre[W];
    // unique list: four lanes per bitmap word, each emits the set bits of one byte
    int* uq_g = planT + LT.o_uniq + (long)t * LT.UQ;
    if (tid < 4 * W) {
        const int w = tid >> 2, by = tid & 3;
        const unsigned word = bm[w];
        unsigned bits = (word >> (8 * by)) & 0xffu;
        int o = pre[w] + __popc(word & ((1u << (8 * by)) - 1u));
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            if (o < LT.UQ) uq_g[o] = base + (w << 5) + 8 * by + bit;
            ++o;
        }
    }
    if (U < LT.UQ && U > 0) {                               // the list's tail repeats its last id = the highest set bit
        int last = 0;
        for (int w = W - 1; w >= 0; --w)
            if (bm[w]) { last = base + (w << 5) + 31 - __clz(bm[w]); break; }
        for (int q = U + tid; q < LT.UQ; q += 1024) uq_g[q] = last;
    }
    // 4. records
    unsigned* rec_g = reinterpret_cast<unsigned*>(planT + LT.o_rec) + toff;
    int* edge_g = planT + LT.o_edge + toff;
    {
        const int eb = sbeg[g];
        auto emit = [&](int r, int e) {
            const int i = e / k, j = i - base;
            const int l = pre[j >> 5] + __popc(bm[j >> 5] & ((1u << (j & 31)) - 1u));
            rec_g[eb + r] = (unsigned)l | ((unsigned)(e - i * k) << 16);
            edge_g[eb + r] = e;
        };
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (sub + 16 * u < dgt) emit(sub + 16 * u, ev[u]);
        for (int r = 64 + sub; r < dgt; r += 16) emit(r, tedge[col + r]);
    }
    if (tid < ((Et + 3) & ~3) - Et) {                       // padding of the range: harmless entries
        rec_g[Et + tid] = 0u;
        edge_g[Et + tid] = base * k;
    }
    if (tid < P) tg_g[tid] = make_int4(spt[tid], sbeg[tid], sdeg[tid], 0);
    if (tid == 0) *hdr_g = make_int4(U, (int)toff, Et, 0);
}

// operator coefficients in tile order: out[p] = coef[edge[p]]; the words between the tile ranges were never written by the
// builder (arbitrary bits): anything that is not an edge id gives (0, 0) and no memory access
__global__ void tileT_permute_kernel(const float2* __restrict__ coefA, const float2* __restrict__ coefB, const int* __restrict__ edge,
                                     long ep, unsigned ne, float2* __restrict__ outA, float2* __restrict__ outB) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t < ep) {
        const unsigned e = (unsigned)edge[t];
        const bool ok = e < ne;
        outA[t] = ok ? coefA[e] : make_float2(0.f, 0.f);
        if (coefB) outB[t] = ok ? coefB[e] : make_float2(0.f, 0.f);
    }
}

int check_plan_args(const char* name, int num_points, int num_clouds, int k, int P) {
    if (num_points < 0 || num_clouds < 0 || k < 2 || k % 2 || k > 64 || (P != 32 && P != 64) || P * k > MAX_PK) {
        dc_set_error("%s: bad size (num_points=%d num_clouds=%d k=%d P=%d; k even, 2 <= k <= 64, P in {32, 64}, P*k <= %d)",
                     name, num_points, num_clouds, k, P, MAX_PK);
        return DC_ERR_ARG;
    }
    return DC_OK;
}
}  // namespace

DC_EXPORT int32_t dc_tile_plan_tiles(int32_t num_points, int32_t num_clouds, int32_t max_cloud, int32_t P) {
    return dc_tile_plan_num_tiles(num_points, num_clouds, max_cloud, P);
}

DC_EXPORT size_t dc_tile_plan_words(int32_t num_tiles, int32_t k, int32_t P) {
    return (size_t)dc_tile_plan_layout(num_tiles, k, P).words;
}

DC_EXPORT int32_t dc_tile_plan_max_cloud(void) { return MAX_CLOUD; }

DC_EXPORT int dc_tile_plan_build(const float* pos, const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds,
                                 int32_t num_points, int32_t max_cloud, int32_t k, int32_t P, int32_t* plan, void* stream) {
    DC_REQUIRE(pos && nbr && cloud_ptr && plan, "dc_tile_plan_build: null pointer");
    if (int rc = check_plan_args("dc_tile_plan_build", num_points, num_clouds, k, P)) return rc;
    DC_REQUIRE(max_cloud >= 1 && max_cloud <= MAX_CLOUD, "dc_tile_plan_build: clouds of more than %d points are not supported (max_cloud=%d)",
               MAX_CLOUD, max_cloud);
    if (num_points == 0 || num_clouds == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DcTilePlan L = dc_tile_plan_layout(dc_tile_plan_num_tiles(num_points, num_clouds, max_cloud, P), k, P);
    hipLaunchKernelGGL(tile_rank_kernel, dim3(num_clouds, dc_cdiv(max_cloud, 64)), dim3(256), 0, s, pos, cloud_ptr, num_clouds, plan, L);
    hipLaunchKernelGGL(tile_unique_kernel, dim3(L.T), dim3(256), 0, s, nbr, cloud_ptr, num_clouds, plan, L);
    DC_CHECK_LAUNCH("dc_tile_plan_build");
    return DC_OK;
}

// ---- transposed plan ---------------------------------------------------------------------------------------------------
DC_EXPORT size_t dc_tile_plan_T_words(int32_t num_points, int32_t num_clouds, int32_t num_tiles, int32_t k, int32_t P) {
    return (size_t)dc_tile_plan_T_layout(num_points, num_clouds, num_tiles, k, P).words;
}
// entries of the edge-ordered sections (= the length of an operator's coefficient array in tile order)
DC_EXPORT int64_t dc_tile_plan_T_edges(int32_t num_points, int32_t num_clouds, int32_t num_tiles, int32_t k) {
    return dc_tile_plan_T_edges((long)num_points, k, num_tiles, num_clouds);
}
// word offset of the edge-id section inside the blob (the permutation handed to dc_csc_permute_coef)
DC_EXPORT int64_t dc_tile_plan_T_edge_offset(int32_t num_points, int32_t num_clouds, int32_t num_tiles, int32_t k, int32_t P) {
    return dc_tile_plan_T_layout(num_points, num_clouds, num_tiles, k, P).o_edge;
}

DC_EXPORT int dc_tile_plan_T_build(const int32_t* plan, const int32_t* tptr, const int32_t* tedge, const int32_t* cloud_ptr,
                                   int32_t num_clouds, int32_t num_points, int32_t max_cloud, int32_t k, int32_t P,
                                   int32_t* planT, void* stream) {
    DC_REQUIRE(plan && tptr && tedge && cloud_ptr && planT, "dc_tile_plan_T_build: null pointer");
    if (int rc = check_plan_args("dc_tile_plan_T_build", num_points, num_clouds, k, P)) return rc;
    DC_REQUIRE(max_cloud >= 1 && max_cloud <= MAX_CLOUD, "dc_tile_plan_T_build: clouds of more than %d points are not supported (max_cloud=%d)",
               MAX_CLOUD, max_cloud);
    DC_REQUIRE((long long)num_points * k < 2147483647LL - 4LL * (num_points / P + 2 * num_clouds) - 16, "dc_tile_plan_T_build: edge ids overflow int32");
    if (num_points == 0 || num_clouds == 0) return DC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T = dc_tile_plan_num_tiles(num_points, num_clouds, max_cloud, P);
    const DcTilePlan L = dc_tile_plan_layout(T, k, P);
    const DcTilePlanT LT = dc_tile_plan_T_layout(num_points, num_clouds, T, k, P);
    hipLaunchKernelGGL(tileT_build_kernel, dim3(T), dim3(1024), 0, s, plan, L, tptr, tedge, cloud_ptr, num_clouds, planT, LT);
    DC_CHECK_LAUNCH("dc_tile_plan_T_build");
    return DC_OK;
}

// coefTt[EP, 2] = the operator's coefficients coef[Nt * k, 2] in the tile order of planT (EP = dc_tile_plan_T_edges).
// Once per batch and operator, like dc_csc_permute_coef for the CSC order; coefB / coefBTt (may be NULL) = a second
// operator over the same graph in the same launch (grad and div of a batch).
DC_EXPORT int dc_tile_plan_T_permute_coef(const float* coef, const float* coefB, const int32_t* planT, int32_t num_points,
                                          int32_t num_clouds, int32_t num_tiles, int32_t k, int32_t P, float* coefTt,
                                          float* coefBTt, void* stream) {
    DC_REQUIRE(coef && planT && coefTt && (!coefB || coefBTt), "dc_tile_plan_T_permute_coef: null pointer");
    if (int rc = check_plan_args("dc_tile_plan_T_permute_coef", num_points, num_clouds, k, P)) return rc;
    if (num_points == 0) return DC_OK;
    const DcTilePlanT LT = dc_tile_plan_T_layout(num_points, num_clouds, num_tiles, k, P);
    hipLaunchKernelGGL(tileT_permute_kernel, dim3(dc_cdiv(LT.EP, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(coef), reinterpret_cast<const float2*>(coefB), planT + LT.o_edge, LT.EP,
                       (unsigned)((long)num_points * k), reinterpret_cast<float2*>(coefTt), reinterpret_cast<float2*>(coefBTt));
    DC_CHECK_LAUNCH("dc_tile_plan_T_permute_coef");
    return DC_OK;
}
