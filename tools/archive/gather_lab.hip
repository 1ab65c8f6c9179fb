// Gather lab: standalone micro-benchmark of variants of the ELL "grad" apply
//   out[2i+a, c] = sum_s G[i,s,a] * x[nbr[i,s], c]     (Nt = B*N points, k neighbours, C channels)
// to find what actually bounds the neighbour-row gathers on MI355X.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_lab.hip -o /tmp/gather_lab && /tmp/gather_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct alignas(16) F4 { float v[4]; };
struct G2 { float a, b; };

__device__ __forceinline__ long xcd_block(int remap) {
    const long b = blockIdx.x, nb = gridDim.x;
    if (!remap) return b;
    const long q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

// ---- A: current production structure (LDS-staged ids/coefs, 16-B gathers, unroll U) ------------
template <int U, int TPB>
__global__ __launch_bounds__(TPB) void k_staged(long total, int groups, int remap, const G2* coef, const int* nbr, int k,
                                                const float* x, long ldx, float* out, long ldo, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long t0 = xcd_block(remap) * TPB;
    if (t0 >= total) return;
    const long tl = min(t0 + (long)TPB, total) - 1;
    const long pf = t0 / groups, pl = tl / groups;
    const int nent = (int)(pl - pf + 1) * k;
    int* ids = (int*)smem;
    G2* cf = (G2*)(smem + ((size_t)nent * 4 + 15) / 16 * 16);
    for (int q = threadIdx.x; q < nent; q += TPB) { ids[q] = nbr[pf * k + q]; cf[q] = coef[pf * k + q]; }
    __syncthreads();
    const long t = t0 + threadIdx.x;
    if (t >= total) return;
    const long i = t / groups;
    const int c0 = (int)(t - i * groups) * 4;
    const int off = (int)(i - pf) * k;
    F4 au = {0, 0, 0, 0}, av = {0, 0, 0, 0};
#pragma unroll U
    for (int s = 0; s < k; ++s) {
        const G2 g = cf[off + s];
        const F4 xv = *(const F4*)(x + (long)ids[off + s] * ldx + c0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { au.v[q] = fmaf(g.a, xv.v[q], au.v[q]); av.v[q] = fmaf(g.b, xv.v[q], av.v[q]); }
    }
    if (mode == 1) {  // gather-only: defeat the stores (keep a data dependence)
        if (au.v[0] == 12345.f) *(F4*)(out + (2 * i) * ldo + c0) = au;
        return;
    }
    *(F4*)(out + (2 * i) * ldo + c0) = au;
    *(F4*)(out + (2 * i + 1) * ldo + c0) = av;
}

// ---- B: LDS slab: block = (cloud, 8-channel slab); whole cloud slab in LDS; thread = point ----------
// sliced operator layout: nbrS/coefS[(tile*k + s)*64 + lane], tile = 64 consecutive points
template <int S>  // channels per slab (multiple of 4)
__global__ __launch_bounds__(1024) void k_slab(const unsigned short* nbrS, const G2* coefS, int N, int k, const float* x,
                                               long ldx, float* out, long ldo, int slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    F4* lds = (F4*)smem;  // [N][S/4]
    constexpr int Q = S / 4;
    // blockIdx.x -> (cloud, slab) with all slabs of a cloud adjacent on one XCD
    const long b = xcd_block(1);
    const int cloud = (int)(b / slabs), slab = (int)(b % slabs);
    const int c0 = slab * S;
    const float* xc = x + (long)cloud * N * ldx + c0;
    for (int idx = threadIdx.x; idx < N * Q; idx += blockDim.x) {
        const int r = idx / Q, q = idx % Q;
        lds[idx] = *(const F4*)(xc + (long)r * ldx + q * 4);
    }
    __syncthreads();
    const int tiles = N / 64;
    for (int p = threadIdx.x; p < N; p += blockDim.x) {
        const int tile = p >> 6, lane = p & 63;
        const long base = ((long)(cloud * tiles + tile) * k) * 64 + lane;
        F4 au[Q], av[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { au[q] = F4{0, 0, 0, 0}; av[q] = F4{0, 0, 0, 0}; }
#pragma unroll 4
        for (int s = 0; s < k; ++s) {
            const int j = nbrS[base + (long)s * 64];
            const G2 g = coefS[base + (long)s * 64];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const F4 xv = lds[j * Q + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) { au[q].v[e] = fmaf(g.a, xv.v[e], au[q].v[e]); av[q].v[e] = fmaf(g.b, xv.v[e], av[q].v[e]); }
            }
        }
        float* o = out + (2 * ((long)cloud * N + p)) * ldo + c0;
#pragma unroll
        for (int q = 0; q < Q; ++q) { *(F4*)(o + q * 4) = au[q]; *(F4*)(o + ldo + q * 4) = av[q]; }
    }
}


// ---- D: block-local dedup: tile = 16 points x 64 channels; the UNIQUE neighbour rows of the tile are
// staged into LDS once (coalesced 256-B rows through the TA), all k*16 gathers then read LDS rows
// (a 16-lane group reads one aligned 256-B row = all 64 banks once: conflict-free ds_read_b128).
__global__ __launch_bounds__(256) void k_dedup(const int* tile_ptr, const int* uniq, const unsigned short* loc,
                                               const G2* coef, int k, const float* x, long ldx, float* out, long ldo,
                                               int slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    F4* rows = (F4*)smem;  // [U][16]
    const long b = xcd_block(1);
    const long tile = b / slabs;
    const int c0 = (int)(b % slabs) * 64;
    const int u0 = tile_ptr[tile], U = tile_ptr[tile + 1] - u0;
    const int lane16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    for (int r = grp; r < U; r += 16) rows[r * 16 + lane16] = *(const F4*)(x + (long)uniq[u0 + r] * ldx + c0 + lane16 * 4);
    __syncthreads();
    const long i = tile * 16 + grp;
    const unsigned short* lp = loc + i * k;
    const G2* cp = coef + i * k;
    F4 au = {0, 0, 0, 0}, av = {0, 0, 0, 0};
#pragma unroll 4
    for (int s = 0; s < k; ++s) {
        const G2 g = cp[s];
        const F4 xv = rows[(int)lp[s] * 16 + lane16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { au.v[q] = fmaf(g.a, xv.v[q], au.v[q]); av.v[q] = fmaf(g.b, xv.v[q], av.v[q]); }
    }
    *(F4*)(out + (2 * i) * ldo + c0 + lane16 * 4) = au;
    *(F4*)(out + (2 * i + 1) * ldo + c0 + lane16 * 4) = av;
}

// ---- C: plain streaming copy of the same byte volume (in + out) for reference ------------------
__global__ void k_copy(const F4* in, F4* out, long n_in, long n_out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
    F4 acc = {0, 0, 0, 0};
    for (long i = t; i < n_in; i += st) { const F4 v = in[i]; acc.v[0] += v.v[0]; }
    for (long i = t; i < n_out; i += st) out[i] = acc;
}

template <class F>
float timeit(F f, int iters = 50) {
    for (int i = 0; i < 5; ++i) f();
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const int B = 32, N = 1024, k = 20;
    const long Nt = (long)B * N, E = Nt * k;
    std::mt19937 rng(1);
    // kNN-like graph: neighbours = random points of the same cloud (worst case locality) or a window
    for (int local = 0; local < 2; ++local) {
        std::vector<int> nbr(E);
        std::vector<G2> coef(E);
        for (long i = 0; i < Nt; ++i) {
            const long cb = (i / N) * N;
            for (int s = 0; s < k; ++s) {
                int j = local ? (int)((i - cb + s * 3 + (rng() % 5)) % N) : (int)(rng() % N);
                nbr[i * k + s] = (int)(cb + j);
                coef[i * k + s] = G2{(float)(rng() % 100) * 0.01f, (float)(rng() % 100) * 0.01f};
            }
        }
        // sliced layout (cloud-local ids)
        std::vector<unsigned short> nbrS(E);
        std::vector<G2> coefS(E);
        for (long i = 0; i < Nt; ++i) {
            const long tile = i / 64, lane = i % 64, cb = (i / N) * N;
            for (int s = 0; s < k; ++s) {
                nbrS[(tile * k + s) * 64 + lane] = (unsigned short)(nbr[i * k + s] - cb);
                coefS[(tile * k + s) * 64 + lane] = coef[i * k + s];
            }
        }
        int *d_nbr; G2 *d_coef, *d_coefS; unsigned short* d_nbrS;
        CK(hipMalloc(&d_nbr, E * 4)); CK(hipMalloc(&d_coef, E * 8)); CK(hipMalloc(&d_coefS, E * 8)); CK(hipMalloc(&d_nbrS, E * 2));
        CK(hipMemcpy(d_nbr, nbr.data(), E * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_coef, coef.data(), E * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_coefS, coefS.data(), E * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_nbrS, nbrS.data(), E * 2, hipMemcpyHostToDevice));
        for (int C : {64, 128}) {
            float *d_x, *d_out, *d_out2;
            std::vector<float> hx(Nt * C);
            for (auto& v : hx) v = (float)(rng() % 1000) * 1e-3f;
            CK(hipMalloc(&d_x, Nt * C * 4)); CK(hipMalloc(&d_out, 2 * Nt * C * 4)); CK(hipMalloc(&d_out2, 2 * Nt * C * 4));
            CK(hipMemcpy(d_x, hx.data(), Nt * C * 4, hipMemcpyHostToDevice));
            const int groups = C / 4;
            const long total = Nt * groups;
            const double mb = (12.0 * C * Nt + 12.0 * E) / 1e6;
            printf("== graph=%s C=%d  algorithmic %.1f MB\n", local ? "windowed" : "random", C, mb);
            auto lds_bytes = [&](int tpb) { return (size_t)((tpb + groups - 1) / groups + 1) * k * 12 + 16; };
#define RUN_STAGED(U, TPB, REMAP, MODE, LABEL)                                                                         \
    {                                                                                                                  \
        float us = timeit([&] {                                                                                        \
            hipLaunchKernelGGL((k_staged<U, TPB>), dim3((total + TPB - 1) / TPB), dim3(TPB), lds_bytes(TPB), 0, total, groups, \
                               REMAP, d_coef, d_nbr, k, d_x, (long)C, d_out, (long)C, MODE);                           \
        });                                                                                                            \
        printf("  %-44s %8.2f us  %7.1f GB/s\n", LABEL, us, mb / us * 1e3);                                            \
    }
            RUN_STAGED(4, 256, 1, 0, "staged U4 tpb256 remap")
            RUN_STAGED(10, 256, 1, 0, "staged U10 tpb256 remap")
            RUN_STAGED(20, 256, 1, 0, "staged U20 tpb256 remap")
            RUN_STAGED(4, 128, 1, 0, "staged U4 tpb128 remap")
            RUN_STAGED(4, 512, 1, 0, "staged U4 tpb512 remap")
            RUN_STAGED(4, 256, 0, 0, "staged U4 tpb256 no-remap")
            RUN_STAGED(4, 256, 1, 1, "staged U4 tpb256 remap GATHER-ONLY (no stores)")
            RUN_STAGED(20, 256, 1, 1, "staged U20 tpb256 remap GATHER-ONLY")
            {
                constexpr int S = 8;
                const int slabs = C / S;
                CK(hipFuncSetAttribute((const void*)k_slab<S>, hipFuncAttributeMaxDynamicSharedMemorySize, N * S * 4));
                float us = timeit([&] {
                    hipLaunchKernelGGL((k_slab<S>), dim3(B * slabs), dim3(1024), N * S * 4, 0, d_nbrS, d_coefS, N, k, d_x, (long)C,
                                       d_out2, (long)C, slabs);
                });
                printf("  %-44s %8.2f us  %7.1f GB/s\n", "LDS slab S=8 (1024 thr, thread=point)", us, mb / us * 1e3);
            }
            {
                constexpr int S = 16;
                const int slabs = C / S;
                CK(hipFuncSetAttribute((const void*)k_slab<S>, hipFuncAttributeMaxDynamicSharedMemorySize, N * S * 4));
                float us = timeit([&] {
                    hipLaunchKernelGGL((k_slab<S>), dim3(B * slabs), dim3(1024), N * S * 4, 0, d_nbrS, d_coefS, N, k, d_x, (long)C,
                                       d_out2, (long)C, slabs);
                });
                printf("  %-44s %8.2f us  %7.1f GB/s\n", "LDS slab S=16", us, mb / us * 1e3);
            }
            {
                constexpr int S = 4;
                const int slabs = C / S;
                float us = timeit([&] {
                    hipLaunchKernelGGL((k_slab<S>), dim3(B * slabs), dim3(1024), N * S * 4, 0, d_nbrS, d_coefS, N, k, d_x, (long)C,
                                       d_out2, (long)C, slabs);
                });
                printf("  %-44s %8.2f us  %7.1f GB/s\n", "LDS slab S=4", us, mb / us * 1e3);
            }
            {   // correctness of the slab variant vs the staged one
                std::vector<float> a(2 * Nt * C), b2(2 * Nt * C);
                CK(hipMemcpy(a.data(), d_out, a.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b2.data(), d_out2, b2.size() * 4, hipMemcpyDeviceToHost));
                double md = 0;
                for (size_t i = 0; i < a.size(); ++i) md = std::max(md, (double)fabsf(a[i] - b2[i]));
                printf("  slab-vs-staged max abs diff %.3g\n", md);
            }

            RUN_STAGED(4, 256, 1, 0, "staged U4 tpb256 remap (again)")
            {   // block-dedup prototype (tile = 16 points)
                const long tiles = Nt / 16;
                std::vector<int> tptr(tiles + 1, 0), uq; std::vector<unsigned short> loc(E);
                int umax = 0; double uavg = 0;
                for (long t = 0; t < tiles; ++t) {
                    std::vector<int> ids;
                    for (long i = t * 16; i < t * 16 + 16; ++i) for (int s2 = 0; s2 < k; ++s2) ids.push_back(nbr[i * k + s2]);
                    std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
                    for (long i = t * 16; i < t * 16 + 16; ++i) for (int s2 = 0; s2 < k; ++s2)
                        loc[i * k + s2] = (unsigned short)(std::lower_bound(ids.begin(), ids.end(), nbr[i * k + s2]) - ids.begin());
                    tptr[t + 1] = tptr[t] + (int)ids.size(); umax = std::max(umax, (int)ids.size()); uavg += ids.size();
                    uq.insert(uq.end(), ids.begin(), ids.end());
                }
                int *d_tp, *d_uq; unsigned short* d_loc;
                CK(hipMalloc(&d_tp, tptr.size() * 4)); CK(hipMalloc(&d_uq, uq.size() * 4)); CK(hipMalloc(&d_loc, E * 2));
                CK(hipMemcpy(d_tp, tptr.data(), tptr.size() * 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(d_uq, uq.data(), uq.size() * 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(d_loc, loc.data(), E * 2, hipMemcpyHostToDevice));
                const int slabs = C / 64;
                const size_t ldsb = (size_t)umax * 256;
                CK(hipFuncSetAttribute((const void*)k_dedup, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
                float us = timeit([&] {
                    hipLaunchKernelGGL(k_dedup, dim3(tiles * slabs), dim3(256), ldsb, 0, d_tp, d_uq, d_loc, d_coef, k, d_x, (long)C,
                                       d_out2, (long)C, slabs);
                });
                char lab[128]; snprintf(lab, sizeof lab, "block-dedup LDS (U avg %.0f max %d, %zu KB)", uavg / tiles, umax, ldsb / 1024);
                printf("  %-44s %8.2f us  %7.1f GB/s\n", lab, us, mb / us * 1e3);
                std::vector<float> a(2 * Nt * C), b2(2 * Nt * C);
                CK(hipMemcpy(a.data(), d_out, a.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b2.data(), d_out2, b2.size() * 4, hipMemcpyDeviceToHost));
                double md = 0; for (size_t q = 0; q < a.size(); ++q) md = std::max(md, (double)fabsf(a[q] - b2[q]));
                printf("  dedup-vs-staged max abs diff %.3g\n", md);
                CK(hipFree(d_tp)); CK(hipFree(d_uq)); CK(hipFree(d_loc));
            }
            {
                float us = timeit([&] {
                    hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const F4*)d_x, (F4*)d_out, Nt * C / 4, 2 * Nt * C / 4);
                });
                printf("  %-44s %8.2f us  %7.1f GB/s\n", "stream copy (read x, write out)", us, 12.0 * C * Nt / 1e6 / us * 1e3);
            }
            CK(hipFree(d_x)); CK(hipFree(d_out)); CK(hipFree(d_out2));
        }
        CK(hipFree(d_nbr)); CK(hipFree(d_coef)); CK(hipFree(d_coefS)); CK(hipFree(d_nbrS));
    }
    return 0;
}
