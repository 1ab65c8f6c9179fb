// Gather lab 2: does block-local de-duplication of neighbour rows through LDS beat the L1/TA gather path of the
// production ELL applies, on a REAL kNN graph over spatially ordered points?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_lab2.hip -o /tmp/gather_lab2 && /tmp/gather_lab2
// Workload = the graded kernel (dc_apply_div_curl_norm: v[2Nt,C] -> [div|curl|norm] [Nt,3C] inside a 4C-wide
// buffer) and the plain grad apply, B clouds x 1024 points, k = 20, C = 64 / 128, on points of the bench's synthetic
// surface.  Variants:
//   staged   : production structure (ids/coefficients staged in LDS, 16-byte gathers through L1), points in
//              generation (random) order and in Morton order;
//   dedup P  : a block = P consecutive (Morton-ordered) points x all channels; the UNIQUE neighbour rows of the
//              tile are loaded once into LDS (whole rows, coalesced), ids (as tile-local uint16) and coefficients
//              are staged too, so the inner loop touches LDS only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct alignas(16) F4 { float v[4]; };
struct G2 { float a, b; };

__device__ __forceinline__ long xcd_block() {
    const long b = blockIdx.x, nb = gridDim.x;
    const long q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

// ---- production structure --------------------------------------------------------------------------------------
template <int MODE>  // 0: grad (x[Nt,C] -> out[2Nt,C]); 1: divcurlnorm (v[2Nt,C] -> out[Nt, 3C] with ldo)
__global__ __launch_bounds__(256) void k_staged(long total, int groups, const G2* coef, const int* nbr, int k,
                                                const float* x, long ldx, float* out, long ldo, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long t0 = xcd_block() * 256;
    if (t0 >= total) return;
    const long tl = min(t0 + 256L, total) - 1;
    const long pf = t0 / groups, pl = tl / groups;
    const int nent = (int)(pl - pf + 1) * k;
    int* ids = (int*)smem;
    G2* cf = (G2*)(smem + ((size_t)nent * 4 + 15) / 16 * 16);
    for (int q = threadIdx.x; q < nent; q += 256) { ids[q] = nbr[pf * k + q]; cf[q] = coef[pf * k + q]; }
    __syncthreads();
    const long t = t0 + threadIdx.x;
    if (t >= total) return;
    const long i = t / groups;
    const int c0 = (int)(t - i * groups) * 4;
    const int off = (int)(i - pf) * k;
    F4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    if (MODE == 0) {
#pragma unroll 4
        for (int s = 0; s < k; ++s) {
            const G2 g = cf[off + s];
            const F4 xv = *(const F4*)(x + (long)ids[off + s] * ldx + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { a0.v[q] = fmaf(g.a, xv.v[q], a0.v[q]); a1.v[q] = fmaf(g.b, xv.v[q], a1.v[q]); }
        }
        *(F4*)(out + (2 * i) * ldo + c0) = a0;
        *(F4*)(out + (2 * i + 1) * ldo + c0) = a1;
    } else {
#pragma unroll 4
        for (int s = 0; s < k; ++s) {
            const G2 d = cf[off + s];
            const long j = ids[off + s];
            const F4 vu = *(const F4*)(x + (2 * j) * ldx + c0), vv = *(const F4*)(x + (2 * j + 1) * ldx + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a0.v[q] = fmaf(d.a, vu.v[q], a0.v[q]); a0.v[q] = fmaf(d.b, vv.v[q], a0.v[q]);
                a1.v[q] = fmaf(d.a, vv.v[q], a1.v[q]); a1.v[q] = fmaf(-d.b, vu.v[q], a1.v[q]);
            }
        }
        const F4 ou = *(const F4*)(x + (2 * i) * ldx + c0), ov = *(const F4*)(x + (2 * i + 1) * ldx + c0);
        F4 nv;
#pragma unroll
        for (int q = 0; q < 4; ++q) nv.v[q] = sqrtf(fmaf(ou.v[q], ou.v[q], ov.v[q] * ov.v[q]));
        *(F4*)(out + i * ldo + c0) = a0;
        *(F4*)(out + i * ldo + C + c0) = a1;
        *(F4*)(out + i * ldo + 2 * C + c0) = nv;
    }
}

// ---- tile-local de-duplication -----------------------------------------------------------------------------------
// block = tile of P points x CS channels (CS/4 lanes per point); blockDim = P * CS / 4.
// LDS: rows [U][R][CS] (R = 1 grad, 2 divcurlnorm: u and v rows), loc [P*k] u16, coef [P*k].
template <int MODE, int CS>
__global__ void k_dedup(const int* __restrict__ tile_ptr, const int* __restrict__ uniq, const unsigned short* __restrict__ loc,
                        const G2* __restrict__ coef, int P, int k, long Nt, const float* __restrict__ x, long ldx,
                        float* __restrict__ out, long ldo, int C, int slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = MODE == 0 ? 1 : 2;
    constexpr int L = CS / 4;            // lanes per row
    const long b = xcd_block();
    const long tile = b / slabs;
    const int cb = (int)(b - tile * slabs) * CS;
    const int u0 = tile_ptr[tile], U = tile_ptr[tile + 1] - u0;
    F4* rows = (F4*)smem;                                   // [U][R][L]
    G2* cf = (G2*)(smem + (size_t)U * R * CS * 4);          // [P*k]
    unsigned short* lc = (unsigned short*)(cf + P * k);     // [P*k]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const long p0 = tile * P;
    const int npts = (int)min((long)P, Nt - p0);
    for (int q = tid; q < npts * k; q += nthr) { cf[q] = coef[p0 * k + q]; lc[q] = loc[p0 * k + q]; }
    const int lane = tid % L, grp = tid / L, ngrp = nthr / L;
    for (int r = grp; r < U * R; r += ngrp) {
        const int u = r / R, h = r - u * R;
        const long row = R == 1 ? (long)uniq[u0 + u] : 2L * uniq[u0 + u] + h;
        rows[r * L + lane] = *(const F4*)(x + row * ldx + cb + lane * 4);
    }
    __syncthreads();
    if (grp >= npts) return;
    const long i = p0 + grp;
    const G2* cp = cf + grp * k;
    const unsigned short* lp = lc + grp * k;
    F4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    if (MODE == 0) {
#pragma unroll 4
        for (int s = 0; s < k; ++s) {
            const G2 g = cp[s];
            const F4 xv = rows[(int)lp[s] * L + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a0.v[q] = fmaf(g.a, xv.v[q], a0.v[q]); a1.v[q] = fmaf(g.b, xv.v[q], a1.v[q]); }
        }
        *(F4*)(out + (2 * i) * ldo + cb + lane * 4) = a0;
        *(F4*)(out + (2 * i + 1) * ldo + cb + lane * 4) = a1;
    } else {
#pragma unroll 4
        for (int s = 0; s < k; ++s) {
            const G2 d = cp[s];
            const int l = lp[s];
            const F4 vu = rows[(l * 2) * L + lane], vv = rows[(l * 2 + 1) * L + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a0.v[q] = fmaf(d.a, vu.v[q], a0.v[q]); a0.v[q] = fmaf(d.b, vv.v[q], a0.v[q]);
                a1.v[q] = fmaf(d.a, vv.v[q], a1.v[q]); a1.v[q] = fmaf(-d.b, vu.v[q], a1.v[q]);
            }
        }
        // own rows: slot 0 of a kNN list is the point itself
        const int l0 = lp[0];
        const F4 ou = rows[(l0 * 2) * L + lane], ov = rows[(l0 * 2 + 1) * L + lane];
        F4 nv;
#pragma unroll
        for (int q = 0; q < 4; ++q) nv.v[q] = sqrtf(fmaf(ou.v[q], ou.v[q], ov.v[q] * ov.v[q]));
        *(F4*)(out + i * ldo + cb + lane * 4) = a0;
        *(F4*)(out + i * ldo + C + cb + lane * 4) = a1;
        *(F4*)(out + i * ldo + 2 * C + cb + lane * 4) = nv;
    }
}

__global__ void k_copy(const F4* in, F4* out, long n_in, long n_out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
    F4 acc = {0, 0, 0, 0};
    for (long i = t; i < n_in; i += st) { const F4 v = in[i]; acc.v[0] += v.v[0]; }
    for (long i = t; i < n_out; i += st) out[i] = acc;
}

template <class F>
float timeit(F f, int iters = 40) {
    for (int i = 0; i < 4; ++i) f();
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

static uint32_t part1by2(uint32_t x) {
    x &= 0x3ff; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249;
    return x;
}

struct Graph { std::vector<int> nbr; std::vector<G2> coef; };

// kNN (incl. self, ascending distance) per cloud by brute force on the host
static Graph build_graph(const std::vector<float>& pos, int B, int N, int k, std::mt19937& rng) {
    Graph g; g.nbr.resize((size_t)B * N * k); g.coef.resize((size_t)B * N * k);
    std::vector<std::pair<float, int>> d(N);
    for (int b = 0; b < B; ++b) {
        const float* p = pos.data() + (size_t)b * N * 3;
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j) {
                const float dx = p[3 * i] - p[3 * j], dy = p[3 * i + 1] - p[3 * j + 1], dz = p[3 * i + 2] - p[3 * j + 2];
                d[j] = {dx * dx + dy * dy + dz * dz, j};
            }
            std::partial_sort(d.begin(), d.begin() + k, d.end());
            for (int s = 0; s < k; ++s) {
                g.nbr[((size_t)b * N + i) * k + s] = b * N + d[s].second;
                g.coef[((size_t)b * N + i) * k + s] = G2{(float)(rng() % 200) * 0.01f - 1.f, (float)(rng() % 200) * 0.01f - 1.f};
            }
        }
    }
    return g;
}

struct Dedup { std::vector<int> tptr, uq; std::vector<unsigned short> loc; int umax; double uavg; };
static Dedup build_dedup(const Graph& g, long Nt, int k, int P) {
    Dedup d; const long tiles = (Nt + P - 1) / P;
    d.tptr.assign(tiles + 1, 0); d.loc.resize((size_t)Nt * k); d.umax = 0; d.uavg = 0;
    for (long t = 0; t < tiles; ++t) {
        std::vector<int> ids;
        const long e = std::min(Nt, (t + 1) * P);
        for (long i = t * P; i < e; ++i) for (int s = 0; s < k; ++s) ids.push_back(g.nbr[i * k + s]);
        std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        for (long i = t * P; i < e; ++i) for (int s = 0; s < k; ++s)
            d.loc[i * k + s] = (unsigned short)(std::lower_bound(ids.begin(), ids.end(), g.nbr[i * k + s]) - ids.begin());
        d.tptr[t + 1] = d.tptr[t] + (int)ids.size(); d.umax = std::max(d.umax, (int)ids.size()); d.uavg += ids.size();
        d.uq.insert(d.uq.end(), ids.begin(), ids.end());
    }
    d.uavg /= tiles;
    return d;
}

template <class T> T* upload(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

template <int MODE, int CS>
float run_dedup(const Dedup& dd, int* d_tp, int* d_uq, unsigned short* d_loc, G2* d_coef, int P, int k, long Nt, const float* d_x,
                long ldx, float* d_out, long ldo, int C) {
    constexpr int R = MODE == 0 ? 1 : 2;
    const int slabs = C / CS;
    const long tiles = (Nt + P - 1) / P;
    const size_t lds = (size_t)dd.umax * R * CS * 4 + (size_t)P * k * 10 + 16;
    CK(hipFuncSetAttribute((const void*)k_dedup<MODE, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return timeit([&] {
        hipLaunchKernelGGL((k_dedup<MODE, CS>), dim3(tiles * slabs), dim3(P * CS / 4), lds, 0, d_tp, d_uq, d_loc, d_coef, P, k, Nt, d_x, ldx,
                           d_out, ldo, C, slabs);
    });
}

int main(int argc, char** argv) {
    const int N = 1024, k = 20;
    std::mt19937 rng(1);
    for (int B : {32, 512}) {
        const long Nt = (long)B * N, E = Nt * k;
        // points on r = 1 + 0.25 sin(3 theta) cos(2 phi) (the bench's synthetic surface); clouds of a big batch repeat
        const int Bu = std::min(B, 32);
        std::vector<float> pos((size_t)Bu * N * 3);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (size_t i = 0; i < (size_t)Bu * N; ++i) {
            float x = nd(rng), y = nd(rng), z = nd(rng); const float n = sqrtf(x * x + y * y + z * z); x /= n; y /= n; z /= n;
            const float th = acosf(z), ph = atan2f(y, x), r = 1.f + 0.25f * sinf(3 * th) * cosf(2 * ph);
            pos[3 * i] = r * x; pos[3 * i + 1] = r * y; pos[3 * i + 2] = r * z;
        }
        // Morton order per cloud
        std::vector<float> pos_m(pos.size());
        for (int b = 0; b < Bu; ++b) {
            std::vector<std::pair<uint32_t, int>> key(N);
            for (int i = 0; i < N; ++i) {
                const float* p = &pos[((size_t)b * N + i) * 3];
                auto qz = [](float v) { return (uint32_t)std::min(1023.f, std::max(0.f, (v + 1.3f) / 2.6f * 1024.f)); };
                key[i] = {part1by2(qz(p[0])) | (part1by2(qz(p[1])) << 1) | (part1by2(qz(p[2])) << 2), i};
            }
            std::sort(key.begin(), key.end());
            for (int i = 0; i < N; ++i) for (int c = 0; c < 3; ++c) pos_m[((size_t)b * N + i) * 3 + c] = pos[((size_t)b * N + key[i].second) * 3 + c];
        }
        Graph gr = build_graph(pos, Bu, N, k, rng), gm = build_graph(pos_m, Bu, N, k, rng);
        auto tile_up = [&](Graph& g) {   // replicate the first Bu clouds up to B
            g.nbr.resize(E); g.coef.resize(E);
            for (long e = (long)Bu * N * k; e < E; ++e) {
                const long src = e % ((long)Bu * N * k);
                g.nbr[e] = g.nbr[src] + (int)((e / ((long)Bu * N * k)) * Bu * N);
                g.coef[e] = g.coef[src];
            }
        };
        tile_up(gr); tile_up(gm);
        int* d_nbr_r = upload(gr.nbr); G2* d_coef_r = upload(gr.coef);
        int* d_nbr_m = upload(gm.nbr); G2* d_coef_m = upload(gm.coef);
        for (int C : {64, 128}) {
            if (B == 512 && C == 128) continue;
            std::vector<float> hx((size_t)2 * Nt * C);
            for (auto& v : hx) v = (float)(rng() % 2000) * 1e-3f - 1.f;
            float* d_x = upload(hx);
            float *d_out, *d_out2;
            CK(hipMalloc(&d_out, (size_t)Nt * 4 * C * 4)); CK(hipMalloc(&d_out2, (size_t)Nt * 4 * C * 4));
            CK(hipMemset(d_out, 0, (size_t)Nt * 4 * C * 4)); CK(hipMemset(d_out2, 0, (size_t)Nt * 4 * C * 4));
            const int groups = C / 4;
            const long total = Nt * groups;
            const size_t lds_st = (size_t)((256 + groups - 1) / groups + 1) * k * 12 + 16;
            for (int mode = 0; mode < 2; ++mode) {
                const double mb = mode == 0 ? (12.0 * C * Nt + 12.0 * E) / 1e6 : (20.0 * C * Nt + 12.0 * E) / 1e6;
                const long ldo = mode == 0 ? C : 4 * C;
                float* o1 = mode == 0 ? d_out : d_out + C;
                float* o2 = mode == 0 ? d_out2 : d_out2 + C;
                printf("== B=%d C=%d %s  algorithmic %.1f MB\n", B, C, mode ? "divcurlnorm" : "grad", mb);
                auto staged = [&](int* nb, G2* cf, float* o) {
                    return timeit([&] {
                        if (mode == 0) hipLaunchKernelGGL((k_staged<0>), dim3((total + 255) / 256), dim3(256), lds_st, 0, total, groups, cf, nb, k, d_x, (long)C, o, ldo, C);
                        else hipLaunchKernelGGL((k_staged<1>), dim3((total + 255) / 256), dim3(256), lds_st, 0, total, groups, cf, nb, k, d_x, (long)C, o, ldo, C);
                    });
                };
                float us = staged(d_nbr_r, d_coef_r, o1);
                printf("  %-52s %8.2f us  %7.1f GB/s  frac %.3f\n", "staged, random point order", us, mb / us * 1e3, mb / us * 1e3 / 8000);
                us = staged(d_nbr_m, d_coef_m, o1);
                printf("  %-52s %8.2f us  %7.1f GB/s  frac %.3f\n", "staged, Morton order", us, mb / us * 1e3, mb / us * 1e3 / 8000);
                for (int P : {16, 32, 64}) {
                    Dedup dd = build_dedup(gm, Nt, k, P);
                    int* d_tp = upload(dd.tptr); int* d_uq = upload(dd.uq); unsigned short* d_loc = upload(dd.loc);
                    for (int CS : {64, 32}) {
                        if (CS > C || P * CS / 4 > 1024 || P * CS / 4 < 64) continue;
                        float t;
                        if (mode == 0) t = CS == 64 ? run_dedup<0, 64>(dd, d_tp, d_uq, d_loc, d_coef_m, P, k, Nt, d_x, C, o2, ldo, C)
                                                    : run_dedup<0, 32>(dd, d_tp, d_uq, d_loc, d_coef_m, P, k, Nt, d_x, C, o2, ldo, C);
                        else t = CS == 64 ? run_dedup<1, 64>(dd, d_tp, d_uq, d_loc, d_coef_m, P, k, Nt, d_x, C, o2, ldo, C)
                                          : run_dedup<1, 32>(dd, d_tp, d_uq, d_loc, d_coef_m, P, k, Nt, d_x, C, o2, ldo, C);
                        char lab[128];
                        snprintf(lab, sizeof lab, "dedup P=%d CS=%d (U avg %.0f max %d of %d)", P, CS, dd.uavg, dd.umax, P * k);
                        printf("  %-52s %8.2f us  %7.1f GB/s  frac %.3f\n", lab, t, mb / t * 1e3, mb / t * 1e3 / 8000);
                    }
                    if (P == 16 && B == 32) {   // correctness vs staged on the same (Morton) graph
                        const size_t nout = (size_t)Nt * 4 * C;
                        std::vector<float> a(nout), b2(nout);
                        staged(d_nbr_m, d_coef_m, o1);
                        CK(hipMemcpy(a.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(b2.data(), d_out2, nout * 4, hipMemcpyDeviceToHost));
                        double md = 0; for (size_t q = 0; q < (mode == 0 ? (size_t)2 * Nt * C : nout); ++q) md = std::max(md, (double)fabsf(a[q] - b2[q]));
                        printf("  dedup-vs-staged max abs diff %.3g\n", md);
                    }
                    CK(hipFree(d_tp)); CK(hipFree(d_uq)); CK(hipFree(d_loc));
                }
                const long n_in = mode == 0 ? Nt * C / 4 : 2 * Nt * C / 4, n_out = mode == 0 ? 2 * Nt * C / 4 : 3 * Nt * C / 4;
                us = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const F4*)d_x, (F4*)d_out, n_in, n_out); });
                printf("  %-52s %8.2f us  %7.1f GB/s\n", "stream copy of the same in + out bytes", us, (n_in + n_out) * 16.0 / 1e6 / us * 1e3);
            }
            CK(hipFree(d_x)); CK(hipFree(d_out)); CK(hipFree(d_out2));
        }
        CK(hipFree(d_nbr_r)); CK(hipFree(d_coef_r)); CK(hipFree(d_nbr_m)); CK(hipFree(d_coef_m));
    }
    return 0;
}
