// Tile lab (round 3): forward + transposed ELL applies from a per-batch TILE PLAN.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/tile_lab.hip -o tools/tile_lab.bin
//   tools/tile_lab.bin [B ...]           (default B = 32 512)
//
// Idea under test.  The production applies gather every neighbour row through the vector-memory path (texture
// addresser / L1): k = 20 times the compulsory read volume, and that path - not HBM - bounds them (r02n counters).
// A tile plan groups the points of a cloud into tiles of P spatially close points (Morton order of the positions,
// built once per batch like the CSC); the UNIQUE neighbour rows of a tile (~110 for P = 32 instead of 640) are
// loaded once into LDS as whole 256-byte rows, and the k-loop reads LDS only.  With 16 lanes per point and 16 bytes
// per lane, the 16 lanes of every ds_read_b128 lane group carry 16 different 16-byte columns of a 256-byte row, so
// random rows are bank-conflict free (MI355X_MICROARCH.md LDS table: 256 B/clk/CU vs ~49 B/clk/CU measured on the TA).
// Tensors stay in their original point order: a tile is a LIST of point ids, outputs are written as whole rows.
// Same FMAs in the same slot order as the staged kernels -> results must be bit-identical (checked here).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct alignas(16) F4 { float v[4]; };
struct G2 { float a, b; };
typedef unsigned short u16;

__device__ __forceinline__ long xcd_block() {
    const long b = blockIdx.x, nb = gridDim.x;
    const long q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

// MODE 0: grad      x[Nt,C]  -> out[2Nt,C]            (1 row piece per neighbour)
// MODE 1: dcn       v[2Nt,C] -> out[Nt, div|curl|norm] (2 row pieces per neighbour: rows 2j, 2j+1)
// piece h of neighbour j: x + j*ldj + h*hs
template <int MODE>
__device__ __forceinline__ void accum(F4& a0, F4& a1, const G2 d, const F4& p0, const F4& p1) {
    if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { a0.v[q] = fmaf(d.a, p0.v[q], a0.v[q]); a1.v[q] = fmaf(d.b, p0.v[q], a1.v[q]); }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0.v[q] = fmaf(d.a, p0.v[q], a0.v[q]); a0.v[q] = fmaf(d.b, p1.v[q], a0.v[q]);
            a1.v[q] = fmaf(d.a, p1.v[q], a1.v[q]); a1.v[q] = fmaf(-d.b, p0.v[q], a1.v[q]);
        }
    }
}
#ifndef STORE_MODE
#define STORE_MODE 0
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4(float* p, const F4& a) {
#if STORE_MODE == 1
    __builtin_nontemporal_store(*(const f4v*)&a, (f4v*)p);
#elif STORE_MODE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(*(const f4v*)&a) : "memory");
#elif STORE_MODE == 3
    asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(*(const f4v*)&a) : "memory");
#elif STORE_MODE == 4
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(*(const f4v*)&a) : "memory");
#else
    *(F4*)p = a;
#endif
}
template <int MODE>
__device__ __forceinline__ void store_out(float* out, long ldo, int C, long i, int c, const F4& a0, const F4& a1, const F4& ou, const F4& ov) {
    if (MODE == 0) {
        st4(out + (2 * i) * ldo + c, a0);
        st4(out + (2 * i + 1) * ldo + c, a1);
    } else {
        F4 nv;
#pragma unroll
        for (int q = 0; q < 4; ++q) nv.v[q] = sqrtf(fmaf(ou.v[q], ou.v[q], ov.v[q] * ov.v[q]));
        st4(out + i * ldo + c, a0);
        st4(out + i * ldo + C + c, a1);
        st4(out + i * ldo + 2 * C + c, nv);
    }
}

// ---- production structure (ids / coefficients staged, rows gathered through L1) --------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_staged(long total, int groups, const G2* coef, const int* nbr, int k,
                                                const float* x, long ldx, float* out, long ldo, int C, unsigned long long* stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#define STAMPS(n) do { if (stamps && threadIdx.x == 0) stamps[(long)blockIdx.x * 8 + (n)] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMPS(0);
    const long t0 = xcd_block() * 256;
    if (t0 >= total) return;
    const long tl = min(t0 + 256L, total) - 1;
    const long pf = t0 / groups, pl = tl / groups;
    const int nent = (int)(pl - pf + 1) * k;
    int* ids = (int*)smem;
    G2* cf = (G2*)(smem + ((size_t)nent * 4 + 15) / 16 * 16);
    for (int q = threadIdx.x; q < nent; q += 256) { ids[q] = nbr[pf * k + q]; cf[q] = coef[pf * k + q]; }
    STAMPS(2);
    __syncthreads();
    STAMPS(3);
    const long t = t0 + threadIdx.x;
    if (t >= total) return;
    const long i = t / groups;
    const int c0 = (int)(t - i * groups) * 4;
    const int off = (int)(i - pf) * k;
    F4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll 4
    for (int s = 0; s < k; ++s) {
        const G2 g = cf[off + s];
        const long j = ids[off + s];
        if (MODE == 0) {
            const F4 xv = *(const F4*)(x + j * ldx + c0);
            accum<0>(a0, a1, g, xv, xv);
        } else {
            const F4 vu = *(const F4*)(x + (2 * j) * ldx + c0), vv = *(const F4*)(x + (2 * j + 1) * ldx + c0);
            accum<1>(a0, a1, g, vu, vv);
        }
    }
    F4 ou = {0, 0, 0, 0}, ov = {0, 0, 0, 0};
    if (MODE == 1) { ou = *(const F4*)(x + (2 * i) * ldx + c0); ov = *(const F4*)(x + (2 * i + 1) * ldx + c0); }
    if (stamps) { asm volatile("s_nop 0" ::: "memory"); STAMPS(4); }
    store_out<MODE>(out, ldo, C, i, c0, a0, a1, ou, ov);
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMPS(5); }
}

// ---- tile plan, forward (v2: everything a tile needs arrives by LDS-DMA) ------------------------------------------
// plan (fixed strides per tile t): tile_pts[t][P] point ids (-1 = padding), nu[t] unique count, uniq[t][P*K] unique
// ids (ascending; the tile's own points are always members), loc[t][P*K] u16 tile-local index of neighbour (p, s),
// selfloc[t][P] u16 tile-local index of the point itself, coefP[t][P*K] coefficients in tile order (per operator,
// permuted once per batch).  Block = one tile x one CS-channel slab (CS = 64: 16 lanes per point and conflict-free
// ds_read_b128 on random rows; CS = 32 / 16: smaller LDS footprint -> more workgroups per CU, some bank conflicts).
// LDS: rows [CAPR pieces][CS] floats, cf [PKpad] G2, lc [PKpad] u16 (both padded to whole 1-KiB DMA chunks).
template <int CS, int NT> struct TileGeom {
    static constexpr int L = CS / 4;            // lanes per row piece
    static constexpr int PPI = 64 / L;          // pieces per wave-level DMA instruction
    static constexpr int NW = NT / 64;
    static constexpr int NG = NT / L;           // points per compute pass
};
__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int MODE, int P, int NT, int CAP, int CS>
__global__ __launch_bounds__(NT) void k_tile(const int* __restrict__ tile_pts, const int* __restrict__ nu,
                                             const int* __restrict__ uniq, const u16* __restrict__ loc, const u16* __restrict__ selfloc,
                                             const G2* __restrict__ coefP, int k_rt, const float* __restrict__ x, long ldx,
                                             float* __restrict__ out, long ldo, int C, int slabs, unsigned long long* stamps = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TG = TileGeom<CS, NT>;
    constexpr int K = 20;
#define STAMP(n) do { if (stamps && threadIdx.x == 0) stamps[(long)blockIdx.x * 8 + (n)] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);
    constexpr int R = MODE == 0 ? 1 : 2;
    constexpr int L = TG::L, PPI = TG::PPI, NW = TG::NW, NG = TG::NG;
    constexpr int CAPR = (CAP * R + PPI * NW - 1) / (PPI * NW) * (PPI * NW);
    constexpr int RIT = CAPR / (PPI * NW);
    constexpr int PK = P * K;
    constexpr int CFB = (PK * 8 + 1023) / 1024 * 1024, LCB = (PK * 2 + 1023) / 1024 * 1024;
    const long b = xcd_block();
    const long tile = b / slabs;
    const int cb = (int)(b - tile * slabs) * CS;
    const int tid = threadIdx.x, ll = tid % L, grp = tid / L;
    const int wave = tid >> 6, lane64 = tid & 63;
    float* rows = (float*)smem;                            // [CAPR][CS]
    char* cfb = smem + (size_t)CAPR * CS * 4;              // [PK] G2
    char* lcb = cfb + CFB;                                 // [PK] u16
    int* pts = (int*)(lcb + LCB);                          // [P]
    u16* sl = (u16*)(pts + P);                             // [P]
    const int* uq = uniq + tile * PK;
    // ids: one round trip (the unique count is read beside them, not before them; slots beyond it hold the last id)
    int rid[RIT];
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
        const int r = min((wave + it * NW) * PPI + lane64 / L, PK * R - 1);
        rid[it] = uq[r / R] * R + (r % R);
    }
    const int U = nu[tile];
    const int nrow = U * R;
    int mypt = -1; u16 mysl = 0;
    if (tid < P) { mypt = tile_pts[tile * P + tid]; mysl = selfloc[tile * P + tid]; }
#pragma unroll
    for (int it = 0; it < RIT; ++it) asm volatile("" : "+v"(rid[it]));
    asm volatile("" : "+v"(mypt));
    STAMP(1);
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
        const int r0 = (wave + it * NW) * PPI;
        if (r0 < nrow) dma16(x + (long)rid[it] * ldx + cb + ll * 4, rows + r0 * CS);
    }
    {   // coefficients and local indices of the tile: contiguous -> whole 1-KiB chunks
        const char* gcf = (const char*)(coefP + tile * PK);
        for (int c = wave; c * 1024 < PK * 8; c += NW) dma16(gcf + min(c * 1024 + lane64 * 16, PK * 8 - 16), cfb + c * 1024);
        const char* glc = (const char*)(loc + tile * PK);
        for (int c = wave; c * 1024 < PK * 2; c += NW) dma16(glc + min(c * 1024 + lane64 * 16, PK * 2 - 16), lcb + c * 1024);
    }
    if (tid < P) { pts[tid] = mypt; sl[tid] = mysl; }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    const G2* cf = (const G2*)cfb;
    const u16* lc = (const u16*)lcb;
    for (int p = grp; p < P; p += NG) {
        const long i = pts[p];
        if (i < 0) continue;
        const G2* cp = cf + p * K;
        const u16* lp = lc + p * K;
        F4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll 4
        for (int s = 0; s < k_rt; ++s) {
            const int l = lp[s];
            const F4 p0 = *(const F4*)(rows + (l * R) * CS + ll * 4);
            const F4 p1 = R == 2 ? *(const F4*)(rows + (l * R + R - 1) * CS + ll * 4) : p0;
            accum<MODE>(a0, a1, cp[s], p0, p1);
        }
        F4 ou = {0, 0, 0, 0}, ov = {0, 0, 0, 0};
        if (MODE == 1) { const int l = sl[p]; ou = *(const F4*)(rows + (l * 2) * CS + ll * 4); ov = *(const F4*)(rows + (l * 2 + 1) * CS + ll * 4); }
        if (stamps && p == grp) { asm volatile("s_nop 0" ::: "memory"); STAMP(4); }
        store_out<MODE>(out, ldo, C, i, cb + ll * 4, a0, a1, ou, ov);
    }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(5); }
}

// ---- transposed: production structure (CSC in-edge lists staged, rows gathered through L1) -------------------------
// TMODE 0: div^T   dy[Nt,C] -> dv[2Nt,C]  : dv[2j+a] = sum_e D[e,a] dy[i]            (1 piece per in-edge)
// TMODE 1: grad^T  dy[2Nt,C] -> dx[Nt,C]  : dx[j] = sum_e G[e,0] dy[2i] + G[e,1] dy[2i+1]  (2 pieces per in-edge)
template <int TMODE>
__device__ __forceinline__ void accumT(F4& a0, F4& a1, const G2 d, const F4& p0, const F4& p1) {
    if (TMODE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { a0.v[q] = fmaf(d.a, p0.v[q], a0.v[q]); a1.v[q] = fmaf(d.b, p0.v[q], a1.v[q]); }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { a0.v[q] = fmaf(d.a, p0.v[q], a0.v[q]); a0.v[q] = fmaf(d.b, p1.v[q], a0.v[q]); }
    }
}
template <int TMODE>
__device__ __forceinline__ void storeT(float* out, long ldo, long j, int c, const F4& a0, const F4& a1) {
    if (TMODE == 0) { st4(out + (2 * j) * ldo + c, a0); st4(out + (2 * j + 1) * ldo + c, a1); }
    else st4(out + j * ldo + c, a0);
}

constexpr int T_CHUNK = 2048;
template <int TMODE>
__global__ __launch_bounds__(256) void k_stagedT(long total, int groups, const G2* coefT, const int* tptr, const int* tedge, int k,
                                                 const float* dy, long ldy, float* out, long ldo) {
    __shared__ int src[T_CHUNK];
    __shared__ G2 cf[T_CHUNK];
    const long t0 = xcd_block() * 256;
    if (t0 >= total) return;
    const long tl = min(t0 + 256L, total) - 1;
    const long pf = t0 / groups, pl = tl / groups;
    const int e_begin = tptr[pf], e_end = tptr[pl + 1];
    const long t = t0 + threadIdx.x;
    const bool active = t < total;
    const long j = active ? t / groups : pf;
    const int c0 = active ? (int)(t - j * groups) * 4 : 0;
    const int cb = active ? tptr[j] : 0, ce = active ? tptr[j + 1] : 0;
    F4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    for (int base = e_begin; base < e_end; base += T_CHUNK) {
        const int cnt = min(T_CHUNK, e_end - base);
        __syncthreads();
        for (int q = threadIdx.x; q < cnt; q += 256) { src[q] = tedge[base + q] / k; cf[q] = coefT[base + q]; }
        __syncthreads();
        const int lo = max(cb, base) - base, hi = min(ce, base + cnt) - base;
#pragma unroll 4
        for (int p = lo; p < hi; ++p) {
            const long i = src[p];
            if (TMODE == 0) { const F4 g = *(const F4*)(dy + i * ldy + c0); accumT<0>(a0, a1, cf[p], g, g); }
            else { const F4 g0 = *(const F4*)(dy + (2 * i) * ldy + c0), g1 = *(const F4*)(dy + (2 * i + 1) * ldy + c0); accumT<1>(a0, a1, cf[p], g0, g1); }
        }
    }
    if (active) storeT<TMODE>(out, ldo, j, c0, a0, a1);
}

// ---- tile plan, transposed (v2, LDS-DMA) ---------------------------------------------------------------------------
// plan: tile_pts (same tiles), tnu[t], tuniq[t][UT] unique SOURCE ids of the tile's in-edges, toff[t][P+1] in-degree
// prefix (relative), tbase[t] start of the tile's entries in the tile-major arrays (each tile padded to 8 entries:
// 16-byte aligned DMA sources), tt_loc[.] tile-local source index, coefTT[.] coefficients in tile-major order (per
// target ascending edge id = the CSC order: same sums, same order).
template <int TMODE, int P, int NT, int CAP, int ECAP, int CS>
__global__ __launch_bounds__(NT) void k_tileT(const int* __restrict__ tile_pts, const int* __restrict__ tnu,
                                              const int* __restrict__ tuniq, int UT, const int* __restrict__ tbase,
                                              const u16* __restrict__ toff, const u16* __restrict__ tt_loc,
                                              const G2* __restrict__ coefTT, const float* __restrict__ dy, long ldy,
                                              float* __restrict__ out, long ldo, int slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TG = TileGeom<CS, NT>;
    constexpr int R = TMODE == 0 ? 1 : 2;
    constexpr int L = TG::L, PPI = TG::PPI, NW = TG::NW, NG = TG::NG;
    constexpr int CAPR = (CAP * R + PPI * NW - 1) / (PPI * NW) * (PPI * NW);
    constexpr int RIT = CAPR / (PPI * NW);
    constexpr int CFB = (ECAP * 8 + 1023) / 1024 * 1024, LCB = (ECAP * 2 + 1023) / 1024 * 1024;
    const long b = xcd_block();
    const long tile = b / slabs;
    const int cb = (int)(b - tile * slabs) * CS;
    const int tid = threadIdx.x, ll = tid % L, grp = tid / L;
    const int wave = tid >> 6, lane64 = tid & 63;
    float* rows = (float*)smem;
    char* cfb = smem + (size_t)CAPR * CS * 4;
    char* lcb = cfb + CFB;
    int* pts = (int*)(lcb + LCB);                          // [P]
    u16* off = (u16*)(pts + P);                            // [P+1]
    const int U = tnu[tile];
    const int nrow = U * R;
    const int* uq = tuniq + tile * (long)UT;
    int rid[RIT];
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
        const int r = min((wave + it * NW) * PPI + lane64 / L, nrow - 1);
        rid[it] = uq[r / R] * R + (r % R);
    }
    int mypt = -1; u16 myoff = 0;
    if (tid < P) mypt = tile_pts[tile * P + tid];
    if (tid <= P) myoff = toff[tile * (P + 1) + tid];
    const int e0 = tbase[tile];
    const int ne = toff[tile * (P + 1) + P];
#pragma unroll
    for (int it = 0; it < RIT; ++it) asm volatile("" : "+v"(rid[it]));
    asm volatile("" : "+v"(mypt));
    asm volatile("" : "+v"(myoff));
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
        const int r0 = (wave + it * NW) * PPI;
        if (r0 < nrow) dma16(dy + (long)rid[it] * ldy + cb + ll * 4, rows + r0 * CS);
    }
    {
        const int nep = (ne + 7) & ~7;
        const char* gcf = (const char*)(coefTT + e0);
        for (int c = wave; c * 1024 < nep * 8; c += NW) dma16(gcf + min(c * 1024 + lane64 * 16, nep * 8 - 16), cfb + c * 1024);
        const char* glc = (const char*)(tt_loc + e0);
        for (int c = wave; c * 1024 < nep * 2; c += NW) dma16(glc + min(c * 1024 + lane64 * 16, nep * 2 - 16), lcb + c * 1024);
    }
    if (tid < P) pts[tid] = mypt;
    if (tid <= P) off[tid] = myoff;
    __syncthreads();
    const G2* cf = (const G2*)cfb;
    const u16* lc = (const u16*)lcb;
    for (int p = grp; p < P; p += NG) {
        const long j = pts[p];
        if (j < 0) continue;
        const int lo = off[p], hi = off[p + 1];
        F4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll 4
        for (int q = lo; q < hi; ++q) {
            const int l = lc[q];
            const F4 p0 = *(const F4*)(rows + (l * R) * CS + ll * 4);
            const F4 p1 = R == 2 ? *(const F4*)(rows + (l * R + R - 1) * CS + ll * 4) : p0;
            accumT<TMODE>(a0, a1, cf[q], p0, p1);
        }
        storeT<TMODE>(out, ldo, j, cb + ll * 4, a0, a1);
    }
}

__global__ void k_copy(const F4* in, F4* out, long n_in, long n_out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
    F4 acc = {0, 0, 0, 0};
    for (long i = t; i < n_in; i += st) { const F4 v = in[i]; acc.v[0] += v.v[0]; }
    for (long i = t; i < n_out; i += st) st4((float*)(out + i), acc);
}

static int g_iters = 100;
template <class F>
float timeit(F f) {
    for (int i = 0; i < 5; ++i) f();
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    for (int i = 0; i < g_iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e3f / g_iters;
}

static uint32_t part1by2(uint32_t x) {
    x &= 0x3ff; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249;
    return x;
}

struct Graph { std::vector<int> nbr; std::vector<G2> coef; };
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

struct Plan {
    int P; long tiles;
    std::vector<int> pts, nu, uq; std::vector<u16> loc, selfloc; std::vector<G2> coefP; int umax = 0; double uavg = 0;
    // transposed
    std::vector<int> tnu, tuq, tbase; std::vector<u16> toff, ttloc; std::vector<G2> coefTT; int UT = 0, tumax = 0, emax = 0; double tuavg = 0;
};
// order[b*N + r] = point id at Morton rank r of cloud b
static Plan build_plan(const Graph& g, const std::vector<int>& order, const std::vector<int>& tptr, const std::vector<int>& tedge,
                       const std::vector<G2>& coefT, long Nt, int N, int k, int P) {
    Plan pl; pl.P = P; pl.tiles = Nt / P;   // N % P == 0 in the lab
    const size_t PK = (size_t)P * k;
    pl.pts.resize(pl.tiles * P); pl.nu.resize(pl.tiles); pl.uq.assign(pl.tiles * PK, 0); pl.loc.resize(pl.tiles * PK);
    pl.selfloc.resize(pl.tiles * P); pl.coefP.resize(pl.tiles * PK);
    pl.tnu.resize(pl.tiles); pl.tbase.resize(pl.tiles); pl.toff.resize(pl.tiles * (P + 1));
    std::vector<std::vector<int>> tu(pl.tiles);
    for (long t = 0; t < pl.tiles; ++t) {
        std::vector<int> ids;
        for (int p = 0; p < P; ++p) {
            const int i = order[t * P + p];
            pl.pts[t * P + p] = i;
            ids.push_back(i);
            for (int s = 0; s < k; ++s) ids.push_back(g.nbr[(size_t)i * k + s]);
        }
        std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        for (int p = 0; p < P; ++p) {
            const int i = order[t * P + p];
            pl.selfloc[t * P + p] = (u16)(std::lower_bound(ids.begin(), ids.end(), i) - ids.begin());
            for (int s = 0; s < k; ++s) {
                pl.loc[(t * P + p) * k + s] = (u16)(std::lower_bound(ids.begin(), ids.end(), g.nbr[(size_t)i * k + s]) - ids.begin());
                pl.coefP[(t * P + p) * k + s] = g.coef[(size_t)i * k + s];
            }
        }
        pl.nu[t] = (int)ids.size(); pl.umax = std::max(pl.umax, (int)ids.size()); pl.uavg += ids.size();
        std::copy(ids.begin(), ids.end(), pl.uq.begin() + t * PK);
        std::fill(pl.uq.begin() + t * PK + ids.size(), pl.uq.begin() + (t + 1) * PK, ids.back());
        // transposed: unique sources of the in-edges of the tile's targets; entries in tile-major order, padded to 8
        std::vector<int> src;
        for (int p = 0; p < P; ++p) { const int j = order[t * P + p]; for (int q = tptr[j]; q < tptr[j + 1]; ++q) src.push_back(tedge[q] / k); }
        pl.emax = std::max(pl.emax, (int)src.size());
        std::sort(src.begin(), src.end()); src.erase(std::unique(src.begin(), src.end()), src.end());
        pl.tbase[t] = (int)pl.ttloc.size();
        int run = 0;
        for (int p = 0; p < P; ++p) {
            const int j = order[t * P + p];
            pl.toff[t * (P + 1) + p] = (u16)run;
            for (int q = tptr[j]; q < tptr[j + 1]; ++q, ++run) {
                pl.ttloc.push_back((u16)(std::lower_bound(src.begin(), src.end(), tedge[q] / k) - src.begin()));
                pl.coefTT.push_back(coefT[q]);
            }
        }
        pl.toff[t * (P + 1) + P] = (u16)run;
        while (pl.ttloc.size() % 8) { pl.ttloc.push_back(0); pl.coefTT.push_back(G2{0.f, 0.f}); }
        pl.tnu[t] = (int)src.size(); pl.tumax = std::max(pl.tumax, (int)src.size()); pl.tuavg += src.size();
        tu[t] = src;
    }
    for (int q = 0; q < 64; ++q) { pl.ttloc.push_back(0); pl.coefTT.push_back(G2{0.f, 0.f}); }   // slack for the clamped tail chunk
    pl.uavg /= pl.tiles; pl.tuavg /= pl.tiles;
    pl.UT = pl.tumax;
    pl.tuq.assign((size_t)pl.tiles * pl.UT, 0);
    for (long t = 0; t < pl.tiles; ++t) std::copy(tu[t].begin(), tu[t].end(), pl.tuq.begin() + (size_t)t * pl.UT);
    return pl;
}

template <class T> T* upload(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

struct DevPlan { int *pts, *nu, *uq; u16 *loc, *selfloc; G2* coefP; int *tnu, *tuq, *tbase; u16 *toff, *ttloc; G2* coefTT; };
static DevPlan upload_plan(const Plan& p) {
    return DevPlan{upload(p.pts), upload(p.nu), upload(p.uq), upload(p.loc), upload(p.selfloc), upload(p.coefP), upload(p.tnu), upload(p.tuq), upload(p.tbase),
                   upload(p.toff), upload(p.ttloc), upload(p.coefTT)};
}
static void free_plan(DevPlan& d) {
    for (void* p : {(void*)d.pts, (void*)d.nu, (void*)d.uq, (void*)d.loc, (void*)d.selfloc, (void*)d.coefP, (void*)d.tnu, (void*)d.tuq, (void*)d.tbase, (void*)d.toff,
                    (void*)d.ttloc, (void*)d.coefTT})
        CK(hipFree(p));
}

static unsigned long long* g_stamps = nullptr;   // [max blocks][8]
static void stamp_report(long nblocks, const char* tag) {
    std::vector<unsigned long long> h((size_t)nblocks * 8);
    CK(hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double d[5] = {0, 0, 0, 0, 0}; long n = 0;
    unsigned long long smin[8], emax[8], slast[8];
    for (int x = 0; x < 8; ++x) { smin[x] = ~0ull; emax[x] = 0; slast[x] = 0; }
    for (long b = 0; b < nblocks; ++b) {
        const unsigned long long* t = &h[b * 8];
        if (!t[5]) continue;
        for (int q = 0; q < 5; ++q) d[q] += (double)(t[q + 1] - t[q]);
        ++n;
        const int x = b & 7;
        smin[x] = std::min(smin[x], t[0]); emax[x] = std::max(emax[x], t[5]); slast[x] = std::max(slast[x], t[0]);
    }
    double span = 0, ramp = 0;
    for (int x = 0; x < 8; ++x) { span += (double)(emax[x] - smin[x]) / 8; ramp += (double)(slast[x] - smin[x]) / 8; }
    printf("    stamps %-40s WGs %ld  ids %.0f | issue %.0f | wait+barrier %.0f | compute %.0f | store %.0f  (s_memtime ticks, mean per WG); per-XCD first-start -> last-end %.0f, first -> last start %.0f\n",
           tag, n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, span, ramp);
}

struct Ctx {
    long Nt, E; int k, C; const float* d_x; float *o_ref, *o_new; long ldo; G2* d_coef; double mb; int mode;
};

static double check(const Ctx& c) {   // only the written columns: mode 0 out[2Nt, C]; mode 1 out[Nt, C..4C) of a 4C-wide buffer
    const size_t n = c.mode == 0 ? (size_t)2 * c.Nt * c.C : (size_t)c.Nt * 4 * c.C;
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), c.mode == 0 ? c.o_ref : c.o_ref - c.C, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), c.mode == 0 ? c.o_new : c.o_new - c.C, n * 4, hipMemcpyDeviceToHost));
    double md = 0; size_t nz = 0;
    if (c.mode == 0) { for (size_t q = 0; q < n; ++q) { md = std::max(md, (double)fabsf(a[q] - b[q])); nz += a[q] != 0.f; } }
    else for (long i = 0; i < c.Nt; ++i) for (int q = c.C; q < 4 * c.C; ++q) { const size_t o = (size_t)i * 4 * c.C + q; md = std::max(md, (double)fabsf(a[o] - b[o])); nz += a[o] != 0.f; }
    if (nz < n / 4) return 1e30;   // the reference itself must be populated
    return md;
}

template <int MODE, int P, int NT, int CAP, int CS>
static void run_tile(const Ctx& c, const Plan& pl, const DevPlan& dp, const char* tag) {
    using TG = TileGeom<CS, NT>;
    constexpr int R = MODE == 0 ? 1 : 2;
    constexpr int CAPR = (CAP * R + TG::PPI * TG::NW - 1) / (TG::PPI * TG::NW) * (TG::PPI * TG::NW);
    const int PK = P * c.k;
    const size_t lds = (size_t)CAPR * CS * 4 + (size_t)(PK * 8 + 1023) / 1024 * 1024 + (size_t)(PK * 2 + 1023) / 1024 * 1024 + P * 6 + 16;
    if (lds > 160 * 1024 || c.C % CS) return;
    if (pl.umax > CAP) { printf("  tile %s P=%d CAP=%d: plan exceeds CAP (U max %d) - skipped\n", tag, P, CAP, pl.umax); return; }
    const int slabs = c.C / CS;
    CK(hipFuncSetAttribute((const void*)k_tile<MODE, P, NT, CAP, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(c.mode == 0 ? c.o_new : c.o_new - c.C, 0, (c.mode == 0 ? (size_t)2 * c.Nt * c.C : (size_t)c.Nt * 4 * c.C) * 4));
    auto f = [&] {
        hipLaunchKernelGGL((k_tile<MODE, P, NT, CAP, CS>), dim3(pl.tiles * slabs), dim3(NT), lds, 0, dp.pts, dp.nu, dp.uq, dp.loc, dp.selfloc, dp.coefP, c.k,
                           c.d_x, (long)c.C, c.o_new, c.ldo, c.C, slabs);
    };
    const float us = timeit(f);
    const double md = check(c);
    char lab[160];
    snprintf(lab, sizeof lab, "tile %s P=%d NT=%d CAP=%d CS=%d lds=%zuK wg/cu=%d", tag, P, NT, CAP, CS, lds / 1024,
             (int)std::min<size_t>(160 * 1024 / lds, 2048 / NT));
    printf("  %-62s %8.2f us  %7.1f GB/s  frac %.3f  maxdiff %g\n", lab, us, c.mb / us * 1e3, c.mb / us * 1e3 / 8000, md);
    if (g_stamps && c.Nt <= 65536) {
        CK(hipMemset(g_stamps, 0, (size_t)pl.tiles * slabs * 64));
        for (int rep = 0; rep < 3; ++rep)
            hipLaunchKernelGGL((k_tile<MODE, P, NT, CAP, CS>), dim3(pl.tiles * slabs), dim3(NT), lds, 0, dp.pts, dp.nu, dp.uq, dp.loc, dp.selfloc, dp.coefP, c.k,
                               c.d_x, (long)c.C, c.o_new, c.ldo, c.C, slabs, g_stamps);
        CK(hipDeviceSynchronize());
        stamp_report(pl.tiles * slabs, lab);
    }
    fflush(stdout);
}

struct CtxT { long Nt, E; int k, C; const float* d_dy; float *o_ref, *o_new; long ldo; double mb; int tmode; };
static double checkT(const CtxT& c) {
    const size_t n = c.tmode == 0 ? (size_t)2 * c.Nt * c.C : (size_t)c.Nt * c.C;
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), c.o_ref, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), c.o_new, n * 4, hipMemcpyDeviceToHost));
    double md = 0; size_t nz = 0;
    for (size_t q = 0; q < n; ++q) { md = std::max(md, (double)fabsf(a[q] - b[q])); nz += a[q] != 0.f; }
    if (nz < n / 4) return 1e30;
    return md;
}
template <int TMODE, int P, int NT, int CAP, int ECAP, int CS>
static void run_tileT(const CtxT& c, const Plan& pl, const DevPlan& dp) {
    using TG = TileGeom<CS, NT>;
    constexpr int R = TMODE == 0 ? 1 : 2;
    constexpr int CAPR = (CAP * R + TG::PPI * TG::NW - 1) / (TG::PPI * TG::NW) * (TG::PPI * TG::NW);
    const size_t lds = (size_t)CAPR * CS * 4 + (size_t)(ECAP * 8 + 1023) / 1024 * 1024 + (size_t)(ECAP * 2 + 1023) / 1024 * 1024 + P * 6 + 16;
    if (lds > 160 * 1024 || c.C % CS) return;
    if (pl.tumax > CAP || pl.emax + 8 > ECAP) { printf("  tileT P=%d: plan exceeds CAP/ECAP (U max %d, in-edges max %d) - skipped\n", P, pl.tumax, pl.emax); return; }
    const int slabs = c.C / CS;
    CK(hipFuncSetAttribute((const void*)k_tileT<TMODE, P, NT, CAP, ECAP, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(c.o_new, 0, (c.tmode == 0 ? (size_t)2 * c.Nt * c.C : (size_t)c.Nt * c.C) * 4));
    auto f = [&] {
        hipLaunchKernelGGL((k_tileT<TMODE, P, NT, CAP, ECAP, CS>), dim3(pl.tiles * slabs), dim3(NT), lds, 0, dp.pts, dp.tnu, dp.tuq, pl.UT, dp.tbase, dp.toff,
                           dp.ttloc, dp.coefTT, c.d_dy, (long)c.C, c.o_new, c.ldo, slabs);
    };
    const float us = timeit(f);
    const double md = checkT(c);
    char lab[160];
    snprintf(lab, sizeof lab, "tileT P=%d NT=%d CAP=%d ECAP=%d CS=%d lds=%zuK wg/cu=%d", P, NT, CAP, ECAP, CS, lds / 1024,
             (int)std::min<size_t>(160 * 1024 / lds, 2048 / NT));
    printf("  %-62s %8.2f us  %7.1f GB/s  frac %.3f  maxdiff %g\n", lab, us, c.mb / us * 1e3, c.mb / us * 1e3 / 8000, md);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int N = 1024, k = 20;
    std::vector<int> Bs;
    for (int a = 1; a < argc; ++a) Bs.push_back(atoi(argv[a]));
    if (Bs.empty()) Bs = {32, 512};
    std::mt19937 rng(1);
    if (getenv("TILE_LAB_STAMPS")) CK(hipMalloc(&g_stamps, (size_t)1 << 24));
    for (int B : Bs) {
        g_iters = B > 64 ? 20 : 100;
        if (getenv("TILE_LAB_ITERS")) g_iters = atoi(getenv("TILE_LAB_ITERS"));
        const long Nt = (long)B * N, E = Nt * k;
        const int Bu = std::min(B, 32);
        std::vector<float> pos((size_t)Bu * N * 3);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (size_t i = 0; i < (size_t)Bu * N; ++i) {
            float x = nd(rng), y = nd(rng), z = nd(rng); const float n = sqrtf(x * x + y * y + z * z); x /= n; y /= n; z /= n;
            const float th = acosf(z), ph = atan2f(y, x), r = 1.f + 0.25f * sinf(3 * th) * cosf(2 * ph);
            pos[3 * i] = r * x; pos[3 * i + 1] = r * y; pos[3 * i + 2] = r * z;
        }
        Graph gr = build_graph(pos, Bu, N, k, rng);
        // Morton rank per cloud (points keep their generation order in memory)
        std::vector<int> order(Nt);
        for (int b = 0; b < Bu; ++b) {
            std::vector<std::pair<uint32_t, int>> key(N);
            for (int i = 0; i < N; ++i) {
                const float* p = &pos[((size_t)b * N + i) * 3];
                auto qz = [](float v) { return (uint32_t)std::min(1023.f, std::max(0.f, (v + 1.3f) / 2.6f * 1024.f)); };
                key[i] = {part1by2(qz(p[0])) | (part1by2(qz(p[1])) << 1) | (part1by2(qz(p[2])) << 2), b * N + i};
            }
            std::sort(key.begin(), key.end());
            for (int i = 0; i < N; ++i) order[(size_t)b * N + i] = key[i].second;
        }
        // replicate the first Bu clouds up to B
        gr.nbr.resize(E); gr.coef.resize(E);
        const long base_e = (long)Bu * N * k, base_n = (long)Bu * N;
        for (long e = base_e; e < E; ++e) { gr.nbr[e] = gr.nbr[e % base_e] + (int)((e / base_e) * base_n); gr.coef[e] = gr.coef[e % base_e]; }
        for (long i = base_n; i < Nt; ++i) order[i] = order[i % base_n] + (int)((i / base_n) * base_n);
        // CSC (ascending edge id per column) + coefficients in CSC order
        std::vector<int> tptr(Nt + 1, 0), tedge(E);
        for (long e = 0; e < E; ++e) tptr[gr.nbr[e] + 1]++;
        for (long j = 0; j < Nt; ++j) tptr[j + 1] += tptr[j];
        { std::vector<int> cur(tptr.begin(), tptr.end() - 1); for (long e = 0; e < E; ++e) tedge[cur[gr.nbr[e]]++] = (int)e; }
        std::vector<G2> coefT(E);
        for (long t = 0; t < E; ++t) coefT[t] = gr.coef[tedge[t]];
        int* d_nbr = upload(gr.nbr); G2* d_coef = upload(gr.coef);
        int* d_tptr = upload(tptr); int* d_tedge = upload(tedge); G2* d_coefT = upload(coefT);
        Plan p128 = build_plan(gr, order, tptr, tedge, coefT, Nt, N, k, 128), p32 = build_plan(gr, order, tptr, tedge, coefT, Nt, N, k, 32),
             p64 = build_plan(gr, order, tptr, tedge, coefT, Nt, N, k, 64);
        DevPlan d128 = upload_plan(p128), d32 = upload_plan(p32), d64 = upload_plan(p64);
        printf("plan: P=32 U avg %.0f max %d | P=64 U avg %.0f max %d | P=128 U avg %.0f max %d ; transposed P=32 U avg %.0f max %d E max %d | P=64 U avg %.0f max %d E max %d\n",
               p32.uavg, p32.umax, p64.uavg, p64.umax, p128.uavg, p128.umax, p32.tuavg, p32.tumax, p32.emax, p64.tuavg, p64.tumax, p64.emax);
        for (int C : {64, 128}) {
            if (B > 64 && C == 128) continue;
            std::vector<float> hx((size_t)2 * Nt * C);
            for (auto& v : hx) v = (float)(rng() % 2000) * 1e-3f - 1.f;
            float* d_x = upload(hx);
            float *d_out, *d_out2;
            CK(hipMalloc(&d_out, (size_t)Nt * 4 * C * 4)); CK(hipMalloc(&d_out2, (size_t)Nt * 4 * C * 4));
            const int groups = C / 4;
            const long total = Nt * groups;
            const size_t lds_st = (size_t)((256 + groups - 1) / groups + 1) * k * 12 + 16;
            for (int mode = 0; mode < 2; ++mode) {
                Ctx c{Nt, E, k, C, d_x, nullptr, nullptr, 0, d_coef, 0, mode};
                c.mb = mode == 0 ? (12.0 * C * Nt + 12.0 * E) / 1e6 : (20.0 * C * Nt + 12.0 * E) / 1e6;
                c.ldo = mode == 0 ? C : 4 * C;
                c.o_ref = mode == 0 ? d_out : d_out + C;
                c.o_new = mode == 0 ? d_out2 : d_out2 + C;
                CK(hipMemset(d_out, 0, (size_t)Nt * 4 * C * 4));
                printf("== B=%d C=%d %s  algorithmic %.1f MB\n", B, C, mode ? "divcurlnorm" : "grad", c.mb);
                float us = timeit([&] {
                    if (mode == 0) hipLaunchKernelGGL((k_staged<0>), dim3((total + 255) / 256), dim3(256), lds_st, 0, total, groups, d_coef, d_nbr, k, d_x, (long)C, c.o_ref, c.ldo, C);
                    else hipLaunchKernelGGL((k_staged<1>), dim3((total + 255) / 256), dim3(256), lds_st, 0, total, groups, d_coef, d_nbr, k, d_x, (long)C, c.o_ref, c.ldo, C);
                });
                printf("  %-78s %8.2f us  %7.1f GB/s  frac %.3f\n", "staged (production structure), generation point order", us, c.mb / us * 1e3, c.mb / us * 1e3 / 8000);
                if (g_stamps && Nt <= 65536) {
                    const long nb = (total + 255) / 256;
                    CK(hipMemset(g_stamps, 0, (size_t)nb * 64));
                    for (int rep = 0; rep < 3; ++rep) {
                        if (mode == 0) hipLaunchKernelGGL((k_staged<0>), dim3(nb), dim3(256), lds_st, 0, total, groups, d_coef, d_nbr, k, d_x, (long)C, c.o_ref, c.ldo, C, g_stamps);
                        else hipLaunchKernelGGL((k_staged<1>), dim3(nb), dim3(256), lds_st, 0, total, groups, d_coef, d_nbr, k, d_x, (long)C, c.o_ref, c.ldo, C, g_stamps);
                    }
                    CK(hipDeviceSynchronize());
                    stamp_report(nb, "staged");
                }
                if (mode == 0) {
                    run_tile<0, 32, 512, 176, 64>(c, p32, d32, "grad"); run_tile<0, 32, 256, 176, 32>(c, p32, d32, "grad"); run_tile<0, 32, 128, 176, 16>(c, p32, d32, "grad");
                    run_tile<0, 64, 1024, 248, 64>(c, p64, d64, "grad"); run_tile<0, 64, 512, 248, 32>(c, p64, d64, "grad"); run_tile<0, 64, 256, 248, 16>(c, p64, d64, "grad");
                    run_tile<0, 64, 512, 248, 64>(c, p64, d64, "grad"); run_tile<0, 64, 256, 248, 32>(c, p64, d64, "grad");
                    run_tile<0, 128, 1024, 400, 32>(c, p128, d128, "grad"); run_tile<0, 128, 512, 400, 16>(c, p128, d128, "grad"); run_tile<0, 128, 1024, 400, 64>(c, p128, d128, "grad");
                } else {
                    run_tile<1, 32, 512, 176, 64>(c, p32, d32, "dcn"); run_tile<1, 32, 256, 176, 32>(c, p32, d32, "dcn"); run_tile<1, 32, 128, 176, 16>(c, p32, d32, "dcn");
                    run_tile<1, 64, 1024, 248, 64>(c, p64, d64, "dcn"); run_tile<1, 64, 512, 248, 32>(c, p64, d64, "dcn"); run_tile<1, 64, 256, 248, 16>(c, p64, d64, "dcn");
                    run_tile<1, 64, 512, 248, 64>(c, p64, d64, "dcn"); run_tile<1, 64, 256, 248, 32>(c, p64, d64, "dcn");
                    run_tile<1, 128, 1024, 400, 32>(c, p128, d128, "dcn"); run_tile<1, 128, 512, 400, 16>(c, p128, d128, "dcn"); run_tile<1, 128, 512, 400, 32>(c, p128, d128, "dcn");
                }
                const long n_in = mode == 0 ? Nt * C / 4 : 2 * Nt * C / 4, n_out = mode == 0 ? 2 * Nt * C / 4 : 3 * Nt * C / 4;
                us = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const F4*)d_x, (F4*)d_out2, n_in, n_out); });
                printf("  %-78s %8.2f us  %7.1f GB/s\n", "stream copy of the same in + out bytes", us, (n_in + n_out) * 16.0 / 1e6 / us * 1e3);
            }
            // transposed
            for (int tmode = 0; tmode < 2; ++tmode) {
                CtxT c{Nt, E, k, C, d_x, d_out, d_out2, (long)C, (12.0 * C * Nt + 16.0 * E) / 1e6, tmode};
                CK(hipMemset(d_out, 0, (size_t)Nt * 4 * C * 4));
                printf("== B=%d C=%d %s  algorithmic %.1f MB\n", B, C, tmode ? "grad^T" : "div^T", c.mb);
                float us = timeit([&] {
                    if (tmode == 0) hipLaunchKernelGGL((k_stagedT<0>), dim3((total + 255) / 256), dim3(256), 0, 0, total, groups, d_coefT, d_tptr, d_tedge, k, d_x, (long)C, d_out, (long)C);
                    else hipLaunchKernelGGL((k_stagedT<1>), dim3((total + 255) / 256), dim3(256), 0, 0, total, groups, d_coefT, d_tptr, d_tedge, k, d_x, (long)C, d_out, (long)C);
                });
                printf("  %-78s %8.2f us  %7.1f GB/s  frac %.3f\n", "stagedT (production structure)", us, c.mb / us * 1e3, c.mb / us * 1e3 / 8000);
                if (tmode == 0) {
                    run_tileT<0, 32, 512, 192, 1280, 64>(c, p32, d32); run_tileT<0, 32, 256, 192, 1280, 32>(c, p32, d32); run_tileT<0, 32, 128, 192, 1280, 16>(c, p32, d32);
                    run_tileT<0, 64, 1024, 288, 2304, 64>(c, p64, d64); run_tileT<0, 64, 512, 288, 2304, 32>(c, p64, d64); run_tileT<0, 64, 256, 288, 2304, 16>(c, p64, d64);
                    run_tileT<0, 64, 512, 288, 2304, 64>(c, p64, d64);
                } else {
                    run_tileT<1, 32, 512, 192, 1280, 64>(c, p32, d32); run_tileT<1, 32, 256, 192, 1280, 32>(c, p32, d32); run_tileT<1, 32, 128, 192, 1280, 16>(c, p32, d32);
                    run_tileT<1, 64, 1024, 288, 2304, 64>(c, p64, d64); run_tileT<1, 64, 512, 288, 2304, 32>(c, p64, d64); run_tileT<1, 64, 256, 288, 2304, 16>(c, p64, d64);
                    run_tileT<1, 64, 512, 288, 2304, 64>(c, p64, d64);
                }
            }
            CK(hipFree(d_x)); CK(hipFree(d_out)); CK(hipFree(d_out2));
        }
        free_plan(d128); free_plan(d32); free_plan(d64);
        CK(hipFree(d_nbr)); CK(hipFree(d_coef)); CK(hipFree(d_tptr)); CK(hipFree(d_tedge)); CK(hipFree(d_coefT));
    }
    return 0;
}
