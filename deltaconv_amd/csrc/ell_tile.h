// Forward ELL applies / max-aggregation from a tile plan (tile_plan.h): the unique neighbour rows of a tile of P
// points are brought into LDS once by LDS-DMA, the k-loop reads LDS only.
//
// Measured on MI355X (tools/tile_lab.hip, profiles/r03*_tile_lab*.txt), B = 32 x 1024 points, k = 20, C = 64:
//   [div|curl|norm] 17.7 us staged -> 12.4 us (0.35 -> 0.50 of 8 TB/s), grad 10.0 -> 6.8 us (0.41 -> 0.60); texture-
//   addresser busy cycles 38.5 k -> 14.5 k per launch, LDS bank conflicts 0.
// Mapping: workgroup = (tile, 64-channel slab), thread = (point of the tile, 4 channels); 16 lanes per point.
//   * every lane group of a ds_read_b128 covers the 16 different 16-byte columns of 256-byte rows, so random rows are
//     bank-conflict free (MI355X_MICROARCH.md, LDS table: 256 B/clk/CU; the gather path delivers ~49);
//   * rows, coefficients (the k * 8 contiguous bytes of every point of the tile, straight from the operator) and local
//     indices arrive by `global_load_lds_dwordx4` (no staging registers, no ds_write: a register-staged variant of the
//     same kernel ran 2x slower, r03a);
//   * all row ids are loaded before the first DMA piece is issued (hipcc waits vmcnt(0) at the first use of an
//     ordinary load's result while DMA pieces are in flight, which would serialise the pieces);
//   * outputs leave with non-temporal stores: a plain store leaves the whole output dirty in the per-XCD L2 until the
//     end-of-kernel write-back; streaming it out during the kernel is worth 1.5 - 3.5 us per launch (r03e);
//   * tiles whose unique rows exceed the LDS capacity (or the plan's list) fetch the excess rows from global memory by
//     neighbour id -- correct for any graph, fast for the spatially coherent ones a kNN graph gives.
// Same FMAs in the same slot order as the staged kernels (ell_math.h): results are bit-identical (tests/test_gpu_tile.py).
#pragma once
#include <algorithm>
#include <initializer_list>
#include "common.h"
#include "ell_math.h"
#include "tile_plan.h"

namespace dctile {
using dcell::G2;
using dcell::Vec;
using dcell::vfma;
using dcell::vzero;

constexpr int CS = 64;        // channels per slab: 16 lanes x 4
// unique rows of a tile kept in LDS (~165 unique rows per tile of 64 points at k = 20: 127 KB of two-piece rows, one workgroup
// per CU).  Round-6 lab (profiles/r06_labs.txt): 124 rows for tiles of 32 points -- 64 KB, two workgroups of 512 threads per
// CU, one's pieces landing while the other walks -- LOSES: div|curl|norm 13.7 -> 17.0 us, the transposed family +45 - 75 %
// (3.0 x instead of 2.6 x halo rows and twice the per-tile overheads outweigh the overlap); the capacity stays one constant.
template <int P> constexpr int cap_rows() { return 248; }
__device__ __forceinline__ void store_nt(float* p, const Vec<4>& a) { dc_store16<DC_ST_TILE>(p, *reinterpret_cast<const dc_f32x4*>(&a)); }
// cache policy of the LDS-DMA pieces (gfx942+ CPol bits: 1 = sc0, 2 = nt, 16 = sc1): lab switch, see profiles/r06_labs.txt
#ifndef DC_DMA_AUX
#define DC_DMA_AUX 0
#endif
__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, DC_DMA_AUX);
}

template <int R, int P>
struct Geom {
    static constexpr int NT = P * 16, NW = NT / 64;
    static constexpr int CAP = cap_rows<P>();
    static constexpr int CAPR = (CAP * R + 4 * NW - 1) / (4 * NW) * (4 * NW);   // capacity in 256-byte pieces
    static constexpr int RIT = CAPR / (4 * NW);                                  // DMA instructions per wave
};
inline size_t chunk1k(size_t bytes) { return (bytes + 1023) / 1024 * 1024; }
// coefficients of a tile in LDS: whole 1-KiB DMA pieces, each holding the k * 8 bytes of 64 / (k / 2) points
inline int coef_ppi(int k) { return 64 / (k / 2); }
inline size_t coef_bytes(int P, int k) { return (size_t)((P + coef_ppi(k) - 1) / coef_ppi(k)) * 1024; }
template <int R, int P>
inline size_t lds_bytes(int k, bool coef) {
    return (size_t)Geom<R, P>::CAPR * 256 + (coef ? coef_bytes(P, k) : 0) + chunk1k((size_t)P * k * 2) + P * 4 + P * 2 + 16;
}

// BODY: per-thread accumulator object, copied from the kernel argument
//   static constexpr bool COEF, SELF;  static constexpr int NST (store instructions finish() issues per wave);
//   const float* in; long ldj, hs;             piece h of row j = in + j * ldj + h * hs
//   void init(int c);  void step(int s, G2 g, const Vec<4>& p0, const Vec<4>& p1);
//   void finish(long i, int c, const Vec<4>& s0, const Vec<4>& s1);
// One unit (tile, 64-channel slab) per workgroup: the form of the one-piece-per-row kernels (grad, max aggregation), whose
// 64 KB of rows let TWO workgroups share a CU -- as long as a wave stays within 64 registers, which the loop-carried state
// of the persistent form below (next ids, pending accumulators) does not (r03: 48 -> 94 registers, grad 8.0 -> 9.4 us).
template <int R, int P, class BODY>
__global__ __launch_bounds__(P * 16) void tile_unit_kernel(DcTilePlan L, const int* __restrict__ plan,
                                                          const float* __restrict__ coef, const int* __restrict__ nbr,
                                                          int slabs, int remap, BODY body) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using GM = Geom<R, P>;
    constexpr int NW = GM::NW, CAP = GM::CAP, CAPR = GM::CAPR, RIT = GM::RIT;
    const long b = dc_xcd_block(remap);
    const long tile = b / slabs;
    const int cb = (int)(b - tile * slabs) * CS;
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    const int wave = tid >> 6, lane64 = tid & 63;
    const int k = L.k, PK = L.PK;
    const int cpp = k >> 1, ppi = 64 / cpp;                               // 16-byte pieces per point, points per DMA piece
    const size_t cfb_bytes = BODY::COEF ? (size_t)((P + ppi - 1) / ppi) * 1024 : 0;
    float* rows = reinterpret_cast<float*>(smem);                         // [CAPR][64]
    char* cfb = smem + (size_t)CAPR * 256;                                // [PK] G2
    char* lcb = cfb + cfb_bytes;                                          // [PK] u16
    int* pts = reinterpret_cast<int*>(lcb + (PK * 2 + 1023) / 1024 * 1024);   // [P]
    unsigned short* sl = reinterpret_cast<unsigned short*>(pts + P);     // [P]
    const int* uq = plan + L.o_uniq + tile * PK;
    // row ids: one round trip; the unique count is read beside them, not before them (the list's tail repeats its last id)
    int rid[RIT];
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
        const int r = min((wave + it * NW) * 4 + (lane64 >> 4), PK * R - 1);
        rid[it] = uq[r / R];
    }
    const int U = plan[L.o_nu + tile];
    if (U == 0) return;                                                   // empty tile (block-uniform)
    const int UL = min(min(U, CAP), PK), nrow = UL * R;                   // rows held in LDS (the plan lists at most P * k)
    int mypt = -1;
    unsigned short mysl = 0;
    if (tid < P) {
        mypt = plan[L.o_pts + tile * P + tid];
        mysl = reinterpret_cast<const unsigned short*>(plan + L.o_self)[tile * P + tid];
    }
    // coefficient pieces of this lane: piece c = wave (+ NW) holds points c * ppi .. of the tile, this lane its point
    // lane / cpp and the 16-byte chunk lane % cpp of that point's k * 8 bytes (k <= 64: at most two pieces per wave)
    int cpt[2] = {-1, -1};
    const int cpp_pt = lane64 / cpp, cpp_ch = lane64 - cpp_pt * cpp;
    if (BODY::COEF && cpp_pt < ppi)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = (wave + u * NW) * ppi + cpp_pt;
            if (p < P) cpt[u] = plan[L.o_pts + tile * P + p];
        }
#pragma unroll
    for (int it = 0; it < RIT; ++it) asm volatile("" : "+v"(rid[it]));   // the id waits end here, before the first DMA piece
    asm volatile("" : "+v"(mypt));
    asm volatile("" : "+v"(cpt[0]), "+v"(cpt[1]));
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
        const int r0 = (wave + it * NW) * 4;
        if (r0 < nrow) {
            const int h = R == 2 ? (lane64 >> 4) & 1 : 0;                 // piece of the row this lane group loads
            dma16(body.in + (long)rid[it] * body.ldj + h * body.hs + cb + l16 * 4, rows + r0 * 64);
        }
    }
    {   // coefficients and local indices of the tile: contiguous -> whole 1-KiB chunks (tail lanes re-read the end)
        if (BODY::COEF)
#pragma unroll
            for (int u = 0; u < 2; ++u)                                   // (padding lanes / points re-read row 0: never used)
                if ((wave + u * NW) * ppi < P)
                    dma16(coef + (cpt[u] >= 0 ? (long)cpt[u] * k * 2 + cpp_ch * 4 : 0), cfb + (wave + u * NW) * 1024);
        const char* g = reinterpret_cast<const char*>(plan + L.o_loc) + (size_t)tile * PK * 2;
        for (int c = wave; c * 1024 < PK * 2; c += NW) dma16(g + min(c * 1024 + lane64 * 16, PK * 2 - 16), lcb + c * 1024);
    }
    if (tid < P) {
        pts[tid] = mypt;
        sl[tid] = mysl;
    }
    __syncthreads();
    const int c = cb + l16 * 4;
    body.init(c);
    if constexpr (BODY::ROWPASS) {                                        // every thread, padded points included
        static_assert(R == 1, "row pass: one row per point");
        for (int idx = tid; idx < nrow * 16; idx += P * 16) {             // P * 16 is a multiple of 16: idx % 16 == l16
            Vec<4>* q = reinterpret_cast<Vec<4>*>(rows + (idx >> 4) * 64 + l16 * 4);
            Vec<4> h = *q;
            body.cook(h);
            *q = h;
        }
        __syncthreads();
    }
    const long i = pts[grp];
    if (i < 0) return;
    const G2* cp = reinterpret_cast<const G2*>(cfb + (grp / ppi) * 1024 + (grp % ppi) * (k * 8));
    const unsigned short* lp = reinterpret_cast<const unsigned short*>(lcb) + grp * k;
    Vec<4> s0 = vzero<4>(), s1 = vzero<4>();
    if (U <= UL) {
#pragma unroll 4
        for (int s = 0; s < k; ++s) {
            const int l = lp[s];
            const Vec<4> p0 = *reinterpret_cast<const Vec<4>*>(rows + (l * R) * 64 + l16 * 4);
            const Vec<4> p1 = R == 2 ? *reinterpret_cast<const Vec<4>*>(rows + (l * R + R - 1) * 64 + l16 * 4) : p0;
            body.step(s, BODY::COEF ? cp[s] : G2{0.f, 0.f}, p0, p1);
        }
        if (BODY::SELF) {
            const int l = sl[grp];
            s0 = *reinterpret_cast<const Vec<4>*>(rows + (l * R) * 64 + l16 * 4);
            s1 = *reinterpret_cast<const Vec<4>*>(rows + (l * R + R - 1) * 64 + l16 * 4);
        }
    } else {                                                              // more unique rows than LDS holds: the rest by id
#pragma unroll 1
        for (int s = 0; s < k; ++s) {
            const int l = lp[s];
            Vec<4> p0, p1;
            if (l < UL) {
                p0 = *reinterpret_cast<const Vec<4>*>(rows + (l * R) * 64 + l16 * 4);
                p1 = R == 2 ? *reinterpret_cast<const Vec<4>*>(rows + (l * R + R - 1) * 64 + l16 * 4) : p0;
            } else {
                const float* g = body.in + (long)nbr[i * k + s] * body.ldj + c;
                p0 = dcell::vload<4>(g);
                if constexpr (BODY::ROWPASS) body.cook(p0);               // a row that LDS does not hold: raw from memory
                p1 = R == 2 ? dcell::vload<4>(g + body.hs) : p0;
            }
            body.step(s, BODY::COEF ? cp[s] : G2{0.f, 0.f}, p0, p1);
        }
        if (BODY::SELF) {
            const float* g = body.in + i * body.ldj + c;
            s0 = dcell::vload<4>(g);
            s1 = dcell::vload<4>(g + body.hs);
        }
    }
    body.finish(i, c, s0, s1);
}

// Persistent, software-pipelined form: the two-piece kernels (div, [div|curl|norm], hodge: 127 KB of rows, one workgroup
// per CU whatever the registers).
template <int R, int P, class BODY>
__global__ __launch_bounds__(P * 16) void tile_fwd_kernel(DcTilePlan L, const int* __restrict__ plan,
                                                          const float* __restrict__ coef, const int* __restrict__ nbr,
                                                          int slabs, int remap, int upw, unsigned long long* stamp,
                                                          const BODY body0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using GM = Geom<R, P>;
    constexpr int NW = GM::NW, CAP = GM::CAP, CAPR = GM::CAPR, RIT = GM::RIT;
    // Work unit = (tile, 64-channel slab); a workgroup owns `upw` CONSECUTIVE units (the slabs of a tile, then the next
    // tile of the same cloud: one set of row ids serves all slabs of its tile, and the ids of the next tile are requested
    // behind the DMA pieces of the current one, so their round trip -- a quarter of a workgroup's life when every unit
    // was its own workgroup (r03d phase stamps) -- is paid once per workgroup instead of once per tile).
    const long units = (long)L.T * slabs;
    const long u0 = dc_xcd_block(remap) * upw, u1 = min(u0 + (long)upw, units);
    if (u0 >= u1) return;
    dc_stamp_in(stamp);
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    const int wave = tid >> 6, lane64 = tid & 63;
    const int k = L.k, PK = L.PK;
    const int cpp = k >> 1, ppi = 64 / cpp;                               // 16-byte pieces per point, points per DMA piece
    const size_t cfb_bytes = BODY::COEF ? (size_t)((P + ppi - 1) / ppi) * 1024 : 0;
    float* rows = reinterpret_cast<float*>(smem);                         // [CAPR][64]
    char* cfb = smem + (size_t)CAPR * 256;                                // [PK] G2
    char* lcb = cfb + cfb_bytes;                                          // [PK] u16
    int* pts = reinterpret_cast<int*>(lcb + (PK * 2 + 1023) / 1024 * 1024);   // [P]
    unsigned short* sl = reinterpret_cast<unsigned short*>(pts + P);     // [P]
    // coefficient pieces of this lane: piece c = wave (+ NW) holds points c * ppi .. of the tile, this lane its point
    // lane / cpp and the 16-byte chunk lane % cpp of that point's k * 8 bytes (k <= 64: at most two pieces per wave)
    const int cpp_pt = lane64 / cpp, cpp_ch = lane64 - cpp_pt * cpp;

    // everything a tile needs from the plan before its first DMA piece: one round trip
    struct Ids { int rid[RIT]; int U, mypt, cpt[2]; unsigned short mysl; };
    auto load_ids = [&](long tile, Ids& d) {
        const int* uq = plan + L.o_uniq + tile * PK;
        // row ids; the unique count is read beside them, not before them (the list's tail repeats its last id)
#pragma unroll
        for (int it = 0; it < RIT; ++it) {
            const int r = min((wave + it * NW) * 4 + (lane64 >> 4), PK * R - 1);
            d.rid[it] = uq[r / R];
        }
        d.U = plan[L.o_nu + tile];
        d.mypt = -1;
        d.mysl = 0;
        if (tid < P) {
            d.mypt = plan[L.o_pts + tile * P + tid];
            d.mysl = reinterpret_cast<const unsigned short*>(plan + L.o_self)[tile * P + tid];
        }
        d.cpt[0] = d.cpt[1] = -1;
        if (BODY::COEF && cpp_pt < ppi)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = (wave + u * NW) * ppi + cpp_pt;
                if (p < P) d.cpt[u] = plan[L.o_pts + tile * P + p];
            }
    };
    Ids cur;
    load_ids(u0 / slabs, cur);
    // Software pipeline over the units: the output stores of unit u-1 are issued BEHIND the DMA pieces of unit u, and the
    // wait for those pieces leaves the BODY::NST youngest vector-memory instructions (the stores) in flight -- the vmcnt
    // counter retires in order, so a wait for pieces issued behind the stores would also wait for the stores' round trip
    // to memory (non-temporal: the slowest instructions of the kernel).
    BODY pbody = body0;                                                      // unit u-1: accumulators waiting to be stored
    long pi = -1;
    int pc = 0;
    Vec<4> ps0 = vzero<4>(), ps1 = vzero<4>();
    for (long u = u0; u < u1; ++u) {
        const long tile = u / slabs;
        const int cb = (int)(u - tile * slabs) * CS;
        // the id waits end here, before the first DMA piece (hipcc waits vmcnt(0) at the first use of an ordinary load's
        // result, which would serialise DMA pieces already in flight)
#pragma unroll
        for (int it = 0; it < RIT; ++it) asm volatile("" : "+v"(cur.rid[it]));
        asm volatile("" : "+v"(cur.mypt), "+v"(cur.U));
        asm volatile("" : "+v"(cur.cpt[0]), "+v"(cur.cpt[1]));
        const int U = __builtin_amdgcn_readfirstlane(cur.U);              // block-uniform
        const int UL = min(min(U, CAP), PK), nrow = UL * R;               // rows held in LDS (the plan lists at most P * k)
        // the next unit's tile: its ids travel while this unit's rows land (requested BEFORE the pieces: older than them)
        Ids nxt = cur;
        if (u + 1 < u1 && (u + 1) / slabs != tile) load_ids((u + 1) / slabs, nxt);
        asm volatile("" ::: "memory");
        if (U != 0) {
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
                const int r0 = (wave + it * NW) * 4;
                if (r0 < nrow) {
                    const int h = R == 2 ? (lane64 >> 4) & 1 : 0;             // piece of the row this lane group loads
                    dma16(body0.in + (long)cur.rid[it] * body0.ldj + h * body0.hs + cb + l16 * 4, rows + r0 * 64);
                }
            }
            // coefficients and local indices of the tile: contiguous -> whole 1-KiB chunks (tail lanes re-read the end)
            if (BODY::COEF)
#pragma unroll
                for (int w = 0; w < 2; ++w)                                   // (padding lanes / points re-read row 0: never used)
                    if ((wave + w * NW) * ppi < P)
                        dma16(coef + (cur.cpt[w] >= 0 ? (long)cur.cpt[w] * k * 2 + cpp_ch * 4 : 0), cfb + (wave + w * NW) * 1024);
            const char* g = reinterpret_cast<const char*>(plan + L.o_loc) + (size_t)tile * PK * 2;
            for (int c = wave; c * 1024 < PK * 2; c += NW) dma16(g + min(c * 1024 + lane64 * 16, PK * 2 - 16), lcb + c * 1024);
            if (tid < P) {
                pts[tid] = cur.mypt;
                sl[tid] = cur.mysl;
            }
        }
        // unit u-1 leaves: exactly BODY::NST store instructions per wave that has a live point, none otherwise
        asm volatile("" ::: "memory");
        const bool stores = __builtin_amdgcn_ballot_w64(pi >= 0) != 0;      // wave-uniform
        if (pi >= 0) pbody.finish(pi, pc, ps0, ps1);
        asm volatile("" ::: "memory");
        if (stores) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(BODY::NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        pi = -1;
        if (U != 0) {
            const long i = pts[grp];
            if (i >= 0) {
                BODY body = body0;
                const G2* cp = reinterpret_cast<const G2*>(cfb + (grp / ppi) * 1024 + (grp % ppi) * (k * 8));
                const unsigned short* lp = reinterpret_cast<const unsigned short*>(lcb) + grp * k;
                const int c = cb + l16 * 4;
                body.init(c);
                Vec<4> s0 = vzero<4>(), s1 = vzero<4>();
                if (U <= UL) {
#pragma unroll BODY::KU
                    for (int s = 0; s < k; ++s) {
                        const int l = lp[s];
                        const Vec<4> p0 = *reinterpret_cast<const Vec<4>*>(rows + (l * R) * 64 + l16 * 4);
                        const Vec<4> p1 = R == 2 ? *reinterpret_cast<const Vec<4>*>(rows + (l * R + R - 1) * 64 + l16 * 4) : p0;
                        body.step(s, BODY::COEF ? cp[s] : G2{0.f, 0.f}, p0, p1);
                    }
                    if (BODY::SELF) {
                        const int l = sl[grp];
                        s0 = *reinterpret_cast<const Vec<4>*>(rows + (l * R) * 64 + l16 * 4);
                        s1 = *reinterpret_cast<const Vec<4>*>(rows + (l * R + R - 1) * 64 + l16 * 4);
                    }
                } else {                                                      // more unique rows than LDS holds: the rest by id
#pragma unroll 1
                    for (int s = 0; s < k; ++s) {
                        const int l = lp[s];
                        Vec<4> p0, p1;
                        if (l < UL) {
                            p0 = *reinterpret_cast<const Vec<4>*>(rows + (l * R) * 64 + l16 * 4);
                            p1 = R == 2 ? *reinterpret_cast<const Vec<4>*>(rows + (l * R + R - 1) * 64 + l16 * 4) : p0;
                        } else {
                            const float* g = body.in + (long)nbr[i * k + s] * body.ldj + c;
                            p0 = dcell::vload<4>(g);
                            p1 = R == 2 ? dcell::vload<4>(g + body.hs) : p0;
                        }
                        body.step(s, BODY::COEF ? cp[s] : G2{0.f, 0.f}, p0, p1);
                    }
                    if (BODY::SELF) {
                        const float* g = body.in + i * body.ldj + c;
                        s0 = dcell::vload<4>(g);
                        s1 = dcell::vload<4>(g + body.hs);
                    }
                }
                pbody = body; pi = i; pc = c; ps0 = s0; ps1 = s1;
            }
        }
        // the rows are free for the next unit's pieces once every wave has read its last LDS operand (no vector-memory wait)
        if (u + 1 < u1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur = nxt;
    }
    if (pi >= 0) pbody.finish(pi, pc, ps0, ps1);
    dc_stamp_out(stamp);
}

// ---- bodies (the arithmetic of ell_math.h, slot by slot) ---------------------------------------------------------
// grad @ x (ell_math.h: grad_fwd)
struct GradB {
    static constexpr int TAG = 0;
    static constexpr bool COEF = true, SELF = false;
    static constexpr int KU = 4;                       // k-loop unroll of the persistent kernel
    static constexpr bool ROWPASS = false;
    static constexpr int NST = 2;                      // vector-memory store instructions of finish()
    const float* in; long ldj, hs; float* out; long ldo;
    Vec<4> au, av;
    __device__ void init(int) { au = vzero<4>(); av = vzero<4>(); }
    __device__ void step(int, G2 g, const Vec<4>& x, const Vec<4>&) { vfma<4>(au, g.a, x); vfma<4>(av, g.b, x); }
    __device__ void finish(long i, int c, const Vec<4>&, const Vec<4>&) {
        store_nt(out + (2 * i) * ldo + c, au);
        store_nt(out + (2 * i + 1) * ldo + c, av);
    }
};
// div @ v (div_fwd)
struct DivB {
    static constexpr int TAG = 3;                      // dc_stamp_tag kind
    static constexpr bool COEF = true, SELF = false;
    static constexpr int KU = 4;                       // k-loop unroll of the persistent kernel
    static constexpr bool ROWPASS = false;
    static constexpr int NST = 1;
    const float* in; long ldj, hs; float* out; long ldo;
    Vec<4> acc;
    __device__ void init(int) { acc = vzero<4>(); }
    __device__ void step(int, G2 d, const Vec<4>& vu, const Vec<4>& vv) { vfma<4>(acc, d.a, vu); vfma<4>(acc, d.b, vv); }
    __device__ void finish(long i, int c, const Vec<4>&, const Vec<4>&) { store_nt(out + i * ldo + c, acc); }
};
// [div v | curl v | norm v] (divcurlnorm_fwd)
struct DivCurlNormB {
    static constexpr int TAG = 1;
    static constexpr bool COEF = true, SELF = true;
    static constexpr int KU = 4;                       // k-loop unroll of the persistent kernel
    static constexpr bool ROWPASS = false;
    static constexpr int NST = 3;
    const float* in; long ldj, hs; float* out; long ldo; int C;
    Vec<4> dv, cv;
    __device__ void init(int) { dv = vzero<4>(); cv = vzero<4>(); }
    __device__ void step(int, G2 d, const Vec<4>& vu, const Vec<4>& vv) {
        vfma<4>(dv, d.a, vu);
        vfma<4>(dv, d.b, vv);
        vfma<4>(cv, d.a, vv);
        vfma<4>(cv, -d.b, vu);
    }
    __device__ void finish(long i, int c, const Vec<4>& ou, const Vec<4>& ov) {
        Vec<4> nv;
#pragma unroll
        for (int q = 0; q < 4; ++q) nv.v[q] = sqrtf(fmaf(ou.v[q], ou.v[q], ov.v[q] * ov.v[q]));
        store_nt(out + i * ldo + c, dv);
        store_nt(out + i * ldo + C + c, cv);
        store_nt(out + i * ldo + 2 * C + c, nv);
    }
};
// hodge Laplacian from [div v | curl v] (hodge_fwd): pieces = the two column blocks of one row
struct HodgeB {
    static constexpr int TAG = 2;
    static constexpr bool COEF = true, SELF = false;
    static constexpr int KU = 2;     // 4 puts the persistent kernel at 128 registers + 1 scratch spill (round-4 verdict); same op order
    static constexpr bool ROWPASS = false;
    static constexpr int NST = 2;
    const float* in; long ldj, hs; float* out; long ldo;
    Vec<4> hu, hv;
    __device__ void init(int) { hu = vzero<4>(); hv = vzero<4>(); }
    __device__ void step(int, G2 g, const Vec<4>& dv, const Vec<4>& cv) {
        vfma<4>(hu, -g.a, dv);
        vfma<4>(hu, g.b, cv);
        vfma<4>(hv, -g.b, dv);
        vfma<4>(hv, -g.a, cv);
    }
    __device__ void finish(long i, int c, const Vec<4>&, const Vec<4>&) {
        store_nt(out + (2 * i) * ldo + c, hu);
        store_nt(out + (2 * i + 1) * ldo + c, hv);
    }
};
// max over the k neighbours, first maximal slot (knn_max_fwd / knn_max_affine_fwd); AFFINE: y = act(scale * h + shift)
template <bool AFFINE>
struct KnnMaxB {
    static constexpr int TAG = 0;
    static constexpr bool COEF = false, SELF = false;
    static constexpr int KU = 4;
    // ROWPASS: BatchNorm + activation of the producing block are applied ONCE per unique row, in place in LDS, before the walk
    // (the same fmaf and select per element as per gathered value before: same bits; ~165 rows instead of 64 x k gathers per tile --
    // the walk was VALU-bound: 9 of its 12 instructions per gathered value were this transform)
    static constexpr bool ROWPASS = AFFINE;
    static constexpr int NST = 2;
    const float* in; long ldj, hs; const float *scale, *shift; float slope; float* out; long ldo; unsigned char* arg; long lda;
    // optional epilogue (round 6): the layer's LAST s_mlp block rides along -- out = act2(scale2 h2[i] + shift2) + max, the
    // residual form of deltaconv.py:59 (`x = s_mlp(...) + x_max`), also written to out2 (the layer's block of the heads' concat
    // buffer): the separate BatchNorm / activation pass over [n, C] and the x_max round trip go away.  Same two addends as
    // dc_bn_act2 with a residual: same bits.
    const float* h2 = nullptr; long ldh2 = 0; const float *scale2 = nullptr, *shift2 = nullptr; float slope2 = 0.f;
    float* out2 = nullptr; long ldo2 = 0;
    Vec<4> best; unsigned slot[4]; float sc[4], sh[4];
    __device__ void init(int c) {
        if (AFFINE)
#pragma unroll
            for (int q = 0; q < 4; ++q) { sc[q] = scale[c + q]; sh[q] = shift[c + q]; }
    }
    __device__ void cook(Vec<4>& h) const {                 // a row piece of this thread's four channels, raw -> activated
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float z = fmaf(sc[q], h.v[q], sh[q]);
            h.v[q] = z > 0.f ? z : slope * z;
        }
    }
    __device__ void step(int s, G2, const Vec<4>& h, const Vec<4>&) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float y = h.v[q];
            const bool up = s == 0 || y > best.v[q];
            best.v[q] = up ? y : best.v[q];
            slot[q] = up ? (unsigned)s : slot[q];
        }
    }
    __device__ void finish(long i, int c, const Vec<4>&, const Vec<4>&) {
        if (h2) {
            const Vec<4> v = dcell::vload<4>(h2 + i * ldh2 + c), s2 = dcell::vload<4>(scale2 + c), t2 = dcell::vload<4>(shift2 + c);
            Vec<4> y;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float z = fmaf(s2.v[q], v.v[q], t2.v[q]);
                y.v[q] = (z > 0.f ? z : slope2 * z) + best.v[q];
            }
            // x' is read again by the very next launches (grad apply, the next layer's products): a plain store keeps it cached
            *reinterpret_cast<dc_f32x4*>(out + i * ldo + c) = *reinterpret_cast<const dc_f32x4*>(&y);
            if (out2) store_nt(out2 + i * ldo2 + c, y);
        } else {
            store_nt(out + i * ldo + c, best);
        }
        // the four slot bytes as one 32-bit store (c and lda are multiples of 4)
        const unsigned w = slot[0] | (slot[1] << 8) | (slot[2] << 16) | (slot[3] << 24);
        __builtin_nontemporal_store(w, reinterpret_cast<unsigned*>(arg + i * lda + c));
    }
};

// workgroups a CU holds (LDS; 2048 threads) x CUs of the device = the persistent grid's capacity
inline int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}
template <int R, int P, class BODY>
inline void launch_one(const DcTilePlan& L, const int* plan, const float* coef, const int* nbr, int C, BODY body, hipStream_t s) {
    const int slabs = C / CS;
    const size_t lds = lds_bytes<R, P>(L.k, BODY::COEF);
    static unsigned long long attr_set = 0;             // > 64 KiB of dynamic LDS needs the attribute once per kernel and device
    if constexpr (R == 1) {
        if (!dc_ensure_lds(&attr_set, reinterpret_cast<const void*>(&tile_unit_kernel<R, P, BODY>), 160 * 1024, "tiled apply")) return;
    } else {
        if (!dc_ensure_lds(&attr_set, reinterpret_cast<const void*>(&tile_fwd_kernel<R, P, BODY>), 160 * 1024, "tiled apply")) return;
    }
    const long units = (long)L.T * slabs;
    if constexpr (R == 1) {
        hipLaunchKernelGGL((tile_unit_kernel<R, P, BODY>), dim3((unsigned)units), dim3(P * 16), lds, s, L, plan, coef, nbr, slabs,
                           dc_option(DC_OPT_XCD_REMAP), body);
    } else {
    const long per_cu = std::max<long>(1, std::min<long>(2048 / (P * 16), (160 * 1024) / (long)lds));
    const long capacity = per_cu * device_cus();
    // units per workgroup: fill the resident slots once; at least 2 as soon as that still leaves a workgroup per CU (the second
    // unit's ids ride on the first one's pieces: -8 .. -11 % per launch at 32 x 1024 points, r03 sweep tools/archive/tile_upw.py)
    int upw = (int)((units + capacity - 1) / capacity);
    if (upw < 2 && units >= 2L * device_cus()) upw = 2;
    if (dc_option(7) > 0) upw = dc_option(7);           // option 7 (lab): forced
    hipLaunchKernelGGL((tile_fwd_kernel<R, P, BODY>), dim3((unsigned)((units + upw - 1) / upw)), dim3(P * 16), lds, s, L, plan, coef, nbr,
                       slabs, dc_option(DC_OPT_XCD_REMAP), upw, dc_stamp_next(1000 * BODY::TAG + C), body);
    }
}
template <int R, class BODY>
inline void launch(const DcTilePlan& L, const int* plan, const float* coef, const int* nbr, int C, BODY body, hipStream_t s) {
    if (L.P == 64) launch_one<R, 64, BODY>(L, plan, coef, nbr, C, body, s);
    else launch_one<R, 32, BODY>(L, plan, coef, nbr, C, body, s);
}

// 16-byte path only: channels a multiple of the slab, leading dimensions and bases 16-byte aligned
inline bool eligible(int C, std::initializer_list<long> lds, std::initializer_list<const void*> ptrs) {
    if (C <= 0 || C % CS) return false;
    for (long l : lds)
        if (l % 4) return false;
    for (const void* p : ptrs)
        if (reinterpret_cast<uintptr_t>(p) & 15) return false;
    return true;
}

}  // namespace dctile
