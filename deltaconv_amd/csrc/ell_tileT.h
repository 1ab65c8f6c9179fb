// Transposed ELL applies / max-aggregation backward from the transposed tile plan (tile_plan.h, second half): the unique
// SOURCE rows of a tile of P target points come into LDS once by LDS-DMA, every target walks its in-edge list (ascending
// edge id, the CSC order) over LDS only.  Backward of `SparseTensor @ dense` (torch_sparse autograd spmm with A^T; call
// sites /root/reference/deltaconv/nn/deltaconv.py:57,66, geometry/operators.py:27,33,40,43) and of
// torch_scatter.scatter(reduce='max') (nn/deltaconv.py:52,54).
//
// Why (round 3 counters, profiles/r03b_tile_lab_pmc.txt): the gather kernels (ell_stage.h: ell_T_kernel,
// aggregate.hip: knn_max_bwd_kernel) sat at 0.13 - 0.25 of the HBM roofline with the texture addresser busy only 22 k of
// 49 k cycles: a wavefront walks to the LARGEST in-degree of its four targets (mean 20, max 45+) through a serial chain of
// dependent L2 gathers.  Here
//   * the plan orders the targets of a tile by in-degree, so the four targets of a wavefront finish together;
//   * the rows of the in-edges' sources, the tile's edge records (local source index | slot) and the operator's coefficients
//     in tile order arrive by `global_load_lds_dwordx4` (contiguous 1-KiB pieces for the edge data);
//   * the walk reads LDS only: 16 lanes x 16 bytes per 256-byte row piece, conflict-free as in ell_tile.h.
// Same FMAs in the same (ascending edge id) order per target as the gather kernels: results are BIT-IDENTICAL
// (tests/test_gpu_tileT.py); no floating-point atomics anywhere.
// Mapping: workgroup = persistent over consecutive units (tile, 64-channel slab) like tile_fwd_kernel; thread = (target
// lane group, 4 channels).  Tiles whose unique sources exceed the LDS capacity or whose edge list exceeds ECAP read the
// excess from global memory (correct for any graph).
#pragma once
#include <algorithm>
#include <initializer_list>
#include "common.h"
#include "ell_math.h"
#include "ell_tile.h"
#include "tile_plan.h"

namespace dctileT {
using dcell::G2;
using dcell::Vec;
using dcell::vfma;
using dcell::vzero;
using dctile::CS;
using dctile::dma16;
using dctile::Geom;

// in-edge entries of a tile kept in LDS (mean P * k: 1280 at P = 64, k = 20; 1920 at k = 30)
template <int P> constexpr int ecap_entries() { return 2560; }

template <int FAMILY>
__device__ __forceinline__ void st16(float* p, const Vec<4>& a) { dc_store16<FAMILY>(p, *reinterpret_cast<const dc_f32x4*>(&a)); }

// capacity of a body's kernel: BODY::CAPT unique source rows / BODY::ECAPT in-edge entries in LDS, BODY::WGS waves per SIMD asked of
// the compiler (HIP's second __launch_bounds__ argument: the register budget).  Default: the forward kernels' 248 rows, 2560 entries, one workgroup per CU.
template <int R, int P, class BODY>
struct TGeom {
    static constexpr int NW = P * 16 / 64;
    static constexpr int CAP = BODY::CAPT, ECAP = BODY::ECAPT;
    static constexpr int CAPR = (CAP * R + 4 * NW - 1) / (4 * NW) * (4 * NW);
    static constexpr int RIT = CAPR / (4 * NW);
};
template <int R, int P, class BODY>
inline size_t lds_bytes() {
    using TG = TGeom<R, P, BODY>;
    return (size_t)TG::CAPR * 256 + (BODY::COEF ? (size_t)TG::ECAP * 8 : 0) + (size_t)TG::ECAP * 4 + (BODY::ARG ? (size_t)(TG::CAP + 8) * 64 : 0) + 16;
}

// BODY (per-thread accumulator object, copied from the kernel argument):
//   static constexpr bool COEF (operator coefficients used), ARG (slot words of the sources staged beside the rows);
//   static constexpr int NST (vector-memory store instructions finish() issues per wave);
//   const float* in; long ldj, hs;          piece h of source row i = in + i * ldj + h * hs
//   const unsigned char* arg; long lda;     (ARG only) slot bytes of source i at arg + i * lda
//   void init();
//   void step(int s, G2 g, const Vec<4>& p0, const Vec<4>& p1, unsigned aw);
//   void finish(long j, int c);
template <int R, int P, class BODY>
__global__ __launch_bounds__(P * 16, BODY::WGS) void tileT_kernel(DcTilePlanT L, const int* __restrict__ plan, const float* __restrict__ coefT,
                                                       int slabs, int remap, int upw, unsigned long long* stamp,
                                                       const BODY body0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using GM = TGeom<R, P, BODY>;
    constexpr int NW = GM::NW, CAP = GM::CAP, CAPR = GM::CAPR, RIT = GM::RIT, ECAP = GM::ECAP;
    const long units = (long)L.T * slabs;
    const long u0 = dc_xcd_block(remap) * upw, u1 = min(u0 + (long)upw, units);
    if (u0 >= u1) return;
    dc_stamp_in(stamp);
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    const int wave = tid >> 6, lane64 = tid & 63;
    const int k = L.k;
    float* rows = reinterpret_cast<float*>(smem);                         // [CAPR][64]
    char* cfb = smem + (size_t)CAPR * 256;                                // [ECAP] G2 (COEF)
    char* rcb = cfb + (BODY::COEF ? (size_t)ECAP * 8 : 0);                // [ECAP] u32 records
    char* agb = rcb + (size_t)ECAP * 4;                                   // [CAP + 8][16] u32 slot words (ARG)
    const unsigned* rec_g = reinterpret_cast<const unsigned*>(plan + L.o_rec);
    const int* edge_g = plan + L.o_edge;

    struct Ids { int rid[RIT]; int aid; int4 hdr, tg; };
    auto load_ids = [&](long tile, Ids& d) {
        const int* uq = plan + L.o_uniq + tile * L.UQ;
#pragma unroll
        for (int it = 0; it < RIT; ++it) {
            const int r = min((wave + it * NW) * 4 + (lane64 >> 4), L.UQ * R - 1);
            d.rid[it] = uq[r / R];
        }
        d.aid = BODY::ARG ? uq[min(wave * 16 + (lane64 >> 2), L.UQ - 1)] : 0;   // slot words: 4 lanes x 16 bytes per source
        d.hdr = reinterpret_cast<const int4*>(plan + L.o_hdr)[tile];
        d.tg = reinterpret_cast<const int4*>(plan + L.o_tg)[tile * P + grp];
    };
    Ids cur;
    load_ids(u0 / slabs, cur);
    BODY pbody = body0;                                                   // unit u-1: accumulators waiting to be stored
    long pj = -1;
    int pc = 0;
    for (long u = u0; u < u1; ++u) {
        const long tile = u / slabs;
        const int cb = (int)(u - tile * slabs) * CS;
#pragma unroll
        for (int it = 0; it < RIT; ++it) asm volatile("" : "+v"(cur.rid[it]));
        asm volatile("" : "+v"(cur.aid), "+v"(cur.hdr.x), "+v"(cur.hdr.y), "+v"(cur.hdr.z));
        asm volatile("" : "+v"(cur.tg.x), "+v"(cur.tg.y), "+v"(cur.tg.z));
        const int U = __builtin_amdgcn_readfirstlane(cur.hdr.x);          // block-uniform
        const int toff = __builtin_amdgcn_readfirstlane(cur.hdr.y);
        const int Et = __builtin_amdgcn_readfirstlane(cur.hdr.z);
        const int UL = min(U, CAP), nrow = UL * R;
        const int EL = min((Et + 3) & ~3, ECAP);                          // entries staged in LDS (whole 16-byte units)
        Ids nxt = cur;
        if (u + 1 < u1 && (u + 1) / slabs != tile) load_ids((u + 1) / slabs, nxt);
        asm volatile("" ::: "memory");
        if (Et != 0) {
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
                const int r0 = (wave + it * NW) * 4;
                if (r0 < nrow) {
                    const int h = R == 2 ? (lane64 >> 4) & 1 : 0;
                    dma16(body0.in + (long)cur.rid[it] * body0.ldj + h * body0.hs + cb + l16 * 4, rows + r0 * 64);
                }
            }
            if (BODY::ARG) {                                              // 64 slot bytes per source and slab
                if (wave * 16 < UL) dma16(body0.arg + (long)cur.aid * body0.lda + cb + (lane64 & 3) * 16, agb + wave * 1024);
                if (NW * 16 < CAP)                                        // (P = 32: 8 waves cover 128 sources per pass)
                    for (int c = wave + NW; c * 16 < UL; c += NW) {
                        const int a2 = plan[L.o_uniq + tile * L.UQ + min(c * 16 + (lane64 >> 2), L.UQ - 1)];
                        dma16(body0.arg + (long)a2 * body0.lda + cb + (lane64 & 3) * 16, agb + c * 1024);
                    }
            }
            {   // edge records (and coefficients) of the tile: contiguous -> whole 1-KiB pieces, tail lanes re-read the end
                const char* g = reinterpret_cast<const char*>(rec_g + toff);
                for (int c = wave; c * 1024 < EL * 4; c += NW) dma16(g + min(c * 1024 + lane64 * 16, EL * 4 - 16), rcb + c * 1024);
                if (BODY::COEF) {
                    const char* gc = reinterpret_cast<const char*>(coefT + 2L * toff);
                    for (int c = wave; c * 1024 < EL * 8; c += NW) dma16(gc + min(c * 1024 + lane64 * 16, EL * 8 - 16), cfb + c * 1024);
                }
            }
        }
        // unit u-1 leaves BEHIND this unit's pieces: finish() loads what it needs of the target's own rows (their latency
        // hides under the pieces' flight), then issues exactly BODY::NST store instructions per wave with a live target;
        // the wait below leaves those stores in flight (the vmcnt counter retires in order: see tile_fwd_kernel)
        asm volatile("" ::: "memory");
        const bool stores = __builtin_amdgcn_ballot_w64(pj >= 0) != 0;      // wave-uniform
        if (pj >= 0) pbody.finish(pj, pc);
        asm volatile("" ::: "memory");
        if (stores) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(BODY::NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const long j = cur.tg.x;
        const int c = cb + l16 * 4;
        pj = -1;
        if (j >= 0) {
            BODY body = body0;
            body.init();
            const int eo = cur.tg.y, dg = cur.tg.z;
            const unsigned* rc = reinterpret_cast<const unsigned*>(rcb);
            const G2* cf = reinterpret_cast<const G2*>(cfb);
            const unsigned* aw = reinterpret_cast<const unsigned*>(agb);
            if (U <= CAP && Et <= ECAP) {
#pragma unroll 4
                for (int r = 0; r < dg; ++r) {
                    const unsigned rec = rc[eo + r];
                    const int l = (int)(rec & 0xffffu);
                    const Vec<4> p0 = *reinterpret_cast<const Vec<4>*>(rows + (l * R) * 64 + l16 * 4);
                    const Vec<4> p1 = R == 2 ? *reinterpret_cast<const Vec<4>*>(rows + (l * R + R - 1) * 64 + l16 * 4) : p0;
                    body.step((int)(rec >> 16), BODY::COEF ? cf[eo + r] : G2{0.f, 0.f}, p0, p1, BODY::ARG ? aw[l * 16 + l16] : 0u);
                }
            } else {                                                      // rare: everything that is not in LDS by id
#pragma unroll 1
                for (int r = 0; r < dg; ++r) {
                    const unsigned rec = rec_g[toff + eo + r];
                    const int l = (int)(rec & 0xffffu);
                    const G2 g2 = BODY::COEF ? reinterpret_cast<const G2*>(coefT)[toff + eo + r] : G2{0.f, 0.f};
                    Vec<4> p0, p1;
                    unsigned a = 0u;
                    if (l < UL) {
                        p0 = *reinterpret_cast<const Vec<4>*>(rows + (l * R) * 64 + l16 * 4);
                        p1 = R == 2 ? *reinterpret_cast<const Vec<4>*>(rows + (l * R + R - 1) * 64 + l16 * 4) : p0;
                        if (BODY::ARG) a = aw[l * 16 + l16];
                    } else {
                        const long i = edge_g[toff + eo + r] / k;
                        const float* g = body.in + i * body.ldj + c;
                        p0 = dcell::vload<4>(g);
                        p1 = R == 2 ? dcell::vload<4>(g + body.hs) : p0;
                        if (BODY::ARG) a = *reinterpret_cast<const unsigned*>(body.arg + i * body.lda + c);
                    }
                    body.step((int)(rec >> 16), g2, p0, p1, a);
                }
            }
            pbody = body; pj = j; pc = c;
        }
        if (u + 1 < u1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur = nxt;
    }
    if (pj >= 0) pbody.finish(pj, pc);
    dc_stamp_out(stamp);
}

// ---- bodies (the accumulators of ell_math.h: GradT, GradTSum, DivT, DivCurlNormT, HodgeT, KnnMaxT) ----------------------
// finish() runs one unit late, behind the next unit's pieces: what it loads of the target's own rows (old values of an
// accumulating output, the other gradient terms, v and d|v| of the norm backward) travels under the pieces' flight.
constexpr int ST = DC_ST_ELL;   // store family of the transposed applies (common.h: plain stores unless the mask says otherwise)

// grad^T : dx[j] (+)= sum_e G[e,0] dy[2i] + G[e,1] dy[2i+1]; SUM: out[j] = a[j] (+ b[j]) + the sum (dc_apply_grad_T_sum)
template <bool SUM>
struct GradTB {
    static constexpr int TAG = 13;                     // dc_stamp_tag kind
    static constexpr int CAPT = 248, ECAPT = 2560, WGS = 1;
    static constexpr bool COEF = true, ARG = false;
    static constexpr int NST = 1;
    const float* in; long ldj, hs; const unsigned char* arg; long lda;
    const float* a; long lda_; const float* b; long ldb; float* out; long ldo; int accumulate;
    Vec<4> acc;
    __device__ void init() { acc = vzero<4>(); }
    __device__ void step(int, G2 g, const Vec<4>& yu, const Vec<4>& yv, unsigned) { vfma<4>(acc, g.a, yu); vfma<4>(acc, g.b, yv); }
    __device__ void finish(long j, int c) {
        if (SUM) {
            Vec<4> o = dcell::vload<4>(a + j * lda_ + c);
            if (b) {
                const Vec<4> ob = dcell::vload<4>(b + j * ldb + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) o.v[q] += ob.v[q];              // (a + b) + grad^T dy: the order of GradTSum
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) o.v[q] += acc.v[q];
            st16<ST>(out + j * ldo + c, o);
        } else if (accumulate) {
            Vec<4> o = dcell::vload<4>(out + j * ldo + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) o.v[q] += acc.v[q];
            st16<ST>(out + j * ldo + c, o);
        } else {
            st16<ST>(out + j * ldo + c, acc);
        }
    }
};
// div^T : dv[2j+a] (+)= sum_e D[e,a] dy[i]
struct DivTB {
    static constexpr int TAG = 15;                     // dc_stamp_tag kind
    static constexpr int CAPT = 248, ECAPT = 2560, WGS = 1;
    static constexpr bool COEF = true, ARG = false;
    static constexpr int NST = 2;
    const float* in; long ldj, hs; const unsigned char* arg; long lda;
    float* dv; long ldv; int accumulate;
    Vec<4> au, av;
    __device__ void init() { au = vzero<4>(); av = vzero<4>(); }
    __device__ void step(int, G2 d, const Vec<4>& g, const Vec<4>&, unsigned) { vfma<4>(au, d.a, g); vfma<4>(av, d.b, g); }
    __device__ void finish(long j, int c) {
        if (accumulate) {
            const Vec<4> ou = dcell::vload<4>(dv + (2 * j) * ldv + c), ov = dcell::vload<4>(dv + (2 * j + 1) * ldv + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) { au.v[q] = ou.v[q] + au.v[q]; av.v[q] = ov.v[q] + av.v[q]; }
        }
        st16<ST>(dv + (2 * j) * ldv + c, au);
        st16<ST>(dv + (2 * j + 1) * ldv + c, av);
    }
};
// backward of [div v | curl v | norm v] (ell_math.h: DivCurlNormT): pieces = dout[i, 0:C], dout[i, C:2C]
struct DivCurlNormTB {
    static constexpr int TAG = 11;                     // dc_stamp_tag kind
    static constexpr int CAPT = 248, ECAPT = 2560, WGS = 1;
    static constexpr bool COEF = true, ARG = false;
    static constexpr int NST = 2;
    const float* in; long ldj, hs; const unsigned char* arg; long lda;
    const float* v; long ldv; float* dv; long lddv; int accumulate; int C;
    Vec<4> au, av;
    __device__ void init() { au = vzero<4>(); av = vzero<4>(); }
    __device__ void step(int, G2 d, const Vec<4>& dd, const Vec<4>& dcu, unsigned) {
        vfma<4>(au, d.a, dd);
        vfma<4>(au, -d.b, dcu);
        vfma<4>(av, d.b, dd);
        vfma<4>(av, d.a, dcu);
    }
    __device__ void finish(long j, int c) {
        const Vec<4> dn = dcell::vload<4>(in + j * ldj + 2 * C + c);
        const Vec<4> vu = dcell::vload<4>(v + (2 * j) * ldv + c), vv = dcell::vload<4>(v + (2 * j + 1) * ldv + c);
        Vec<4> ou = vzero<4>(), ov = vzero<4>();
        if (accumulate) { ou = dcell::vload<4>(dv + (2 * j) * lddv + c); ov = dcell::vload<4>(dv + (2 * j + 1) * lddv + c); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float nrm = sqrtf(fmaf(vu.v[q], vu.v[q], vv.v[q] * vv.v[q]));
            const float sc = nrm > 0.f ? dn.v[q] / nrm : 0.f;            // subgradient 0 at |v| = 0 (as torch)
            au.v[q] = fmaf(sc, vu.v[q], au.v[q]);
            av.v[q] = fmaf(sc, vv.v[q], av.v[q]);
        }
        if (accumulate)
#pragma unroll
            for (int q = 0; q < 4; ++q) { au.v[q] = ou.v[q] + au.v[q]; av.v[q] = ov.v[q] + av.v[q]; }
        st16<ST>(dv + (2 * j) * lddv + c, au);
        st16<ST>(dv + (2 * j + 1) * lddv + c, av);
    }
};
// backward of the Hodge-Laplacian apply (HodgeT): pieces = dh[2i], dh[2i+1]
struct HodgeTB {
    static constexpr int TAG = 12;                     // dc_stamp_tag kind
    static constexpr int CAPT = 248, ECAPT = 2560, WGS = 1;
    static constexpr bool COEF = true, ARG = false;
    static constexpr int NST = 2;
    const float* in; long ldj, hs; const unsigned char* arg; long lda;
    float* ddc; long ldd; int accumulate; int C;
    Vec<4> ad, ac;
    __device__ void init() { ad = vzero<4>(); ac = vzero<4>(); }
    __device__ void step(int, G2 g, const Vec<4>& hu, const Vec<4>& hv, unsigned) {
        vfma<4>(ad, -g.a, hu);
        vfma<4>(ad, -g.b, hv);
        vfma<4>(ac, g.b, hu);
        vfma<4>(ac, -g.a, hv);
    }
    __device__ void finish(long j, int c) {
        if (accumulate) {
            const Vec<4> od = dcell::vload<4>(ddc + j * ldd + c), oc = dcell::vload<4>(ddc + j * ldd + C + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) { ad.v[q] = od.v[q] + ad.v[q]; ac.v[q] = oc.v[q] + ac.v[q]; }
        }
        st16<ST>(ddc + j * ldd + c, ad);
        st16<ST>(ddc + j * ldd + C + c, ac);
    }
};
// max-aggregation backward (KnnMaxT): dh[j,c] (+)= sum over in-edges (i,s) with arg[i,c] == s of dout[i,c]
struct KnnMaxTB {
    static constexpr int TAG = 14;                     // dc_stamp_tag kind
#if defined(DC_KNNMAXT_2WG) && DC_KNNMAXT_2WG
    static constexpr int CAPT = 192, ECAPT = 1536, WGS = 8;   // lab: 66 KB of LDS, 8 waves per SIMD = <= 64 registers: two workgroups per CU
#else
    static constexpr int CAPT = 248, ECAPT = 2560, WGS = 1;
#endif
    static constexpr bool COEF = false, ARG = true;
    static constexpr int NST = 1;
    const float* in; long ldj, hs; const unsigned char* arg; long lda;
    float* dh; long ldh; int accumulate;
    Vec<4> acc;
    __device__ void init() { acc = vzero<4>(); }
    __device__ void step(int s, G2, const Vec<4>& g, const Vec<4>&, unsigned w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc.v[q] += ((w >> (8 * q)) & 0xffu) == (unsigned)s ? g.v[q] : 0.f;
    }
    __device__ void finish(long j, int c) {
        if (accumulate) {
            const Vec<4> o = dcell::vload<4>(dh + j * ldh + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc.v[q] = o.v[q] + acc.v[q];
        }
        st16<ST>(dh + j * ldh + c, acc);
    }
};

template <int R, int P, class BODY>
inline void launch_one(const DcTilePlanT& L, const int* plan, const float* coefT, int C, BODY body, hipStream_t s) {
    const int slabs = C / CS;
    const size_t lds = lds_bytes<R, P, BODY>();
    static unsigned long long attr_set = 0;             // > 64 KiB of dynamic LDS: once per kernel and device (common.h)
    if (!dc_ensure_lds(&attr_set, reinterpret_cast<const void*>(&tileT_kernel<R, P, BODY>), 160 * 1024, "tiled transposed apply")) return;
    const long units = (long)L.T * slabs;
    const long per_cu = std::max<long>(1, std::min<long>(2048 / (P * 16), (160 * 1024) / (long)lds));
    const long capacity = per_cu * dctile::device_cus();
    int upw = (int)((units + capacity - 1) / capacity);
    if (upw < 2 && units >= 2L * dctile::device_cus() && per_cu < 2) upw = 2;   // (two workgroups per CU overlap by themselves)
    if (dc_option(7) > 0) upw = dc_option(7);
    hipLaunchKernelGGL((tileT_kernel<R, P, BODY>), dim3((unsigned)((units + upw - 1) / upw)), dim3(P * 16), lds, s, L, plan, coefT, slabs,
                       dc_option(DC_OPT_XCD_REMAP), upw, dc_stamp_next(1000 * BODY::TAG + C), body);
}
template <int R, class BODY>
inline void launch(const DcTilePlanT& L, const int* plan, const float* coefT, int C, BODY body, hipStream_t s) {
    if (L.P == 64) launch_one<R, 64, BODY>(L, plan, coefT, C, body, s);
    else launch_one<R, 32, BODY>(L, plan, coefT, C, body, s);
}

}  // namespace dctileT
