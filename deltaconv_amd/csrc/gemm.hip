// Dense per-point GEMMs of the scalar / vector MLP stream on the matrix cores, LDS-staged.  fp32 tensors in and out, fp32
// accumulation; two multiply paths over the same tiles, staging, prologues and epilogues:
//   * split products (X3, the default on whole 128-column tiles): every fp32 fragment is cut in registers into three bfloat16
//     planes and each accumulator gets six v_mfma_f32_32x32x16_bf16 per 16-deep k-step -- error against fp64 at or below the
//     exact chain's, 1.24-1.31 x its speed (comment at split_pair / the X3 loop below);
//   * the exact chain described next (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain): ragged shapes, 64-column tiles,
//     option DC_OPT_GEMM_EXACT.
//
//   forward          Y[M,N]  = X[M,K] W[N,K]^T          (every Linear(no bias): /root/reference/deltaconv/nn/mlp.py:9,15)
//   input gradient   dX[M,K] (+)= dY[M,N] W[N,K]        (ATen mm in the autograd of the same lines)
// M = points (32768..65536), N, K = features (64..1024): tall and skinny, so the row-major activations are
// streamed once and the small weight matrix stays in L2.  (The weight gradient dW = dY^T X has the long
// dimension as its reduction and lives in gemm_tn.hip.)
//
// Mapping (CDNA4): v_mfma_f32_32x32x2_f32 -- exact fp32, lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31].  A workgroup = 4 waves (2 x 2) on a BM x BN tile of the output, K walked in tiles of 32
// through double-buffered LDS (one barrier per tile; the next tile's global loads are in flight while the
// current one is multiplied):
//   * an operand whose reduction index is contiguous in memory (X, dY, and W in the forward product) is staged
//     as [rows][32 + 4 pad] and read back with ONE ds_read_b128 per four k-steps: the lane half h = l>>5 takes
//     k = 8j + 4h + t for t = 0..3, i.e. MFMA step t pairs columns {8j + t, 8j + 4 + t} -- a fixed permutation of
//     the reduction order, applied to both operands alike (pad 4: the 16 rows of a read group land on 16
//     different 16-byte bank slots);
//   * W in the input-gradient product has the reduction index as its ROW index: staged as [32][BN] and read with
//     conflict-free ds_read_b32 at the same permuted k.
// Global loads are 16 bytes per lane, whole 128-byte row segments per 8 lanes (full cache lines through the
// texture path once, fragments come from LDS).  Accumulators: (BM/64) x (BN/64) tiles of 32 x 32 per wave.
// Output: the MFMA C/D layout (lane = column, registers = rows) would be 64 dword stores per lane, each touching
// two 128-byte row pieces -- store-issue bound.  Each wave instead transposes its tile through LDS (the operand
// buffers are dead by then) and writes whole rows with 16-byte stores (r02a: the dword epilogue cost ~20 % of a
// 128 x 128 x 512 workgroup's time).
// The weight gradient dW[M,N] = dY[R,M]^T X[R,N] (both operands reduction-major, R = points) runs through the same
// kernel with the reduction split over row slabs (grid.y) into per-slab partial tiles, summed in slab order by
// gemm_tn_reduce_kernel (gemm_tn.hip): deterministic, no atomics.
//
// Fused epilogues (forward): the per-column sum / sum of squares of the tile (BatchNorm statistics of the
// Linear output, nn/nonlin.py:24-35), or of the per-point vector norms of an interleaved (P_c, Q_c) output
// (VectorNonLin statistics, nn/nonlin.py:63-79) are reduced in fp64 inside the workgroup and written as one
// partial per (column, row tile); the ordered final stage of colreduce.h turns them into scale / shift.  The
// separate statistics pass over the [M, N] output (one full read) disappears.  Deterministic: fixed order.
//
// Bound: MFMA.  Exact chain: 157 TFLOP/s fp32; per 32-deep K tile a 128 x 128 workgroup issues 64 MFMAs per wave (4096
// cycles) against 32 KB of global loads (8 B/clk/CU) and 16 ds_read_b128 per wave.  Split products: 48 bf16 MFMAs per wave
// (1536 cycles of a 2.5 PFLOP/s pipe that runs power-limited at ~1.65 GHz) + 288 split instructions on the VALU.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "nn_math.h"
#include "colreduce.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef DC_X3_PLAIN
#define DC_X3_PLAIN 2          // split-product loop of the plain products: 1 = simple, 2 = pipelined (two plane sets)
#endif
#ifndef DC_X3_PRO
#define DC_X3_PRO 1            // ... of the BatchNorm-backward prologue variants (no registers for a second plane set)
#endif
#ifdef DC_LAB_STAMPS
__device__ unsigned long long dc_lab_stamps[8192 * 8];
#define DC_STAMP(n) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192) dc_lab_stamps[blockIdx.x * 8 + (n)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DC_STAMP(n) do { } while (0)
#endif
constexpr int BK = 32;         // reduction tile
constexpr int LDK = BK + 4;    // row stride (floats) of a k-contiguous operand tile in LDS
constexpr int NT = 256;        // threads per workgroup (4 waves, 2 x 2)

enum { EPI_NONE = 0, EPI_COLSTATS = 1, EPI_VNSTATS = 2 /* interleaved (P_c, Q_c) columns */, EPI_VNSTATS0 = 3 /* plain [2n, co] */ };
enum { B_NK = 0 /* W[N,K]: forward */, B_KN = 1 /* reduction-major: input gradient, weight gradient */ };
enum { A_MK = 0 /* X[M,K]: reduction index contiguous */, A_KM = 1 /* dY[R,M]: reduction-major (weight gradient) */ };

struct GemmP {
    const float* A; long lda;
    const float* B; long ldb;
    float* C; long ldc;
    long M; int N; long K;
    int tiles_n, remap, accumulate;
    long k_per_slab, slab_stride;          // split reduction: blockIdx.y = slab, C += slab * slab_stride
    // BatchNorm-backward prologue (PRO): the A operand is dh = bn_act_backward(dy, h), formed while staging:
    //   dz = dy * (c_sc h + c_sh > 0 ? 1 : slope);   dh = c_g dz + c_a h + c_b     (dc_bn_act_backward_reduce)
    const float* A2; long lda2;            // h, same shape / layout as A (= dy)
    const float* pc; int pcn;              // packed per-column coefficients [5][pcn]: c_sc, c_sh, c_g, c_a, c_b
    float slope;
    double* part; int chunks, stat_cols;   // statistics partials [2][stat_cols][chunks]
    int stagger, resident;                 // first-round phase shift (shader cycles) of every second workgroup of a CU
    // BP kernels: the B operand (a weight matrix) arrives PRE-SPLIT: three bf16 planes [rows = output columns][reduction index]
    // (row stride ldb elements, plane stride bps elements), written once per step by dc_presplit_weights -- no B staging, no B
    // tile in LDS, no B split in the K loop: every wave loads the plane fragments of its columns straight from L2
    const unsigned short* Bp; long bps;
};

// Fast-path load: buffer_load through a descriptor built from the wave-uniform tile origin (SGPRs), a wave-uniform
// byte offset (soff: which of the thread's loads) and ONE 32-bit per-thread byte offset per operand (voff) -- no 64-bit
// per-load address registers and no VALU address arithmetic in the K loop.  num_records = 2^31: no range clipping.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 uload4(const float* ubase, unsigned voff, unsigned soff = 0) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ubase), 0, 0x7FFFFFFF, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(f32x4, v);
}

// Guarded forms (ragged shapes): the SAME loads with the per-thread offset of an out-of-range element replaced by
// 2^31 -- beyond num_records, so the hardware range check returns 0.0 without a memory access and the K loop stays
// branch-free.  Vector form: all four elements in or out together (extents and leading dimensions multiples of 4,
// 16-byte aligned bases); scalar form: four dword loads with one validity each (anything else).
constexpr unsigned POISON = 0x80000000u;
__device__ __forceinline__ f32x4 gload4v(const float* ubase, unsigned voff, unsigned soff, bool ok) {
    return uload4(ubase, ok ? voff : POISON, soff);
}
__device__ __forceinline__ f32x4 gload4s(const float* ubase, unsigned voff, unsigned soff, bool ok_row, long col, long ncols) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ubase), 0, 0x7FFFFFFF, 0x00020000);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                   r, (int)((ok_row && col + e < ncols) ? voff + 4u * e : POISON), (int)soff, 0));
    return v;
}
// one operand piece in any mode: MODE 0 = no guards, 1 = guarded vector, 2 = guarded scalar
template <int MODE>
__device__ __forceinline__ f32x4 pload4(const float* ubase, unsigned voff, unsigned soff, bool ok_row, long col, long ncols) {
    if (MODE == 0) return uload4(ubase, voff, soff);
    if (MODE == 1) return gload4v(ubase, voff, soff, ok_row && col < ncols);
    return gload4s(ubase, voff, soff, ok_row, col, ncols);
}

__device__ __forceinline__ f32x4 bn_bwd_vec(const f32x4 dy, const f32x4 h, const f32x4 (&cf)[5], float slope) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float z = fmaf(cf[0][e], h[e], cf[1][e]);
        const float dz = dy[e] * (z > 0.f ? 1.f : slope);
        o[e] = fmaf(cf[2][e], dz, fmaf(cf[3][e], h[e], cf[4][e]));
    }
    return o;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter (vmcnt(0)),
// which would make every K tile wait for the global loads issued for the tiles AFTER the next one.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);      // nothing is scheduled across (the MFMAs are not memory operations)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ---- split products (X3): an fp32 value x is cut into three bfloat16 planes, x = hi + mid + lo up to 2^-25 |x|
// (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); both differences are exact in fp32), and the product
// of two fp32 operands is accumulated from the six partial products of weight >= 2^-16 (lo.hi, hi.lo, mid.mid, mid.hi,
// hi.mid, hi.hi) on the bf16 matrix pipe with fp32 accumulation: v_mfma_f32_32x32x16_bf16 retires 16x the
// multiply-adds per cycle of v_mfma_f32_32x32x2_f32, so six of them cost 6/16 of the exact chain.  The dropped terms
// (mid.lo, lo.mid, lo.lo) are below 2^-23 of |a||b|, and each instruction sums 16 products before the one rounding
// into the accumulator: measured error against fp64 is BELOW the fp32 chain's (profiles/r03o_bf16x3_lab.txt).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Planes { u32x4 h, m, l; };       // 8 bfloat16 each: element j of the MFMA operand = k index 8 (lane >> 5) + j
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xFFFF0000u);
    m = pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xFFFF0000u);
    l = pk_bf16(s0, s1);
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, Planes& o) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h, m, l;
        split_pair(x[2 * j], x[2 * j + 1], h, m, l);
        o.h[j] = h; o.m[j] = m; o.l[j] = l;
    }
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int BM, int BN, int AL, int BL, int MODE, int EPI, int PRO = 0, int X3 = 0, int BP = 0>
__device__ __forceinline__ void gemm_body(const GemmP& p, const unsigned bid, const unsigned gemm_blocks) {
    static_assert(!BP || (X3 == 2 && AL == A_MK && BL == B_NK && MODE == 0), "pre-split B planes: pipelined split loop, K-contiguous operands, whole tiles");
    constexpr bool FAST = MODE == 0;               // no guards anywhere (loads, statistics, stores)
    // guarded modes per operand: 1 = 16-byte loads, 2 = dword loads.  MODE 1: both vector, 2: both scalar, 3: A vector /
    // B scalar, 4: A scalar / B vector
    constexpr int MA = MODE == 0 ? 0 : (MODE == 1 || MODE == 3 ? 1 : 2), MB = MODE == 0 ? 0 : (MODE == 1 || MODE == 4 ? 1 : 2);       // 2 workgroups per CU: <= 256 registers per lane
    constexpr int WM = BM / 2, WN = BN / 2;        // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;      // 32 x 32 accumulators per wave
    constexpr int A_FL = AL == A_MK ? BM * LDK : BK * BM;
    constexpr int B_FL = BL == B_NK ? BN * LDK : BK * BN;
    constexpr int A_IT = BM / 32, B_IT = BP ? 0 : BN / 32;  // 16-byte loads per thread and K tile (BP: B is never staged)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][A_FL]
    float* Bs = smem + 2 * A_FL;       // [2][B_FL]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // Phase shift.  The two workgroups of a CU start together and stay in lockstep: both in the K loop (sharing the
    // matrix pipe, each at half speed -- one wave per SIMD already saturates it), then both in the epilogue / the next
    // tile's first loads (pipe idle).  Holding back the workgroup in the ODD wave slot (HW_ID.WAVE_ID: the second
    // workgroup placed on the CU) by about half a K loop in the first round puts one workgroup's epilogue under the
    // other's K loop for the rest of the launch.
    DC_STAMP(0);
    if (p.stagger > 0 && (long)blockIdx.y * gemm_blocks + bid < p.resident) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);      // HW_REG_HW_ID bits [3:0]
        if (slot & 1) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(16);
        }
    }
    long blk = bid;                                  // XCD-aware placement over the multiplying workgroups (common.h: dc_xcd_block)
    if (p.remap) {
        const long q = gemm_blocks >> 3, r = gemm_blocks & 7, xcd = blk & 7, idx = blk >> 3;
        blk = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    const long kbeg = (long)blockIdx.y * p.k_per_slab;
    const long kend = min(p.K, kbeg + p.k_per_slab);
    const long tm = blk / p.tiles_n;
    const int tn = (int)(blk - tm * p.tiles_n);
    const long m0 = tm * BM;
    const int n0 = tn * BN;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // Staging registers global -> LDS.  Two sets (plain GEMM): tile kt+2 is requested while tile kt is multiplied and
    // reaches LDS during tile kt+1, so a whole K tile of MFMAs (~2.6k cycles) covers the load latency before the first
    // s_waitcnt (r02o counters: with one set the waves sat in waitcnt/barrier 12 % of their cycles, the library 4 %).
    // One set where a second does not fit the 256-register budget (2 workgroups per CU) without spilling: the
    // prologue variants (h tile + coefficients) and the 128 x 128 tile with a reduction-major B operand.
    constexpr int STG = (X3 || PRO || (BM == 128 && BN == 128 && BL == B_KN)) ? 1 : 2;
    f32x4 sa[STG][A_IT], sb[STG][B_IT ? B_IT : 1];
    f32x4 sa2[PRO ? A_IT : 1], cf[5];          // prologue: h tile, per-column coefficients of this thread's 4 columns
    if (PRO && AL == A_KM) {                    // reduction-major A: the thread's columns never change
#pragma unroll
        for (int q = 0; q < 5; ++q)
            cf[q] = pload4<MODE == 0 ? 0 : 2>(p.pc + (long)q * p.pcn + m0, (unsigned)((tid % (BM / 4)) * 16), 0, true,
                                              m0 + (tid % (BM / 4)) * 4, p.M);
    }
    // fast path: per-thread byte offsets inside a K tile (one per operand) and the row step between two loads
    const unsigned voa = AL == A_MK ? (unsigned)(((tid >> 3) * p.lda + (tid & 7) * 4) * 4)
                                    : (unsigned)(((tid / (BM / 4)) * p.lda + (tid % (BM / 4)) * 4) * 4);
    const unsigned voa2 = AL == A_MK ? (unsigned)(((tid >> 3) * p.lda2 + (tid & 7) * 4) * 4)
                                     : (unsigned)(((tid / (BM / 4)) * p.lda2 + (tid % (BM / 4)) * 4) * 4);
    const unsigned vob = BL == B_NK ? (unsigned)(((tid >> 3) * p.ldb + (tid & 7) * 4) * 4)
                                    : (unsigned)(((tid / (BN / 4)) * p.ldb + (tid % (BN / 4)) * 4) * 4);
    constexpr int A_STEP = AL == A_MK ? NT / 8 : NT / (BM / 4);      // rows between the loads `it` and `it + 1`
    constexpr int B_STEP = BL == B_NK ? NT / 8 : NT / (BN / 4);
    // The K tile moves in PIECES (one 16-byte load / LDS store per thread, one fragment register per lane), so the
    // loop body below can place every memory instruction at a chosen MFMA.
    constexpr int NST = A_IT + B_IT;                                         // LDS stores per tile
    constexpr int NLD = NST + (PRO ? A_IT + (AL == A_MK ? 5 : 0) : 0);       // global loads per tile
    constexpr int NRP = TM + TN;                                             // fragment reads per set
    // global load n of the K tile at k0 -> staging set S.  n: A pieces, B pieces, then (prologue) h pieces, coefficients
    auto load_piece = [&](int n, long k0, auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
        // (row, col) of the piece's first element; validity only matters in the guarded modes
        if (n < A_IT || (PRO && n >= NST && n < NST + A_IT)) {
            const bool second = n >= NST;                      // the prologue's h tile: same shape / layout as A
            const int it = second ? n - NST : n, idx = tid + NT * it;
            const float* base = second ? p.A2 : p.A;
            const long ld = second ? p.lda2 : p.lda;
            const float* ua = AL == A_MK ? base + m0 * ld + k0 : base + k0 * ld + m0;              // wave-uniform
            const unsigned soff = (unsigned)(it * A_STEP * 4) * (unsigned)ld;
            const long row = AL == A_MK ? m0 + (idx >> 3) : k0 + idx / (BM / 4);
            const long col = AL == A_MK ? k0 + (idx & 7) * 4 : m0 + (idx % (BM / 4)) * 4;
            const f32x4 v = pload4<MA>(ua, second ? voa2 : voa, soff, row < (AL == A_MK ? p.M : kend), col,
                                         AL == A_MK ? kend : p.M);
            if (second) sa2[it] = v;
            else sa[S][it] = v;
        } else if (n < NST) {
            const int it = n - A_IT, idx = tid + NT * it;
            const float* ub = BL == B_NK ? p.B + (long)n0 * p.ldb + k0 : p.B + k0 * p.ldb + n0;
            const long row = BL == B_NK ? n0 + (idx >> 3) : k0 + idx / (BN / 4);
            const long col = BL == B_NK ? k0 + (idx & 7) * 4 : n0 + (idx % (BN / 4)) * 4;
            sb[S][it] = pload4<MB>(ub, vob, (unsigned)(it * B_STEP * 4) * (unsigned)p.ldb,
                                     row < (BL == B_NK ? (long)p.N : kend), col, BL == B_NK ? kend : (long)p.N);
        } else if (PRO && AL == A_MK) {         // K-contiguous A: the columns are the reduction index of this tile
            const int q = n - NST - A_IT;
            cf[q] = pload4<MODE == 0 ? 0 : 2>(p.pc + (long)q * p.pcn + k0, (unsigned)((tid & 7) * 16), 0, true,
                                              k0 + (tid & 7) * 4, kend);
        }
    };
    // LDS store n (A pieces, then B pieces) of staging set S -> buffer buf
    auto store_piece = [&](int n, int buf, auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
        if (n < A_IT) {
            const int it = n, idx = tid + NT * it;
            float* a = As + buf * A_FL;
            const f32x4 va = PRO ? bn_bwd_vec(sa[S][it], sa2[it], cf, p.slope) : sa[S][it];
            if (AL == A_MK)
                *reinterpret_cast<f32x4*>(a + (idx >> 3) * LDK + (idx & 7) * 4) = va;
            else
                *reinterpret_cast<f32x4*>(a + (idx / (BM / 4)) * BM + (idx % (BM / 4)) * 4) = va;
        } else {
            const int it = n - A_IT, idx = tid + NT * it;
            float* b = Bs + buf * B_FL;
            if (BL == B_NK)
                *reinterpret_cast<f32x4*>(b + (idx >> 3) * LDK + (idx & 7) * 4) = sb[S][it];
            else
                *reinterpret_cast<f32x4*>(b + (idx / (BN / 4)) * BN + (idx % (BN / 4)) * 4) = sb[S][it];
        }
    };
    // Fragments: lane (li, lh) holds k = 8 j + 4 lh + t (t = 0..3) of fragment set j for its row / column -- one
    // ds_read_b128 on a K-contiguous tile, four ds_read_b32 on a reduction-major one.  Two register sets.
    f32x4 fa[2][TM], fb[2][TN];
    auto read_piece = [&](int n, int buf, int j, f32x4 (&ra)[TM], f32x4 (&rb)[TN]) {
        if (n < TM) {
            const int i = n;
            const float* a = AL == A_MK ? As + buf * A_FL + (wm0 + li) * LDK + 4 * lh
                                        : As + buf * A_FL + (4 * lh) * BM + wm0 + li;
            if (AL == A_MK) {
                ra[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK + 8 * j);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) ra[i][t] = a[(8 * j + t) * BM + i * 32];
            }
        } else {
            const int jn = n - TM;
            const float* b = BL == B_NK ? Bs + buf * B_FL + (wn0 + li) * LDK + 4 * lh
                                        : Bs + buf * B_FL + (4 * lh) * BN + wn0 + li;
            if (BL == B_NK) {
                rb[jn] = *reinterpret_cast<const f32x4*>(b + jn * 32 * LDK + 8 * j);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) rb[jn][t] = b[(8 * j + t) * BN + jn * 32];
            }
        }
    };
    // MFMA s of a fragment set: k-step t = s / (TM TN), accumulator (i, jn) = the rest.  The first TM TN MFMAs touch
    // every fragment register of the set once.
    constexpr int NMF = 4 * TM * TN, LEAD = TM * TN, NFR = NMF - LEAD;
    auto mfma_one = [&](int s, const f32x4 (&ra)[TM], const f32x4 (&rb)[TN]) {
        const int t = s / LEAD, i = (s % LEAD) / TN, jn = s % TN;
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i][t], rb[jn][t], acc[i][jn], 0, 0, 0);
    };

    DC_STAMP(1);
    const int nk = (int)((kend - kbeg + BK - 1) / BK);
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, STG - 1>;
    if constexpr (X3 != 0) {
        // Split-product loop.  A K tile = two k-steps of 16; per k-step a lane reads 8 consecutive k of its row / column
        // per fragment as fp32 (the SAME LDS tiles and traffic as the exact loop), cuts them into planes in registers and
        // issues 6 MFMAs per accumulator.  One staging set; the two workgroups of a CU run in antiphase (one wave of a SIMD
        // splits on the VALU while the other feeds the matrix pipe).
        f32x4 qa[TM][2], qb[TN][2];
        Planes pa[TM], pb[TN];
        auto read_raw = [&](int buf, int ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (AL == A_MK) {
                    const float* a = As + buf * A_FL + (wm0 + 32 * i + li) * LDK + 16 * ks + 8 * lh;
                    qa[i][0] = *reinterpret_cast<const f32x4*>(a);
                    qa[i][1] = *reinterpret_cast<const f32x4*>(a + 4);
                } else {
                    const float* a = As + buf * A_FL + (16 * ks + 8 * lh) * BM + wm0 + 32 * i + li;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { qa[i][0][t] = a[t * BM]; qa[i][1][t] = a[(4 + t) * BM]; }
                }
            }
#pragma unroll
            for (int jn = 0; jn < (BP ? 0 : TN); ++jn) {
                if (BL == B_NK) {
                    const float* b = Bs + buf * B_FL + (wn0 + 32 * jn + li) * LDK + 16 * ks + 8 * lh;
                    qb[jn][0] = *reinterpret_cast<const f32x4*>(b);
                    qb[jn][1] = *reinterpret_cast<const f32x4*>(b + 4);
                } else {
                    const float* b = Bs + buf * B_FL + (16 * ks + 8 * lh) * BN + wn0 + 32 * jn + li;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { qb[jn][0][t] = b[t * BN]; qb[jn][1][t] = b[(4 + t) * BN]; }
                }
            }
        };
        auto split_all = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i) split8(qa[i][0], qa[i][1], pa[i]);
#pragma unroll
            for (int jn = 0; jn < (BP ? 0 : TN); ++jn) split8(qb[jn][0], qb[jn][1], pb[jn]);
        };
        // BP: plane fragments of the wave's columns for the k-step at reduction index k, straight from global memory (L2).
        // The planes are stored FRAGMENT-MAJOR (dc_presplit_weights): the 64 x 16 bytes that the lanes of a wavefront hold of
        // (32 columns, 16 reduction indices) are 1 KiB contiguous -- one fully coalesced load per plane (a row-major layout
        // costs 32 cache lines of 32 useful bytes per load: the first build ran 10-30 % SLOWER than the in-loop split on it).
        const __amdgpu_buffer_rsrc_t rbp = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(BP ? p.Bp : nullptr), 0,
                                                                              0x7FFFFFFF, 0x00020000);
        const unsigned vobp = (unsigned)(lane * 16);
        const unsigned ksteps = (unsigned)(p.ldb >> 4);                       // k-steps of the whole reduction (ldb = its length)
        auto load_bp = [&](int jn, Planes& o, long k) {
            const unsigned so = (((unsigned)((n0 + wn0) >> 5) + (unsigned)jn) * ksteps + (unsigned)(k >> 4)) * 1024u;
            o.h = __builtin_amdgcn_raw_buffer_load_b128(rbp, (int)vobp, (int)so, 0);
            o.m = __builtin_amdgcn_raw_buffer_load_b128(rbp, (int)vobp, (int)(so + (unsigned)(p.bps * 2)), 0);
            o.l = __builtin_amdgcn_raw_buffer_load_b128(rbp, (int)vobp, (int)(so + (unsigned)(p.bps * 4)), 0);
        };
        // smallest partial products first; consecutive MFMAs go to different accumulators
        auto mfma_all = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_bf16(pa[i].l, pb[jn].h, acc[i][jn]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_bf16(pa[i].h, pb[jn].l, acc[i][jn]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_bf16(pa[i].m, pb[jn].m, acc[i][jn]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_bf16(pa[i].m, pb[jn].h, acc[i][jn]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_bf16(pa[i].h, pb[jn].m, acc[i][jn]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_bf16(pa[i].h, pb[jn].h, acc[i][jn]);
        };
        if constexpr (X3 == 2) {
            // Pipelined form: two plane sets.  A k-step multiplies one set while the next k-step's fragments (read from
            // LDS a step earlier) are cut into the other, fragment by fragment -- 6 MFMAs, then the 36 VALU instructions
            // of one fragment, then the LDS reads that refill its fp32 registers for the step after -- so the split runs
            // in the shadow of the matrix pipe inside ONE wave.  The LDS ring is two tiles deep: tile kt+2 is stored into
            // the buffer of tile kt during tile kt (all of that buffer was read before the previous barrier).
            Planes pa1[TM], pb1[TN], pb2[TN];             // (pb2: third B set of the pre-split form)
            constexpr bool RING3 = BP && !PRO;
            constexpr int NF = TM + TN, NM = 6 * TM * TN, VPM = 36 * NF / NM;
            auto read_frag = [&](int f, int buf, int ks) {
                if (f < TM) {
                    const int i = f;
                    if (AL == A_MK) {
                        const float* a = As + buf * A_FL + (wm0 + 32 * i + li) * LDK + 16 * ks + 8 * lh;
                        qa[i][0] = *reinterpret_cast<const f32x4*>(a);
                        qa[i][1] = *reinterpret_cast<const f32x4*>(a + 4);
                    } else {
                        const float* a = As + buf * A_FL + (16 * ks + 8 * lh) * BM + wm0 + 32 * i + li;
#pragma unroll
                        for (int t = 0; t < 4; ++t) { qa[i][0][t] = a[t * BM]; qa[i][1][t] = a[(4 + t) * BM]; }
                    }
                } else {
                    const int jn = f - TM;
                    if (BL == B_NK) {
                        const float* b = Bs + buf * B_FL + (wn0 + 32 * jn + li) * LDK + 16 * ks + 8 * lh;
                        qb[jn][0] = *reinterpret_cast<const f32x4*>(b);
                        qb[jn][1] = *reinterpret_cast<const f32x4*>(b + 4);
                    } else {
                        const float* b = Bs + buf * B_FL + (16 * ks + 8 * lh) * BN + wn0 + 32 * jn + li;
#pragma unroll
                        for (int t = 0; t < 4; ++t) { qb[jn][0][t] = b[t * BN]; qb[jn][1][t] = b[(4 + t) * BN]; }
                    }
                }
            };
            // one k-step: multiply the planes (ca, cb); cut the fragments in flight into (na, nb); refill them from
            // (rbuf, rks) -- unconditionally: past the last tile they read stale LDS and nothing consumes the result
            // `stage(f)`: the share of fragment group f in moving the operand ring on (LDS stores of tile kt+2, global loads of
            // tile kt+3): issued between the MFMAs of the k-step instead of in a burst behind it, where the matrix pipe idled
            // (BP: `kb` = reduction index of the k-step being PREPARED; its B plane fragments are requested in the first slots
            //  of this step -- a whole k-step of MFMAs covers their L2 round trip -- the A fragments are split in the last ones)
            auto step = [&](const Planes (&ca)[TM], const Planes (&cb)[TN], Planes (&na)[TM], Planes (&nb)[TN], int rbuf, int rks,
                            auto stage, long kb = 0) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int fr = BP ? (f < TN ? TM + f : f - TN) : f;           // fragment handled in slot f
#pragma unroll
                    for (int g = f * NM / NF; g < (f + 1) * NM / NF; ++g) {
                        const int pr = g / (TM * TN), r = g % (TM * TN), i = r / TN, jn = r % TN;
                        // smallest partial products first: l.h  h.l  m.m  m.h  h.m  h.h
                        const u32x4 a = pr == 0 ? ca[i].l : (pr == 2 || pr == 3 ? ca[i].m : ca[i].h);
                        const u32x4 b = pr == 1 ? cb[jn].l : (pr == 2 || pr == 4 ? cb[jn].m : cb[jn].h);
                        acc[i][jn] = mfma_bf16(a, b, acc[i][jn]);
                    }
#if defined(DC_LAB_X3) && DC_LAB_X3 >= 3                 /* lab: no split at all (wrong results, timing only) */
                    if (f < TM) { na[f].h = __builtin_bit_cast(u32x4, qa[f][0]); na[f].m = __builtin_bit_cast(u32x4, qa[f][1]); na[f].l = na[f].h; }
                    else { nb[f - TM].h = __builtin_bit_cast(u32x4, qb[f - TM][0]); nb[f - TM].m = __builtin_bit_cast(u32x4, qb[f - TM][1]); nb[f - TM].l = nb[f - TM].h; }
                    read_frag(f, rbuf, rks);
#elif defined(DC_LAB_X3) && DC_LAB_X3 >= 1               /* lab: the B fragments are not split (1) and not even read (2) */
                    if (f < TM) split8(qa[f][0], qa[f][1], na[f]);
                    else { nb[f - TM].h = __builtin_bit_cast(u32x4, qb[f - TM][0]); nb[f - TM].m = __builtin_bit_cast(u32x4, qb[f - TM][1]); nb[f - TM].l = nb[f - TM].h; }
                    if (DC_LAB_X3 == 1 || f < TM) read_frag(f, rbuf, rks);
#else
                    if (fr < TM) {
                        split8(qa[fr][0], qa[fr][1], na[fr]);
                        read_frag(fr, rbuf, rks);
                    } else if (BP) {
                        load_bp(fr - TM, nb[fr - TM], kb);
                    } else {
                        split8(qb[fr - TM][0], qb[fr - TM][1], nb[fr - TM]);
                        read_frag(fr, rbuf, rks);
                    }
#endif
                    stage(f);
                }
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    if (BP && f < TN) {                                      // a B slot of the pre-split form: three plane loads
                        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
#pragma unroll
                        for (int g = 0; g < NM / NF; ++g) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    } else {
#pragma unroll
                        for (int g = 0; g < NM / NF; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, BP ? 36 * NF / NM : VPM, 0); // its share of the split
                        }
                    }
                    if (PRO) {                                                               // LDS stores of the group
                        if (f == 0) __builtin_amdgcn_sched_group_barrier(0x200, NST, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x200, (NST + NF - 1) / NF, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x020, (NLD + NF - 1) / NF, 0);     // global loads of the group
                }
            };
#pragma unroll
            for (int n = 0; n < NLD; ++n) load_piece(n, kbeg, S0{});
#pragma unroll
            for (int n = 0; n < NST; ++n) store_piece(n, 0, S0{});
            if (nk > 1) {
#pragma unroll
                for (int n = 0; n < NLD; ++n) load_piece(n, kbeg + BK, S0{});
#pragma unroll
                for (int n = 0; n < NST; ++n) store_piece(n, 1, S0{});
            }
            if (nk > 2) {
#pragma unroll
                for (int n = 0; n < NLD; ++n) load_piece(n, kbeg + 2 * BK, S0{});
            }
            lds_barrier();
            DC_STAMP(7);
            read_raw(0, 0);
            split_all();
            if (BP) {                                       // the planes of the first two k-steps
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) load_bp(jn, pb[jn], kbeg);
                if (RING3) {
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) load_bp(jn, pb1[jn], kbeg + 16);
                }
            }
            read_raw(0, 1);
            // one K tile = two k-steps.  ba / bb: the B plane sets the two steps multiply; la / lb: the sets they PREPARE -- in the
            // in-loop form the set of the next step (cut from the fragments in flight), with pre-split planes (BP) the set of the
            // step after next, loaded from global memory: a ring of three sets, two k-steps of MFMAs over every L2 round trip
            auto tile_steps = [&](int kt, const Planes (&ba)[TN], const Planes (&bb)[TN], Planes (&la)[TN], Planes (&lb)[TN]) {
                const int cur = kt & 1;
                // (unconditional: past the last tiles the stores refill a buffer nobody reads again and the loads re-read the last
                // tile -- no branches inside the pinned instruction stream)
                const long k3 = kbeg + (long)min(kt + 3, nk - 1) * BK;
                auto none = [&](int) {};
                auto stores_loads = [&](int f) {                           // (piece n is stored before its registers are reloaded)
                    if (PRO) {
                        // the prologue forms load more pieces (h tile, coefficients) than they store and every store reads the
                        // coefficients: ALL stores go in front of the first reload (per-slot shares would reload sa[1..] in slot 0
                        // before slot 1 stored them -- found by the pre-split-plane A/B at 64-row tiles, round 4)
                        if (f == 0) {
#pragma unroll
                            for (int n = 0; n < NST; ++n) store_piece(n, cur, S0{});
                        }
                    } else {
#pragma unroll
                        for (int n = f * NST / NF; n < (f + 1) * NST / NF; ++n) store_piece(n, cur, S0{});
                    }
#pragma unroll
                    for (int n = f * NLD / NF; n < (f + 1) * NLD / NF; ++n) load_piece(n, k3, S0{});
                };
                // (the weight gradient's reduction-major form spills with the pieces inside the stream: it keeps the burst behind it)
                constexpr bool INSIDE = AL == A_MK;
                // (BP: the planes requested now belong to tile kt + 1 -- ring of three sets -- or, where the prologue's registers
                //  leave room for two sets only, to the next k-step; past the end the last tile again)
                const long kn = kbeg + (long)min(kt + 1, nk - 1) * BK;
                const long kb1 = RING3 ? kn : kbeg + (long)kt * BK + 16, kb2 = RING3 ? kn + 16 : kn;
                __builtin_amdgcn_sched_barrier(0);
                step(pa, ba, pa1, la, cur ^ 1, 0, none, kb1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (INSIDE) {
                    step(pa1, bb, pa, lb, cur ^ 1, 1, stores_loads, kb2);
                } else {
                    step(pa1, bb, pa, lb, cur ^ 1, 1, none);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 2 < nk) {
#pragma unroll
                        for (int n = 0; n < NST; ++n) store_piece(n, cur, S0{});
                    }
                    if (kt + 3 < nk) {
#pragma unroll
                        for (int n = 0; n < NLD; ++n) load_piece(n, kbeg + (long)(kt + 3) * BK, S0{});
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                lds_barrier();
            };
            if constexpr (RING3) {
                int kt = 0;
                for (; kt + 3 <= nk; kt += 3) {
                    tile_steps(kt, pb, pb1, pb2, pb);
                    tile_steps(kt + 1, pb2, pb, pb1, pb2);
                    tile_steps(kt + 2, pb1, pb2, pb, pb1);
                }
                if (kt < nk) tile_steps(kt, pb, pb1, pb2, pb);
                if (kt + 1 < nk) tile_steps(kt + 1, pb2, pb, pb1, pb2);
            } else {
                for (int kt = 0; kt < nk; ++kt) tile_steps(kt, pb, pb1, pb1, pb);
            }
        } else {
#pragma unroll
        for (int n = 0; n < NLD; ++n) load_piece(n, kbeg, S0{});
#pragma unroll
        for (int n = 0; n < NST; ++n) store_piece(n, 0, S0{});
        if (nk > 1) {
#pragma unroll
            for (int n = 0; n < NLD; ++n) load_piece(n, kbeg + BK, S0{});
        }
        lds_barrier();
        read_raw(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            split_all();
            read_raw(cur, 1);
            mfma_all();
            split_all();
            if (kt + 1 < nk) {             // tile kt+1 (requested a whole tile ago) -> the other buffer
#pragma unroll
                for (int n = 0; n < NST; ++n) store_piece(n, cur ^ 1, S0{});
            }
            if (kt + 2 < nk) {
#pragma unroll
                for (int n = 0; n < NLD; ++n) load_piece(n, kbeg + (long)(kt + 2) * BK, S0{});
            }
            lds_barrier();                 // buffer `cur` is consumed (its fragments are in registers), cur ^ 1 is written
            if (kt + 1 < nk) read_raw(cur ^ 1, 0);
            mfma_all();
        }
        }   // simple / pipelined split loop
    } else {
#pragma unroll
    for (int n = 0; n < NLD; ++n) load_piece(n, kbeg, S0{});
#pragma unroll
    for (int n = 0; n < NST; ++n) store_piece(n, 0, S0{});
    if (STG == 2 && nk > 1) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) load_piece(n, kbeg + BK, S1{});
    }
    lds_barrier();
#pragma unroll
    for (int n = 0; n < NRP; ++n) read_piece(n, 0, 0, fa[0], fb[0]);
    // One K tile = 4 fragment sets of NMF MFMAs each; the instruction order is written out and pinned (a
    // sched_barrier after every MFMA: the scheduler otherwise re-sorts the reads and MFMAs by its own latency model):
    //   set 0: [LEAD MFMAs] [all reads of set 1] then one MFMA per slot, the global loads of tile kt+STG in the first slots
    //   set 1: [LEAD MFMAs] reads of set 2 in the first slots
    //   set 2: [LEAD MFMAs] reads of set 3 in the first slots, the LDS stores of tile kt+1 in the LAST slots
    //   barrier;  [reads of set 0 of tile kt+1]  set 3
    // Every set starts with the LEAD MFMAs that touch all of its fragment registers: the waits on its reads (issued a
    // whole set earlier) then sit in front of anything newly queued on the in-order LDS counter.  The stores come as
    // late as possible (most time for the loads to land), the barrier before the LAST set, whose operands are in
    // registers by then.  par_tag: parity of kt (static: it selects the staging set); more_tag: tile kt+1 exists;
    // load_tag: tile kt+STG exists.
    constexpr int CL = (NLD + NFR - 1) / NFR, CR = (NRP + NFR - 1) / NFR, CW = (NST + NFR - 1) / NFR;   // pieces per slot
    constexpr int WSLOTS = (NST + CW - 1) / CW;                              // slots that carry stores
    auto tile_body = [&](int kt, auto par_tag, auto more_tag, auto load_tag) {
        constexpr bool MORE = decltype(more_tag)::value, LOAD = decltype(load_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;
        using SL = std::integral_constant<int, STG == 2 ? PAR : 0>;          // set the new loads land in
        using SS = std::integral_constant<int, STG == 2 ? PAR ^ 1 : 0>;      // set holding tile kt+1
        const int cur = kt & 1;
        const long knext = kbeg + (long)(kt + STG) * BK;
#pragma unroll
        for (int s = 0; s < NMF; ++s) {                                      // ---- set 0
            if (s == LEAD) {
#pragma unroll
                for (int n = 0; n < NRP; ++n) read_piece(n, cur, 1, fa[1], fb[1]);
            }
            if (LOAD && s >= LEAD) {
#pragma unroll
                for (int c = 0; c < CL; ++c)
                    if ((s - LEAD) * CL + c < NLD) load_piece((s - LEAD) * CL + c, knext, SL{});
            }
            mfma_one(s, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NMF; ++s) {                                      // ---- set 1
            if (s >= LEAD) {
#pragma unroll
                for (int c = 0; c < CR; ++c)
                    if ((s - LEAD) * CR + c < NRP) read_piece((s - LEAD) * CR + c, cur, 2, fa[0], fb[0]);
            }
            mfma_one(s, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NMF; ++s) {                                      // ---- set 2
            if (s >= LEAD) {
#pragma unroll
                for (int c = 0; c < CR; ++c)
                    if ((s - LEAD) * CR + c < NRP) read_piece((s - LEAD) * CR + c, cur, 3, fa[1], fb[1]);
                if (MORE && s - LEAD >= NFR - WSLOTS) {
#pragma unroll
                    for (int c = 0; c < CW; ++c)
                        if ((s - LEAD - (NFR - WSLOTS)) * CW + c < NST)
                            store_piece((s - LEAD - (NFR - WSLOTS)) * CW + c, cur ^ 1, SS{});
                }
            }
            mfma_one(s, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();     // tile `cur` is consumed (its last fragments are in registers), tile cur^1 is written
        if (MORE) {
#pragma unroll
            for (int n = 0; n < NRP; ++n) read_piece(n, cur ^ 1, 0, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NMF; ++s) {                                      // ---- set 3
            mfma_one(s, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using T = std::true_type;
        using F = std::false_type;
        if (STG == 2) {
            int kt = 0;
            for (; kt + 3 < nk; kt += 2) {          // steady state, two tiles per trip (static staging sets)
                tile_body(kt, P0{}, T{}, T{});
                tile_body(kt + 1, P1{}, T{}, T{});
            }
            const int rem = nk - kt;                // 1 .. 3 tiles left, kt even
            if (rem == 3) {
                tile_body(kt, P0{}, T{}, T{});
                tile_body(kt + 1, P1{}, T{}, F{});
                tile_body(kt + 2, P0{}, F{}, F{});
            } else if (rem == 2) {
                tile_body(kt, P0{}, T{}, F{});
                tile_body(kt + 1, P1{}, F{}, F{});
            } else {
                tile_body(kt, P0{}, F{}, F{});
            }
        } else {
            for (int kt = 0; kt + 1 < nk; ++kt) tile_body(kt, P0{}, T{}, T{});
            tile_body(nk - 1, P0{}, F{}, F{});
        }
    }
    }   // exact / split loop
    DC_STAMP(2);
    __syncthreads();                  // (the staging below reuses the operand buffers)

    // ---- statistics epilogue (rows beyond M hold zeros and add nothing)
    if (EPI != EPI_NONE) {
        constexpr int SC = EPI == EPI_VNSTATS ? BN / 2 : BN;        // statistic columns of the tile
        double* sst = reinterpret_cast<double*>(smem);               // [2 quantities][2 wave rows][SC]; tiles are dead
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (EPI == EPI_COLSTATS) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const double x = (double)acc[i][jn][q];
                        s0 += x;
                        s1 += x * x;
                    }
                } else if (EPI == EPI_VNSTATS0) {   // rows (2r, 2r+1) = (u, v) components = registers (q, q+1), q even
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const double nr = (double)dcnn::vn_norm(acc[i][jn][q], acc[i][jn][q + 1]);
                        s0 += nr;
                        s1 += nr * nr;
                    }
                } else {   // rows (2r, 2r+1) = registers (q, q+1), q even; columns (2c, 2c+1) = (P_c, Q_c) = lanes (l, l+1)
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const float pu = acc[i][jn][q], pv = acc[i][jn][q + 1];
                        const float qu = __shfl_xor(pu, 1, 64), qv = __shfl_xor(pv, 1, 64);
                        float yu, yv;
                        dcnn::vn_combine(pu, qu, pv, qv, yu, yv);
                        const double nr = (double)dcnn::vn_norm(yu, yv);
                        s0 += nr;
                        s1 += nr * nr;
                    }
                }
            }
            s0 += __shfl_xor(s0, 32, 64);
            s1 += __shfl_xor(s1, 32, 64);
            if (lh == 0) {
                if (EPI == EPI_COLSTATS || EPI == EPI_VNSTATS0) {
                    const int c = wn0 + jn * 32 + li;
                    sst[(0 * 2 + (wave >> 1)) * SC + c] = s0;
                    sst[(1 * 2 + (wave >> 1)) * SC + c] = s1;
                } else if ((li & 1) == 0) {
                    const int c = (wn0 + jn * 32 + li) >> 1;
                    sst[(0 * 2 + (wave >> 1)) * SC + c] = s0;
                    sst[(1 * 2 + (wave >> 1)) * SC + c] = s1;
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * SC; idx += NT) {
            const int q = idx / SC, c = idx - q * SC;
            const int gc = (EPI == EPI_VNSTATS ? n0 / 2 : n0) + c;
            if (gc < p.stat_cols)
                p.part[((long)q * p.stat_cols + gc) * p.chunks + tm] = sst[(q * 2 + 0) * SC + c] + sst[(q * 2 + 1) * SC + c];
        }
        __syncthreads();   // sst is about to be overwritten by the output staging
    }

    DC_STAMP(3);
    // ---- store: each wave transposes its WM x WN tile through LDS and writes whole rows, 16 bytes per lane.
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5).  The staging row
    // stride is exactly WN floats: the ds_read_b128 lane groups then cover all 16 slots of a 256-byte bank row.
    float* stg = smem + wave * (WM * WN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                stg[(i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh) * WN + jn * 32 + li] = acc[i][jn][q];
    __syncthreads();
    float* cbase = p.C + (long)blockIdx.y * p.slab_stride;
    constexpr int RL = WN / 4;                         // lanes per output row
#pragma unroll
    for (int it = 0; it < WM * WN / 4 / 64; ++it) {
        const int idx = it * 64 + lane;
        const int r = idx / RL, c4 = (idx % RL) * 4;
        const long row = m0 + wm0 + r;
        const int col = n0 + wn0 + c4;
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * WN + c4);
        float* dst = cbase + row * p.ldc + col;
        if (FAST) {
            if (p.accumulate) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
                v += o;
            }
            dc_store16<DC_ST_GEMM>(dst, v);
        } else if (row < p.M) {
            if (col + 3 < p.N && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                if (p.accumulate) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
                    v += o;
                }
                dc_store16<DC_ST_GEMM>(dst, v);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < p.N) dst[e] = p.accumulate ? dst[e] + v[e] : v[e];
            }
        }
    }
    DC_STAMP(4);
#ifdef DC_LAB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DC_STAMP(5);
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192) dc_lab_stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((16 << 11) | 4);
#endif
}

template <int BM, int BN, int AL, int BL, int MODE, int EPI, int PRO = 0, int X3 = 0, int BP = 0>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(GemmP p) {
    gemm_body<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>(p, blockIdx.x, gridDim.x);
}
// Two products of the SAME instantiation in one launch (round 6: the max-aggregation stream's and the s_mlp's last products of a
// DeltaConv layer are independent): the first nb0 workgroups multiply p0, the rest p1 -- the body is the single kernel's.
template <int BM, int BN, int AL, int BL, int MODE, int EPI, int PRO = 0, int X3 = 0, int BP = 0>
__global__ __launch_bounds__(NT, 2) void gemm_pair_kernel(GemmP p0, GemmP p1, unsigned nb0) {
    if (blockIdx.x < nb0) gemm_body<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>(p0, blockIdx.x, nb0);
    else gemm_body<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>(p1, blockIdx.x - nb0, gridDim.x - nb0);
}

struct Tile { int bm, bn; };

Tile pick_tile(long M, int N, int K, int tile) {
    switch (tile) {
        case 1: return {128, 128};
        case 2: return {128, 64};
        case 3: return {64, 64};
        case 4: return {64, 128};
        default: break;
    }
    // column tile: 128 when it divides N (or N is large and ragged anyway), else 64 when THAT divides N -- a tile that
    // divides the problem runs the unguarded fast path (e.g. N = 192: 3 x 64 instead of 2 x 128 with guards)
    int bn = N > 64 ? 128 : 64;
    if (N % 128 != 0 && N % 64 == 0) bn = 64;
    long tn = (N + bn - 1) / bn;
    // row tile: enough workgroups for 256 CUs x 2; 64-row tiles for short problems
    int bm = ((M + 127) / 128) * tn >= 384 ? 128 : 64;
    if (M % bm != 0 && M % 64 == 0) bm = 64;
    // few rows (the per-rank steps of 8-GPU strong scaling: 4096 points): 64 x 128 tiles leave 64 workgroups for 256 CUs at
    // N = 128 -- 64-column tiles while the launch has fewer than one workgroup per CU (round 6, profiles/r06_labs.txt item 9)
    if (bm == 64 && bn == 128 && N % 64 == 0 && ((M + 63) / 64) * tn < 256 && dc_option(DC_OPT_WIDE_TILES) == 0) bn = 64;
    return {bm, bn};
}

size_t lds_bytes(int bm, int bn, int al, int bl) {
    const size_t a = al == A_MK ? (size_t)bm * LDK : (size_t)BK * bm, b = bl == B_NK ? (size_t)bn * LDK : (size_t)BK * bn;
    return std::max(2 * (a + b), (size_t)bm * bn) * sizeof(float);      // operand ring | output staging
}

// ---- deferred products (common.h: deferred finalisers): a forward product with the statistics epilogue whose caller asked for it
// waits in this queue until dc_finalisers_end; two queued products of the same instantiation run as ONE launch (gemm_pair_kernel).
typedef void (*GemmSingleFn)(const GemmP&, unsigned, size_t, hipStream_t);
typedef void (*GemmPairFn)(const GemmP&, const GemmP&, unsigned, unsigned, size_t, hipStream_t);
struct GemmPending { GemmP p; unsigned blocks; size_t lds; GemmSingleFn single; GemmPairFn pair; };
static thread_local GemmPending g_gemm_q[DC_FIN_MAX];
static thread_local int g_gemm_n = 0;

template <int BM, int BN, int AL, int BL, int MODE, int EPI, int PRO, int X3, int BP>
void gemm_single_thunk(const GemmP& p, unsigned blocks, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((gemm_kernel<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>), dim3(blocks, 1), dim3(NT), lds, s, p);
}
template <int BM, int BN, int AL, int BL, int MODE, int EPI, int PRO, int X3, int BP>
void gemm_pair_thunk(const GemmP& p0, const GemmP& p1, unsigned b0, unsigned b1, size_t lds, hipStream_t s) {
    static unsigned long long configured = 0;
    if (!dc_ensure_lds(&configured, reinterpret_cast<const void*>(&gemm_pair_kernel<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>), lds, "dense product pair"))
        return;
    GemmP q1 = p1;
    q1.stagger = 0;                         // (the phase shift is a first-round device: the second product's workgroups come later)
    hipLaunchKernelGGL((gemm_pair_kernel<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>), dim3(b0 + b1, 1), dim3(NT), lds, s, p0, q1, b0);
}

template <int BM, int BN, int AL, int BL, int MODE, int EPI, int PRO, int X3, int BP = 0>
void launch_one(const GemmP& p, long tiles_m, int slabs, hipStream_t s) {
    static unsigned long long configured = 0;
    const size_t lds = lds_bytes(BM, BN, AL, BL);
    if (!dc_ensure_lds(&configured, reinterpret_cast<const void*>(&gemm_kernel<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>), lds, "dense product")) return;
    // the pairable kind: forward product from the step's weight planes with the BatchNorm-statistics epilogue (whole tiles)
    constexpr bool PAIRABLE = EPI == EPI_COLSTATS && BP == 1 && MODE == 0 && PRO == 0 && AL == A_MK && BL == B_NK;
    if (dc_gemm_take_request()) {           // (always consumed, whatever the kind)
        if constexpr (PAIRABLE) {
            if (slabs == 1 && g_gemm_n < DC_FIN_MAX) {
                GemmPending& e = g_gemm_q[g_gemm_n++];
                e.p = p; e.blocks = (unsigned)(tiles_m * p.tiles_n); e.lds = lds;
                e.single = &gemm_single_thunk<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>;
                e.pair = &gemm_pair_thunk<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>;
                return;
            }
        }
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, AL, BL, MODE, EPI, PRO, X3, BP>),
                       dim3((unsigned)(tiles_m * p.tiles_n), (unsigned)slabs), dim3(NT), lds, s, p);
}

template <int AL, int BL, int MODE, int EPI, int PRO, int X3 = 0, int BP = 0>
void launch_tile(const GemmP& p, Tile t, long tiles_m, int slabs, hipStream_t s) {
    if (t.bm == 128 && t.bn == 128) launch_one<128, 128, AL, BL, MODE, EPI, PRO, X3, BP>(p, tiles_m, slabs, s);
    else if (t.bm == 128) launch_one<128, 64, AL, BL, MODE, EPI, PRO, X3, BP>(p, tiles_m, slabs, s);
    else if (t.bn == 128) launch_one<64, 128, AL, BL, MODE, EPI, PRO, X3, BP>(p, tiles_m, slabs, s);
    else launch_one<64, 64, AL, BL, MODE, EPI, PRO, X3, BP>(p, tiles_m, slabs, s);
}

// The unguarded path (whole tiles: every hot shape of the reference models) multiplies through the split products where
// they win (r03 A/B, profiles/r03_x3_ab.txt): output tiles 128 columns wide (forward / input gradient; a 64-wide tile
// splits 12 VALU instructions per MFMA against 6) and 128 x 128 tiles of the weight gradient, whose reduction-major
// fragments cost 8 ds_read_b32 each.  Option DC_OPT_GEMM_EXACT = 1 forces the exact fp32 chain everywhere; ragged
// shapes always run it.
template <int AL, int BL, int EPI, int PRO = 0>
void launch_fast(const GemmP& p, Tile t, long tiles_m, int slabs, int mode, hipStream_t s) {
    // (DC_OPT_GEMM_EXACT = 2, lab: split on every unguarded tile)
    const bool split = mode == 0 && (dc_option(DC_OPT_GEMM_EXACT) == 2 ||
                                     (dc_option(DC_OPT_GEMM_EXACT) == 0 && t.bn == 128 && (AL == A_MK || t.bm == 128)));
    // Pre-split weight planes (round 4): the B split and its LDS traffic are gone, so 64-column tiles (6 split instructions per
    // MFMA like today's 128-column ones) take the split path too.  Plain products only (the BatchNorm-prologue loaders keep the
    // simple loop: their registers).
    if constexpr (AL == A_MK && BL == B_NK) {
        if (p.Bp && mode == 0 && dc_option(DC_OPT_GEMM_EXACT) == 0) {
            launch_tile<AL, BL, 0, EPI, PRO, 2, 1>(p, t, tiles_m, slabs, s);
            return;
        }
    }
    if (split) launch_tile<AL, BL, 0, EPI, PRO, (PRO ? DC_X3_PRO : DC_X3_PLAIN)>(p, t, tiles_m, slabs, s);
    else if (mode == 0) launch_tile<AL, BL, 0, EPI, PRO>(p, t, tiles_m, slabs, s);
    else if (mode == 1) launch_tile<AL, BL, 1, EPI, PRO>(p, t, tiles_m, slabs, s);
    else if (mode == 2) launch_tile<AL, BL, 2, EPI, PRO>(p, t, tiles_m, slabs, s);
    else if (mode == 3) launch_tile<AL, BL, 3, EPI, PRO>(p, t, tiles_m, slabs, s);
    else launch_tile<AL, BL, 4, EPI, PRO>(p, t, tiles_m, slabs, s);
}

// operand-load mode: 0 = no guards (whole tiles, aligned: every hot shape of the reference models); otherwise per
// operand guarded 16-byte loads (ragged tile edges, but its extents / leading dimension multiples of 4 and an aligned
// base) or guarded dword loads (anything): 1 = both vector, 2 = both scalar, 3 = A vector / B scalar, 4 = A scalar / B vector
int load_mode(bool whole_tiles, bool a4, bool b4) {
    if (whole_tiles && a4 && b4) return 0;
    return a4 && b4 ? 1 : (a4 ? 3 : (b4 ? 4 : 2));
}

// first-round phase shift (see the kernel): only where two 128 x 128 workgroups share a CU and the launch has a second
// round to profit from it.  Option DC_OPT_GEMM_STAGGER = percent of the estimated K-loop time of one workgroup running
// alone (0: the default, 50; negative: off).  r03 sweep (profiles/r03_x3_stagger.txt): 25 .. 100 are equivalent.
void set_stagger(GemmP& p, Tile t, long workgroups, long k_per_wg) {
    p.stagger = 0;
    p.resident = 512;
    int pct = dc_option(DC_OPT_GEMM_STAGGER);          // 0: default (50), < 0: off
    if (pct == 0) pct = 50;
    if (pct < 0 || t.bm != 128 || t.bn != 128 || workgroups < 2 * p.resident) return;
    const long per_tile = dc_option(DC_OPT_GEMM_EXACT) ? 4200 : 1250;       // shader cycles per K tile, alone on the CU
    p.stagger = (int)std::min<long>((k_per_wg + BK - 1) / BK * per_tile * pct / 100, 400000);
}

bool al16p(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// bl: operand layout of W; epi: fused statistics; part/chunks filled by the caller for epi != 0
struct Prologue { const float* h; long ldh; const float* coefs; int ncoef; float slope; };
// Hint for the NEXT product enqueued by this thread: the bf16 planes of its weight operand (dc_presplit_weights).
//   transposed == 0: planes of W[N, K] as stored (forward products: the rows are the output columns);
//   transposed != 0: planes of W^T [K, N] (input-gradient products dX = dY W: their output columns are W's columns).
// ld = reduction length of the planes (K resp. N of W), plane_elems = distance between two planes = N * K.  Consumed (and cleared) by the next
// dc_linear_* call whatever its outcome; ignored where the planes do not apply (ragged shapes, exact chain, prologue forms) and when
// that product's weight operand is not `src` (a hint left behind by a call that failed before its product was enqueued).
struct BHint { const void* src; const unsigned short* p; long stride, ld; int transposed; };
thread_local BHint g_bhint = {nullptr, nullptr, 0, 0, 0};

int run_gemm(const char* name, int bl, int epi, const float* A, long lda, const float* B, long ldb, long M, int N, int K,
             float* C, long ldc, int accumulate, int tile, double* part, int stat_cols, hipStream_t s,
             const Prologue* pro = nullptr) {
    BHint hint = g_bhint;
    g_bhint = BHint{nullptr, nullptr, 0, 0, 0};
    if (hint.src != static_cast<const void*>(B)) hint.p = nullptr;
    // (round 4 lab: N = q * 128 + 64 output columns as two launches -- q * 128 columns on 128-column split tiles + 64 on 64-column
    //  tiles -- is 27 us faster in isolation for 32768 x 448 x 1024 and 8 us SLOWER inside the step, same-box A/B: not used)
    const Tile t = pick_tile(M, N, K, tile);
    const long tiles_m = (M + t.bm - 1) / t.bm;
    GemmP p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.tiles_n = (N + t.bn - 1) / t.bn;
    p.remap = dc_option(DC_OPT_XCD_REMAP);
    p.accumulate = accumulate;
    p.k_per_slab = K; p.slab_stride = 0;
    p.part = part; p.chunks = (int)tiles_m; p.stat_cols = stat_cols;
    p.A2 = nullptr; p.lda2 = 0; p.pc = nullptr; p.pcn = 0; p.slope = 0.f;
    p.Bp = nullptr; p.bps = 0;
    set_stagger(p, t, tiles_m * p.tiles_n, (long)K);
    if (lda >= (1 << 21) || ldb >= (1 << 21) || (pro && pro->ldh >= (1 << 21))) {     // 32-bit in-tile byte offsets
        dc_set_error("%s: leading dimension above 2^21 elements", name);
        return DC_ERR_ARG;
    }
    const bool whole = M % t.bm == 0 && N % t.bn == 0 && K % BK == 0 && ldc % 4 == 0 && al16p(C);
    bool a4 = K % 4 == 0 && lda % 4 == 0 && al16p(A);                       // 4-vectors run along K
    const bool b4 = (bl == B_NK ? K % 4 == 0 : N % 4 == 0) && ldb % 4 == 0 && al16p(B);
    if (pro) {
        p.A2 = pro->h; p.lda2 = pro->ldh; p.pc = pro->coefs; p.pcn = pro->ncoef; p.slope = pro->slope;
        a4 = a4 && pro->ldh % 4 == 0 && al16p(pro->h);
        const int fast = load_mode(whole && al16p(pro->coefs) && pro->ncoef % 4 == 0, a4, b4);
        if (hint.p && hint.transposed && hint.ld == K && N % 32 == 0 && K % 16 == 0 && al16p(hint.p) && hint.stride == (long)N * K &&
            hint.stride * 6 < (1L << 31) && load_mode(whole && al16p(pro->coefs) && pro->ncoef % 4 == 0, a4, true) == 0 &&
            dc_option(DC_OPT_GEMM_EXACT) == 0 && dc_option(DC_OPT_NO_PLANES) == 0) {
            p.Bp = hint.p; p.bps = hint.stride; p.ldb = hint.ld; p.B = nullptr;       // forward form over W^T's planes
            launch_fast<A_MK, B_NK, EPI_NONE, 1>(p, t, tiles_m, 1, 0, s);
        } else {
            launch_fast<A_MK, B_KN, EPI_NONE, 1>(p, t, tiles_m, 1, fast, s);
        }
    } else if (hint.p && hint.transposed == (bl == B_KN) && hint.ld == K && N % 32 == 0 && K % 16 == 0 && al16p(hint.p) &&
               hint.stride == (long)N * K && hint.stride * 6 < (1L << 31) &&
               load_mode(whole, a4, true) == 0 && dc_option(DC_OPT_GEMM_EXACT) == 0 && dc_option(DC_OPT_NO_PLANES) == 0) {
        // pre-split planes: forward products as they are; input-gradient products in the forward form over W^T's planes
        p.Bp = hint.p; p.bps = hint.stride; p.ldb = hint.ld; p.B = nullptr;
        if (epi == EPI_COLSTATS) launch_fast<A_MK, B_NK, EPI_COLSTATS>(p, t, tiles_m, 1, 0, s);
        else if (epi == EPI_VNSTATS) launch_fast<A_MK, B_NK, EPI_VNSTATS>(p, t, tiles_m, 1, 0, s);
        else if (epi == EPI_VNSTATS0) launch_fast<A_MK, B_NK, EPI_VNSTATS0>(p, t, tiles_m, 1, 0, s);
        else launch_fast<A_MK, B_NK, EPI_NONE>(p, t, tiles_m, 1, 0, s);
    } else if (bl == B_NK) {
        const int fast = load_mode(whole, a4, b4);
        if (epi == EPI_COLSTATS) launch_fast<A_MK, B_NK, EPI_COLSTATS>(p, t, tiles_m, 1, fast, s);
        else if (epi == EPI_VNSTATS) launch_fast<A_MK, B_NK, EPI_VNSTATS>(p, t, tiles_m, 1, fast, s);
        else if (epi == EPI_VNSTATS0) launch_fast<A_MK, B_NK, EPI_VNSTATS0>(p, t, tiles_m, 1, fast, s);
        else launch_fast<A_MK, B_NK, EPI_NONE>(p, t, tiles_m, 1, fast, s);
    } else {
        launch_fast<A_MK, B_KN, EPI_NONE>(p, t, tiles_m, 1, load_mode(whole, a4, b4), s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dc_set_error("%s: %s", name, hipGetErrorString(e));
        return DC_ERR_LAUNCH;
    }
    return DC_OK;
}

int chunks_for(long M, int N, int K, int tile) {
    const Tile t = pick_tile(M, N, K, tile);
    return (int)((M + t.bm - 1) / t.bm);
}

}  // namespace

void dc_gemm_flush(void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < g_gemm_n;) {
        GemmPending& a = g_gemm_q[i];
        if (i + 1 < g_gemm_n && g_gemm_q[i + 1].pair == a.pair && g_gemm_q[i + 1].lds == a.lds) {
            a.pair(a.p, g_gemm_q[i + 1].p, a.blocks, g_gemm_q[i + 1].blocks, a.lds, s);
            i += 2;
        } else {
            a.single(a.p, a.blocks, a.lds, s);
            i += 1;
        }
    }
    g_gemm_n = 0;
}
void dc_gemm_discard() { g_gemm_n = 0; }

// ---- weight gradient through the LDS-staged kernel (called by dc_gemm_tn, gemm_tn.hip) ----------------------------
// partial[slab][M][N] = A[rows of the slab, M]^T B[rows of the slab, N];  returns the number of slabs.
struct DcTnPlan { int bm, bn, slabs; long rows_per_slab; };
int dc_tn_lds_launch(const float* A, long lda, const float* B, long ldb, long R, int M, int N, float* partial, hipStream_t s,
                     const float* h = nullptr, long ldh = 0, const float* coefs = nullptr, float slope = 0.f);
DcTnPlan dc_tn_lds_plan(long R, int M, int N) {
    DcTnPlan pl;
    // (round 4 lab: N - 64 output columns on 128 x 128 split tiles + 64 on 64 x 64 tiles over the same slabs was no faster than
    //  128 x 64 exact tiles for the [1024, 448] embedding weight, 237 vs 237 us: removed, profiles/r04_gemm_planes_ab.txt)
    // r02r sweep (profiles/r02r_tn_sweep.txt): outputs below 64K elements run best on 64 x 64 tiles (more workgroups
    // per slab, shorter epilogues; 4 fit a CU) with ~768 workgroups; larger ones on 128 x 128 tiles with at most 512
    // workgroups = ONE resident wave of 2 per CU (576 cost +25 %: a second, nearly empty round); <= 128 slabs
    // (r03 lab, profiles/r03_tn_x3_lab.txt: with the split products a 128 x 256 / 256 x 128 / 128 x 384 output runs 10-13 % faster on
    // 128 x 128 tiles than on 64 x 64 ones -- the bound between the two plans moved from 64K to 32K outputs)
    // (round 6: few ROWS -- the per-rank steps of 8-GPU strong scaling, R <= 8192 -- take the small plan whatever the output: 128 x 128
    //  tiles over 128-row slabs are all prologue and epilogue; C5 per rank -3.4 %, profiles/r06_labs.txt item 9; switch 11 = 1: off)
    const bool small = (long)M * N < 32768 || (R <= 8192 && dc_option(DC_OPT_WIDE_TILES) == 0);
    pl.bm = (M > 64 && !small) ? 128 : 64;
    pl.bn = (N > 64 && !small) ? 128 : 64;
    if (M % pl.bm != 0 && M % 64 == 0) pl.bm = 64;        // a tile that divides the output runs the unguarded loads
    if (N % pl.bn != 0 && N % 64 == 0) pl.bn = 64;        // (e.g. N = 448 = 7 x 64)
    if (const int t = dc_option(DC_OPT_TN_TILE)) {
        pl.bm = (t == 1 || t == 2) ? 128 : 64;
        pl.bn = (t == 1 || t == 4) ? 128 : 64;
    }
    const long tiles = (long)((M + pl.bm - 1) / pl.bm) * ((N + pl.bn - 1) / pl.bn);
    long slabs = std::min<long>(128, std::max<long>(1, (small ? 768 : 512) / tiles));
    if (const int f = dc_option(DC_OPT_TN_SLABS)) slabs = f;
    long rps = (R + slabs - 1) / slabs;
    rps = std::max<long>((rps + BK - 1) / BK * BK, 4 * BK);
    pl.rows_per_slab = rps;
    pl.slabs = (int)((R + rps - 1) / rps);
    return pl;
}
int dc_tn_lds_launch(const float* A, long lda, const float* B, long ldb, long R, int M, int N, float* partial,
                     hipStream_t s, const float* h, long ldh, const float* coefs, float slope) {
    const DcTnPlan pl = dc_tn_lds_plan(R, M, N);
    const Tile t{pl.bm, pl.bn};
    const int ldpart = N;                                      // row stride of the partial tiles
    const long tiles_m = (M + t.bm - 1) / t.bm;
    GemmP p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = partial; p.ldc = ldpart;
    p.M = M; p.N = N; p.K = R;
    p.tiles_n = (N + t.bn - 1) / t.bn;
    p.remap = 0;                       // tiles x slabs: every workgroup streams its own rows, nothing to co-locate
    p.accumulate = 0;
    p.k_per_slab = pl.rows_per_slab; p.slab_stride = (long)M * ldpart;
    p.part = nullptr; p.chunks = 0; p.stat_cols = 0;
    p.A2 = h; p.lda2 = ldh; p.pc = coefs; p.pcn = M; p.slope = slope;
    set_stagger(p, t, tiles_m * p.tiles_n * pl.slabs, pl.rows_per_slab);
    if (lda >= (1 << 21) || ldb >= (1 << 21) || ldh >= (1 << 21)) return -1;      // 32-bit in-tile byte offsets
    const bool whole = M % t.bm == 0 && N % t.bn == 0 && R % BK == 0 && pl.rows_per_slab % BK == 0 && ldpart % 4 == 0 &&
                       al16p(partial);
    bool a4 = M % 4 == 0 && lda % 4 == 0 && al16p(A);                       // reduction-major: 4-vectors run along M / N
    const bool b4 = N % 4 == 0 && ldb % 4 == 0 && al16p(B);
    if (h) {
        a4 = a4 && ldh % 4 == 0 && al16p(h);
        launch_fast<A_KM, B_KN, EPI_NONE, 1>(p, t, tiles_m, pl.slabs, load_mode(whole && al16p(coefs), a4, b4), s);
    } else {
        launch_fast<A_KM, B_KN, EPI_NONE>(p, t, tiles_m, pl.slabs, load_mode(whole, a4, b4), s);
    }
    return pl.slabs;
}

// Y[M,N] (ldy) = X[M,K] (ldx) W[N,K]^T (ldw).  tile: 0 = automatic; 1..4 = 128x128, 128x64, 64x64, 64x128.
DC_EXPORT int dc_linear_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int32_t N,
                                int32_t K, float* Y, int64_t ldy, int32_t tile, void* stream) {
    DC_REQUIRE(X && W && Y, "dc_linear_forward: null pointer");
    DC_REQUIRE(M >= 0 && N >= 1 && K >= 1 && ldx >= K && ldw >= K && ldy >= N, "dc_linear_forward: bad size");
    if (M == 0) return DC_OK;
    return run_gemm("dc_linear_forward", B_NK, EPI_NONE, X, ldx, W, ldw, M, N, K, Y, ldy, 0, tile, nullptr, 0,
                    static_cast<hipStream_t>(stream));
}

// dX[M,K] (lddx) (+)= dY[M,N] (lddy) W[N,K] (ldw)
DC_EXPORT int dc_linear_backward_input(const float* dY, int64_t lddy, const float* W, int64_t ldw, int64_t M, int32_t N,
                                       int32_t K, float* dX, int64_t lddx, int32_t accumulate, int32_t tile,
                                       void* stream) {
    DC_REQUIRE(dY && W && dX, "dc_linear_backward_input: null pointer");
    DC_REQUIRE(M >= 0 && N >= 1 && K >= 1 && lddy >= N && ldw >= K && lddx >= K, "dc_linear_backward_input: bad size");
    if (M == 0) return DC_OK;
    // as a product: C[M, K] = A[M, N] B[N, K] -- reduction over N, B stored reduction-major
    return run_gemm("dc_linear_backward_input", B_KN, EPI_NONE, dY, lddy, W, ldw, M, K, N, dX, lddx, accumulate, tile,
                    nullptr, 0, static_cast<hipStream_t>(stream));
}

// dX[M,K] (lddx) (+)= dh[M,N] W[N,K] with dh = BatchNorm/activation backward of (dy, h) formed in the operand loader:
// the [M,N] tensor dh is never written (coefs: dc_bn_act_backward_reduce).
DC_EXPORT int dc_linear_bn_backward_input(const float* dy, int64_t lddy, const float* h, int64_t ldh, const float* coefs,
                                          float slope, const float* W, int64_t ldw, int64_t M, int32_t N, int32_t K,
                                          float* dX, int64_t lddx, int32_t accumulate, int32_t tile, void* stream) {
    DC_REQUIRE(dy && h && coefs && W && dX, "dc_linear_bn_backward_input: null pointer");
    DC_REQUIRE(M >= 0 && N >= 1 && K >= 1 && lddy >= N && ldh >= N && ldw >= K && lddx >= K,
               "dc_linear_bn_backward_input: bad size");
    if (M == 0) return DC_OK;
    const Prologue pro{h, (long)ldh, coefs, N, slope};
    return run_gemm("dc_linear_bn_backward_input", B_KN, EPI_NONE, dy, lddy, W, ldw, M, K, N, dX, lddx, accumulate, tile,
                    nullptr, 0, static_cast<hipStream_t>(stream), &pro);
}

DC_EXPORT size_t dc_linear_stats_workspace_bytes(int64_t M, int32_t N, int32_t K, int32_t tile) {
    return (size_t)chunks_for(M, N, K, tile) * 2 * (size_t)N * sizeof(double);
}

// Linear + BatchNorm batch statistics in one pass over the output: Y = X W^T and, from the tile sums of the GEMM
// epilogue, mean / invstd / scale = gamma * invstd / shift = beta - mean * scale (+ running statistics), exactly
// what dc_bn_stats computes from Y (nn/nonlin.py:24-35).
DC_EXPORT int dc_linear_bn_stats_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int32_t N,
                                         int32_t K, float* Y, int64_t ldy, const float* gamma, const float* beta,
                                         float eps, float momentum, float* running_mean, float* running_var,
                                         float* mean, float* invstd, float* scale, float* shift, int32_t tile,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(X && W && Y && mean && invstd && scale && shift, "dc_linear_bn_stats_forward: null pointer");
    DC_REQUIRE(M >= 1 && N >= 1 && K >= 1 && ldx >= K && ldw >= K && ldy >= N, "dc_linear_bn_stats_forward: bad size");
    if (!workspace || workspace_bytes < dc_linear_stats_workspace_bytes(M, N, K, tile)) {
        dc_set_error("dc_linear_bn_stats_forward: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(workspace);
    if (int rc = run_gemm("dc_linear_bn_stats_forward", B_NK, EPI_COLSTATS, X, ldx, W, ldw, M, N, K, Y, ldy, 0, tile, part,
                          N, s))
        return rc;
    const dccol::BnFin fin{(long)M, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
    dccol::finalise_or_defer(DC_FIN_BN, part, chunks_for(M, N, K, tile), N, fin, s);     // (queued when the caller asked: common.h)
    DC_CHECK_LAUNCH("dc_linear_bn_stats_forward");
    return DC_OK;
}

// Linear + VectorNonLin statistics (nn/nonlin.py:63-79).  interleaved != 0: PQ[2n, 2co] = V[2n, K] Wst[2co, K]^T with
// interleaved (P_c, Q_c) columns, statistics of |y| over the n points, (y_u, y_v) = (P_u - Q_v, P_v + Q_u) -- what
// dc_vn_stats(combine = 2) computes from PQ (first block of a VectorMLP, the I_J fold).  interleaved == 0:
// Y[2n, co] = V[2n, K] W[co, K]^T, statistics of |(Y_2i, Y_2i+1)| -- dc_vn_stats(combine = 0) (deeper blocks).
DC_EXPORT int dc_linear_vn_stats_forward(const float* V, int64_t ldv, const float* Wst, int64_t ldw, int64_t n,
                                         int32_t co, int32_t K, float* PQ, int64_t ldpq, int32_t interleaved,
                                         const float* gamma, const float* beta, float eps, float momentum,
                                         float* running_mean, float* running_var, float* mean, float* invstd,
                                         float* scale, float* shift, int32_t tile, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    DC_REQUIRE(V && Wst && PQ && mean && invstd && scale && shift, "dc_linear_vn_stats_forward: null pointer");
    const long M = 2 * n;
    const int N = interleaved ? 2 * co : co;
    DC_REQUIRE(n >= 1 && co >= 1 && K >= 1 && ldv >= K && ldw >= K && ldpq >= N, "dc_linear_vn_stats_forward: bad size");
    if (!workspace || workspace_bytes < dc_linear_stats_workspace_bytes(M, N, K, tile)) {
        dc_set_error("dc_linear_vn_stats_forward: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(workspace);
    if (int rc = run_gemm("dc_linear_vn_stats_forward", B_NK, interleaved ? EPI_VNSTATS : EPI_VNSTATS0, V, ldv, Wst, ldw, M,
                          N, K, PQ, ldpq, 0, tile, part, co, s))
        return rc;
    const dccol::BnFin fin{(long)n, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
    hipLaunchKernelGGL((dccol::colreduce_final_kernel<dccol::BnFin>), dim3(co), dim3(64), 0, s, part,
                       chunks_for(M, N, K, tile), co, fin);
    DC_CHECK_LAUNCH("dc_linear_vn_stats_forward");
    return DC_OK;
}

// The same two products with the statistics cut at the reduction (synchronised BatchNorm, data parallel): the fp64 column sums
// sums[2 C] = (sum_0, sum_1) of THIS rank's rows instead of finalised coefficients -- the host all-reduces them and
// dc_bn_coeffs_from_sums continues (deltaconv_amd/dp.py).  Same tile sums, same ordered final stage.
DC_EXPORT int dc_linear_bn_sums_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int32_t N, int32_t K,
                                        float* Y, int64_t ldy, double* sums, int32_t tile, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    DC_REQUIRE(X && W && Y && sums, "dc_linear_bn_sums_forward: null pointer");
    DC_REQUIRE(M >= 1 && N >= 1 && K >= 1 && ldx >= K && ldw >= K && ldy >= N, "dc_linear_bn_sums_forward: bad size");
    if (!workspace || workspace_bytes < dc_linear_stats_workspace_bytes(M, N, K, tile)) {
        dc_set_error("dc_linear_bn_sums_forward: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(workspace);
    if (int rc = run_gemm("dc_linear_bn_sums_forward", B_NK, EPI_COLSTATS, X, ldx, W, ldw, M, N, K, Y, ldy, 0, tile, part, N, s))
        return rc;
    hipLaunchKernelGGL((dccol::colreduce_final_kernel<dccol::SumsFin>), dim3(N), dim3(64), 0, s, part, chunks_for(M, N, K, tile), N,
                       dccol::SumsFin{sums, N, (double)M});
    DC_CHECK_LAUNCH("dc_linear_bn_sums_forward");
    return DC_OK;
}
DC_EXPORT int dc_linear_vn_sums_forward(const float* V, int64_t ldv, const float* Wst, int64_t ldw, int64_t n, int32_t co, int32_t K,
                                        float* PQ, int64_t ldpq, int32_t interleaved, double* sums, int32_t tile, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    DC_REQUIRE(V && Wst && PQ && sums, "dc_linear_vn_sums_forward: null pointer");
    const long M = 2 * n;
    const int N = interleaved ? 2 * co : co;
    DC_REQUIRE(n >= 1 && co >= 1 && K >= 1 && ldv >= K && ldw >= K && ldpq >= N, "dc_linear_vn_sums_forward: bad size");
    if (!workspace || workspace_bytes < dc_linear_stats_workspace_bytes(M, N, K, tile)) {
        dc_set_error("dc_linear_vn_sums_forward: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(workspace);
    if (int rc = run_gemm("dc_linear_vn_sums_forward", B_NK, interleaved ? EPI_VNSTATS : EPI_VNSTATS0, V, ldv, Wst, ldw, M, N, K, PQ,
                          ldpq, 0, tile, part, co, s))
        return rc;
    hipLaunchKernelGGL((dccol::colreduce_final_kernel<dccol::SumsFin>), dim3(co), dim3(64), 0, s, part, chunks_for(M, N, K, tile), co,
                       dccol::SumsFin{sums, co, (double)n});
    DC_CHECK_LAUNCH("dc_linear_vn_sums_forward");
    return DC_OK;
}

// ---- pre-split weight planes (round 4) ------------------------------------------------------------------------------------
// A weight matrix is the B operand of hundreds of workgroups per product and of several products per step; cutting it into
// its three bf16 planes once per step (instead of in every wave's K loop) removes half of the split instructions and the whole
// B tile from LDS: profiles/r04_gemm_split_bounds.txt (lab1) put that at 10-15 % of the split products, and it makes the split
// path pay on 64-column tiles as well.  The planes are bit-identical to what split_pair() produces in the K loop (same
// roundings), the MFMA order is unchanged: a product from planes returns the SAME BITS as the in-loop split.
namespace {
struct PresplitEntry { const float* src; unsigned short* fwd; unsigned short* bwd; long n, k, ld; };
__device__ __forceinline__ void split1(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned hp = pk_bf16(x, 0.f);
    const float r = x - __builtin_bit_cast(float, hp << 16);
    const unsigned mp = pk_bf16(r, 0.f);
    const float q = r - __builtin_bit_cast(float, mp << 16);
    h = (unsigned short)(hp & 0xffffu); m = (unsigned short)(mp & 0xffffu); l = (unsigned short)(pk_bf16(q, 0.f) & 0xffffu);
}
// one workgroup = one 32 x 32 tile of one matrix (rows and columns are multiples of 32); chunk_start[e] = first tile of entry e.
// Thread (r, c4) reads four consecutive columns of row r (coalesced 128-byte rows), cuts them, and
//   * stores its 4 consecutive plane elements of the FORWARD layout with one 8-byte store per plane (fragment-major: (row block
//     of 32, k-step of 16) -> 64 lanes x 8 elements, lane = 32 * (k-half) + row % 32: the 32 rows of a tile are 32 consecutive lanes);
//   * passes the tile through LDS for the planes of the TRANSPOSE: there a lane holds 8 consecutive ROWS of one column -- 16-byte
//     stores, 32 columns = 32 consecutive lanes.  (The first version wrote 2-byte elements at scattered addresses: 15 us per step.)
__global__ __launch_bounds__(256) void presplit_kernel(const PresplitEntry* __restrict__ table, const int* __restrict__ chunk_start,
                                                       int n_entries) {
    __shared__ unsigned short tl[3][32][34];
    int lo = 0, hi = n_entries;                             // entry of this tile (<= ~6 steps on cached words)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const PresplitEntry e = table[lo];
    const long total = e.n * e.k;
    const int tiles_c = (int)(e.k >> 5);
    const int t = (int)blockIdx.x - chunk_start[lo];
    const long rb = t / tiles_c, cb = t - rb * tiles_c;     // tile = rows 32 rb .., columns 32 cb ..
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(e.src + (rb * 32 + r) * e.ld + cb * 32 + c0);
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split1(x[q], h[q], m[q], l[q]);
    if (e.fwd) {
        const long c = cb * 32 + c0;                        // global column of element 0: same k-step / k-half for all four
        const long j = (((rb * (e.k >> 4) + (c >> 4)) * 64 + ((c & 15) >> 3) * 32 + r) * 8) + (c & 7);
        *reinterpret_cast<uint2*>(e.fwd + j) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        *reinterpret_cast<uint2*>(e.fwd + total + j) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
        *reinterpret_cast<uint2*>(e.fwd + 2 * total + j) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
    }
    if (e.bwd) {                                            // block-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) { tl[0][r][c0 + q] = h[q]; tl[1][r][c0 + q] = m[q]; tl[2][r][c0 + q] = l[q]; }
        __syncthreads();
        // transposed planes: rows <-> columns.  Item = (plane, column c, octet of rows): 3 x 32 x 4 = 384 items of 16 bytes
        for (int idx = threadIdx.x; idx < 384; idx += 256) {
            const int p = idx >> 7, rem = idx & 127, c = rem & 31, oct = rem >> 5;
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = tl[p][oct * 8 + 2 * q][c] | ((unsigned)tl[p][oct * 8 + 2 * q + 1][c] << 16);
            const long rg = rb * 32 + oct * 8;              // first of the eight rows: k-step rg / 16, k-half (rg % 16) / 8
            const long j = ((cb * (e.n >> 4) + (rg >> 4)) * 64 + ((rg & 15) >> 3) * 32 + c) * 8;
            *reinterpret_cast<uint4*>(e.bwd + (long)p * total + j) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
}  // namespace

// table: device array of n_entries records {src, fwd planes or NULL, bwd (transposed) planes or NULL, rows, cols, row stride}
// (six 64-bit words each); chunk_start: device int32 [n_entries + 1], chunk_start[e + 1] - chunk_start[e] = rows * cols / 1024
// (tiles of 32 x 32; rows, cols multiples of 32, src rows 16-byte aligned).
// fwd = three planes of rows x cols bf16, bwd = three planes of the transpose, both FRAGMENT-MAJOR (see presplit_kernel); rows
// and cols multiples of 32.
DC_EXPORT int dc_presplit_weights(const int64_t* table, const int32_t* chunk_start, int32_t n_entries, int32_t total_chunks,
                                  void* stream) {
    DC_REQUIRE(table && chunk_start && n_entries >= 0 && total_chunks >= 0, "dc_presplit_weights: bad arguments");
    if (n_entries == 0 || total_chunks == 0) return DC_OK;
    static_assert(sizeof(PresplitEntry) == 48, "table record = six 64-bit words");
    hipLaunchKernelGGL(presplit_kernel, dim3(total_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const PresplitEntry*>(table), chunk_start, n_entries);
    DC_CHECK_LAUNCH("dc_presplit_weights");
    return DC_OK;
}

// Planes of the weight operand of the NEXT dc_linear_* product enqueued by the calling thread (see BHint above).
DC_EXPORT int dc_gemm_next_b_planes(const void* weight, const void* planes, int64_t plane_elems, int64_t ld, int32_t transposed) {
    g_bhint = BHint{weight, static_cast<const unsigned short*>(planes), (long)plane_elems, (long)ld, transposed ? 1 : 0};
    return DC_OK;
}

#ifdef DC_LAB_STAMPS
// (lab build only) copies the phase stamps of the last launches to the host: [8192][8] = entry, first-loads issued?, K loop done,
// statistics done, stores issued, stores complete, HW_ID
DC_EXPORT int dc_lab_read_stamps(unsigned long long* host, int32_t n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(dc_lab_stamps), (size_t)n_words * 8) == hipSuccess ? DC_OK : DC_ERR_LAUNCH;
}
#endif
