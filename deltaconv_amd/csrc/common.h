// Shared helpers for the DeltaConv HIP kernels (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define DC_EXPORT extern "C" __attribute__((visibility("default")))

enum {
    DC_OK = 0,
    DC_ERR_ARG = -1,      // bad argument (null pointer, unsupported size, misaligned leading dimension)
    DC_ERR_LAUNCH = -2,   // HIP reported a launch error (see dc_last_error)
    DC_ERR_WORKSPACE = -3 // workspace too small
};

void dc_set_error(const char* fmt, ...);

#define DC_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            dc_set_error(__VA_ARGS__);        \
            return DC_ERR_ARG;                \
        }                                     \
    } while (0)

// Large dynamic LDS (> 64 KiB) needs a per-kernel, per-device opt-in: dc_ensure_lds() sets it once per device and, when the
// device cannot give the bytes (anything but gfx950's 160 KiB per workgroup), records a clear message that the next
// DC_CHECK_LAUNCH returns instead of an opaque launch failure.
bool dc_ensure_lds(unsigned long long* done_mask, const void* kernel, size_t bytes, const char* what);
bool dc_take_lds_failure();

#define DC_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        if (dc_take_lds_failure()) {                                            \
            (void)hipGetLastError();                                            \
            return DC_ERR_LAUNCH;                                               \
        }                                                                       \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            dc_set_error("%s: %s", name, hipGetErrorString(e_));                \
            return DC_ERR_LAUNCH;                                               \
        }                                                                       \
    } while (0)

// Runtime experiment switches (dc_set_option): index -> value.
enum { DC_OPT_XCD_REMAP = 0, DC_OPT_GATHER_BATCH = 1, DC_OPT_TN_LDS = 2 /* 2: weight-gradient GEMM through the direct-load kernel of round 1 (lab) */, DC_OPT_GEMM_EXACT = 3 /* 1: dense products through the exact fp32 MFMA chain instead of the bf16 split products */, DC_OPT_GEMM_STAGGER = 4 /* first-round phase shift of every second 128 x 128 workgroup of a CU, percent of a K loop */, DC_OPT_TN_TILE = 5 /* lab: 1..4 forces the weight-gradient tile */, DC_OPT_TN_SLABS = 6 /* lab: forces the slab count */, DC_OPT_TILE_UPW = 7, DC_OPT_CSC_ONE_WG = 8 /* 1: CSC count / scan / fill by one workgroup per cloud (round 3) instead of eight column ranges */, DC_OPT_NO_PLANES = 9 /* 1: ignore pre-split weight planes (every product splits its B operand in the K loop: round 3) */, DC_OPT_CE_TWO_LAUNCH = 10 /* 1: cross-entropy of <= 64 rows through the two-launch form (A/B, bit-identity test) */, DC_OPT_WIDE_TILES = 11 /* 1: dense products on few rows keep 128-column tiles (A/B of the round-6 rule: 64-column tiles while a launch has fewer than 256 workgroups) */, DC_OPT_COUNT = 12 };
int dc_option(int key);

static inline int dc_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Output stores ----------------------------------------------------------------------------------------------------
// A plain store leaves a kernel's output dirty in the per-XCD L2 until the end-of-kernel write-back, which is then
// exposed in full; a non-temporal store streams it out while the kernel still runs (measured alone, r03e: a 42 MB
// stream copy 8.98 -> 5.76 us, the tiled div|curl|norm apply 16.0 -> 12.4 us).  Inside a step the picture is mixed: the
// consumer of a streamed-out tensor no longer finds it in L2, so the policy is chosen per kernel family from A/B runs
// of the whole step (profiles/r03*_store_policy.txt); DC_NT_MASK holds the families that stream.  16-byte stores only.
enum { DC_ST_TILE = 1 /* ell_tile.h */, DC_ST_ELL = 2 /* ell_math.h: staged applies, transposes, edge kernels */,
       DC_ST_NN = 4 /* colreduce.h stv: BatchNorm / activation / vector non-linearity passes */, DC_ST_GEMM = 8 /* gemm.hip */ };
#ifndef DC_NT_MASK
#define DC_NT_MASK (DC_ST_TILE)
#endif
typedef float dc_f32x4 __attribute__((ext_vector_type(4)));
template <int FAMILY>
__device__ __forceinline__ void dc_store16(float* p, dc_f32x4 v) {
    if constexpr ((DC_NT_MASK & FAMILY) != 0) __builtin_nontemporal_store(v, reinterpret_cast<dc_f32x4*>(p));
    else *reinterpret_cast<dc_f32x4*>(p) = v;
}

// Device-clock stamps of selected kernels: what bench.py reads the graded apply's duration INSIDE the replayed training step
// from (HIP events cannot bracket one kernel of a captured graph).  dc_stamp_buffer(buf, slots) arms it: every later
// launch of a stamped kernel family takes the next record of `buf` -- 4 x u64: [earliest workgroup entry | latest workgroup
// exit with its stores complete | unused | unused], constant 100 MHz clock (s_memrealtime) -- at ENQUEUE time, so a launch
// captured into a HIP graph keeps its record across replays; the caller resets the records (min = huge, max = 0) before a
// replay.  Off (buf = NULL): the kernels get a null pointer and skip two scalar branches.
// ---- deferred finalisers (round 6): between dc_finalisers_begin() and dc_finalisers_end() an entry point whose caller asked for it
// (dc_finaliser_defer_next(): one-shot) queues the second stage of its column reduction instead of launching it; the end call
// launches ONE kernel for all queued finalisers (colreduce.h: colreduce_final_many_kernel).  Two independent products of a layer
// node (max-aggregation stream, s_mlp) thus share one finaliser launch.  Host-side state, thread-local (error.hip).
constexpr int DC_FIN_MAX = 4, DC_FIN_BLOB = 128;
enum { DC_FIN_BN = 0, DC_FIN_BWD_COEF = 1 };
struct DcFinPending {
    int kind, chunks, C;
    const double* partial;
    alignas(8) unsigned char blob[DC_FIN_BLOB];       // the finaliser object (dccol::BnFin / BwdCoefFin), copied by bytes
    // the reduction's FIRST stage, when that is queued too (BatchNorm-backward reductions: two of them run as one launch)
    int stage1, rpc;                                  // stage1 != 0: not launched yet
    long R;
    alignas(8) unsigned char functor[DC_FIN_BLOB];    // the row functor (nn.hip: BnBwdF<4>), copied by bytes
};
bool dc_gemm_take_request();                          // true once after dc_gemm_defer_next() inside an open batch whose finaliser request is pending too
void dc_gemm_flush(void* stream);                     // gemm.hip: launches the queued dense products (two of one kind as one launch)
void dc_gemm_discard();
bool dc_fin_take_request();                           // true once after dc_finaliser_defer_next() inside an open batch with room
void dc_fin_push(int kind, const double* partial, int chunks, int C, const void* fin, size_t bytes, const void* functor = nullptr,
                 size_t functor_bytes = 0, long R = 0, int rpc = 0);
int dc_fin_pending(DcFinPending** out);               // -> count (and the queue)
void dc_fin_clear();
unsigned long long* dc_stamp_next(int tag);        // host side; nullptr when stamping is off or the records are used up
__device__ __forceinline__ void dc_stamp_in(unsigned long long* st) {
    if (st && threadIdx.x == 0) atomicMin(st, (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void dc_stamp_out(unsigned long long* st) {
    if (st && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicMax(st + 1, (unsigned long long)wall_clock64());
    }
}

// Wave-level helpers -------------------------------------------------------------------------
// Stream-ordered zero fill as a kernel (not hipMemsetAsync: a memset node inside a captured HIP graph
// is a different code path from a kernel node; every entry point enqueues kernels only).
namespace {
__global__ void dc_zero_words_kernel(unsigned* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
}  // namespace
static inline void dc_zero_words(void* p, long n_words, hipStream_t s) {
    if (n_words <= 0) return;
    hipLaunchKernelGGL(dc_zero_words_kernel, dim3(dc_cdiv(n_words, 256)), dim3(256), 0, s, static_cast<unsigned*>(p), n_words);
}

__device__ __forceinline__ double dc_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dc_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dc_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware block remap (MI355X: 8 XCDs, block b is dispatched to XCD b % 8, each XCD has its own
// 4 MiB L2).  Returns a logical block id such that every XCD owns a CONTIGUOUS range of logical
// blocks (= contiguous points = whole clouds), so the neighbour gathers of a cloud hit one L2.
// Pure performance: any placement gives the same result.
__device__ __forceinline__ long dc_xcd_block(int remap) {
    const long b = blockIdx.x, nb = gridDim.x;
    if (!remap) return b;
    const long q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}
