// Tall-skinny transposed GEMM on the fp32 matrix cores:  C[M,N] = A^T B,  A [R,M], B [R,N], R >> M,N.
//
// This is the weight-gradient GEMM of every per-point Linear layer (dW = dY^T X with R = Nt or 2Nt
// = 32768..65536 rows and M,N = 64..512 features): ATen/rocBLAS `mm` in the reference's autograd of
// /root/reference/deltaconv/nn/mlp.py:9,15.  The vendor library reaches 27-98 TFLOP/s on these
// shapes (tuned); the reduction dimension is the long one, so the work is split over row slabs.
//
// Mapping (CDNA4): v_mfma_f32_32x32x2_f32 takes A[i][k] and B[k][j] with lane l holding
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].  With both operands row-major over the SAME
// row index r (= k), lane l simply loads A[r0 + (l>>5)][m0 + (l&31)] and B[r0 + (l>>5)][n0 + (l&31)]:
// two coalesced 128-byte row segments per instruction, no LDS, no transposes.  A workgroup is
// 4 waves in a 2x2 arrangement, each wave owns a 64x64 (or 32-wide at the edges) block of C in 4
// independent 32x32 accumulators and streams its row slab with a software-pipelined (double
// buffered) register prefetch of 8 k-steps.  Partials [slab][M][N] are then summed in slab order
// by a second kernel: deterministic, no atomics.  Exact fp32 (the MFMA is an fmaf chain).
// Bound: MFMA (157 TFLOP/s fp32 peak); operand traffic is 1 KB per 256 MFMA-cycles per wave.
#include <algorithm>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KU = 8;  // k-steps (of 2 rows) per pipeline stage 

template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, long lda,
                                                      const float* __restrict__ B, long ldb, long R, int M, int N,
                                                      int tiles_n, int rows_per_slab, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x, slab = blockIdx.y;
    const int m0 = (tile / tiles_n) * (64 * WM) + (wave >> 1) * (32 * WM);
    const int n0 = (tile % tiles_n) * (64 * WN) + (wave & 1) * (32 * WN);
    if (m0 >= M || n0 >= N) return;  // wave-uniform
    bool mv[WM], nv[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) mv[i] = m0 + 32 * i < M;
#pragma unroll
    for (int j = 0; j < WN; ++j) nv[j] = n0 + 32 * j < N;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const long r_begin = (long)slab * rows_per_slab;
    const long r_end = min(R, r_begin + rows_per_slab);
    const int kh = lane >> 5, cl = lane & 31;

    float a_cur[KU][WM], b_cur[KU][WN], a_nxt[KU][WM], b_nxt[KU][WN];
    // Branch-free loads: out-of-range 32-column blocks / rows are read from a clamped (valid) address
    // and zeroed with a select, so the stage is a straight run of global_load_dword.
    const float* ap[WM];
    const float* bp[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) ap[i] = A + (mv[i] ? m0 + 32 * i : m0) + cl;
#pragma unroll
    for (int j = 0; j < WN; ++j) bp[j] = B + (nv[j] ? n0 + 32 * j : n0) + cl;
    // Full stages: 2*KU valid rows, a straight run of unconditional loads (invalid 32-column blocks
    // read a clamped valid column and their products are simply never stored).
    auto load_full = [&](long r0, float (&a)[KU][WM], float (&b)[KU][WN]) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const long row = r0 + 2 * u + kh;
#pragma unroll
            for (int i = 0; i < WM; ++i) a[u][i] = ap[i][row * lda];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[u][j] = bp[j][row * ldb];
        }
    };
    auto mfma_stage = [&](const float (&a)[KU][WM], const float (&b)[KU][WN]) {
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    };

    // Ping-pong register buffers, loop unrolled by two stages: no register copies, so the MFMAs of
    // stage s only wait for stage s while the loads of stage s+1 stay in flight (vmcnt = one stage).
    const long full_end = r_begin + (r_end - r_begin) / (2 * KU) * (2 * KU);
    const long S = 2 * KU;
    if (r_begin < full_end) load_full(r_begin, a_cur, b_cur);
    for (long r = r_begin; r < full_end; r += 2 * S) {
        if (r + S < full_end) load_full(r + S, a_nxt, b_nxt);
        mfma_stage(a_cur, b_cur);
        if (r + S < full_end) {
            if (r + 2 * S < full_end) load_full(r + 2 * S, a_cur, b_cur);
            mfma_stage(a_nxt, b_nxt);
        }
    }
    if (full_end < r_end) {   // ragged tail of the slab: clamp the row, zero the value
        const long last_row = r_end - 1;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const long row = full_end + 2 * u + kh;
            const bool ok = row <= last_row;
            const long rr = ok ? row : last_row;
#pragma unroll
            for (int i = 0; i < WM; ++i) { const float v = ap[i][rr * lda]; a_cur[u][i] = ok ? v : 0.f; }
#pragma unroll
            for (int j = 0; j < WN; ++j) { const float v = bp[j][rr * ldb]; b_cur[u][j] = ok ? v : 0.f; }
        }
        mfma_stage(a_cur, b_cur);
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* p = partial + (long)slab * M * N;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (!mv[i] || !nv[j]) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * kh;
                p[(long)(m0 + 32 * i + row) * N + n0 + 32 * j + cl] = acc[i][j][q];
            }
        }
}

// C[m][n] = sum over slabs.  Block = 16 waves x 64 consecutive elements: wave w sums slabs w, w+16, ...
// (coalesced 256-B reads, all loads of a thread in flight), then the 16 partial sums are added in
// wave order through LDS -- a fixed association, bit-reproducible.
constexpr int RED_WAVES = 16;
__global__ __launch_bounds__(64 * RED_WAVES) void gemm_tn_reduce_kernel(const float* __restrict__ partial, int slabs,
                                                                        long mn, int N, float* __restrict__ C, long ldc,
                                                                        int accumulate) {
    __shared__ float sm[RED_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (e < mn) {
#pragma unroll 8
        for (int sl = w; sl < slabs; sl += RED_WAVES) s += partial[(long)sl * mn + e];
    }
    sm[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < mn) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < RED_WAVES; ++q) t += sm[q][lane];
        float* dst = C + (e / N) * ldc + (e % N);
        *dst = accumulate ? *dst + t : t;
    }
}

// At most 16 slabs: every chain of the kernel above holds at most one slab, its result is the plain ordered sum over the slabs
// starting from 0.f -- the same bits from a streaming form: thread = 4 consecutive elements (16-byte loads, all slabs of a thread
// in flight), no LDS, no idle waves (the 1024 x 448 embedding gradient has 9 slabs: 7 of the 16 waves above had nothing to do,
// 20 -> ~6 us).  mn, N and ldc multiples of 4, 16-byte aligned bases.
__global__ __launch_bounds__(256) void gemm_tn_reduce_few_kernel(const float* __restrict__ partial, int slabs, long mn, int N,
                                                                 float* __restrict__ C, long ldc, int accumulate) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= mn) return;
    float v[16][4];
#pragma unroll
    for (int sl = 0; sl < 16; ++sl)
        if (sl < slabs) {
            const float4 q = *reinterpret_cast<const float4*>(partial + (long)sl * mn + e);
            v[sl][0] = q.x; v[sl][1] = q.y; v[sl][2] = q.z; v[sl][3] = q.w;
        }
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < 16; ++sl)
        if (sl < slabs)
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] += (0.f + v[sl][j]);          // (chain q: 0.f + p_q, then the chains in order)
    float* dst = C + (e / N) * ldc + (e % N);
    float4 o = {t[0], t[1], t[2], t[3]};
    if (accumulate) {
        const float4 d = *reinterpret_cast<const float4*>(dst);
        o.x = d.x + o.x; o.y = d.y + o.y; o.z = d.z + o.z; o.w = d.w + o.w;
    }
    *reinterpret_cast<float4*>(dst) = o;
}
// the ordered slab reduction: streaming form where it applies, else the 16-chain form
static void launch_tn_reduce(const float* partial, int slabs, long mn, int N, float* C, long ldc, int accumulate, hipStream_t s) {
    const bool few = slabs <= 16 && mn % 4 == 0 && N % 4 == 0 && ldc % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(partial) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                     dc_option(DC_OPT_TN_LDS) != 3;            // (option 2 = 3, lab: always the 16-chain form)
    if (few)
        hipLaunchKernelGGL(gemm_tn_reduce_few_kernel, dim3(dc_cdiv(mn / 4, 256)), dim3(256), 0, s, partial, slabs, mn, N, C, ldc, accumulate);
    else
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(dc_cdiv(mn, 64)), dim3(64 * RED_WAVES), 0, s, partial, slabs, mn, N, C, ldc,
                           accumulate);
}

// Several ordered slab reductions in ONE launch (round 6): an autograd node that forms three or four weight gradients (a DeltaConv
// layer: v_mlp, s_mlp, max-aggregation Linear) queues their partial tiles and sums them all at the end of its backward -- one
// launch instead of one per weight (inside a replayed step a launch costs ~4.7 us whatever it does).  Same association as the two
// kernels above, hence the same bits: chain q = 0.f + p_q + p_(q+16) + ..., then the chains in order on top of 0.f (empty chains
// add +0.f, which changes nothing: no sum here is ever -0.f).  Table by value in the kernel arguments; each entry runs in the form
// the single launch would have taken (first version: one serial loop per thread over up to 128 slabs -- the step got 0.22 ms SLOWER,
// profiles/r06_labs.txt item 8).
constexpr int TNR_MAX = 16;
struct TnReduceTable {
    const float* partial[TNR_MAX];
    float* C[TNR_MAX];
    long mn[TNR_MAX], ldc[TNR_MAX];
    int N[TNR_MAX], slabs[TNR_MAX], accumulate[TNR_MAX];
    int first_block[TNR_MAX + 1];          // prefix sums of the entries' block counts
    int count;
};
// an entry is "streamed" (block = 1024 elements, every chain holds one slab, all 16-byte loads of a thread in flight) when it has
// at most 16 slabs and 16-byte geometry; else "chained" (block = 64 elements x 16 chains over 4 waves, scalar accesses)
__host__ __device__ inline bool tnr_streamed(const float* partial, const float* C, long mn, long ldc, int N, int slabs) {
    return slabs <= 16 && mn % 4 == 0 && N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(C) & 15) == 0;
}
__global__ __launch_bounds__(256) void gemm_tn_reduce_many_kernel(TnReduceTable t) {
    __shared__ float sm[16][64];
    int lo = 0, hi = t.count;
    const int blk = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.first_block[mid] <= blk) lo = mid;
        else hi = mid;
    }
    const float* __restrict__ partial = t.partial[lo];
    float* __restrict__ C = t.C[lo];
    const long mn = t.mn[lo], ldc = t.ldc[lo];
    const int N = t.N[lo], slabs = t.slabs[lo], accumulate = t.accumulate[lo];
    const int local = blk - t.first_block[lo];
    if (tnr_streamed(partial, C, mn, ldc, N, slabs)) {               // uniform over the entry
        const long e = ((long)local * 256 + threadIdx.x) * 4;
        if (e >= mn) return;
        float v[16][4];
#pragma unroll
        for (int sl = 0; sl < 16; ++sl)
            if (sl < slabs) {
                const float4 q = *reinterpret_cast<const float4*>(partial + (long)sl * mn + e);
                v[sl][0] = q.x; v[sl][1] = q.y; v[sl][2] = q.z; v[sl][3] = q.w;
            }
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 16; ++sl)
            if (sl < slabs)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[j] += (0.f + v[sl][j]);
        float* dst = C + (e / N) * ldc + (e % N);
        float4 o = {sum[0], sum[1], sum[2], sum[3]};
        if (accumulate) {
            const float4 d = *reinterpret_cast<const float4*>(dst);
            o.x = d.x + o.x; o.y = d.y + o.y; o.z = d.z + o.z; o.w = d.w + o.w;
        }
        *reinterpret_cast<float4*>(dst) = o;
        return;
    }
    // chained: wave w carries chains w, w + 4, w + 8, w + 12 (chain q = slabs q, q + 16, ... in order), four independent
    // accumulators with their loads in flight together; then the 16 chains in order through LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long e = (long)local * 64 + lane;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < mn) {
#pragma unroll 2
        for (int s0 = 0; s0 < slabs; s0 += 16) {
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sl = s0 + w + 4 * j;
                x[j] = sl < slabs ? partial[(long)sl * mn + e] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (s0 + w + 4 * j < slabs) c[j] += x[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sm[w + 4 * j][lane] = c[j];
    __syncthreads();
    if (w == 0 && e < mn) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += sm[q][lane];
        float* dst = C + (e / N) * ldc + (e % N);
        *dst = accumulate ? *dst + sum : sum;
    }
}

struct Plan {
    int wm, wn, tiles_m, tiles_n, slabs, rows_per_slab;
};
Plan make_plan(long R, int M, int N) {
    Plan p;
    p.wm = (M % 128 == 0 || M > 64) ? 2 : 1;
    p.wn = (N % 128 == 0 || N > 64) ? 2 : 1;
    p.tiles_m = dc_cdiv(M, 64 * p.wm);
    p.tiles_n = dc_cdiv(N, 64 * p.wn);
    const int tiles = p.tiles_m * p.tiles_n;
    int slabs = std::min(128, std::max(1, 512 / tiles));        // ~2 workgroups per CU, bounded partial traffic
    long rps = (R + slabs - 1) / slabs;
    rps = std::max<long>((rps + 2 * KU - 1) / (2 * KU) * (2 * KU), 2 * KU * 4);
    p.rows_per_slab = (int)rps;
    p.slabs = (int)((R + rps - 1) / rps);
    return p;
}

}  // namespace

// LDS-staged variant (gemm.hip): full-line 16-byte loads into LDS, fragments from LDS, 16-byte stores
struct DcTnPlan { int bm, bn, slabs; long rows_per_slab; };
DcTnPlan dc_tn_lds_plan(long R, int M, int N);
int dc_tn_lds_launch(const float* A, long lda, const float* B, long ldb, long R, int M, int N, float* partial, hipStream_t s,
                     const float* h = nullptr, long ldh = 0, const float* coefs = nullptr, float slope = 0.f);

DC_EXPORT size_t dc_gemm_tn_workspace_bytes(int64_t R, int32_t M, int32_t N) {
    const Plan p = make_plan(R, M, N);
    const DcTnPlan q = dc_tn_lds_plan(R, M, N);
    return (size_t)std::max(p.slabs, q.slabs) * M * N * sizeof(float);
}

// C[M,N] (ldc) (+)= A^T B with A [R,M] (lda), B [R,N] (ldb); M and N multiples of 32.
DC_EXPORT int dc_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t R, int32_t M, int32_t N,
                         float* C, int64_t ldc, int32_t accumulate, void* workspace, size_t workspace_bytes,
                         void* stream) {
    DC_REQUIRE(A && B && C, "dc_gemm_tn: null pointer");
    DC_REQUIRE(R >= 1 && M >= 1 && N >= 1, "dc_gemm_tn: bad size");
    DC_REQUIRE(lda >= M && ldb >= N && ldc >= N, "dc_gemm_tn: leading dimension smaller than the row");
    DC_REQUIRE(lda < (1 << 21) && ldb < (1 << 21), "dc_gemm_tn: leading dimension above 2^21 elements");
    if (!workspace || workspace_bytes < dc_gemm_tn_workspace_bytes(R, M, N)) {
        dc_set_error("dc_gemm_tn: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* partial = static_cast<float*>(workspace);
    // The LDS-staged kernel of gemm.hip (any M, N, leading dimension) is the product path: with the r02r tile / slab
    // plan it matches or beats the direct-load kernel of round 1 on every measured shape.  That kernel stays
    // reachable for A/B runs (option DC_OPT_TN_LDS = 2, multiples of 32 only).
    const bool direct = dc_option(DC_OPT_TN_LDS) == 2 && M % 32 == 0 && N % 32 == 0;
    if (!direct) {
        const int slabs = dc_tn_lds_launch(A, (long)lda, B, (long)ldb, (long)R, M, N, partial, s);
        const long mn2 = (long)M * N;
        launch_tn_reduce(partial, slabs, mn2, N, C, (long)ldc, accumulate, s);
        DC_CHECK_LAUNCH("dc_gemm_tn");
        return DC_OK;
    }
    const Plan p = make_plan(R, M, N);
    dim3 grid(p.tiles_m * p.tiles_n, p.slabs);
#define DC_TN_LAUNCH(WM, WN)                                                                                       \
    hipLaunchKernelGGL((gemm_tn_kernel<WM, WN>), grid, dim3(256), 0, s, A, (long)lda, B, (long)ldb, (long)R, M, N, \
                       p.tiles_n, p.rows_per_slab, partial)
    if (p.wm == 2 && p.wn == 2) DC_TN_LAUNCH(2, 2);
    else if (p.wm == 2) DC_TN_LAUNCH(2, 1);
    else if (p.wn == 2) DC_TN_LAUNCH(1, 2);
    else DC_TN_LAUNCH(1, 1);
    const long mn = (long)M * N;
    launch_tn_reduce(partial, p.slabs, mn, N, C, (long)ldc, accumulate, s);
    DC_CHECK_LAUNCH("dc_gemm_tn");
    return DC_OK;
}

// dW[N,K] (lddw) (+)= dh[R,N]^T X[R,K] with dh = BatchNorm/activation backward of (dy, h) formed in the operand loader
// (LDS-staged kernel of gemm.hip, reduction split over row slabs, ordered reduction).  Workspace:
// dc_gemm_tn_workspace_bytes(R, N, K).
DC_EXPORT int dc_linear_bn_backward_weight(const float* dy, int64_t lddy, const float* h, int64_t ldh, const float* coefs,
                                           float slope, const float* X, int64_t ldx, int64_t R, int32_t N, int32_t K,
                                           float* dW, int64_t lddw, int32_t accumulate, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    DC_REQUIRE(dy && h && coefs && X && dW, "dc_linear_bn_backward_weight: null pointer");
    DC_REQUIRE(R >= 1 && N >= 1 && K >= 1 && lddy >= N && ldh >= N && ldx >= K && lddw >= K,
               "dc_linear_bn_backward_weight: bad size");
    DC_REQUIRE(lddy < (1 << 21) && ldh < (1 << 21) && ldx < (1 << 21),
               "dc_linear_bn_backward_weight: leading dimension above 2^21 elements");
    if (!workspace || workspace_bytes < dc_gemm_tn_workspace_bytes(R, N, K)) {
        dc_set_error("dc_linear_bn_backward_weight: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* partial = static_cast<float*>(workspace);
    const int slabs = dc_tn_lds_launch(dy, (long)lddy, X, (long)ldx, (long)R, N, K, partial, s, h, (long)ldh, coefs, slope);
    const long mn = (long)N * K;
    launch_tn_reduce(partial, slabs, mn, K, dW, (long)lddw, accumulate, s);
    DC_CHECK_LAUNCH("dc_linear_bn_backward_weight");
    return DC_OK;
}

// ---- the same weight gradients with the slab reduction deferred: product now, ONE reduction launch for several weights later ----
// dc_gemm_tn_slabs / dc_linear_bn_backward_weight_slabs: the partial tiles [slabs][M][N] only (workspace as above); *slabs <- how
// many there are (host int).  dc_gemm_tn_reduce_many: C_i[M_i, N_i] (ldc_i) (+)= the ordered sum of the slabs of entry i, for
// `count` entries (host arrays), bit-identical to what dc_gemm_tn / dc_linear_bn_backward_weight write.  Stream-ordered, capturable.
DC_EXPORT int dc_gemm_tn_slabs(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t R, int32_t M, int32_t N,
                               void* workspace, size_t workspace_bytes, int32_t* slabs, void* stream) {
    DC_REQUIRE(A && B && slabs, "dc_gemm_tn_slabs: null pointer");
    DC_REQUIRE(R >= 1 && M >= 1 && N >= 1, "dc_gemm_tn_slabs: bad size");
    DC_REQUIRE(lda >= M && ldb >= N, "dc_gemm_tn_slabs: leading dimension smaller than the row");
    DC_REQUIRE(lda < (1 << 21) && ldb < (1 << 21), "dc_gemm_tn_slabs: leading dimension above 2^21 elements");
    if (!workspace || workspace_bytes < dc_gemm_tn_workspace_bytes(R, M, N)) {
        dc_set_error("dc_gemm_tn_slabs: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    *slabs = dc_tn_lds_launch(A, (long)lda, B, (long)ldb, (long)R, M, N, static_cast<float*>(workspace),
                              static_cast<hipStream_t>(stream));
    DC_CHECK_LAUNCH("dc_gemm_tn_slabs");
    return DC_OK;
}

DC_EXPORT int dc_linear_bn_backward_weight_slabs(const float* dy, int64_t lddy, const float* h, int64_t ldh, const float* coefs,
                                                 float slope, const float* X, int64_t ldx, int64_t R, int32_t N, int32_t K,
                                                 void* workspace, size_t workspace_bytes, int32_t* slabs, void* stream) {
    DC_REQUIRE(dy && h && coefs && X && slabs, "dc_linear_bn_backward_weight_slabs: null pointer");
    DC_REQUIRE(R >= 1 && N >= 1 && K >= 1 && lddy >= N && ldh >= N && ldx >= K, "dc_linear_bn_backward_weight_slabs: bad size");
    DC_REQUIRE(lddy < (1 << 21) && ldh < (1 << 21) && ldx < (1 << 21),
               "dc_linear_bn_backward_weight_slabs: leading dimension above 2^21 elements");
    if (!workspace || workspace_bytes < dc_gemm_tn_workspace_bytes(R, N, K)) {
        dc_set_error("dc_linear_bn_backward_weight_slabs: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    *slabs = dc_tn_lds_launch(dy, (long)lddy, X, (long)ldx, (long)R, N, K, static_cast<float*>(workspace),
                              static_cast<hipStream_t>(stream), h, (long)ldh, coefs, slope);
    DC_CHECK_LAUNCH("dc_linear_bn_backward_weight_slabs");
    return DC_OK;
}

DC_EXPORT int dc_gemm_tn_reduce_many(const int64_t* partials, const int64_t* outs, const int64_t* ldc, const int32_t* rows,
                                     const int32_t* cols, const int32_t* slabs, const int32_t* accumulate, int32_t count,
                                     void* stream) {
    DC_REQUIRE(count >= 0 && (count == 0 || (partials && outs && ldc && rows && cols && slabs)), "dc_gemm_tn_reduce_many: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int t0 = 0; t0 < count; t0 += TNR_MAX) {
        TnReduceTable t;
        t.count = 0;
        int blocks = 0;
        for (int i = t0; i < count && i < t0 + TNR_MAX; ++i) {
            DC_REQUIRE(partials[i] && outs[i] && rows[i] >= 1 && cols[i] >= 1 && slabs[i] >= 1 && ldc[i] >= cols[i],
                       "dc_gemm_tn_reduce_many: bad entry");
            const int c = t.count++;
            t.partial[c] = reinterpret_cast<const float*>(partials[i]);
            t.C[c] = reinterpret_cast<float*>(outs[i]);
            t.mn[c] = (long)rows[i] * cols[i];
            t.ldc[c] = (long)ldc[i];
            t.N[c] = cols[i];
            t.slabs[c] = slabs[i];
            t.accumulate[c] = accumulate ? accumulate[i] : 0;
            t.first_block[c] = blocks;
            blocks += tnr_streamed(t.partial[c], t.C[c], t.mn[c], t.ldc[c], t.N[c], t.slabs[c]) ? dc_cdiv(dc_cdiv(t.mn[c], 4L), 256L)
                                                                                               : dc_cdiv(t.mn[c], 64L);
        }
        if (!t.count) continue;
        t.first_block[t.count] = blocks;
        hipLaunchKernelGGL(gemm_tn_reduce_many_kernel, dim3(blocks), dim3(256), 0, s, t);
    }
    DC_CHECK_LAUNCH("dc_gemm_tn_reduce_many");
    return DC_OK;
}
