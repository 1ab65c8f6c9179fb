// MLP blocks on a HANDFUL of rows: the classification head (one row per cloud, B = 32 rows at the bench batch):
//   MLP([2048, 512]) -> Dropout -> MLP([512, 256]) -> Dropout -> Linear(256, num_classes)
// (/root/reference/deltaconv/models/deltanet_classification.py:34-36,51; blocks = nn/mlp.py:7-11: Linear(no bias) ->
// BatchNorm over the rows -> LeakyReLU).  With so few rows every step of a block is launch latency: the products were
// the last vendor-library GEMMs of the step (three Cijk_* launches forward, four backward, 63 us), each followed by a
// statistics reduction, its finaliser and an activation pass -- 31 launches of ~4.6 us for the head (r03h timeline).
//
// Here a block is ONE kernel forward and one backward (+ the input gradient through the MFMA GEMM of gemm.hip):
//   forward  : a wavefront owns CW output columns; lanes split the reduction index (X staged through LDS in pieces of
//              256 columns and shared by the block's waves), a recursive-halving butterfly leaves lane (c, m) with
//              h[m, c]; the BatchNorm statistics over the rows of a column are then a reduction over 32 (64) lanes:
//              statistics, finalisation (fp64, the formulas of colreduce.h: BnFin), running statistics, activation and
//              the store happen in the same wavefront.  No partial sums, no finaliser, no activation pass.
//   backward : the same mapping: lane (c, m) forms dz, the two BatchNorm sums over the rows (fp64), d gamma / d beta,
//              dh; then the wavefront's columns of the weight gradient dW[c, :] = sum_m dh[m, c] X[m, :] with dh
//              broadcast from the lanes.  d X = dH W goes through dc_linear_backward_input (ragged M is guarded there).
// Deterministic: fixed summation orders everywhere (lane-sequential over k, fixed butterflies over lanes / rows).
// mode 0 = Linear (+ bias) only; 1 = BatchNorm with batch statistics; 2 = BatchNorm with running statistics.
#include "common.h"
#include "nn_math.h"

namespace {
using dcnn::act;
using dcnn::dact;

constexpr int KC = 256;              // reduction-index piece staged in LDS: one float4 per lane
constexpr int TPB = 256;

template <int CW>
struct RB {
    static constexpr int RM = 64 / CW;          // rows a lane group can hold (lane = c * RM + m)
};

// recursive halving over the 64 lanes: in: v[j], j = 0..63 partial sums of this lane; out: the full sum of index j = lane
template <int OFF>
__device__ __forceinline__ void halve_step(float (&v)[64], int lane) {
    const bool up = (lane & OFF) != 0;
#pragma unroll
    for (int i = 0; i < OFF; ++i) {
        const float send = up ? v[i] : v[i + OFF];
        const float keep = up ? v[i + OFF] : v[i];
        v[i] = keep + __shfl_xor(send, OFF, 64);
    }
}
__device__ __forceinline__ float halve64(float (&v)[64], int lane) {
    halve_step<32>(v, lane);
    halve_step<16>(v, lane);
    halve_step<8>(v, lane);
    halve_step<4>(v, lane);
    halve_step<2>(v, lane);
    halve_step<1>(v, lane);
    return v[0];
}

// sum over the RM lanes of a lane group (lanes sharing lane / RM)
template <int RM>
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
    for (int off = RM / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// X[0..M, kc .. kc+KC) -> LDS as [M][64] float4 (zero beyond K), in two halves so that the global loads of piece i+1 are
// in flight while piece i is multiplied: all loads of a piece are issued together (a load -> store loop would wait for
// every load in turn: 8 round trips per piece, 37 us for the [32 x 2048] block of the head).
template <int SI>
__device__ __forceinline__ void x_load(float4 (&stg)[SI], const float* __restrict__ X, long ldx, int M, int K, int kc, int tid) {
#pragma unroll
    for (int it = 0; it < SI; ++it) {
        const int idx = tid + it * TPB;
        const int m = idx >> 6, k = kc + (idx & 63) * 4;
        stg[it] = (m < M && k < K) ? *reinterpret_cast<const float4*>(X + (long)m * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int SI>
__device__ __forceinline__ void x_store(float4* xs, const float4 (&stg)[SI], int tid) {
#pragma unroll
    for (int it = 0; it < SI; ++it) xs[tid + it * TPB] = stg[it];
}

// K-slices per workgroup: the pieces of the reduction index are a serial chain per wavefront (stage -> barrier -> multiply,
// ~1.3 us each: 8 pieces at K = 2048), so a workgroup runs KS groups of 4 wavefronts, group s taking the pieces s, s + KS, ...
// with its own LDS staging area; forward: the KS partial results of a column meet in LDS after the lane butterfly and are
// added in slice order; backward (dW[n, k]): the slices own disjoint k ranges, nothing to combine.
// forward: 4 x 32 KB (<= 32 rows) or 2 x 64 KB of staging; 1024-thread workgroups leave 128 registers per lane, so the
// 4-slice form loads each piece right before it stores it (no register prefetch: two rounds at K = 2048).  backward keeps
// 64 broadcast dh values per lane: 2 slices (512 threads, 256 registers).
template <int CW> struct KSlices { static constexpr int KS = CW == 2 ? 4 : 2, KSB = 2; };

template <int CW>
__global__ __launch_bounds__(TPB * KSlices<CW>::KS) void rowblock_fwd_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ W, long ldw, const float* __restrict__ bias, int M, int N,
    int K, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, float* __restrict__ rmean,
    float* __restrict__ rvar, int mode, float slope, float* __restrict__ H, long ldh, float* __restrict__ coef,
    float* __restrict__ Y, long ldy, float drop_p, unsigned seed, const long long* __restrict__ step, unsigned salt,
    unsigned char* __restrict__ mask) {
    constexpr int RM = RB<CW>::RM, KS = KSlices<CW>::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ks = threadIdx.x / TPB, tid = threadIdx.x - ks * TPB;           // k-slice, thread inside its 4 wavefronts
    float4* xs = reinterpret_cast<float4*>(smem) + (size_t)ks * RM * 64;
    const int lane = tid & 63, wave = tid >> 6;
    const int n0 = (blockIdx.x * (TPB / 64) + wave) * CW;
    float v[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] = 0.f;
    constexpr int SI = RM * 64 / TPB;
    constexpr bool PREFETCH = KS < 4;
    float4 stg[SI];
    if (PREFETCH) x_load<SI>(stg, X, ldx, M, K, ks * KC, tid);
    const int rounds = (K + KC * KS - 1) / (KC * KS);                           // the same for every slice: barriers match
    for (int r = 0; r < rounds; ++r) {
        const int kc = (r * KS + ks) * KC;                                      // >= K: an all-zero piece
        if (!PREFETCH) x_load<SI>(stg, X, ldx, M, K, kc, tid);
        const int k = kc + lane * 4;
        float4 w4[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c)
            w4[c] = (k < K && n0 + c < N) ? *reinterpret_cast<const float4*>(W + (long)(n0 + c) * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        x_store<SI>(xs, stg, tid);
        __syncthreads();
        if (PREFETCH && r + 1 < rounds) x_load<SI>(stg, X, ldx, M, K, kc + KC * KS, tid);
        if (kc < K) {
#pragma unroll
            for (int m = 0; m < RM; ++m)
                if (m < M) {
                    const float4 x4 = xs[m * 64 + lane];
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        float a = v[c * RM + m];
                        a = fmaf(w4[c].x, x4.x, a);
                        a = fmaf(w4[c].y, x4.y, a);
                        a = fmaf(w4[c].z, x4.z, a);
                        a = fmaf(w4[c].w, x4.w, a);
                        v[c * RM + m] = a;
                    }
                }
        }
    }
    float hp = halve64(v, lane);
    // the KS partial results of (column, row): through LDS, added in slice order by slice 0
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                                // [KS][TPB]
    red[ks * TPB + tid] = hp;
    __syncthreads();
    if (ks != 0) return;
    float h = red[tid];
#pragma unroll
    for (int q = 1; q < KS; ++q) h += red[q * TPB + tid];
    const int c = lane / RM, m = lane - c * RM, n = n0 + c;
    const bool col = n < N, ok = col && m < M;
    if (mode == 0) {
        if (bias && col) h += bias[n];
        if (ok) {
            H[(long)m * ldh + n] = h;
            if (Y != H) Y[(long)m * ldy + n] = h;
        }
        return;
    }
    float scale, shift;
    if (mode == 1) {                                       // batch statistics over the M rows of the column (BnFin)
        const double hv = ok ? (double)h : 0.0;
        const double s0 = group_sum<RM>(hv), s1 = group_sum<RM>(hv * hv);
        const double mu = s0 / (double)M;
        double var = s1 / (double)M - mu * mu;
        if (var < 0) var = 0;
        const double is = 1.0 / sqrt(var + (double)eps);
        const float g = (gamma && col) ? gamma[n] : 1.f, b = (beta && col) ? beta[n] : 0.f;
        scale = (float)(g * is);
        shift = (float)(b - mu * g * is);
        if (m == 0 && col) {
            coef[n] = (float)mu;
            coef[N + n] = (float)is;
            coef[2 * N + n] = scale;
            coef[3 * N + n] = shift;
            if (rmean) {
                const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
                rmean[n] = (float)((1.0 - momentum) * rmean[n] + momentum * mu);
                rvar[n] = (float)((1.0 - momentum) * rvar[n] + momentum * unb);
            }
        }
    } else {                                               // running statistics (bn_eval_coeffs_kernel)
        const float is = col ? 1.f / sqrtf(rvar[n] + eps) : 0.f;
        const float g = (gamma && col) ? gamma[n] : 1.f, b = (beta && col) ? beta[n] : 0.f;
        const float mu = col ? rmean[n] : 0.f;
        scale = g * is;
        shift = b - mu * g * is;
        if (m == 0 && col) {
            coef[n] = mu;
            coef[N + n] = is;
            coef[2 * N + n] = scale;
            coef[3 * N + n] = shift;
        }
    }
    if (ok) {
        H[(long)m * ldh + n] = h;
        float y = act(fmaf(scale, h, shift), slope);
        if (drop_p > 0.f) {                                // Dropout(p) behind the block: mask kept for the backward pass
            const bool keep = dcnn::dropout_keep(seed, *step, salt, (unsigned)(m * N + n), drop_p);
            mask[(long)m * N + n] = keep ? 1 : 0;
            y = keep ? y * (1.f / (1.f - drop_p)) : 0.f;
        }
        Y[(long)m * ldy + n] = y;
    }
}

template <int CW>
__global__ __launch_bounds__(TPB * KSlices<CW>::KSB) void rowblock_bwd_kernel(
    const float* __restrict__ dY, long lddy, const float* __restrict__ H, long ldh, const float* __restrict__ coef,
    const float* __restrict__ gamma, float slope, int mode, const float* __restrict__ X, long ldx, int M, int N, int K,
    float* __restrict__ dW, long lddw, float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dH, long lddh, const unsigned char* __restrict__ mask, float keep_scale) {
    constexpr int RM = RB<CW>::RM, KS = KSlices<CW>::KSB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ks = threadIdx.x / TPB, tid = threadIdx.x - ks * TPB;
    float4* xs = reinterpret_cast<float4*>(smem) + (size_t)ks * RM * 64;
    const int lane = tid & 63, wave = tid >> 6;
    const int n0 = (blockIdx.x * (TPB / 64) + wave) * CW;
    const int c = lane / RM, m = lane - c * RM, n = n0 + c;
    const bool col = n < N, ok = col && m < M;
    float dy = ok ? dY[(long)m * lddy + n] : 0.f;
    if (mask && ok) dy = mask[(long)m * N + n] ? dy * keep_scale : 0.f;      // backward of the Dropout behind the block
    float dh;
    if (mode == 0) {
        dh = dy;
        const double s0 = group_sum<RM>((double)dy);
        if (dbias && m == 0 && col && ks == 0) dbias[n] = (float)s0;
    } else {
        const float h = ok ? H[(long)m * ldh + n] : 0.f;
        const float mu = col ? coef[n] : 0.f, is = col ? coef[N + n] : 0.f, sc = col ? coef[2 * N + n] : 0.f,
                    sh = col ? coef[3 * N + n] : 0.f;
        float dz, dzx;
        dcnn::bn_bwd_terms(dy, h, sc, sh, mu, is, slope, dz, dzx);
        if (!ok) { dz = 0.f; dzx = 0.f; }
        const double s0 = group_sum<RM>((double)dz), s1 = group_sum<RM>((double)dzx);
        if (m == 0 && col && ks == 0) {
            if (dbeta) dbeta[n] = (float)s0;
            if (dgamma) dgamma[n] = (float)s1;
        }
        const float gi = ((gamma && col) ? gamma[n] : 1.f) * is;
        dh = ok ? dcnn::bn_bwd_dh(dy, h, sc, sh, mu, is, slope, gi, (float)(s0 / (double)M), (float)(s1 / (double)M), mode == 1) : 0.f;
    }
    if (ok && ks == 0) dH[(long)m * lddh + n] = dh;
    if (!dW) return;
    // dW[n, :] = sum_m dh[m, n] X[m, :]: dh of every row of the wavefront's columns, broadcast from its lane
    float dhr[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) dhr[j] = __shfl(dh, j, 64);
    constexpr int SI = RM * 64 / TPB;
    float4 stg[SI];
    x_load<SI>(stg, X, ldx, M, K, ks * KC, tid);
    const int rounds = (K + KC * KS - 1) / (KC * KS);
    for (int r = 0; r < rounds; ++r) {
        const int kc = (r * KS + ks) * KC;
        __syncthreads();
        x_store<SI>(xs, stg, tid);
        __syncthreads();
        if (r + 1 < rounds) x_load<SI>(stg, X, ldx, M, K, kc + KC * KS, tid);
        const int k = kc + lane * 4;
        if (k >= K) continue;
        float4 acc[CW];
#pragma unroll
        for (int cc = 0; cc < CW; ++cc) acc[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mm = 0; mm < RM; ++mm)
            if (mm < M) {
                const float4 x4 = xs[mm * 64 + lane];
#pragma unroll
                for (int cc = 0; cc < CW; ++cc) {
                    const float d = dhr[cc * RM + mm];
                    acc[cc].x = fmaf(d, x4.x, acc[cc].x);
                    acc[cc].y = fmaf(d, x4.y, acc[cc].y);
                    acc[cc].z = fmaf(d, x4.z, acc[cc].z);
                    acc[cc].w = fmaf(d, x4.w, acc[cc].w);
                }
            }
#pragma unroll
        for (int cc = 0; cc < CW; ++cc)
            if (n0 + cc < N) *reinterpret_cast<float4*>(dW + (long)(n0 + cc) * lddw + k) = acc[cc];
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

DC_EXPORT int32_t dc_rowblock_max_rows(void) { return 64; }

// H[M,N] = X[M,K] W[N,K]^T (+ bias); mode 1 / 2: coef[4,N] = (mean, invstd, scale, shift) of the BatchNorm over the M
// rows (batch statistics, running statistics updated when given / running statistics), Y = leaky_slope(scale H + shift).
// mode 0: Y = H (Y may alias H).  M <= dc_rowblock_max_rows(), K % 4 == 0, 16-byte aligned rows of X and W.
static int rowblock_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, int32_t M,
                            int32_t N, int32_t K, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, int32_t mode, float slope, float* H, int64_t ldh,
                            float* coef, float* Y, int64_t ldy, float drop_p, unsigned seed, const long long* step,
                            unsigned salt, unsigned char* mask, void* stream) {
    DC_REQUIRE(X && W && H && Y, "dc_rowblock_forward: null pointer");
    DC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || (mode != 0 && step && mask)),
               "dc_rowblock_forward_dropout: 0 <= p < 1, a BatchNorm block, its step counter and a mask buffer");
    DC_REQUIRE(M >= 1 && M <= 64 && N >= 1 && K >= 4 && K % 4 == 0 && ldx >= K && ldw >= K && ldh >= N && ldy >= N,
               "dc_rowblock_forward: bad size (1 <= M <= 64, K %% 4 == 0)");
    DC_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && aligned16(X) && aligned16(W), "dc_rowblock_forward: rows must be 16-byte aligned");
    DC_REQUIRE(mode >= 0 && mode <= 2 && (mode == 0 || coef) && (mode != 2 || (running_mean && running_var)) &&
                   (mode != 1 || !running_mean == !running_var),
               "dc_rowblock_forward: bad mode / missing BatchNorm arguments");
    DC_REQUIRE(mode != 1 || M > 1, "dc_rowblock_forward: batch statistics need more than one row");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = 128 * 1024;              // KS staging areas: 4 x [32][64] or 2 x [64][64] float4
    static unsigned long long attr2 = 0, attr1 = 0;
    if (!(M <= 32 ? dc_ensure_lds(&attr2, reinterpret_cast<const void*>(&rowblock_fwd_kernel<2>), lds, "dc_rowblock_forward")
                  : dc_ensure_lds(&attr1, reinterpret_cast<const void*>(&rowblock_fwd_kernel<1>), lds, "dc_rowblock_forward"))) {
        DC_CHECK_LAUNCH("dc_rowblock_forward");
    }
    if (M <= 32)
        hipLaunchKernelGGL((rowblock_fwd_kernel<2>), dim3(dc_cdiv(N, 8)), dim3(TPB * 4), lds, s, X, (long)ldx, W, (long)ldw, bias, M, N, K,
                           gamma, beta, eps, momentum, running_mean, running_var, mode, slope, H, (long)ldh, coef, Y, (long)ldy, drop_p,
                           seed, step, salt, mask);
    else
        hipLaunchKernelGGL((rowblock_fwd_kernel<1>), dim3(dc_cdiv(N, 4)), dim3(TPB * 2), lds, s, X, (long)ldx, W, (long)ldw, bias, M, N, K,
                           gamma, beta, eps, momentum, running_mean, running_var, mode, slope, H, (long)ldh, coef, Y, (long)ldy, drop_p,
                           seed, step, salt, mask);
    DC_CHECK_LAUNCH("dc_rowblock_forward");
    return DC_OK;
}

DC_EXPORT int dc_rowblock_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, int32_t M,
                                  int32_t N, int32_t K, const float* gamma, const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var, int32_t mode, float slope, float* H, int64_t ldh,
                                  float* coef, float* Y, int64_t ldy, void* stream) {
    return rowblock_forward(X, ldx, W, ldw, bias, M, N, K, gamma, beta, eps, momentum, running_mean, running_var, mode, slope, H,
                            ldh, coef, Y, ldy, 0.f, 0u, nullptr, 0u, nullptr, stream);
}

// The block followed by torch.nn.Dropout(p) (deltanet_classification.py:34-36) in the same launch: Y = dropout(act(bn(X W^T))),
// mask[M, N] (1 = kept) for dc_rowblock_backward_dropout.  Draws: Philox-4x32-10 keyed by `seed`, counter (element, salt,
// *step) -- `step` = the block's BatchNorm num_batches_tracked on the device (advances once per training step: a new mask in
// every replay of a captured graph), `salt` tells the dropout layers of a model apart.
DC_EXPORT int dc_rowblock_forward_dropout(const float* X, int64_t ldx, const float* W, int64_t ldw, int32_t M, int32_t N,
                                          int32_t K, const float* gamma, const float* beta, float eps, float momentum,
                                          float* running_mean, float* running_var, int32_t mode, float slope, float* H,
                                          int64_t ldh, float* coef, float* Y, int64_t ldy, float p, int32_t seed,
                                          const int64_t* step, int32_t salt, uint8_t* mask, void* stream) {
    return rowblock_forward(X, ldx, W, ldw, nullptr, M, N, K, gamma, beta, eps, momentum, running_mean, running_var, mode, slope,
                            H, ldh, coef, Y, ldy, p, (unsigned)seed, reinterpret_cast<const long long*>(step), (unsigned)salt,
                            mask, stream);
}

// Backward of dc_rowblock_forward up to the input gradient: dH[M,N] = BatchNorm / activation backward of dY (mode 0:
// dH = dY), d gamma / d beta (mode 0: d bias) and dW[N,K] = dH^T X (skipped when dW is NULL).  d X = dH W is
// dc_linear_backward_input(dH, W).
static int rowblock_backward(const float* dY, int64_t lddy, const float* H, int64_t ldh, const float* coef,
                             const float* gamma, float slope, int32_t mode, const float* X, int64_t ldx, int32_t M,
                             int32_t N, int32_t K, float* dW, int64_t lddw, float* dbias, float* dgamma, float* dbeta,
                             float* dH, int64_t lddh, const unsigned char* mask, float keep_scale, void* stream) {
    DC_REQUIRE(dY && dH && (mode == 0 || (H && coef)) && (!dW || X), "dc_rowblock_backward: null pointer");
    DC_REQUIRE(M >= 1 && M <= 64 && N >= 1 && K >= 4 && K % 4 == 0 && lddy >= N && lddh >= N && (!dW || (ldx >= K && lddw >= K)),
               "dc_rowblock_backward: bad size (1 <= M <= 64, K %% 4 == 0)");
    DC_REQUIRE(!dW || (ldx % 4 == 0 && lddw % 4 == 0 && aligned16(X) && aligned16(dW)), "dc_rowblock_backward: rows must be 16-byte aligned");
    DC_REQUIRE(mode >= 0 && mode <= 2, "dc_rowblock_backward: bad mode");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = 128 * 1024;
    static unsigned long long attr2 = 0, attr1 = 0;
    if (!(M <= 32 ? dc_ensure_lds(&attr2, reinterpret_cast<const void*>(&rowblock_bwd_kernel<2>), lds, "dc_rowblock_backward")
                  : dc_ensure_lds(&attr1, reinterpret_cast<const void*>(&rowblock_bwd_kernel<1>), lds, "dc_rowblock_backward"))) {
        DC_CHECK_LAUNCH("dc_rowblock_backward");
    }
    if (M <= 32)
        hipLaunchKernelGGL((rowblock_bwd_kernel<2>), dim3(dc_cdiv(N, 8)), dim3(TPB * 2), lds, s, dY, (long)lddy, H, (long)ldh, coef, gamma,
                           slope, mode, X, (long)ldx, M, N, K, dW, (long)lddw, dbias, dgamma, dbeta, dH, (long)lddh, mask, keep_scale);
    else
        hipLaunchKernelGGL((rowblock_bwd_kernel<1>), dim3(dc_cdiv(N, 4)), dim3(TPB * 2), lds, s, dY, (long)lddy, H, (long)ldh, coef, gamma,
                           slope, mode, X, (long)ldx, M, N, K, dW, (long)lddw, dbias, dgamma, dbeta, dH, (long)lddh, mask, keep_scale);
    DC_CHECK_LAUNCH("dc_rowblock_backward");
    return DC_OK;
}

DC_EXPORT int dc_rowblock_backward(const float* dY, int64_t lddy, const float* H, int64_t ldh, const float* coef,
                                   const float* gamma, float slope, int32_t mode, const float* X, int64_t ldx, int32_t M,
                                   int32_t N, int32_t K, float* dW, int64_t lddw, float* dbias, float* dgamma, float* dbeta,
                                   float* dH, int64_t lddh, void* stream) {
    return rowblock_backward(dY, lddy, H, ldh, coef, gamma, slope, mode, X, ldx, M, N, K, dW, lddw, dbias, dgamma, dbeta, dH, lddh,
                             nullptr, 1.f, stream);
}

// Backward of dc_rowblock_forward_dropout: dY is the gradient BEHIND the dropout; mask / p as in the forward call.
DC_EXPORT int dc_rowblock_backward_dropout(const float* dY, int64_t lddy, const float* H, int64_t ldh, const float* coef,
                                           const float* gamma, float slope, int32_t mode, const float* X, int64_t ldx, int32_t M,
                                           int32_t N, int32_t K, float* dW, int64_t lddw, float* dgamma, float* dbeta, float* dH,
                                           int64_t lddh, const uint8_t* mask, float p, void* stream) {
    DC_REQUIRE(mask && p > 0.f && p < 1.f && mode != 0, "dc_rowblock_backward_dropout: mask, 0 < p < 1, a BatchNorm block");
    return rowblock_backward(dY, lddy, H, ldh, coef, gamma, slope, mode, X, ldx, M, N, K, dW, lddw, nullptr, dgamma, dbeta, dH, lddh,
                             mask, 1.f / (1.f - p), stream);
}
