// Training loss in two launches (one for <= 64 rows, round 6): label-smoothed / plain mean cross entropy and its gradient.
// Replaces ~25 tiny ATen launches (log_softmax, gather, sums, mean and their backward) of
// /root/reference/experiments/utils.py:7-24 per step; at [32 x 40] logits those are pure launch latency.
//   ce_rows_kernel : wave = row (lanes stride the classes), writes d loss / d logits and one fp64
//                    partial per block (wave butterflies + fixed-order sums: bit-reproducible)
//   ce_final_kernel: one wave sums the block partials in a fixed order -> loss = sum / R
// Bound: latency.  Bytes: 8 R C + 8 R.
#include "common.h"
#include "loss_math.h"

namespace {

constexpr int WAVES = 4;
// rows one wave walks: a row is a chain of dependent steps (max -> exp-sum -> log -> gradient, each behind a global load
// or a wave reduction: ~1.2 us), so few rows (classification: B logits rows) get one wave each -- 16 rows per wave cost
// 22 us for the [32 x 40] logits of the bench step -- and only per-point logits (segmentation) amortise a longer walk
inline int rows_per_wave(long R) { return R >= 16384 ? 16 : 1; }

// wave = row (lanes stride the classes, coalesced); each wave walks rpw rows
__global__ __launch_bounds__(64 * WAVES) void ce_rows_kernel(const float* __restrict__ x, long ldx,
                                                             const long* __restrict__ label, long R, int C, float eps,
                                                             float* __restrict__ dx, long lddx,
                                                             double* __restrict__ partial, int rpw) {
    __shared__ double sm[WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inv_rows = 1.f / (float)R, q_off = dcloss::ce_q_off(C, eps), q_on = 1.f - eps;
    double acc = 0.0;
    for (int i = 0; i < rpw; ++i) {
        const long r = ((long)blockIdx.x * WAVES + w) * rpw + i;
        if (r >= R) break;   // wave-uniform
        const float* xr = x + r * ldx;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
        m = dc_wave_max(m);
        float se = 0.f, sx = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = xr[c];
            se += expf(v - m);
            sx += v;
        }
        se = dc_wave_sum(se);
        sx = dc_wave_sum(sx);
        const float lse = m + logf(se);
        const long y = label[r];
        const bool ok = y >= 0 && y < C;
        const float bad = ok ? 0.f : nanf("");
        for (int c = lane; c < C; c += 64)
            dx[r * lddx + c] = dcloss::ce_grad(xr[c], lse, (ok && c == y) ? q_on : q_off, inv_rows) + bad;
        acc += (double)(dcloss::ce_row_loss(ok ? xr[y] : 0.f, sx, lse, C, eps) + bad);   // lane-uniform
    }
    if (lane == 0) sm[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) s += sm[q];
        partial[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(64) void ce_final_kernel(const double* __restrict__ partial, int nb, long R,
                                                      float* __restrict__ loss) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) s += partial[b];
    s = dc_wave_sum(s);
    if (threadIdx.x == 0) *loss = (float)(s / (double)R);
}

// <= 64 rows (the classification nets: one logits row per cloud): ONE workgroup, wave = row(s), the row losses meet in LDS and
// one wave adds them in exactly the association of the two-launch form below (blocks of WAVES rows in row order, lane b the
// blocks b, b + 64, ..., then the butterfly): same bits, one launch less in a launch-bound corner of the step.
constexpr int SMALL_ROWS = 64, SMALL_WAVES = 16;
__global__ __launch_bounds__(64 * SMALL_WAVES) void ce_small_kernel(const float* __restrict__ x, long ldx,
                                                                    const long* __restrict__ label, int R, int C, float eps,
                                                                    float* __restrict__ dx, long lddx, float* __restrict__ loss) {
    __shared__ double rowloss[SMALL_ROWS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inv_rows = 1.f / (float)R, q_off = dcloss::ce_q_off(C, eps), q_on = 1.f - eps;
    for (int r = w; r < R; r += SMALL_WAVES) {
        const float* xr = x + (long)r * ldx;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
        m = dc_wave_max(m);
        float se = 0.f, sx = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = xr[c];
            se += expf(v - m);
            sx += v;
        }
        se = dc_wave_sum(se);
        sx = dc_wave_sum(sx);
        const float lse = m + logf(se);
        const long y = label[r];
        const bool ok = y >= 0 && y < C;
        const float bad = ok ? 0.f : nanf("");
        for (int c = lane; c < C; c += 64)
            dx[(long)r * lddx + c] = dcloss::ce_grad(xr[c], lse, (ok && c == y) ? q_on : q_off, inv_rows) + bad;
        if (lane == 0) rowloss[r] = (double)(dcloss::ce_row_loss(ok ? xr[y] : 0.f, sx, lse, C, eps) + bad);
    }
    __syncthreads();
    if (w == 0) {
        const int nb = (R + WAVES - 1) / WAVES;           // the blocks of the two-launch form (rows_per_wave = 1 here)
        double s = 0.0;
        for (int b = lane; b < nb; b += 64) {
            double part = 0.0;
            for (int q = 0; q < WAVES; ++q)
                if (b * WAVES + q < R) part += rowloss[b * WAVES + q];
            s += part;
        }
        s = dc_wave_sum(s);
        if (lane == 0) *loss = (float)(s / (double)R);
    }
}

}  // namespace

DC_EXPORT size_t dc_ce_loss_workspace_bytes(int64_t num_rows) {
    return (size_t)dc_cdiv(num_rows > 0 ? num_rows : 1, WAVES) * sizeof(double);     // one partial per block, rpw >= 1
}

DC_EXPORT int dc_ce_loss(const float* logits, int64_t ld_logits, const int64_t* labels, int64_t num_rows,
                         int32_t num_classes, float smoothing, float* loss, float* dlogits, int64_t ld_dlogits,
                         void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(logits && labels && loss && dlogits, "dc_ce_loss: null pointer");
    DC_REQUIRE(num_rows >= 1 && num_classes >= 1, "dc_ce_loss: empty input (the mean over zero rows is undefined)");
    DC_REQUIRE(ld_logits >= num_classes && ld_dlogits >= num_classes, "dc_ce_loss: leading dimension smaller than the row");
    DC_REQUIRE(smoothing >= 0.f && smoothing < 1.f, "dc_ce_loss: smoothing must be in [0, 1)");
    if (!workspace || workspace_bytes < dc_ce_loss_workspace_bytes(num_rows)) {
        dc_set_error("dc_ce_loss: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (num_rows <= SMALL_ROWS && !dc_option(DC_OPT_CE_TWO_LAUNCH)) {
        hipLaunchKernelGGL(ce_small_kernel, dim3(1), dim3(64 * SMALL_WAVES), 0, s, logits, (long)ld_logits,
                           reinterpret_cast<const long*>(labels), (int)num_rows, num_classes, smoothing, dlogits,
                           (long)ld_dlogits, loss);
        DC_CHECK_LAUNCH("dc_ce_loss");
        return DC_OK;
    }
    const int rpw = rows_per_wave(num_rows);
    const int nb = dc_cdiv(num_rows, WAVES * rpw);
    double* partial = static_cast<double*>(workspace);
    hipLaunchKernelGGL(ce_rows_kernel, dim3(nb), dim3(64 * WAVES), 0, s, logits, (long)ld_logits,
                       reinterpret_cast<const long*>(labels), (long)num_rows, num_classes, smoothing, dlogits,
                       (long)ld_dlogits, partial, rpw);
    hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(64), 0, s, partial, nb, (long)num_rows, loss);
    DC_CHECK_LAUNCH("dc_ce_loss");
    return DC_OK;
}
