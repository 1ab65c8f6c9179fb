// Per-row formula of the training loss (loss.hip), shared with the CPU host-check build.
//
// Reference being restated: /root/reference/experiments/utils.py:7-24 (`calc_loss`): cross entropy
// against the smoothed target q = (1-eps) on the label, eps/(C-1) elsewhere (eps = 0.2 for the
// classification scripts), or plain mean cross entropy (eps = 0: `F.cross_entropy`, segmentation).
//   loss_r = -sum_c q_c * logp_c,   logp = log_softmax(x_r),   d loss / d x_rc = (softmax_c - q_c) / R
#pragma once
#include "point_math.h"

namespace dcloss {

// Element-level pieces used by the wave-per-row kernel (same formulas, reductions done by the wave).
DC_HD float ce_q_off(int C, float eps) { return C > 1 ? eps / (float)(C - 1) : 0.f; }
DC_HD float ce_grad(float x, float lse, float q, float inv_rows) { return (expf(x - lse) - q) * inv_rows; }
DC_HD float ce_row_loss(float x_label, float sum_x, float lse, int C, float eps) {
    const float q_off = ce_q_off(C, eps), q_on = 1.f - eps;
    return -((q_on - q_off) * (x_label - lse) + q_off * (sum_x - (float)C * lse));
}

// One row: returns loss_r (not yet divided by R) and writes dx[c] = (p_c - q_c) * inv_rows.
// label outside [0, C) poisons the row with NaN (ATen would device-assert).
DC_HD float ce_row(const float* x, int C, long label, float eps, float inv_rows, float* dx) {
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float se = 0.f, sx = 0.f;
    for (int c = 0; c < C; ++c) {
        se += expf(x[c] - m);
        sx += x[c];
    }
    const float lse = m + logf(se);
    const bool ok = label >= 0 && label < C;
    const float bad = ok ? 0.f : nanf("");
    const float q_off = ce_q_off(C, eps), q_on = 1.f - eps;
    for (int c = 0; c < C; ++c) dx[c] = ce_grad(x[c], lse, (ok && c == label) ? q_on : q_off, inv_rows) + bad;
    // -sum q logp = -(q_on - q_off) * logp_y - q_off * sum_c logp_c
    return ce_row_loss(ok ? x[label] : 0.f, sx, lse, C, eps) + bad;
}

}  // namespace dcloss
