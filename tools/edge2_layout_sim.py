"""Lane-level emulation (numpy) of the data flow of csrc/edge2.hip for ONE wavefront: validates the index algebra of
the chained v_mfma_f32_16x16x4_f32 products before any GPU time is spent.

  product 1:  y2^T[n, row]  = sum_c W2[n, c] h1[row, c]        (A = W2 fragments, B = h1 as held by the lanes)
  product 2:  du1^T[c, row] = sum_n W2[n, c] dy2[row, n]       (A = W2^T fragments, B = the D registers of product 1)
  product 3:  dW2[n, c]     = sum_row dy2[row, n] h1[row, c]   (through an LDS tile: rows become the K index)

Lane l = 16 q + j holds, for edge row j, the channels S(q) = {16 b + 4 q + r}: register [b][r].  The K index of every
product is permuted (allowed: a sum), so the D layout of one product IS the B layout of the next -- no transposition
between products 1 and 2.
"""
import numpy as np


def mfma16(a, b, c):
    """a, b: [64] per-lane operand; c: [4, 64] accumulator registers.  A[i][k] = a[16k+i], B[k][j] = b[16k+j],
    D[i][j] lives in register i % 4 of lane 16 (i // 4) + j."""
    A = a.reshape(4, 16).T            # [i, k]
    B = b.reshape(4, 16)              # [k, j]
    D = A @ B
    out = c.copy()
    for i in range(16):
        for j in range(16):
            out[i % 4, 16 * (i // 4) + j] += D[i, j]
    return out


def main():
    rng = np.random.default_rng(0)
    W2 = rng.standard_normal((64, 64))
    h1 = rng.standard_normal((16, 64))          # [row, c]
    lanes = np.arange(64)
    q, j = lanes // 16, lanes % 16
    # lane registers: reg[b][r][lane] = X[row j, channel 16b + 4q + r]
    def to_lanes(X):
        R = np.zeros((4, 4, 64))
        for b in range(4):
            for r in range(4):
                R[b, r] = X[j, 16 * b + 4 * q + r]
        return R

    def from_lanes(R):
        X = np.zeros((16, 64))
        for b in range(4):
            for r in range(4):
                X[j, 16 * b + 4 * q + r] = R[b, r]
        return X
    h1r = to_lanes(h1)
    # fragment tables, "lane l of fragment (mb, cb)" holds a float4 over r
    fragA1 = np.zeros((4, 4, 64, 4))            # [mb][cb][lane][r] = W2[16mb + i][16cb + 4q + r],  i = lane % 16, q = lane // 16
    fragA2 = np.zeros((4, 4, 64, 4))            # [cb][mb][lane][r] = W2[16mb + 4q + r][16cb + i]
    for mb in range(4):
        for cb in range(4):
            for r in range(4):
                fragA1[mb, cb, :, r] = W2[16 * mb + j, 16 * cb + 4 * q + r]
                fragA2[cb, mb, :, r] = W2[16 * mb + 4 * q + r, 16 * cb + j]
    # product 1
    acc = np.zeros((4, 4, 64))                  # [mb][r'][lane]
    for cb in range(4):
        for r in range(4):
            for mb in range(4):
                acc[mb] = mfma16(fragA1[mb, cb, :, r], h1r[cb, r], acc[mb])
    y2 = from_lanes(acc)                        # register [mb][r'] <-> n = 16mb + 4q + r'
    assert np.allclose(y2, h1 @ W2.T), "product 1 layout"
    # product 2 (B = any lane-held [row, n] quantity in the same register layout)
    dy2 = rng.standard_normal((16, 64))
    dy2r = to_lanes(dy2)
    acc2 = np.zeros((4, 4, 64))
    for mb in range(4):
        for r in range(4):
            for cb in range(4):
                acc2[cb] = mfma16(fragA2[cb, mb, :, r], dy2r[mb, r], acc2[cb])
    du1 = from_lanes(acc2)
    assert np.allclose(du1, dy2 @ W2), "product 2 layout"
    # product 3: tiles Ty[row, n], Th[row, c] of 64 rows (4 waves); wave w owns n in [16w, 16w+16)
    Ty, Th = rng.standard_normal((64, 64)), rng.standard_normal((64, 64))
    dW = np.zeros((64, 64))
    for w in range(4):
        acc3 = np.zeros((4, 4, 64))             # [cb][r'][lane]
        for ks in range(16):
            a = Ty[4 * ks + q, 16 * w + j]      # A[i = n - 16w][k = row - 4ks]
            for cb in range(4):
                b = Th[4 * ks + q, 16 * cb + j]
                acc3[cb] = mfma16(a, b, acc3[cb])
        # D[i][jj]: register i % 4, lane 16 (i // 4) + jj  ->  dW[16w + 4q + r'][16cb + j]
        for cb in range(4):
            for r in range(4):
                dW[16 * w + 4 * q + r, 16 * cb + j] = acc3[cb, r]
    assert np.allclose(dW, Ty.T @ Th), "product 3 layout"
    print("edge2 layouts OK")


if __name__ == "__main__":
    main()
