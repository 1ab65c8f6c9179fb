// Depth-2 centralised edge MLP + max aggregation of DeltaConv's first layer, without ATen and without the [E, C] forward
// tensors:   out[i] = max_s act2(bn2( W2 act1(bn1( W1 (x_j - x_i) )) )),  j = nbr[i, s],  BatchNorm over all E = N k edges.
// Reference: /root/reference/deltaconv/nn/deltaconv.py:50-52 with s_mlp_max = MLP([ci, 64, 64]) as built by
// /root/reference/experiments/train_shapenet.py:77-89 (mlp_depth = 2) -- index_select, two addmm on E rows, two
// native_batch_norm, two leaky_relu and a scatter_max, each with its backward, over 655 360 x 64 tensors at C4.
// Algebra pinned on the CPU in fp64 by tests/test_edge_mlp2_spec.py; lane algebra by tools/edge2_layout_sim.py.
//
//   z = x W1^T on N rows (W1 (x_j - x_i) = z_j - z_i); statistics of y1 = z_j - z_i: dc_edge_gather_stats (edge.hip).
//   FORWARD  (edge2_fwd_kernel, one pass over the edges): wave = 16 points, one MFMA block = (16 points) x (slot s):
//       h1 = act1(sc1 (z_j - z_i) + sh1) in registers (lane = (edge row, 16 channels)), y2^T = W2 h1^T on
//       v_mfma_f32_16x16x4_f32 (exact fp32, W2 fragments in LDS), and in the D registers: sum y2, sum y2^2 (BatchNorm-2
//       statistics) and the running max of sign(gamma2) y2 with its first slot.  act2(bn2(.)) is monotone per channel, so
//       out = act2(sc2 sel + sh2) needs no second pass.
//   BACKWARD (edge2_bwd_kernel, ONE recompute pass): d z2 sits on the selected edge of every (point, channel): the two
//       BatchNorm-2 sums come from [N, 64] data; per block h1, y2 are recomputed, dy2 = a hot + b + c y2 (per-column
//       coefficients), du1 = (dy2 W2) act1'(.) by a second MFMA product whose B operand IS the D layout of the first
//       (the K index of a product may be permuted: no transposition), dW2 += dy2^T h1 by a third product through an LDS
//       tile (rows become the K index; wave w owns 16 rows of dW2).  BatchNorm-1's backward is LINEAR in du1, so the pass
//       does not wait for its two sums: it writes du1 [E, 64] (the one edge-sized tensor of the whole block, scratch),
//       sum_s du1 and sum_s du1 y1 per point, and
//   edge2_scatter_kernel closes   dz_p = sc1 [ sum_in du1 - sum_s du1 - n1 (indeg - k) - n2 (colhat - rowhat) ]   with the
//       same closed forms of the mean / variance terms as the depth-1 kernel (edge_math.h), in-edges in ascending edge id:
//       bit-reproducible, no fp atomics.   Then dW1 = dz^T x, dx = dz W1 (gemm_tn.hip / gemm.hip).
// Bound: the fp32 matrix pipe (1 + 3 products of E x 64 x 64: 21.5 GFLOP at C4 = 137 us at 157 TFLOP/s).
#include "common.h"
#include "colreduce.h"
#include "nn_math.h"

namespace {
using namespace dccol;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CH = 64;             // channels of both blocks (the kernels are specialised: 64 x 64 weights = 16 KB of fragments)
constexpr int WPB = 4;             // waves per workgroup
constexpr int PPW = 16;            // points per wave = the N dimension of v_mfma_f32_16x16x4_f32
constexpr int PPB = WPB * PPW;     // 64 points per workgroup
constexpr int TLD = CH + 4;        // row stride of the LDS tiles of the third product (floats)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Lane l = 16 q + j of a wave holds, for edge row j, the channels 16 b + 4 q + r  (register [b][r]).
// fragA1[(mb * 4 + cb) * 64 + lane][r] = W2[16 mb + j][16 cb + 4 q + r]:   y2^T = W2 h1^T,  D register [mb][r'] <-> n = 16 mb + 4 q + r'
// fragA2[(cb * 4 + mb) * 64 + lane][r] = W2[16 mb + 4 q + r][16 cb + j]:   du1^T = W2^T dy2^T, D register [cb][r'] <-> c = 16 cb + 4 q + r'
__device__ __forceinline__ void load_frags(const float* __restrict__ W2, f32x4* fragA1, f32x4* fragA2) {
    for (int idx = threadIdx.x; idx < 16 * 64; idx += WPB * 64) {
        const int f = idx >> 6, l = idx & 63, hi = f >> 2, lo = f & 3, li = l & 15, lq = l >> 4;
        fragA1[idx] = ld4(W2 + (16 * hi + li) * CH + 16 * lo + 4 * lq);              // (mb, cb) = (hi, lo)
        if (fragA2) {
            f32x4 v;                                                                   // (cb, mb) = (hi, lo)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = W2[(16 * lo + 4 * lq + r) * CH + 16 * hi + li];
            fragA2[idx] = v;
        }
    }
}

struct Rows {   // the 16 channels of one row held by a lane
    f32x4 v[4];
};
__device__ __forceinline__ Rows load_row(const float* __restrict__ base, long row, int q) {
    Rows r;
    const float* p = base + row * CH + 4 * q;
#pragma unroll
    for (int b = 0; b < 4; ++b) r.v[b] = ld4(p + 16 * b);
    return r;
}
__device__ __forceinline__ Rows lds_row(const float* s, int q) {
    Rows r;
#pragma unroll
    for (int b = 0; b < 4; ++b) r.v[b] = *reinterpret_cast<const f32x4*>(s + 16 * b + 4 * q);
    return r;
}

// Source of y1 = W1 (x_j - x_i) for the lane's 16 channels.  CI = 0: rows of z = x W1^T (any ci), y1 = z_j - z_i.  CI = 1..3:
// the ci input channels themselves (positions, ci = 3, in both reference nets): y1 = W1 (x_j - x_i) evaluated per edge -- the
// reference's own order of operations (the difference of two nearby points is exact in fp32, z_j - z_i loses |z| / |y1| ~ 20x
// of that), 12 gathered bytes per neighbour instead of 256, and 3 fmaf per channel beside 64 MFMAs.
template <int CI>
struct EdgeSrc {
    Rows zi;                          // CI == 0
    float xi[CI > 0 ? CI : 1];        // CI > 0
    f32x4 w[CI > 0 ? CI : 1][4];      // W1[16 b + 4 q + r][d] as [d][b][r]
    const float* base; long ld;
    __device__ __forceinline__ void init(const float* src, long ld_, long row, const float* __restrict__ W1, int q) {
        base = src; ld = ld_;
        if constexpr (CI == 0) zi = load_row(src, row, q);
        else {
#pragma unroll
            for (int d = 0; d < CI; ++d) {
                xi[d] = src[row * ld + d];
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[d][b][r] = W1[(16 * b + 4 * q + r) * CI + d];
            }
        }
    }
    struct Nb { Rows z; float x[CI > 0 ? CI : 1]; };
    __device__ __forceinline__ Nb fetch(long row, int q) const {
        Nb nb;
        if constexpr (CI == 0) nb.z = load_row(base, row, q);
        else {
#pragma unroll
            for (int d = 0; d < CI; ++d) nb.x[d] = base[row * ld + d];
        }
        return nb;
    }
    __device__ __forceinline__ Rows y1(const Nb& nb) const {
        Rows y;
        if constexpr (CI == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) y.v[b] = nb.z.v[b] - zi.v[b];
        } else {
            float dlt[CI];
#pragma unroll
            for (int d = 0; d < CI; ++d) dlt[d] = nb.x[d] - xi[d];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = w[0][b][r] * dlt[0];
#pragma unroll
                    for (int d = 1; d < CI; ++d) a = fmaf(w[d][b][r], dlt[d], a);
                    y.v[b][r] = a;
                }
        }
        return y;
    }
};

// product 1: y2 (register [mb][r']) from h (register [cb][r])
__device__ __forceinline__ void product1(const f32x4* fragA1, int lane, const Rows& h, Rows& y2) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) y2.v[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        f32x4 a[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[mb] = fragA1[(mb * 4 + cb) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) y2.v[mb] = mfma4(a[mb][r], h.v[cb][r], y2.v[mb]);
    }
}

// block-level ordered reduction of two per-lane quantities over the 64 rows of the workgroup -> partial[(q * CH + c) * chunks + blk]
__device__ __forceinline__ void block_sums(float (*red)[2][CH], int wave, int q, int j, const Rows& s0, const Rows& s1, bool live,
                                           double* __restrict__ partial, int chunks, long blk) {
    const int row = wave * PPW + j;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&red[row][0][16 * b + 4 * q]) = live ? s0.v[b] : zero;
        *reinterpret_cast<f32x4*>(&red[row][1][16 * b + 4 * q]) = live ? s1.v[b] : zero;
    }
    __syncthreads();
    if (threadIdx.x < 2 * CH) {
        const int qn = threadIdx.x >> 6, c = threadIdx.x & 63;
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < PPB; ++r) s += (double)red[r][qn][c];
        partial[((long)qn * CH + c) * chunks + blk] = s;
    }
}

// ---- BatchNorm-1 statistics of y1 = W1 (x_j - x_i) from the MOMENTS of the edge differences (ci <= 3) --------------------------
// y1 is linear in d = x_j - x_i:  sum_e y1[c] = W1[c] . (sum_e d),  sum_e y1[c]^2 = W1[c]^T (sum_e d d^T) W1[c]  -- ci + ci (ci + 1) / 2
// numbers (9 for positions) replace the gather pass over 64-channel rows of z = x W1^T (31 us at C4) and the product that makes z.
// Per point the pass also keeps sd = sum_s d (ci floats): the closed forms of the backward pass need sum_s y1 = W1 sd.
// Ordered fp64 reductions (block: fixed serial order over the 256 threads; final: serial over the blocks): bit-reproducible.
constexpr int MOM_TPB = 256;
template <int CI>
__global__ __launch_bounds__(MOM_TPB) void edge2_moments_kernel(const float* __restrict__ x, long ldx, const int* __restrict__ nbr, long n,
                                                                int k, float* __restrict__ sd, double* __restrict__ partial, int chunks) {
    constexpr int NQ = CI + CI * (CI + 1) / 2;
    __shared__ double sm[NQ][MOM_TPB];
    const long p = (long)blockIdx.x * MOM_TPB + threadIdx.x;
    double q[NQ];
#pragma unroll
    for (int a = 0; a < NQ; ++a) q[a] = 0.0;
    if (p < n) {
        float xi[CI], acc[CI];
#pragma unroll
        for (int d = 0; d < CI; ++d) { xi[d] = x[p * ldx + d]; acc[d] = 0.f; }
        const int* ids = nbr + p * k;
#pragma unroll 4
        for (int s = 0; s < k; ++s) {                     // (4 neighbour ids + rows in flight: the pass is load latency)
            const long jn = ids[s];
            float dl[CI];
#pragma unroll
            for (int d = 0; d < CI; ++d) { dl[d] = x[jn * ldx + d] - xi[d]; acc[d] += dl[d]; q[d] += (double)dl[d]; }
            int o = CI;
#pragma unroll
            for (int a = 0; a < CI; ++a)
#pragma unroll
                for (int b = a; b < CI; ++b) q[o++] += (double)dl[a] * (double)dl[b];
        }
#pragma unroll
        for (int d = 0; d < CI; ++d) sd[p * CI + d] = acc[d];
    }
#pragma unroll
    for (int a = 0; a < NQ; ++a) sm[a][threadIdx.x] = q[a];
    __syncthreads();
    if (threadIdx.x < NQ) {
        double t = 0.0;
        for (int i = 0; i < MOM_TPB; ++i) t += sm[threadIdx.x][i];
        partial[(long)threadIdx.x * chunks + blockIdx.x] = t;
    }
}
// one workgroup of 64 threads: moments = ordered sums of the partials, then channel c: mean, invstd, scale, shift (+ running statistics)
template <int CI>
__global__ __launch_bounds__(CH) void edge2_bn1_kernel(const double* __restrict__ partial, int chunks, long E, const float* __restrict__ W1,
                                                       BnFin fin) {
    constexpr int NQ = CI + CI * (CI + 1) / 2;
    __shared__ double mom[NQ];
    __shared__ double lanes[NQ][CH];
    // ordered in two levels: lane l of quantity q sums the partials l, l + 64, ... ascending, then one thread the 64 lane sums
    // ascending (the first version -- one thread walking all partials of a quantity -- took 12 us for 128 x 9 doubles)
    for (int q = 0; q < NQ; ++q) {
        double t = 0.0;
        for (int b = threadIdx.x; b < chunks; b += CH) t += partial[(long)q * chunks + b];
        lanes[q][threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        double t = 0.0;
#pragma unroll 8
        for (int l = 0; l < CH; ++l) t += lanes[threadIdx.x][l];
        mom[threadIdx.x] = t;
    }
    __syncthreads();
    const int c = threadIdx.x;
    double w[CI];
#pragma unroll
    for (int d = 0; d < CI; ++d) w[d] = (double)W1[c * CI + d];
    double s0 = 0.0, s1 = 0.0;
    int o = CI;
#pragma unroll
    for (int a = 0; a < CI; ++a) {
        s0 += w[a] * mom[a];
#pragma unroll
        for (int b = a; b < CI; ++b) s1 += (a == b ? 1.0 : 2.0) * w[a] * w[b] * mom[o++];
    }
    fin.R = E;
    fin(c, s0, s1);          // s0 = sum_e y1, s1 = sum_e y1^2: the finaliser of every BatchNorm of the library (colreduce.h)
}

// ---- forward ---------------------------------------------------------------------------------------------------------
template <int CI>
__global__ __launch_bounds__(WPB * 64, 2) void edge2_fwd_kernel(const float* __restrict__ src, long ldsrc, const float* __restrict__ W1,
                                                             const int* __restrict__ nbr, long n, int k,
                                                             const float* __restrict__ W2, const float* __restrict__ scale1,
                                                             const float* __restrict__ shift1, float slope1,
                                                             const float* __restrict__ gamma2, float* __restrict__ ysel,
                                                             unsigned char* __restrict__ arg, double* __restrict__ partial,
                                                             int chunks, int remap) {
    __shared__ f32x4 fragA1[16 * 64];
    __shared__ __attribute__((aligned(16))) float s_sc[CH], s_sh[CH], s_sg[CH];
    __shared__ __attribute__((aligned(16))) float red[PPB][2][CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
    const long blk = dc_xcd_block(remap);
    load_frags(W2, fragA1, nullptr);
    if (threadIdx.x < CH) {
        s_sc[threadIdx.x] = scale1[threadIdx.x];
        s_sh[threadIdx.x] = shift1[threadIdx.x];
        s_sg[threadIdx.x] = (gamma2 && gamma2[threadIdx.x] < 0.f) ? -1.f : 1.f;
    }
    __syncthreads();
    const long p = blk * PPB + wave * PPW + j;
    const bool live = p < n;
    const long pc = live ? p : n - 1;
    const Rows sc = lds_row(s_sc, q), sh = lds_row(s_sh, q), sg = lds_row(s_sg, q);
    EdgeSrc<CI> es;
    es.init(src, ldsrc, pc, W1, q);
    const int* ids = nbr + pc * k;
    typename EdgeSrc<CI>::Nb zn = es.fetch(ids[0], q);
    int id_next = ids[k > 1 ? 1 : 0];
    Rows s0, s1, best;
    unsigned bestarg[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < 4; ++b) s0.v[b] = s1.v[b] = best.v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < k; ++s) {
        const Rows y1 = es.y1(zn);
        if (s + 1 < k) {
            zn = es.fetch(id_next, q);
            id_next = ids[s + 2 < k ? s + 2 : k - 1];
        }
        Rows h, y2;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) h.v[b][r] = dcnn::act(fmaf(sc.v[b][r], y1.v[b][r], sh.v[b][r]), slope1);
        product1(fragA1, lane, h, y2);
        const unsigned sbytes = (unsigned)s * 0x01010101u;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = y2.v[b][r];
                s0.v[b][r] += y;
                s1.v[b][r] = fmaf(y, y, s1.v[b][r]);
                const float m = sg.v[b][r] * y;
                const bool up = s == 0 || m > best.v[b][r];       // strict: ties keep the first slot
                best.v[b][r] = up ? m : best.v[b][r];
                const unsigned mask = up ? (0xffu << (8 * r)) : 0u;
                bestarg[b] = (bestarg[b] & ~mask) | (sbytes & mask);
            }
    }
    if (live) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = sg.v[b][r] * best.v[b][r];
            *reinterpret_cast<f32x4*>(ysel + p * CH + 16 * b + 4 * q) = o;
            *reinterpret_cast<unsigned*>(arg + p * CH + 16 * b + 4 * q) = bestarg[b];
        }
    }
    if (partial) block_sums(red, wave, q, j, s0, s1, live, partial, chunks, blk);
}

// ---- backward, the recompute pass -------------------------------------------------------------------------------------
// coefs2 = the five rows [scale2 | shift2 | a | c | b] of BwdCoefFin (colreduce.h):  dy2 = a hot + c y2 + b
template <int CI>
__global__ __launch_bounds__(WPB * 64, 2) void edge2_bwd_kernel(const float* __restrict__ src, long ldsrc, const float* __restrict__ W1,
                                                             const int* __restrict__ nbr, long n, int k,
                                                             const float* __restrict__ W2, const float* __restrict__ scale1,
                                                             const float* __restrict__ shift1, float slope1,
                                                             const float* __restrict__ coefs2, const float* __restrict__ dz2,
                                                             const unsigned char* __restrict__ arg, float* __restrict__ dU,
                                                             float* __restrict__ csum, double* __restrict__ partial,
                                                             float* __restrict__ dWpart, int chunks, int remap) {
    // 68 KiB of LDS (> the 64 KiB a static allocation may take): dynamic, carved here.  Two workgroups per CU.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* fragA1 = reinterpret_cast<f32x4*>(smem);
    f32x4* fragA2 = fragA1 + 16 * 64;
    float* s_sc = reinterpret_cast<float*>(fragA2 + 16 * 64);
    float *s_sh = s_sc + CH, *s_ca = s_sh + CH, *s_cc = s_ca + CH, *s_cb = s_cc + CH;
    float(*tiles)[PPB][TLD] = reinterpret_cast<float(*)[PPB][TLD]>(s_cb + CH);   // Ty = dy2, Th = h1 (also the scratch of block_sums)
    static_assert(sizeof(float) * 2 * PPB * TLD >= sizeof(float) * PPB * 2 * CH, "tiles double as the reduction scratch");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
    const long blk = dc_xcd_block(remap);
    load_frags(W2, fragA1, fragA2);
    if (threadIdx.x < CH) {
        const int c = threadIdx.x;
        s_sc[c] = scale1[c];
        s_sh[c] = shift1[c];
        s_ca[c] = coefs2[2 * CH + c];
        s_cc[c] = coefs2[3 * CH + c];
        s_cb[c] = coefs2[4 * CH + c];
    }
    __syncthreads();
    const long p = blk * PPB + wave * PPW + j;
    const bool live = p < n;
    const long pc = live ? p : n - 1;
    const int row = wave * PPW + j;
    const Rows dzv = load_row(dz2, pc, q);
    EdgeSrc<CI> es;
    es.init(src, ldsrc, pc, W1, q);
    unsigned argv[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) argv[b] = *reinterpret_cast<const unsigned*>(arg + pc * CH + 16 * b + 4 * q);
    const int* ids = nbr + pc * k;
    typename EdgeSrc<CI>::Nb zn = es.fetch(ids[0], q);
    int id_next = ids[k > 1 ? 1 : 0];
    Rows cs, cy;
    f32x4 dw[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) cs.v[b] = cy.v[b] = dw[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < k; ++s) {
        const Rows y1 = es.y1(zn);
        if (s + 1 < k) {
            zn = es.fetch(id_next, q);
            id_next = ids[s + 2 < k ? s + 2 : k - 1];
        }
        Rows h, y2;
        {
            const Rows sc = lds_row(s_sc, q), sh = lds_row(s_sh, q);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) h.v[b][r] = dcnn::act(fmaf(sc.v[b][r], y1.v[b][r], sh.v[b][r]), slope1);
        }
        product1(fragA1, lane, h, y2);
        {   // dy2 in place of y2
            const Rows ca = lds_row(s_ca, q), cc = lds_row(s_cc, q), cb = lds_row(s_cb, q);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hot = ((argv[b] >> (8 * r)) & 0xffu) == (unsigned)s ? dzv.v[b][r] : 0.f;
                    const float d = fmaf(ca.v[b][r], hot, fmaf(cc.v[b][r], y2.v[b][r], cb.v[b][r]));
                    y2.v[b][r] = live ? d : 0.f;
                }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            *reinterpret_cast<f32x4*>(&tiles[0][row][16 * b + 4 * q]) = y2.v[b];
            *reinterpret_cast<f32x4*>(&tiles[1][row][16 * b + 4 * q]) = h.v[b];
        }
        // product 2: du1 (register [cb][r']) = sum_n W2[n, c] dy2[n]
        Rows du;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) du.v[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            f32x4 a[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) a[cb] = fragA2[(cb * 4 + mb) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) du.v[cb] = mfma4(a[cb][r], y2.v[mb][r], du.v[cb]);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = du.v[b][r] * (h.v[b][r] > 0.f ? 1.f : slope1);      // act1'(pre1): h1 > 0  <=>  pre1 > 0
                du.v[b][r] = d;
                cs.v[b][r] += d;
                cy.v[b][r] = fmaf(d, y1.v[b][r], cy.v[b][r]);
            }
            if (live) *reinterpret_cast<f32x4*>(dU + (p * k + s) * CH + 16 * b + 4 * q) = du.v[b];
        }
        __syncthreads();                                              // the two tiles of all 64 rows are in LDS
        // product 3: wave w owns rows n in [16 w, 16 w + 16) of dW2; K = the 64 edge rows of this slot
#pragma unroll 4
        for (int ks = 0; ks < PPB / 4; ++ks) {
            const float a = tiles[0][4 * ks + q][16 * wave + j];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) dw[cb] = mfma4(a, tiles[1][4 * ks + q][16 * cb + j], dw[cb]);
        }
        __syncthreads();                                              // before the next slot overwrites the tiles
    }
    if (live) {
#pragma unroll
        for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(csum + p * CH + 16 * b + 4 * q) = cs.v[b];
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dWpart[blk * (CH * CH) + (16 * wave + 4 * q + r) * CH + 16 * cb + j] = dw[cb][r];
    block_sums(reinterpret_cast<float(*)[2][CH]>(&tiles[0][0][0]), wave, q, j, cs, cy, live, partial, chunks, blk);
}

// dW2[idx] = sum over the workgroups' partials in a fixed order: block = 16 outputs x 16 slices of the chunk range (each thread
// an ascending serial sum of its slice, loads independent of the adds), then the 16 slice sums ascending.  (First version: one
// thread per output walking all 512 partials = 138 us of dependent latency for 8 MB; profiles/r05b.)
__global__ __launch_bounds__(256) void edge2_dw_reduce_kernel(const float* __restrict__ part, int chunks, float* __restrict__ dW) {
    __shared__ double sm[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + o;
    const int per = (chunks + 15) / 16, b0 = sl * per, b1 = min(b0 + per, chunks);
    double s = 0.0;
#pragma unroll 8
    for (int b = b0; b < b1; ++b) s += (double)part[(long)b * (CH * CH) + idx];
    sm[sl][o] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sm[q][threadIdx.x];
        dW[blockIdx.x * 16 + threadIdx.x] = (float)t;
    }
}

// finaliser of the BatchNorm-1 backward sums:  s0 = sum_e du1,  s1 = sum_e du1 y1
struct Bn1BwdFin {
    long E; const float *mean, *invstd; int training; float *dgamma, *dbeta, *m1, *m2;
    __device__ void operator()(int c, double s0, double s1) const {
        const double dg = (double)invstd[c] * (s1 - (double)mean[c] * s0);          // sum du1 xhat1
        if (dbeta) dbeta[c] = (float)s0;
        if (dgamma) dgamma[c] = (float)dg;
        m1[c] = training ? (float)(s0 / (double)E) : 0.f;
        m2[c] = training ? (float)(dg / (double)E) : 0.f;
    }
};

// d z2 = dout act2'(sc2 sel + sh2) written to dzs, and its two BatchNorm-2 sums (dz2, dz2 xhat2)
template <int V>
struct Bn2BwdF {
    const float *dout, *ysel, *scale, *shift, *mean, *invstd; long lddo; float slope; float* dzs;
    __device__ void operator()(long i, int c0, double (&t)[2][V]) const {
        const FV<V> g = ldv<V>(dout + i * lddo + c0), y = ldv<V>(ysel + i * CH + c0);
        FV<V> dz;
#pragma unroll
        for (int q = 0; q < V; ++q) {
            float b;
            dcnn::bn_bwd_terms(g.v[q], y.v[q], scale[c0 + q], shift[c0 + q], mean[c0 + q], invstd[c0 + q], slope, dz.v[q], b);
            t[0][q] = dz.v[q];
            t[1][q] = b;
        }
        stv<V>(dzs + i * CH + c0, dz);
    }
};

// closing pass: thread = (target point, 4 channels); in-edges in ascending edge id (CSC order).  CI = 0: the mean / variance terms
// from rows of z = x W1^T (T = sum_in z_src, s1 = sum_s (z_j - z_i)); CI = 1..3: from the ci input channels themselves --
// indeg z_p - T_p = W1 (indeg x_p - sum_in x_src) and s1_p = W1 sd_p are linear in positions: 12 gathered bytes per in-edge
// instead of a 256-byte row, and no z at all.
template <int CI>
__global__ __launch_bounds__(256) void edge2_scatter_kernel(long n, int k, int remap, const int* __restrict__ tptr,
                                                            const int* __restrict__ tedge, const float* __restrict__ dU,
                                                            const float* __restrict__ src, long ldsrc, const float* __restrict__ W1,
                                                            const float* __restrict__ csum, const float* __restrict__ s1in,
                                                            const float* __restrict__ scale1, const float* __restrict__ mean1,
                                                            const float* __restrict__ invstd1, const float* __restrict__ m1,
                                                            const float* __restrict__ m2, int training, float* __restrict__ dz,
                                                            long lddz) {
    const long t = dc_xcd_block(remap) * 256 + threadIdx.x;
    if (t >= n * (CH / 4)) return;
    const long jp = t / (CH / 4);
    const int c0 = (int)(t % (CH / 4)) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, T = {0.f, 0.f, 0.f, 0.f};
    float tx[CI > 0 ? CI : 1];
#pragma unroll
    for (int d = 0; d < (CI > 0 ? CI : 1); ++d) tx[d] = 0.f;
    const int p0 = tptr[jp], p1 = tptr[jp + 1];
    for (int e0 = p0; e0 < p1; ++e0) {
        const long e = tedge[e0];
        const long i = e / k;
        const f32x4 d = ld4(dU + e * CH + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += d[r];
        if (training) {
            if constexpr (CI == 0) {
                const f32x4 zi = ld4(src + i * CH + c0);
#pragma unroll
                for (int r = 0; r < 4; ++r) T[r] += zi[r];
            } else {
#pragma unroll
                for (int d2 = 0; d2 < CI; ++d2) tx[d2] += src[i * ldsrc + d2];
            }
        }
    }
    const float indeg = (float)(p1 - p0);
    const f32x4 cs = ld4(csum + jp * CH + c0);
    f32x4 diff = {0.f, 0.f, 0.f, 0.f};                 // (indeg z_p - T_p) - s1_p  per channel
    if (training) {
        if constexpr (CI == 0) {
            const f32x4 zj = ld4(src + jp * CH + c0), s1 = ld4(s1in + jp * CH + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) diff[r] = (indeg * zj[r] - T[r]) - s1[r];
        } else {
            float v[CI];
#pragma unroll
            for (int d2 = 0; d2 < CI; ++d2) v[d2] = (indeg * src[jp * ldsrc + d2] - tx[d2]) - s1in[jp * CI + d2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = W1[(c0 + r) * CI] * v[0];
#pragma unroll
                for (int d2 = 1; d2 < CI; ++d2) a = fmaf(W1[(c0 + r) * CI + d2], v[d2], a);
                diff[r] = a;
            }
        }
    }
    f32x4 out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + r;
        float g = acc[r] - cs[r];
        if (training) {
            // col_hat - row_hat = invstd [ (indeg z_p - T_p - indeg mu) - (s1_p - k mu) ]
            const float hat = (diff[r] - (indeg - (float)k) * mean1[c]) * invstd1[c];
            g -= m1[c] * (indeg - (float)k) + m2[c] * hat;
        }
        out[r] = scale1[c] * g;
    }
    *reinterpret_cast<f32x4*>(dz + jp * lddz + c0) = out;
}

constexpr size_t BWD_LDS = 2 * 16 * 64 * 16 + 5 * CH * 4 + 2 * PPB * TLD * 4;
inline int edge2_chunks(long n) { return dc_cdiv(n, PPB); }
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
}  // namespace

// Workspace of dc_edge2_forward / dc_edge2_backward (bytes): statistics partials (+ in the backward: the [E, 64] scratch of
// du1, sum_s du1 per point, the workgroups' dW2 partials, the per-column means).
DC_EXPORT size_t dc_edge2_workspace_bytes(int32_t n, int32_t k, int32_t backward) {
    const size_t chunks = (size_t)edge2_chunks(n);
    size_t b = al256(std::max(chunks * 2 * CH * 8, ws_need(n, CH))) + al256(2 * CH * 8);
    if (backward)
        b += al256((size_t)n * k * CH * 4) + al256((size_t)n * CH * 4) + al256(chunks * CH * CH * 4) + al256(5 * CH * 4) + al256(2 * CH * 4);
    return b;
}

// BatchNorm-1 of the edge MLP from the moments of the edge differences (ci <= 3): sd [n, ci] = sum_s (x_j - x_i) per point (the
// backward pass's closed forms need it) and mean1 / invstd1 / scale1 / shift1 [64] (+ running statistics) of y1 = W1 (x_j - x_i) over
// all n k edges -- what dc_edge_gather_stats computes from rows of z = x W1^T, without z.  Workspace: dc_edge2_workspace_bytes.
DC_EXPORT int dc_edge2_bn1_stats(const float* x, int64_t ldx, int32_t ci, const float* W1, const int32_t* nbr, int32_t n, int32_t k,
                                 const float* gamma1, const float* beta1, float eps, float momentum, float* running_mean,
                                 float* running_var, float* sd, float* mean1, float* invstd1, float* scale1, float* shift1,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(x && W1 && nbr && sd && mean1 && invstd1 && scale1 && shift1, "dc_edge2_bn1_stats: null pointer");
    DC_REQUIRE(n >= 1 && k >= 1 && ci >= 1 && ci <= 3 && ldx >= ci, "dc_edge2_bn1_stats: bad size (ci <= 3)");
    const int chunks = dc_cdiv(n, MOM_TPB);
    if (!workspace || workspace_bytes < (size_t)chunks * 9 * sizeof(double)) {
        dc_set_error("dc_edge2_bn1_stats: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(workspace);
    const BnFin fin{(long)n * k, gamma1, beta1, eps, momentum, running_mean, running_var, mean1, invstd1, scale1, shift1};
#define DC_EDGE2_MOM(CI)                                                                                                      \
    do {                                                                                                                        \
        hipLaunchKernelGGL(edge2_moments_kernel<CI>, dim3(chunks), dim3(MOM_TPB), 0, s, x, (long)ldx, nbr, (long)n, k, sd, partial, chunks); \
        hipLaunchKernelGGL(edge2_bn1_kernel<CI>, dim3(1), dim3(CH), 0, s, partial, chunks, (long)n * k, W1, fin);               \
    } while (0)
    if (ci == 1) DC_EDGE2_MOM(1);
    else if (ci == 2) DC_EDGE2_MOM(2);
    else DC_EDGE2_MOM(3);
#undef DC_EDGE2_MOM
    DC_CHECK_LAUNCH("dc_edge2_bn1_stats");
    return DC_OK;
}

// Forward: z [n, 64] contiguous (= x W1^T); x [n, ci] (row stride ldx) and W1 [64, ci]: when given and ci <= 3 the kernels
// evaluate y1 = W1 (x_j - x_i) per edge (the reference's order of operations) instead of z_j - z_i; scale1 / shift1 = BatchNorm-1 as an affine map (batch statistics of z_j - z_i from
// dc_edge_gather_stats, or the running ones), W2 [64, 64].  Outputs ysel [n, 64] = the selected pre-BatchNorm-2 value per
// (point, channel), arg uint8 [n, 64] its first slot.  stats_mode 1: BatchNorm-2 batch statistics over all n k edges ->
// mean2 / invstd2 / scale2 / shift2 (+ running statistics); 2: only the fp64 sums -> sums = double [2][2*64 + 1], two identical records
// [sum y2 | sum y2^2 | n k] as every *_sums entry point writes them (258 doubles: the caller sizes the buffer for BOTH)
// (synchronised BatchNorm: all-reduce, then dc_bn_coeffs_from_sums); 0: none (inference: coefficients from the running
// statistics).  out = act2(scale2 ysel + shift2) is one dc_bn_act call of the caller.
DC_EXPORT int dc_edge2_forward(const float* z, const float* x, int64_t ldx, int32_t ci, const float* W1, const int32_t* nbr,
                               int32_t n, int32_t k, const float* W2, const float* scale1,
                               const float* shift1, float slope1, int32_t stats_mode, const float* gamma2, const float* beta2,
                               float eps, float momentum, float* running_mean, float* running_var, float* ysel, uint8_t* arg,
                               float* mean2, float* invstd2, float* scale2, float* shift2, double* sums, void* workspace,
                               size_t workspace_bytes, void* stream) {
    DC_REQUIRE(nbr && W2 && scale1 && shift1 && ysel && arg, "dc_edge2_forward: null pointer");
    DC_REQUIRE(z || (x && W1 && ci >= 1 && ci <= 3 && ldx >= ci), "dc_edge2_forward: z may only be NULL with x, W1 and ci <= 3");
    DC_REQUIRE(n >= 1 && k >= 1 && k <= 255, "dc_edge2_forward: bad size");
    DC_REQUIRE((!z || al16(z)) && al16(W2) && al16(ysel) && (reinterpret_cast<uintptr_t>(arg) & 3) == 0, "dc_edge2_forward: misaligned");
    DC_REQUIRE(stats_mode != 1 || (mean2 && invstd2 && scale2 && shift2), "dc_edge2_forward: null pointer");
    DC_REQUIRE(stats_mode != 2 || sums, "dc_edge2_forward: null pointer");
    if (stats_mode && (!workspace || workspace_bytes < dc_edge2_workspace_bytes(n, k, 0))) {
        dc_set_error("dc_edge2_forward: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int chunks = edge2_chunks(n);
    double* partial = stats_mode ? static_cast<double*>(workspace) : nullptr;
    const int remap = dc_option(DC_OPT_XCD_REMAP);
    const bool direct = x && W1 && ci >= 1 && ci <= 3 && ldx >= ci;
#define DC_EDGE2_FWD(CI, SRC, LD)                                                                                              \
    hipLaunchKernelGGL(edge2_fwd_kernel<CI>, dim3(chunks), dim3(WPB * 64), 0, s, SRC, (long)(LD), W1, nbr, (long)n, k, W2, scale1, \
                       shift1, slope1, gamma2, ysel, arg, partial, chunks, remap)
    if (!direct) DC_EDGE2_FWD(0, z, CH);
    else if (ci == 1) DC_EDGE2_FWD(1, x, ldx);
    else if (ci == 2) DC_EDGE2_FWD(2, x, ldx);
    else DC_EDGE2_FWD(3, x, ldx);
#undef DC_EDGE2_FWD
    if (stats_mode == 1) {
        const BnFin fin{(long)n * k, gamma2, beta2, eps, momentum, running_mean, running_var, mean2, invstd2, scale2, shift2};
        hipLaunchKernelGGL((colreduce_final_kernel<BnFin>), dim3(CH), dim3(64), 0, s, partial, chunks, CH, fin);
    } else if (stats_mode == 2) {
        hipLaunchKernelGGL((colreduce_final_kernel<SumsFin>), dim3(CH), dim3(64), 0, s, partial, chunks, CH, SumsFin{sums, CH, (double)n * k});
    }
    DC_CHECK_LAUNCH("dc_edge2_forward");
    return DC_OK;
}

// Backward of the pair (dc_edge2_forward, out = act2(scale2 ysel + shift2)): dout [n, 64] (row stride lddo) -> dz [n, 64]
// (gradient w.r.t. z = x W1^T, row stride lddz), dW2 [64, 64], dgamma1 / dbeta1 / dgamma2 / dbeta2 [64] (each may be NULL).
// coef1 / coef2 = [mean | invstd | scale | shift] rows (4 x 64) of the two BatchNorms as used in the forward pass;
// training1 / training2: batch statistics (the mean / variance terms of the BatchNorm backward) or running ones;
// tptr / tedge = the CSC of the graph (dc_csc_build); s1pt = sum_s (z_j - z_i) per point (dc_edge_gather_stats).
DC_EXPORT int dc_edge2_backward(const float* dout, int64_t lddo, const float* z, const float* x, int64_t ldx, int32_t ci,
                                const float* W1, const int32_t* nbr, const int32_t* tptr,
                                const int32_t* tedge, int32_t n, int32_t k, const float* W2, const float* coef1,
                                const float* coef2, const float* gamma2, float slope1, float slope2, int32_t training1,
                                int32_t training2, const float* ysel, const uint8_t* arg, const float* s1pt, float* dz,
                                int64_t lddz, float* dW2, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2,
                                void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(dout && nbr && tptr && tedge && W2 && coef1 && coef2 && ysel && arg && dz && dW2, "dc_edge2_backward: null pointer");
    DC_REQUIRE(z || (x && W1 && ci >= 1 && ci <= 3 && ldx >= ci), "dc_edge2_backward: z may only be NULL with x, W1 and ci <= 3");
    DC_REQUIRE(s1pt || !training1, "dc_edge2_backward: the batch-statistics backward needs s1pt");
    DC_REQUIRE(n >= 1 && k >= 1 && k <= 255 && lddo >= CH && lddz >= CH && lddo % 4 == 0 && lddz % 4 == 0, "dc_edge2_backward: bad size");
    DC_REQUIRE(al16(dout) && (!z || al16(z)) && al16(W2) && al16(ysel) && al16(dz) && (reinterpret_cast<uintptr_t>(arg) & 3) == 0,
               "dc_edge2_backward: misaligned");
    if (!workspace || workspace_bytes < dc_edge2_workspace_bytes(n, k, 1)) {
        dc_set_error("dc_edge2_backward: workspace too small");
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int chunks = edge2_chunks(n);
    char* w = static_cast<char*>(workspace);
    const size_t part_bytes = al256(std::max((size_t)chunks * 2 * CH * 8, ws_need(n, CH)));
    double* partial = reinterpret_cast<double*>(w);           w += part_bytes;
    w += al256(2 * CH * 8);
    float* dU = reinterpret_cast<float*>(w);                  w += al256((size_t)n * k * CH * 4);
    float* csum = reinterpret_cast<float*>(w);                w += al256((size_t)n * CH * 4);
    float* dWpart = reinterpret_cast<float*>(w);              w += al256((size_t)chunks * CH * CH * 4);
    float* coefs5 = reinterpret_cast<float*>(w);              w += al256(5 * CH * 4);
    float* m12 = reinterpret_cast<float*>(w);
    const float *mean1 = coef1, *invstd1 = coef1 + CH, *scale1 = coef1 + 2 * CH, *shift1 = coef1 + 3 * CH;
    const float *mean2 = coef2, *invstd2 = coef2 + CH, *scale2 = coef2 + 2 * CH, *shift2 = coef2 + 3 * CH;
    // 1. d z2 (into csum: consumed by the recompute pass before csum is written... no: its own buffer = dz, free until step 4)
    float* dz2 = dz;                                           // [n, 64] contiguous view needs lddz == CH; else the csum slot
    DC_REQUIRE(lddz == CH, "dc_edge2_backward: dz must be contiguous [n, 64]");
    {
        const Ws ws = carve(partial, n, CH);
        const BwdCoefFin fin{(long)n * k, gamma2, scale2, shift2, mean2, invstd2, training2, dgamma2, dbeta2, coefs5, CH};
        run_colreduce<4>(Bn2BwdF<4>{dout, ysel, scale2, shift2, mean2, invstd2, (long)lddo, slope2, dz2}, n, CH, ws, s, fin);
    }
    // 2. the recompute pass
    const int remap = dc_option(DC_OPT_XCD_REMAP);
    const bool direct = x && W1 && ci >= 1 && ci <= 3 && ldx >= ci;
#define DC_EDGE2_BWD(CI, SRC, LD)                                                                                              \
    do {                                                                                                                        \
        static unsigned long long attr_set = 0;                                                                                 \
        if (!dc_ensure_lds(&attr_set, reinterpret_cast<const void*>(&edge2_bwd_kernel<CI>), BWD_LDS, "dc_edge2_backward")) {     \
            DC_CHECK_LAUNCH("dc_edge2_backward");                                                                               \
        }                                                                                                                       \
        hipLaunchKernelGGL(edge2_bwd_kernel<CI>, dim3(chunks), dim3(WPB * 64), BWD_LDS, s, SRC, (long)(LD), W1, nbr, (long)n, k, W2, \
                           scale1, shift1, slope1, coefs5, dz2, arg, dU, csum, partial, dWpart, chunks, remap);                 \
    } while (0)
    if (!direct) DC_EDGE2_BWD(0, z, CH);
    else if (ci == 1) DC_EDGE2_BWD(1, x, ldx);
    else if (ci == 2) DC_EDGE2_BWD(2, x, ldx);
    else DC_EDGE2_BWD(3, x, ldx);
#undef DC_EDGE2_BWD
    // 3. BatchNorm-1 sums, dW2
    hipLaunchKernelGGL((colreduce_final_kernel<Bn1BwdFin>), dim3(CH), dim3(64), 0, s, partial, chunks, CH,
                       Bn1BwdFin{(long)n * k, mean1, invstd1, training1, dgamma1, dbeta1, m12, m12 + CH});
    hipLaunchKernelGGL(edge2_dw_reduce_kernel, dim3(CH * CH / 16), dim3(256), 0, s, dWpart, chunks, dW2);
    // 4. the closing pass (overwrites dz = the d z2 buffer: the recompute pass has consumed it)
    const long total = (long)n * (CH / 4);
#define DC_EDGE2_SCATTER(CI, SRC, LD)                                                                                          \
    hipLaunchKernelGGL(edge2_scatter_kernel<CI>, dim3(dc_cdiv(total, 256)), dim3(256), 0, s, (long)n, k, remap, tptr, tedge, dU, SRC, \
                       (long)(LD), W1, csum, s1pt, scale1, mean1, invstd1, m12, m12 + CH, training1, dz, (long)lddz)
    if (z) {                 // rows of z given: the closed forms from them (s1pt = sum_s (z_j - z_i), [n, 64])
        DC_REQUIRE(!training1 || al16(s1pt), "dc_edge2_backward: misaligned");
        DC_EDGE2_SCATTER(0, z, CH);
    } else if (ci == 1)      // no z: from the input channels (s1pt = sd = sum_s (x_j - x_i), [n, ci]: dc_edge2_bn1_stats)
        DC_EDGE2_SCATTER(1, x, ldx);
    else if (ci == 2) DC_EDGE2_SCATTER(2, x, ldx);
    else DC_EDGE2_SCATTER(3, x, ldx);
#undef DC_EDGE2_SCATTER
    DC_CHECK_LAUNCH("dc_edge2_backward");
    return DC_OK;
}
