// Per-thread bodies of the fixed-degree (ELL) operator-apply and max-aggregation kernels,
// shared by the HIP kernels (apply.hip, aggregate.hip) and the CPU host-check build
// (tests/hostcheck) exactly like point_math.h.
//
// Work decomposition: thread t owns (point, channel group) = (t / groups, t % groups), a group
// being V consecutive channels (V = 4 -> 16-byte loads/stores, V = 1 -> any C / stride).
// Consecutive threads walk one feature row, so the V*groups floats of a gathered neighbour row
// are read by adjacent lanes (coalesced 16 B/lane segments); neighbour ids and the two
// coefficients of an edge are identical across the lanes of a point (broadcast loads).
//
// Layouts: nbr[Nt,k] int32; G/D[Nt,k,2]; scalar fields [Nt, ld]; vector fields [2Nt, ld] with
// row 2i = u-component, row 2i+1 = v-component (reference: geometry/operators.py:4-21).
// Transposed forms walk the CSC of nbr: tptr[Nt+1], tedge[t] = edge id e = i*k + s (ascending).
#pragma once
#include "point_math.h"

namespace dcell {

template <int V>
struct alignas(4 * V) Vec {
    float v[V];
};

template <int V>
DC_HD Vec<V> vload(const float* p) {
    return *reinterpret_cast<const Vec<V>*>(p);
}
template <int V>
DC_HD void vstore(float* p, const Vec<V>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (V == 4) {                               // results are streamed out (common.h: dc_store16)
        dc_store16<DC_ST_ELL>(p, *reinterpret_cast<const dc_f32x4*>(&a));
        return;
    }
#endif
    *reinterpret_cast<Vec<V>*>(p) = a;
}
template <int V>
DC_HD Vec<V> vzero() {
    Vec<V> r;
#pragma unroll
    for (int q = 0; q < V; ++q) r.v[q] = 0.f;
    return r;
}
// acc += s * a
template <int V>
DC_HD void vfma(Vec<V>& acc, float s, const Vec<V>& a) {
#pragma unroll
    for (int q = 0; q < V; ++q) acc.v[q] = fmaf(s, a.v[q], acc.v[q]);
}
template <int V>
DC_HD void vout(float* p, const Vec<V>& a, int accumulate) {
    if (accumulate) {
        Vec<V> o = vload<V>(p);
#pragma unroll
        for (int q = 0; q < V; ++q) o.v[q] += a.v[q];
        vstore<V>(p, o);
    } else {
        vstore<V>(p, a);
    }
}

struct G2 {
    float a, b;
};
DC_HD G2 ldcoef(const float* coef, long e) { return *reinterpret_cast<const G2*>(coef + 2 * e); }

// ---- forward applies -------------------------------------------------------------------------
// The k-list of a point is handed over as two row pointers (ids, coefficients).  On the GPU they
// point into LDS: a block stages the rows of all its points with coalesced loads once, because
// per-lane loads of ids / coefficients through the vector-memory path cost as many texture-
// addresser cycles as the feature gathers themselves (PMC: TA busy 2/3 of the kernel, 2 of every
// 3 VMEM instructions were id/coefficient loads).  On the host-check build they point at the
// global arrays.
struct Row {
    const int* ids;
    const G2* cf;
};
DC_HD Row global_row(const float* coef, const int* nbr, long i, int k) {
    return Row{nbr + i * k, reinterpret_cast<const G2*>(coef) + i * k};
}

// grad @ x : out[2i+a, c] = sum_s G[i,s,a] * x[nbr[i,s], c]          (torch_sparse spmm at
// models/deltanet_base.py:78, nn/deltaconv.py:66)
template <int V>
DC_HD void grad_fwd(long i, int c0, int C, Row r, int k, const float* x, long ldx, float* out, long ldo) {
    Vec<V> au = vzero<V>(), av = vzero<V>();
#pragma unroll 4
    for (int s = 0; s < k; ++s) {
        const G2 g = r.cf[s];
        const Vec<V> xv = vload<V>(x + (long)r.ids[s] * ldx + c0);
        vfma<V>(au, g.a, xv);
        vfma<V>(av, g.b, xv);
    }
    vstore<V>(out + (2 * i) * ldo + c0, au);
    vstore<V>(out + (2 * i + 1) * ldo + c0, av);
}

// div @ v : out[i, c] = sum_s D[i,s,0] * v[2j, c] + D[i,s,1] * v[2j+1, c]   (nn/deltaconv.py:57)
template <int V>
DC_HD void div_fwd(long i, int c0, int C, Row r, int k, const float* v, long ldv, float* out, long ldo) {
    Vec<V> acc = vzero<V>();
#pragma unroll 4
    for (int s = 0; s < k; ++s) {
        const G2 d = r.cf[s];
        const long j = r.ids[s];
        vfma<V>(acc, d.a, vload<V>(v + (2 * j) * ldv + c0));
        vfma<V>(acc, d.b, vload<V>(v + (2 * j + 1) * ldv + c0));
    }
    vstore<V>(out + i * ldo + c0, acc);
}

// Fused [div v | curl v | norm v] -> out[i, 0:C | C:2C | 2C:3C]  (nn/deltaconv.py:57 with
// geometry/operators.py:4-7,23-27: curl = -div(J v), J(v) = (-v_v, v_u)).  v is gathered once.
template <int V>
DC_HD void divcurlnorm_fwd(long i, int c0, int C, Row r, int k, const float* v, long ldv, float* out, long ldo) {
    Vec<V> dv = vzero<V>(), cv = vzero<V>();
#pragma unroll 4
    for (int s = 0; s < k; ++s) {
        const G2 d = r.cf[s];
        const long j = r.ids[s];
        const Vec<V> vu = vload<V>(v + (2 * j) * ldv + c0);
        const Vec<V> vv = vload<V>(v + (2 * j + 1) * ldv + c0);
        vfma<V>(dv, d.a, vu);
        vfma<V>(dv, d.b, vv);
        vfma<V>(cv, d.a, vv);   // -(D0 * (-v_v) + D1 * v_u)
        vfma<V>(cv, -d.b, vu);
    }
    const Vec<V> ou = vload<V>(v + (2 * i) * ldv + c0), ov = vload<V>(v + (2 * i + 1) * ldv + c0);
    Vec<V> nv;
#pragma unroll
    for (int q = 0; q < V; ++q) nv.v[q] = sqrtf(fmaf(ou.v[q], ou.v[q], ov.v[q] * ov.v[q]));
    vstore<V>(out + i * ldo + c0, dv);
    vstore<V>(out + i * ldo + C + c0, cv);
    vstore<V>(out + i * ldo + 2 * C + c0, nv);
}

// Fused Hodge-Laplacian from the already computed [div v | curl v] (geometry/operators.py:35-46
// recomputes them; here they are read from dc[j, 0:C | C:2C]):
//   hodge = -(grad(div v) + J grad(curl v))
//   h_u = -(sum G_u dv_j - sum G_v cv_j),  h_v = -(sum G_v dv_j + sum G_u cv_j)
template <int V>
DC_HD void hodge_fwd(long i, int c0, int C, Row r, int k, const float* dc, long ldd, float* out, long ldo) {
    Vec<V> hu = vzero<V>(), hv = vzero<V>();
#pragma unroll 4
    for (int s = 0; s < k; ++s) {
        const G2 g = r.cf[s];
        const long j = r.ids[s];
        const Vec<V> dv = vload<V>(dc + j * ldd + c0);
        const Vec<V> cv = vload<V>(dc + j * ldd + C + c0);
        vfma<V>(hu, -g.a, dv);
        vfma<V>(hu, g.b, cv);
        vfma<V>(hv, -g.b, dv);
        vfma<V>(hv, -g.a, cv);
    }
    vstore<V>(out + (2 * i) * ldo + c0, hu);
    vstore<V>(out + (2 * i + 1) * ldo + c0, hv);
}

// out[i,c] = max_s h[nbr[i,s], c]; arg[i,c] = first maximal slot s (uint8; k <= 255)
// (torch_scatter.scatter(reduce='max') at nn/deltaconv.py:52,54)
template <int V>
DC_HD void knn_max_fwd(long i, int c0, const int* ids, int k, const float* h, long ldh, float* out, long ldo,
                       unsigned char* arg, long lda) {
    Vec<V> best = vload<V>(h + (long)ids[0] * ldh + c0);
    unsigned char slot[V];
#pragma unroll
    for (int q = 0; q < V; ++q) slot[q] = 0;
#pragma unroll 4
    for (int s = 1; s < k; ++s) {
        const Vec<V> hv = vload<V>(h + (long)ids[s] * ldh + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const bool up = hv.v[q] > best.v[q];
            best.v[q] = up ? hv.v[q] : best.v[q];
            slot[q] = up ? (unsigned char)s : slot[q];
        }
    }
    vstore<V>(out + i * ldo + c0, best);
#pragma unroll
    for (int q = 0; q < V; ++q) arg[i * lda + c0 + q] = slot[q];
}

// sum / mean aggregation (torch_scatter reduce = 'sum' | 'add' | 'mean', nn/deltaconv.py:52,54 with aggr != 'max'):
// out[i,c] = scale * sum_s h[nbr[i,s], c], slots in order (fixed association); scale = 1 or 1/k
template <int V>
DC_HD void knn_sum_fwd(long i, int c0, const int* ids, int k, const float* h, long ldh, float scale, float* out, long ldo) {
    Vec<V> acc = vload<V>(h + (long)ids[0] * ldh + c0);
#pragma unroll 4
    for (int s = 1; s < k; ++s) {
        const Vec<V> hv = vload<V>(h + (long)ids[s] * ldh + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) acc.v[q] += hv.v[q];
    }
#pragma unroll
    for (int q = 0; q < V; ++q) acc.v[q] *= scale;
    vstore<V>(out + i * ldo + c0, acc);
}

// Same with the BatchNorm + activation of the producing MLP block folded in:
//   out[i,c] = max_s y[nbr[i,s], c],  y = act(scale_c * h + shift_c)   (nn/mlp.py:9 + nn/deltaconv.py:54)
// y is evaluated per candidate (2 flops on top of a 16-byte gather), so values, ties and the first-maximal-slot
// rule are exactly those of materialising y first -- minus one [Nt,C] write and read.
template <int V>
DC_HD void knn_max_affine_fwd(long i, int c0, const int* ids, int k, const float* h, long ldh, const float* scale,
                              const float* shift, float slope, float* out, long ldo, unsigned char* arg, long lda) {
    float sc[V], sh[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { sc[q] = scale[c0 + q]; sh[q] = shift[c0 + q]; }
    Vec<V> best = vload<V>(h + (long)ids[0] * ldh + c0);
    unsigned char slot[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
        const float z = fmaf(sc[q], best.v[q], sh[q]);
        best.v[q] = z > 0.f ? z : slope * z;
        slot[q] = 0;
    }
#pragma unroll 4
    for (int s = 1; s < k; ++s) {
        const Vec<V> hv = vload<V>(h + (long)ids[s] * ldh + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const float z = fmaf(sc[q], hv.v[q], sh[q]);
            const float y = z > 0.f ? z : slope * z;
            const bool up = y > best.v[q];
            best.v[q] = up ? y : best.v[q];
            slot[q] = up ? (unsigned char)s : slot[q];
        }
    }
    vstore<V>(out + i * ldo + c0, best);
#pragma unroll
    for (int q = 0; q < V; ++q) arg[i * lda + c0 + q] = slot[q];
}

// The same with the layer's last s_mlp block in its epilogue (round 6): out = act2(scale2 h2[i] + shift2) + max  -- the residual
// form `x = s_mlp(...) + x_max` of deltaconv.py:59; same two addends as dc_bn_act2 with the maximum as residual (same bits);
// out2 (may be null): second copy, the layer's block of the heads' concat buffer.
template <int V>
DC_HD void knn_max_affine_residual_fwd(long i, int c0, const int* ids, int k, const float* h, long ldh, const float* scale,
                                       const float* shift, float slope, const float* h2, long ldh2, const float* scale2,
                                       const float* shift2, float slope2, float* out, long ldo, float* out2, long ldo2,
                                       unsigned char* arg, long lda) {
    float sc[V], sh[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { sc[q] = scale[c0 + q]; sh[q] = shift[c0 + q]; }
    Vec<V> best = vload<V>(h + (long)ids[0] * ldh + c0);
    unsigned char slot[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
        const float z = fmaf(sc[q], best.v[q], sh[q]);
        best.v[q] = z > 0.f ? z : slope * z;
        slot[q] = 0;
    }
#pragma unroll 4
    for (int s = 1; s < k; ++s) {
        const Vec<V> hv = vload<V>(h + (long)ids[s] * ldh + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const float z = fmaf(sc[q], hv.v[q], sh[q]);
            const float y = z > 0.f ? z : slope * z;
            const bool up = y > best.v[q];
            best.v[q] = up ? y : best.v[q];
            slot[q] = up ? (unsigned char)s : slot[q];
        }
    }
    const Vec<V> v2 = vload<V>(h2 + i * ldh2 + c0);
    Vec<V> y;
#pragma unroll
    for (int q = 0; q < V; ++q) {
        const float z = fmaf(scale2[c0 + q], v2.v[q], shift2[c0 + q]);
        y.v[q] = (z > 0.f ? z : slope2 * z) + best.v[q];
    }
    vstore<V>(out + i * ldo + c0, y);
    if (out2) vstore<V>(out2 + i * ldo2 + c0, y);
#pragma unroll
    for (int q = 0; q < V; ++q) arg[i * lda + c0 + q] = slot[q];
}

// ---- transposed applies (backward of the above; the operators carry no gradient) ---------------
// A column of the transposed operator = the in-edges of point j, ascending edge id.  Each op is an
// accumulator: init(), step(i, s, g, c0) once per in-edge (i = source point, s = its slot,
// g = the edge's two coefficients), finish(j, c0).  The kernel (apply.hip) stages the in-edge
// lists of a block's points through LDS in chunks; the host-check walks the CSC directly.
//
// grad^T : dx[j, c] (+)= sum_e G[e,0] * dy[2i, c] + G[e,1] * dy[2i+1, c]
template <int V>
struct GradT {
    const float* dy; long ldy; float* dx; long ldx; int accumulate; int C;
    Vec<V> acc;
    DC_HD void init() { acc = vzero<V>(); }
    DC_HD void step(long i, int, G2 g, int c0) {
        vfma<V>(acc, g.a, vload<V>(dy + (2 * i) * ldy + c0));
        vfma<V>(acc, g.b, vload<V>(dy + (2 * i + 1) * ldy + c0));
    }
    DC_HD void finish(long j, int c0) { vout<V>(dx + j * ldx + c0, acc, accumulate); }
};

// grad^T with the gradient accumulation folded in: out[j, c] = a[j, c] (+ b[j, c]) + sum_e ...  (a, b = the gradients
// of x' that reach a layer from its other consumers; the destination is a fresh tensor: no add pass, no clone)
template <int V>
struct GradTSum {
    const float* dy; long ldy; const float* a; long lda; const float* b; long ldb; float* out; long ldo; int C;
    Vec<V> acc;
    DC_HD void init() { acc = vzero<V>(); }
    DC_HD void step(long i, int, G2 g, int c0) {
        vfma<V>(acc, g.a, vload<V>(dy + (2 * i) * ldy + c0));
        vfma<V>(acc, g.b, vload<V>(dy + (2 * i + 1) * ldy + c0));
    }
    DC_HD void finish(long j, int c0) {
        Vec<V> o = vload<V>(a + j * lda + c0);
        if (b) {
            const Vec<V> ob = vload<V>(b + j * ldb + c0);
#pragma unroll
            for (int q = 0; q < V; ++q) o.v[q] += ob.v[q];      // (a + b) + grad^T dy: the order of the separate passes
        }
#pragma unroll
        for (int q = 0; q < V; ++q) o.v[q] += acc.v[q];
        vstore<V>(out + j * ldo + c0, o);
    }
};

// div^T : dv[2j+a, c] (+)= sum_e D[e,a] * dy[i, c]
template <int V>
struct DivT {
    const float* dy; long ldy; float* dv; long ldv; int accumulate; int C;
    Vec<V> au, av;
    DC_HD void init() { au = vzero<V>(); av = vzero<V>(); }
    DC_HD void step(long i, int, G2 d, int c0) {
        const Vec<V> g = vload<V>(dy + i * ldy + c0);
        vfma<V>(au, d.a, g);
        vfma<V>(av, d.b, g);
    }
    DC_HD void finish(long j, int c0) {
        vout<V>(dv + (2 * j) * ldv + c0, au, accumulate);
        vout<V>(dv + (2 * j + 1) * ldv + c0, av, accumulate);
    }
};

// backward of divcurlnorm_fwd: dout[i, 0:C | C:2C | 2C:3C] = (d_div, d_curl, d_norm)
//   dv_u[j] = sum_e (D0 d_div_i - D1 d_curl_i) + d_norm_j v_u[j] / |v_j|
//   dv_v[j] = sum_e (D1 d_div_i + D0 d_curl_i) + d_norm_j v_v[j] / |v_j|
template <int V>
struct DivCurlNormT {
    const float* dout; long ldo; const float* v; long ldv; float* dv; long lddv; int accumulate; int C;
    Vec<V> au, av;
    DC_HD void init() { au = vzero<V>(); av = vzero<V>(); }
    DC_HD void step(long i, int, G2 d, int c0) {
        const Vec<V> dd = vload<V>(dout + i * ldo + c0);
        const Vec<V> dcu = vload<V>(dout + i * ldo + C + c0);
        vfma<V>(au, d.a, dd);
        vfma<V>(au, -d.b, dcu);
        vfma<V>(av, d.b, dd);
        vfma<V>(av, d.a, dcu);
    }
    DC_HD void finish(long j, int c0) {
        const Vec<V> dn = vload<V>(dout + j * ldo + 2 * C + c0);
        const Vec<V> ou = vload<V>(v + (2 * j) * ldv + c0), ov = vload<V>(v + (2 * j + 1) * ldv + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const float nrm = sqrtf(fmaf(ou.v[q], ou.v[q], ov.v[q] * ov.v[q]));
            const float sc = nrm > 0.f ? dn.v[q] / nrm : 0.f;  // subgradient 0 at |v| = 0 (as torch)
            au.v[q] = fmaf(sc, ou.v[q], au.v[q]);
            av.v[q] = fmaf(sc, ov.v[q], av.v[q]);
        }
        vout<V>(dv + (2 * j) * lddv + c0, au, accumulate);
        vout<V>(dv + (2 * j + 1) * lddv + c0, av, accumulate);
    }
};

// backward of hodge_fwd: ddc[j, 0:C]  (+)= -sum_e (G_u dh_u[i] + G_v dh_v[i])
//                        ddc[j, C:2C] (+)=  sum_e (G_v dh_u[i] - G_u dh_v[i])
template <int V>
struct HodgeT {
    const float* dh; long ldh; float* ddc; long ldd; int accumulate; int C;
    Vec<V> ad, ac;
    DC_HD void init() { ad = vzero<V>(); ac = vzero<V>(); }
    DC_HD void step(long i, int, G2 g, int c0) {
        const Vec<V> hu = vload<V>(dh + (2 * i) * ldh + c0);
        const Vec<V> hv = vload<V>(dh + (2 * i + 1) * ldh + c0);
        vfma<V>(ad, -g.a, hu);
        vfma<V>(ad, -g.b, hv);
        vfma<V>(ac, g.b, hu);
        vfma<V>(ac, -g.a, hv);
    }
    DC_HD void finish(long j, int c0) {
        vout<V>(ddc + j * ldd + c0, ad, accumulate);
        vout<V>(ddc + j * ldd + C + c0, ac, accumulate);
    }
};

// max-aggregation backward: dh[j,c] (+)= sum over in-edges (i,s) of j with arg[i,c] == s of dout[i,c]
template <int V>
struct KnnMaxT {
    const unsigned char* arg; long lda; const float* dout; long ldo; float* dh; long ldh; int accumulate; int C;
    Vec<V> acc;
    DC_HD void init() { acc = vzero<V>(); }
    DC_HD void step(long i, int s, G2, int c0) {
        bool any = false;
        bool hit[V];
        if (V == 4) {   // the four slot bytes in one 32-bit load (c0 and lda are multiples of 4)
            const unsigned w = *reinterpret_cast<const unsigned*>(arg + i * lda + c0);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                hit[q] = ((w >> (8 * q)) & 0xffu) == (unsigned)s;
                any = any || hit[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < V; ++q) {
                hit[q] = arg[i * lda + c0 + q] == (unsigned char)s;
                any = any || hit[q];
            }
        }
        if (any) {
            const Vec<V> g = vload<V>(dout + i * ldo + c0);
#pragma unroll
            for (int q = 0; q < V; ++q) acc.v[q] += hit[q] ? g.v[q] : 0.f;
        }
    }
    DC_HD void finish(long j, int c0) { vout<V>(dh + j * ldh + c0, acc, accumulate); }
};

// sum / mean aggregation backward: dh[j,c] (+)= scale * sum over in-edges (i,s) of j of dout[i,c]  (ascending edge id)
template <int V>
struct KnnSumT {
    const float* dout; long ldo; float* dh; long ldh; float scale; int accumulate; int C;
    Vec<V> acc;
    DC_HD void init() { acc = vzero<V>(); }
    DC_HD void step(long i, int, G2, int c0) {
        const Vec<V> g = vload<V>(dout + i * ldo + c0);
#pragma unroll
        for (int q = 0; q < V; ++q) acc.v[q] += g.v[q];
    }
    DC_HD void finish(long j, int c0) {
#pragma unroll
        for (int q = 0; q < V; ++q) acc.v[q] *= scale;
        vout<V>(dh + j * ldh + c0, acc, accumulate);
    }
};

// Host-side / reference walk of one column (no staging): coefT is in CSC order, tedge gives (i, s).
template <class OP>
DC_HD void walk_column(OP op, long j, int c0, const float* coefT, const int* tptr, const int* tedge, int k) {
    op.init();
    for (int p = tptr[j]; p < tptr[j + 1]; ++p) {
        const int e = tedge[p];
        const int i = e / k;
        const G2 g = coefT ? ldcoef(coefT, p) : G2{0.f, 0.f};
        op.step(i, e - i * k, g, c0);
    }
    op.finish(j, c0);
}

// ---- transposed-structure (CSC) build pieces ------------------------------------------------
// stable in-place insertion sort of one column's edge ids (atomics fill columns in arbitrary
// order; ascending edge id makes every transposed sum run in a fixed order -> deterministic)
DC_HD void sort_column(int* tedge, int lo, int hi) {
    for (int a = lo + 1; a < hi; ++a) {
        const int key = tedge[a];
        int b = a - 1;
        while (b >= lo && tedge[b] > key) {
            tedge[b + 1] = tedge[b];
            --b;
        }
        tedge[b + 1] = key;
    }
}

}  // namespace dcell
