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
// Batched gather: U neighbours per batch -- all ids / coefficients of the batch are loaded first,
// then all U neighbour rows are requested back-to-back (U gathers in flight per lane), then the
// FMAs run.  The tail of the k-list is handled by clamping the slot and zeroing the coefficient.
template <int U>
struct Batch {
    long j[U];
    G2 g[U];
};
template <int U>
DC_HD Batch<U> load_batch(const float* coef, const int* nbr, long i, int k, int s0) {
    Batch<U> b;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int s = s0 + u;
        const bool ok = s < k;
        const long e = i * k + (ok ? s : k - 1);
        b.j[u] = nbr[e];
        const G2 g = ldcoef(coef, e);
        b.g[u].a = ok ? g.a : 0.f;
        b.g[u].b = ok ? g.b : 0.f;
    }
    return b;
}

// grad @ x : out[2i+a, c] = sum_s G[i,s,a] * x[nbr[i,s], c]          (torch_sparse spmm at
// models/deltanet_base.py:78, nn/deltaconv.py:66)
template <int V, int U = 4>
DC_HD void grad_fwd(long t, int groups, const float* G, const int* nbr, int k, const float* x, long ldx, float* out,
                    long ldo) {
    const long i = t / groups;
    const int c0 = (int)(t % groups) * V;
    Vec<V> au = vzero<V>(), av = vzero<V>();
    for (int s0 = 0; s0 < k; s0 += U) {
        const Batch<U> b = load_batch<U>(G, nbr, i, k, s0);
        Vec<V> xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = vload<V>(x + b.j[u] * ldx + c0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vfma<V>(au, b.g[u].a, xv[u]);
            vfma<V>(av, b.g[u].b, xv[u]);
        }
    }
    vstore<V>(out + (2 * i) * ldo + c0, au);
    vstore<V>(out + (2 * i + 1) * ldo + c0, av);
}

// div @ v : out[i, c] = sum_s D[i,s,0] * v[2j, c] + D[i,s,1] * v[2j+1, c]   (nn/deltaconv.py:57)
template <int V, int U = 4>
DC_HD void div_fwd(long t, int groups, const float* D, const int* nbr, int k, const float* v, long ldv, float* out,
                   long ldo) {
    const long i = t / groups;
    const int c0 = (int)(t % groups) * V;
    Vec<V> acc = vzero<V>();
    for (int s0 = 0; s0 < k; s0 += U) {
        const Batch<U> b = load_batch<U>(D, nbr, i, k, s0);
        Vec<V> vu[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vu[u] = vload<V>(v + (2 * b.j[u]) * ldv + c0);
            vv[u] = vload<V>(v + (2 * b.j[u] + 1) * ldv + c0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vfma<V>(acc, b.g[u].a, vu[u]);
            vfma<V>(acc, b.g[u].b, vv[u]);
        }
    }
    vstore<V>(out + i * ldo + c0, acc);
}

// Fused [div v | curl v | norm v] -> out[i, 0:C | C:2C | 2C:3C]  (nn/deltaconv.py:57 with
// geometry/operators.py:4-7,23-27: curl = -div(J v), J(v) = (-v_v, v_u)).  v is gathered once.
template <int V, int U = 4>
DC_HD void divcurlnorm_fwd(long t, int groups, const float* D, const int* nbr, int k, const float* v, long ldv,
                           float* out, long ldo) {
    const long i = t / groups;
    const int c0 = (int)(t % groups) * V;
    const int C = groups * V;
    Vec<V> dv = vzero<V>(), cv = vzero<V>();
    for (int s0 = 0; s0 < k; s0 += U) {
        const Batch<U> b = load_batch<U>(D, nbr, i, k, s0);
        Vec<V> vu[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vu[u] = vload<V>(v + (2 * b.j[u]) * ldv + c0);
            vv[u] = vload<V>(v + (2 * b.j[u] + 1) * ldv + c0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vfma<V>(dv, b.g[u].a, vu[u]);
            vfma<V>(dv, b.g[u].b, vv[u]);
            vfma<V>(cv, b.g[u].a, vv[u]);   // -(D0 * (-v_v) + D1 * v_u)
            vfma<V>(cv, -b.g[u].b, vu[u]);
        }
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
template <int V, int U = 4>
DC_HD void hodge_fwd(long t, int groups, const float* G, const int* nbr, int k, const float* dc, long ldd, float* out,
                     long ldo) {
    const long i = t / groups;
    const int c0 = (int)(t % groups) * V;
    const int C = groups * V;
    Vec<V> hu = vzero<V>(), hv = vzero<V>();
    for (int s0 = 0; s0 < k; s0 += U) {
        const Batch<U> b = load_batch<U>(G, nbr, i, k, s0);
        Vec<V> dv[U], cv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dv[u] = vload<V>(dc + b.j[u] * ldd + c0);
            cv[u] = vload<V>(dc + b.j[u] * ldd + C + c0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vfma<V>(hu, -b.g[u].a, dv[u]);
            vfma<V>(hu, b.g[u].b, cv[u]);
            vfma<V>(hv, -b.g[u].b, dv[u]);
            vfma<V>(hv, -b.g[u].a, cv[u]);
        }
    }
    vstore<V>(out + (2 * i) * ldo + c0, hu);
    vstore<V>(out + (2 * i + 1) * ldo + c0, hv);
}

// ---- transposed applies (backward of the above; the operators carry no gradient) ---------------
// grad^T : dx[j, c] (+)= sum_{e in col j} G[e,0] * dy[2i, c] + G[e,1] * dy[2i+1, c]
template <int V>
DC_HD void grad_T(long t, int groups, const float* G, const int* tptr, const int* tedge, int k, const float* dy,
                  long ldy, float* dx, long ldx, int accumulate) {
    const long j = t / groups;
    const int c0 = (int)(t % groups) * V;
    Vec<V> acc = vzero<V>();
    const int t1 = tptr[j + 1];
#pragma unroll 4
    for (int p = tptr[j]; p < t1; ++p) {
        const long e = tedge[p];
        const long i = e / k;
        const G2 g = ldcoef(G, e);
        vfma<V>(acc, g.a, vload<V>(dy + (2 * i) * ldy + c0));
        vfma<V>(acc, g.b, vload<V>(dy + (2 * i + 1) * ldy + c0));
    }
    vout<V>(dx + j * ldx + c0, acc, accumulate);
}

// div^T : dv[2j+a, c] (+)= sum_{e in col j} D[e,a] * dy[i, c]
template <int V>
DC_HD void div_T(long t, int groups, const float* D, const int* tptr, const int* tedge, int k, const float* dy,
                 long ldy, float* dv, long ldv, int accumulate) {
    const long j = t / groups;
    const int c0 = (int)(t % groups) * V;
    Vec<V> au = vzero<V>(), av = vzero<V>();
    const int t1 = tptr[j + 1];
#pragma unroll 4
    for (int p = tptr[j]; p < t1; ++p) {
        const long e = tedge[p];
        const long i = e / k;
        const G2 d = ldcoef(D, e);
        const Vec<V> g = vload<V>(dy + i * ldy + c0);
        vfma<V>(au, d.a, g);
        vfma<V>(av, d.b, g);
    }
    vout<V>(dv + (2 * j) * ldv + c0, au, accumulate);
    vout<V>(dv + (2 * j + 1) * ldv + c0, av, accumulate);
}

// backward of divcurlnorm_fwd: dout[i, 0:C | C:2C | 2C:3C] = (d_div, d_curl, d_norm)
//   dv_u[j] = sum_e (D0 d_div_i - D1 d_curl_i) + d_norm_j v_u[j] / |v_j|
//   dv_v[j] = sum_e (D1 d_div_i + D0 d_curl_i) + d_norm_j v_v[j] / |v_j|
template <int V>
DC_HD void divcurlnorm_T(long t, int groups, const float* D, const int* tptr, const int* tedge, int k,
                         const float* dout, long ldo, const float* v, long ldv, float* dv, long lddv, int accumulate) {
    const long j = t / groups;
    const int c0 = (int)(t % groups) * V;
    const int C = groups * V;
    Vec<V> au = vzero<V>(), av = vzero<V>();
    const int t1 = tptr[j + 1];
#pragma unroll 4
    for (int p = tptr[j]; p < t1; ++p) {
        const long e = tedge[p];
        const long i = e / k;
        const G2 d = ldcoef(D, e);
        const Vec<V> dd = vload<V>(dout + i * ldo + c0);
        const Vec<V> dcu = vload<V>(dout + i * ldo + C + c0);
        vfma<V>(au, d.a, dd);
        vfma<V>(au, -d.b, dcu);
        vfma<V>(av, d.b, dd);
        vfma<V>(av, d.a, dcu);
    }
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

// backward of hodge_fwd: ddc[j, 0:C]  (+)= -sum_e (G_u dh_u[i] + G_v dh_v[i])
//                        ddc[j, C:2C] (+)=  sum_e (G_v dh_u[i] - G_u dh_v[i])
template <int V>
DC_HD void hodge_T(long t, int groups, const float* G, const int* tptr, const int* tedge, int k, const float* dh,
                   long ldh, float* ddc, long ldd, int accumulate) {
    const long j = t / groups;
    const int c0 = (int)(t % groups) * V;
    const int C = groups * V;
    Vec<V> ad = vzero<V>(), ac = vzero<V>();
    const int t1 = tptr[j + 1];
#pragma unroll 4
    for (int p = tptr[j]; p < t1; ++p) {
        const long e = tedge[p];
        const long i = e / k;
        const G2 g = ldcoef(G, e);
        const Vec<V> hu = vload<V>(dh + (2 * i) * ldh + c0);
        const Vec<V> hv = vload<V>(dh + (2 * i + 1) * ldh + c0);
        vfma<V>(ad, -g.a, hu);
        vfma<V>(ad, -g.b, hv);
        vfma<V>(ac, g.b, hu);
        vfma<V>(ac, -g.a, hv);
    }
    vout<V>(ddc + j * ldd + c0, ad, accumulate);
    vout<V>(ddc + j * ldd + C + c0, ac, accumulate);
}

// ---- max aggregation over the k-list (torch_scatter.scatter(reduce='max') at nn/deltaconv.py:52,54)
// out[i,c] = max_s h[nbr[i,s], c]; arg[i,c] = first maximal slot s (uint8; k <= 255).
template <int V>
DC_HD void knn_max_fwd(long t, int groups, const int* nbr, int k, const float* h, long ldh, float* out, long ldo,
                       unsigned char* arg, long lda) {
    const long i = t / groups;
    const int c0 = (int)(t % groups) * V;
    Vec<V> best = vload<V>(h + (long)nbr[i * k] * ldh + c0);
    unsigned char slot[V];
#pragma unroll
    for (int q = 0; q < V; ++q) slot[q] = 0;
#pragma unroll 4
    for (int s = 1; s < k; ++s) {
        const Vec<V> hv = vload<V>(h + (long)nbr[i * k + s] * ldh + c0);
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

// dh[j,c] (+)= sum over in-edges e=(i,s) of j with arg[i,c] == s of dout[i,c]
template <int V>
DC_HD void knn_max_bwd(long t, int groups, const int* tptr, const int* tedge, int k, const unsigned char* arg,
                       long lda, const float* dout, long ldo, float* dh, long ldh, int accumulate) {
    const long j = t / groups;
    const int c0 = (int)(t % groups) * V;
    Vec<V> acc = vzero<V>();
    const int t1 = tptr[j + 1];
    for (int p = tptr[j]; p < t1; ++p) {
        const long e = tedge[p];
        const long i = e / k;
        const unsigned char s = (unsigned char)(e - i * k);
        bool any = false;
        bool hit[V];
#pragma unroll
        for (int q = 0; q < V; ++q) {
            hit[q] = arg[i * lda + c0 + q] == s;
            any = any || hit[q];
        }
        if (any) {
            const Vec<V> g = vload<V>(dout + i * ldo + c0);
#pragma unroll
            for (int q = 0; q < V; ++q) acc.v[q] += hit[q] ? g.v[q] : 0.f;
        }
    }
    vout<V>(dh + j * ldh + c0, acc, accumulate);
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
