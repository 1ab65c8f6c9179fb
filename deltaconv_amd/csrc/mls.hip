// Moving-least-squares assembly of the gradient / divergence operators in fixed-degree (ELL) form.
// Replaces build_grad_div and its helpers coords_projected, gaussian_weights,
// weighted_least_squares, fit_vector_mapping
// (/root/reference/deltaconv/geometry/grad_div_mls.py:72-277), including the torch_scatter
// segment reductions (:112,165,259) and the torch_sparse construction (:263,275) -- the operators
// are never materialised as COO/CSR: G[Nt,k,2] and D[Nt,k,2] share nbr[Nt,k].
//
// Three launches over a (blocks, cloud) grid:
//   1 mls_avgdist  per-cloud mean edge length          (16 fixed chunks per cloud, combined in order by the consumers)
//   2 mls_fit      per point: 6x6 weighted normal equations, Cholesky, gradient rows, surface
//                  coefficients; per-cloud infinity norm by integer atomicMax (order independent)
//   3 mls_div      per edge: normalise G, contract with the pushed-forward frame map -> D
// Algorithmic bytes: ~36 B/point in + 4 B/edge ids + 16 B/edge out (14.3 MB at B=32,N=1024,k=20);
// ~4 kflop/point in fp64.  Arithmetic: point_math.h.
#include "common.h"
#include "point_math.h"

namespace {

// Per-cloud mean edge length: AVG_CHUNKS fixed partitions per cloud (one workgroup each), combined in chunk order by every
// consumer -- deterministic, and the partition depends on the cloud's OWN size only (a cloud's operators do not depend on
// what else is in the batch).  Round 6: was one workgroup of 1024 threads per cloud -- 96 us for one cloud of 4096 points,
// k = 30 (the per-rank step of 8-GPU strong scaling), 14 us at 32 x 1024.
constexpr int AVG_THREADS = 256, AVG_CHUNKS = 16;
__host__ __device__ inline int avg_chunk_points(int n) {            // points per chunk: >= 256, a multiple of 64, <= 16 chunks
    const int c = ((n + AVG_CHUNKS - 1) / AVG_CHUNKS + 63) / 64 * 64;
    return c < 256 ? 256 : c;
}
__device__ inline double cloud_avg(const double* __restrict__ part, int cloud, int n) {
    if (n <= 0) return 0.0;
    const int cp = avg_chunk_points(n), chunks = (n + cp - 1) / cp;
    double t = 0;
    for (int c = 0; c < chunks; ++c) t += part[cloud * AVG_CHUNKS + c];
    return t / n;                                                    // scatter_mean over the cloud (:112)
}
// sum of `acc` over the workgroup -> part[cloud][chunk] (wave butterflies, then the waves in order)
__device__ inline void avg_chunk_store(double acc, double* __restrict__ part, int cloud, int chunk) {
    __shared__ double wsum[AVG_THREADS / 64];
    acc = dc_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < AVG_THREADS / 64; ++w) t += wsum[w];
        part[cloud * AVG_CHUNKS + chunk] = t;
    }
}

// The first launch of the assembly also does the two pieces of per-cloud / per-point set-up that used to be launches of their
// own (round 6: -2 launches per step): it clears the cloud's infinity-norm word (mls_fit's atomicMax target) and, when
// `normal` is given, writes the tangent frames of build_tangent_basis (grad_div_mls.py:50-69; same function, same bits as
// dc_tangent_basis) that mls_fit / mls_div read afterwards.
__global__ __launch_bounds__(AVG_THREADS) void mls_avgdist_kernel(const float* __restrict__ pos,
                                                                  const int* __restrict__ nbr,
                                                                  const int* __restrict__ cloud_ptr, int k,
                                                                  double* __restrict__ avg, unsigned* __restrict__ inf_bits,
                                                                  const float* __restrict__ normal, float* __restrict__ xb,
                                                                  float* __restrict__ yb) {
    const int cloud = blockIdx.y, chunk = blockIdx.x;
    const int begin = cloud_ptr[cloud], n = cloud_ptr[cloud + 1] - begin;
    const int cp = avg_chunk_points(n), q0 = chunk * cp, q1 = min(q0 + cp, n);
    if (chunk == 0 && threadIdx.x == 0) inf_bits[cloud] = 0u;
    if (q0 >= n) return;                                             // workgroup-uniform
    double acc = 0;
    for (int q = q0 + threadIdx.x; q < q1; q += AVG_THREADS) {
        const long i = begin + q;
        if (normal) dcmath::tangent_basis_point(normal + 3 * i, xb + 3 * i, yb + 3 * i);
        acc += dcmath::point_dist_sum(pos, nbr + i * k, i, k) / k;  // dist.mean(dim=1) (:112)
    }
    avg_chunk_store(acc, avg, cloud, chunk);
}

template <bool SHAPE>
__global__ __launch_bounds__(128) void mls_fit_kernel(const float* __restrict__ pos, const float* __restrict__ normal,
                                                      const float* __restrict__ xb, const float* __restrict__ yb,
                                                      const int* __restrict__ nbr, const int* __restrict__ cloud_ptr,
                                                      int k, double kernel_width, double lambda, double lambda_shape,
                                                      const double* __restrict__ avg, float* __restrict__ G,
                                                      double* __restrict__ coef, unsigned* __restrict__ inf_bits) {
    const int cloud = blockIdx.y;
    const int begin = cloud_ptr[cloud], n = cloud_ptr[cloud + 1] - begin;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    float rownorm = 0.f;
    if (q < n) {
        const long i = begin + q;
        rownorm = dcmath::mls_fit_point<SHAPE>(pos, normal, xb, yb, nbr + i * k, i, k, cloud_avg(avg, cloud, n), kernel_width, lambda,
                                               G + i * k * 2, coef + i * 6, lambda_shape);
    }
    rownorm = dc_wave_max(rownorm);  // non-negative floats order like their bit patterns
    if ((threadIdx.x & 63) == 0 && rownorm > 0.f) atomicMax(inf_bits + cloud, __float_as_uint(rownorm));
}

__global__ __launch_bounds__(256) void mls_div_kernel(const float* __restrict__ pos, const float* __restrict__ normal,
                                                      const float* __restrict__ xb, const float* __restrict__ yb,
                                                      const int* __restrict__ nbr, const int* __restrict__ cloud_ptr,
                                                      int k, int normalized, const double* __restrict__ coef,
                                                      const unsigned* __restrict__ inf_bits, float* __restrict__ G,
                                                      float* __restrict__ D) {
    const int cloud = blockIdx.y;
    const int begin = cloud_ptr[cloud], n = cloud_ptr[cloud + 1] - begin;
    const long le = (long)blockIdx.x * blockDim.x + threadIdx.x;  // edge within the cloud
    if (le >= (long)n * k) return;
    const long e = (long)begin * k + le;
    const long i = e / k;
    const long j = nbr[e];
    const dcmath::Frame fi = dcmath::load_frame(pos, normal, xb, yb, i);
    double c[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) c[a] = coef[i * 6 + a];
    const float inf_norm = normalized ? __uint_as_float(inf_bits[cloud]) : 0.f;
    float g[2] = {G[2 * e], G[2 * e + 1]};
    float d[2];
    dcmath::mls_div_edge(fi, c, dcmath::ld3(pos + 3 * j), dcmath::ld3(xb + 3 * j), dcmath::ld3(yb + 3 * j), inf_norm,
                         g, d);
    G[2 * e] = g[0]; G[2 * e + 1] = g[1];
    D[2 * e] = d[0]; D[2 * e + 1] = d[1];
}

// ---- the stages on their own: coords_projected, gaussian_weights, weighted_least_squares, fit_vector_mapping ----
// The reference exports and tests them one by one (grad_div_mls.py:72,100,119,155; test_grad_div_mls.py:58-275);
// the product path runs them fused (mls_fit / mls_div above) through the SAME dcmath:: functions.

// coords_projected (:72-97): frame of edge e = frame[e / k] (the reference expands the frames k times by position),
// positions by row / col.
__global__ __launch_bounds__(256) void mls_coords_kernel(const float* __restrict__ pos, const float* __restrict__ normal,
                                                         const float* __restrict__ xb, const float* __restrict__ yb,
                                                         const int* __restrict__ row, const int* __restrict__ col,
                                                         long num_edges, int k, float* __restrict__ coords) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= num_edges) return;
    const long f = e / k;
    dcmath::Frame fr{dcmath::ld3(pos + 3 * (long)row[e]), dcmath::ld3(normal + 3 * f), dcmath::ld3(xb + 3 * f),
                     dcmath::ld3(yb + 3 * f)};
    const dcmath::EdgeGeom g = dcmath::edge_geom(fr, dcmath::ld3(pos + 3 * (long)col[e]));
    coords[2 * e] = (float)g.u;
    coords[2 * e + 1] = (float)g.v;
}

// gaussian_weights, the cloud average (:112): mean over the cloud's points of the mean over a point's k distances
__global__ __launch_bounds__(AVG_THREADS) void mls_avg_of_dist_kernel(const float* __restrict__ dist,
                                                                      const int* __restrict__ cloud_ptr, int k,
                                                                      double* __restrict__ avg) {
    const int cloud = blockIdx.y, chunk = blockIdx.x;
    const int begin = cloud_ptr[cloud], n = cloud_ptr[cloud + 1] - begin;
    const int cp = avg_chunk_points(n), q0 = chunk * cp, q1 = min(q0 + cp, n);
    if (q0 >= n) return;
    double acc = 0;
    for (int q = q0 + threadIdx.x; q < q1; q += AVG_THREADS) {
        const float* d = dist + (long)(begin + q) * k;
        double s = 0;
        for (int e = 0; e < k; ++e) s += (double)d[e];
        acc += s / k;
    }
    avg_chunk_store(acc, avg, cloud, chunk);
}

__global__ __launch_bounds__(128) void mls_weights_kernel(const float* __restrict__ dist,
                                                          const int* __restrict__ cloud_ptr, int k, double kernel_width,
                                                          const double* __restrict__ avg, float* __restrict__ weights) {
    const int cloud = blockIdx.y;
    const int begin = cloud_ptr[cloud], n = cloud_ptr[cloud + 1] - begin;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const long i = begin + q;
    dcmath::gaussian_weights_point(dist + i * k, k, cloud_avg(avg, cloud, n), kernel_width, weights + i * k);
}

// weighted_least_squares (:119-144): one point per thread
__global__ __launch_bounds__(128) void mls_wls_kernel(const float* __restrict__ coords, const float* __restrict__ weights,
                                                      int num_points, int k, double lambda, float* __restrict__ wls) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_points) return;
    dcmath::wls_point(coords + i * k * 2, weights + i * k, k, lambda, wls + i * k * 6);
}

// fit_vector_mapping (:155-194): edges come in groups of k with one centre (row) per group; every thread forms its
// group's six surface coefficients c = sum_s wls[s, :] * height_s (scatter_add over row, :165) in slot order
__global__ __launch_bounds__(256) void mls_vmap_kernel(const float* __restrict__ pos, const float* __restrict__ normal,
                                                       const float* __restrict__ xb, const float* __restrict__ yb,
                                                       const int* __restrict__ row, const int* __restrict__ col,
                                                       long num_edges, int k, const float* __restrict__ wls,
                                                       const float* __restrict__ coords, float* __restrict__ out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= num_edges) return;
    const long g0 = e / k * k;
    const long i = row[e];
    const dcmath::Frame fi = dcmath::load_frame(pos, normal, xb, yb, i);
    double c[6] = {0, 0, 0, 0, 0, 0};
    for (int s = 0; s < k; ++s) {
        const dcmath::V3 d = dcmath::sub(dcmath::ld3(pos + 3 * (long)col[g0 + s]), fi.p);
        const double h = dcmath::dot(fi.n, d);                        // patch_f (:163)
#pragma unroll
        for (int a = 0; a < 6; ++a) c[a] += (double)wls[(g0 + s) * 6 + a] * h;
    }
    const long j = col[e];
    double m[4];
    dcmath::vector_map(fi, c, (double)coords[2 * e], (double)coords[2 * e + 1], dcmath::ld3(xb + 3 * j),
                       dcmath::ld3(yb + 3 * j), m);
#pragma unroll
    for (int a = 0; a < 4; ++a) out[4 * e + a] = (float)m[a];
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

DC_EXPORT size_t dc_mls_workspace_bytes(int32_t num_clouds, int32_t num_points) {
    return align_up((size_t)num_clouds * AVG_CHUNKS * 8, 256) + align_up((size_t)num_clouds * 4, 256) + (size_t)num_points * 48;
}

namespace {
int mls_assemble(const char* name, const float* pos, const float* normal, float* x_basis, float* y_basis, bool make_basis,
                 const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points,
                 int32_t max_cloud_size, int32_t k, float kernel_width, float regularizer, bool shape,
                 float shape_regularizer, int32_t normalized, float* G, float* D, void* workspace,
                 size_t workspace_bytes, void* stream) {
    if (!(pos && normal && x_basis && y_basis && nbr && cloud_ptr && G && D)) {
        dc_set_error("%s: null pointer", name);
        return DC_ERR_ARG;
    }
    if (!(k >= 1 && num_clouds >= 0 && num_points >= 0 && max_cloud_size >= 0)) {
        dc_set_error("%s: bad size", name);
        return DC_ERR_ARG;
    }
    if (num_clouds == 0 || num_points == 0) return DC_OK;
    if (!workspace || workspace_bytes < dc_mls_workspace_bytes(num_clouds, num_points)) {
        dc_set_error("%s: workspace too small (%zu < %zu)", name, workspace_bytes,
                     dc_mls_workspace_bytes(num_clouds, num_points));
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    double* avg = reinterpret_cast<double*>(ws);
    unsigned* inf_bits = reinterpret_cast<unsigned*>(ws + align_up((size_t)num_clouds * AVG_CHUNKS * 8, 256));
    double* coef = reinterpret_cast<double*>(ws + align_up((size_t)num_clouds * AVG_CHUNKS * 8, 256) +
                                             align_up((size_t)num_clouds * 4, 256));
    hipLaunchKernelGGL(mls_avgdist_kernel, dim3(AVG_CHUNKS, num_clouds), dim3(AVG_THREADS), 0, s, pos, nbr, cloud_ptr, k, avg, inf_bits,
                       make_basis ? normal : nullptr, x_basis, y_basis);
    const dim3 grid(dc_cdiv(max_cloud_size, 128), num_clouds);
    if (shape)
        hipLaunchKernelGGL(mls_fit_kernel<true>, grid, dim3(128), 0, s, pos, normal, x_basis, y_basis, nbr, cloud_ptr, k,
                           (double)kernel_width, (double)regularizer, (double)shape_regularizer, avg, G, coef, inf_bits);
    else
        hipLaunchKernelGGL(mls_fit_kernel<false>, grid, dim3(128), 0, s, pos, normal, x_basis, y_basis, nbr, cloud_ptr, k,
                           (double)kernel_width, (double)regularizer, 0.0, avg, G, coef, inf_bits);
    hipLaunchKernelGGL(mls_div_kernel, dim3(dc_cdiv((long long)max_cloud_size * k, 256), num_clouds), dim3(256), 0, s,
                       pos, normal, x_basis, y_basis, nbr, cloud_ptr, k, normalized, coef, inf_bits, G, D);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dc_set_error("%s: %s", name, hipGetErrorString(e));
        return DC_ERR_LAUNCH;
    }
    return DC_OK;
}
}  // namespace

DC_EXPORT int dc_mls_assemble(const float* pos, const float* normal, const float* x_basis, const float* y_basis,
                              const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points,
                              int32_t max_cloud_size, int32_t k, float kernel_width, float regularizer,
                              int32_t normalized, float* G, float* D, void* workspace, size_t workspace_bytes,
                              void* stream) {
    return mls_assemble("dc_mls_assemble", pos, normal, const_cast<float*>(x_basis), const_cast<float*>(y_basis), false, nbr,
                        cloud_ptr, num_clouds, num_points, max_cloud_size, k, kernel_width, regularizer, false, 0.f, normalized,
                        G, D, workspace, workspace_bytes, stream);
}

// build_tangent_basis + build_grad_div in one call (deltanet_base.py:59-61,69 with normals given): x_basis / y_basis are
// OUTPUTS, written by the first launch of the assembly -- the model's path; same bits as dc_tangent_basis + dc_mls_assemble
DC_EXPORT int dc_mls_assemble_normals(const float* pos, const float* normal, const int32_t* nbr, const int32_t* cloud_ptr,
                                      int32_t num_clouds, int32_t num_points, int32_t max_cloud_size, int32_t k,
                                      float kernel_width, float regularizer, int32_t normalized, float* x_basis,
                                      float* y_basis, float* G, float* D, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    return mls_assemble("dc_mls_assemble_normals", pos, normal, x_basis, y_basis, true, nbr, cloud_ptr, num_clouds,
                        num_points, max_cloud_size, k, kernel_width, regularizer, false, 0.f, normalized, G, D, workspace,
                        workspace_bytes, stream);
}

// build_grad_div(..., shape_regularizer=s) (grad_div_mls.py:241-244,266-267): the gradient rows come from the fit with
// `regularizer`, the surface (height-field) coefficients behind the divergence rows from a fit with `shape_regularizer`
DC_EXPORT int dc_mls_assemble_shape(const float* pos, const float* normal, const float* x_basis, const float* y_basis,
                                    const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points,
                                    int32_t max_cloud_size, int32_t k, float kernel_width, float regularizer,
                                    float shape_regularizer, int32_t normalized, float* G, float* D, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return mls_assemble("dc_mls_assemble_shape", pos, normal, const_cast<float*>(x_basis), const_cast<float*>(y_basis), false,
                        nbr, cloud_ptr, num_clouds, num_points, max_cloud_size, k, kernel_width, regularizer, true,
                        shape_regularizer, normalized, G, D, workspace, workspace_bytes, stream);
}

// ---- stage entry points (the reference's public helpers, grad_div_mls.py:72,100,119,155) ------------------------
namespace {
int launch_status(const char* name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dc_set_error("%s: %s", name, hipGetErrorString(e));
        return DC_ERR_LAUNCH;
    }
    return DC_OK;
}
}  // namespace

DC_EXPORT int dc_mls_coords(const float* pos, const float* normal, const float* x_basis, const float* y_basis,
                            const int32_t* row, const int32_t* col, int64_t num_edges, int32_t k, float* coords,
                            void* stream) {
    if (num_edges < 0 || k < 1) {
        dc_set_error("dc_mls_coords: bad size");
        return DC_ERR_ARG;
    }
    if (num_edges == 0) return DC_OK;
    if (!(pos && normal && x_basis && y_basis && row && col && coords)) {
        dc_set_error("dc_mls_coords: null pointer");
        return DC_ERR_ARG;
    }
    hipLaunchKernelGGL(mls_coords_kernel, dim3(dc_cdiv((long long)num_edges, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pos, normal, x_basis, y_basis, row, col, (long)num_edges, k,
                       coords);
    return launch_status("dc_mls_coords");
}

DC_EXPORT int dc_mls_gaussian_weights(const float* dist, const int32_t* cloud_ptr, int32_t num_clouds,
                                      int32_t max_cloud_size, int32_t k, float kernel_width, float* weights,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (num_clouds < 0 || max_cloud_size < 0 || k < 1) {
        dc_set_error("dc_mls_gaussian_weights: bad size");
        return DC_ERR_ARG;
    }
    if (num_clouds == 0 || max_cloud_size == 0) return DC_OK;
    if (!(dist && cloud_ptr && weights)) {
        dc_set_error("dc_mls_gaussian_weights: null pointer");
        return DC_ERR_ARG;
    }
    if (!workspace || workspace_bytes < (size_t)num_clouds * AVG_CHUNKS * 8) {
        dc_set_error("dc_mls_gaussian_weights: workspace too small (%zu < %zu)", workspace_bytes,
                     (size_t)num_clouds * AVG_CHUNKS * 8);
        return DC_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* avg = static_cast<double*>(workspace);
    hipLaunchKernelGGL(mls_avg_of_dist_kernel, dim3(AVG_CHUNKS, num_clouds), dim3(AVG_THREADS), 0, s, dist, cloud_ptr, k, avg);
    hipLaunchKernelGGL(mls_weights_kernel, dim3(dc_cdiv(max_cloud_size, 128), num_clouds), dim3(128), 0, s, dist,
                       cloud_ptr, k, (double)kernel_width, avg, weights);
    return launch_status("dc_mls_gaussian_weights");
}

DC_EXPORT int dc_mls_wls(const float* coords, const float* weights, int32_t num_points, int32_t k, float regularizer,
                         float* wls, void* stream) {
    if (num_points < 0 || k < 1) {
        dc_set_error("dc_mls_wls: bad size");
        return DC_ERR_ARG;
    }
    if (num_points == 0) return DC_OK;
    if (!(coords && weights && wls)) {
        dc_set_error("dc_mls_wls: null pointer");
        return DC_ERR_ARG;
    }
    hipLaunchKernelGGL(mls_wls_kernel, dim3(dc_cdiv(num_points, 128)), dim3(128), 0, static_cast<hipStream_t>(stream),
                       coords, weights, num_points, k, (double)regularizer, wls);
    return launch_status("dc_mls_wls");
}

DC_EXPORT int dc_mls_vector_mapping(const float* pos, const float* normal, const float* x_basis, const float* y_basis,
                                    const int32_t* row, const int32_t* col, int64_t num_edges, int32_t k,
                                    const float* wls, const float* coords, float* mapping, void* stream) {
    if (num_edges < 0 || k < 1 || num_edges % k != 0) {
        dc_set_error("dc_mls_vector_mapping: bad size (edges come in groups of k)");
        return DC_ERR_ARG;
    }
    if (num_edges == 0) return DC_OK;
    if (!(pos && normal && x_basis && y_basis && row && col && wls && coords && mapping)) {
        dc_set_error("dc_mls_vector_mapping: null pointer");
        return DC_ERR_ARG;
    }
    hipLaunchKernelGGL(mls_vmap_kernel, dim3(dc_cdiv((long long)num_edges, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pos, normal, x_basis, y_basis, row, col, (long)num_edges, k,
                       wls, coords, mapping);
    return launch_status("dc_mls_vector_mapping");
}
