/* deltaconv_hip.h -- C ABI of libdeltaconv_hip.so (MI355X / gfx950).
 *
 * The reference (rubenwiersma/deltaconv) has no FFI boundary on its hot path: it is Python over
 * third-party torch extensions (torch_cluster / torch_sparse / torch_scatter) and ATen.  The entry
 * points below are what a binding for that path binds INSTEAD of those packages; each one cites
 * the reference call site(s) it replaces (paths relative to /root/reference).  INTEGRATION.md
 * shows the ctypes stub and the reference-side edits.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP), fp32 / int32 / uint8 as typed; no torch types.
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued, nothing syncs,
 *     nothing allocates (callers pass workspaces) -> every entry point is hipGraph-capturable.
 *   - return 0 on success, <0 on error (DC_ERR_*); dc_last_error() gives the message.
 *   - layouts: nbr[Nt,k] centre-major (edge e = i*k+s); G,D [Nt,k,2]; scalar fields [Nt, ld];
 *     vector fields [2Nt, ld] with row 2i = u-, row 2i+1 = v-component
 *     (deltaconv/geometry/operators.py:4-21); `ld*` = row stride in floats (>= row length), so
 *     outputs can land directly in a column block of a wider concat buffer.
 *   - clouds of a batch are contiguous; cloud_ptr[B+1] int32 offsets (sorted `batch` vector of
 *     deltaconv/models/deltanet_base.py:44).
 */
#ifndef DELTACONV_HIP_H
#define DELTACONV_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC_OK 0
#define DC_ERR_ARG (-1)
#define DC_ERR_LAUNCH (-2)
#define DC_ERR_WORKSPACE (-3)

int32_t dc_version(void);
const char* dc_last_error(void);
/* Experiment switches for A/B measurements (product defaults: all 0 except key 0).  key 0: XCD-aware block remap
 * (default 1); 1: edge-at-a-time max-aggregation backward; 2: value 2 = weight gradients through the round-1
 * direct-load kernel; 3: value 1 = dense products through the exact fp32 MFMA chain (bitwise an fmaf chain) instead of
 * the bf16 split products (three bf16 planes per fp32 operand, six partial products, fp32 accumulation: error against
 * fp64 no larger than the chain's); 4: first-round phase shift of every second 128 x 128 GEMM workgroup of a CU in
 * percent of a K loop (0 = default 50, negative = off); 5 / 6: force the weight-gradient tile (1..4) / slab count;
 * 7: units (tile x 64-channel slab) per workgroup of the persistent two-piece tiled applies (0 = launcher's choice);
 * 8: value 1 = CSC count / scan / fill by one workgroup per cloud (round 3) instead of eight column ranges per cloud;
 * 9: value 1 = ignore pre-split weight planes (every product splits its weight operand in the K loop, as in round 3);
 * 10: value 1 = cross-entropy of <= 64 rows through the two-launch form (same bits as the one-launch form);
 * 11: value 1 = dense products with fewer than 256 workgroups keep 128-column tiles (round-6 rule off). */
int dc_set_option(int32_t key, int32_t value);

/* Deferred finalisers (round 6): a column reduction is two launches (partials, finaliser).  Between dc_finalisers_begin() and
 * dc_finalisers_end() a call of dc_linear_bn_stats_forward / dc_bn_act_backward_reduce announced by dc_finaliser_defer_next()
 * (one-shot, per call) queues its finaliser instead of launching it; dc_finalisers_end launches ONE kernel for all of them (at
 * most 4; discard != 0: drops them).  The independent products of a DeltaConv layer (max-aggregation stream and s_mlp,
 * deltaconv/nn/deltaconv.py:50-59) share a finaliser launch this way -- same sums, same bits.  The coefficient outputs of the
 * queued calls are valid behind dc_finalisers_end; their workspaces must stay alive until then.  Thread-local host state. */
int dc_finalisers_begin(void);
int dc_finaliser_defer_next(void);
/* with it: the dense product of the announced dc_linear_bn_stats_forward call (weight planes, whole tiles) waits as well; two
 * queued products of one kernel instantiation run as ONE launch at dc_finalisers_end (same arithmetic per element: same bits) */
int dc_gemm_defer_next(void);
int dc_finalisers_end(int32_t discard, void* stream);

/* Measurement aid (bench.py `roofline.frac`): device-clock stamps of the tiled two-piece forward applies and the tiled transposed applies.  After
 * dc_stamp_buffer(buf, slots) every launch of that family takes the next 4 x uint64 record of `buf` AT ENQUEUE TIME (a launch
 * captured into a HIP graph keeps its record across replays): [0] earliest workgroup entry, [1] latest workgroup exit with
 * its stores complete, constant 100 MHz clock; the caller sets [0] = huge, [1] = 0 before a replay.  dc_stamp_tag(record) =
 * 1000 * kind + channels (forward: kind 1 = div|curl|norm, 2 = hodge, 3 = div; transposed: 11 = div|curl|norm^T, 12 = hodge^T,
 * 13 = grad^T (+ sum), 14 = max-aggregation backward, 15 = div^T, 16 = layer-0 edge MLP backward).  buf = NULL disarms (the product default). */
int dc_stamp_buffer(uint64_t* buf, int32_t slots);
int32_t dc_stamp_count(void);
int32_t dc_stamp_tag(int32_t record);

/* ---- graph ------------------------------------------------------------------------------- */
/* knn_graph(pos, k, batch, loop=True, flow='target_to_source')  (torch_cluster via
 * torch_geometric) -- deltaconv/models/deltanet_base.py:52,63.
 * Order: fp32 ((dx*dx+dy*dy)+dz*dz) ascending, ties by lower index, self included; global ids.
 * Every cloud needs >= k points; k <= 64.  lanes_per_query: 0 = auto, 64 = wave-per-query selection
 * kernel (clouds <= 4096 points), 1 or 8 = sorted-insertion kernel. */
int dc_knn(const float* pos, const int32_t* cloud_ptr, int32_t num_clouds, int32_t max_cloud_size, int32_t k,
           int32_t lanes_per_query, int32_t* nbr, void* stream);

/* Transposed adjacency of nbr (in-edges per point, ascending edge id).  Stands in for the A^T
 * products torch_sparse autograd performs and torch_scatter's arg-indexed backward. */
size_t dc_csc_workspace_bytes(int32_t num_points);   /* upper bound for any k <= 255; 4*Nt*(k+1) suffices */
int dc_csc_build(const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points, int32_t k,
                 int32_t* tptr /*[Nt+1]*/, int32_t* tedge /*[Nt*k]*/, void* workspace, size_t workspace_bytes,
                 void* stream);

/* Same result (identical tptr / tedge) for clouds of at most 4096 points: count, scan and fill of a cloud run in one
 * workgroup on LDS counters (2 launches instead of 5).  workspace: num_points * k int32. */
int dc_csc_build_clouds(const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points,
                        int32_t max_cloud, int32_t k, int32_t* tptr, int32_t* tedge, void* workspace,
                        size_t workspace_bytes, void* stream);

/* coefT[t] = coef[tedge[t]]: G or D in CSC order for the transposed applies (once per batch). */
int dc_csc_permute_coef(const float* coef, const int32_t* tedge, int64_t num_edges, float* coefT, void* stream);

/* ---- tangent frames ------------------------------------------------------------------------ */
/* build_tangent_basis -- deltaconv/geometry/grad_div_mls.py:50-69 */
int dc_tangent_basis(const float* normal, int32_t n, float* x_basis, float* y_basis, void* stream);
/* estimate_basis -- deltaconv/geometry/grad_div_mls.py:10-47 (orientation may be NULL).
 * x_basis sign convention: largest-|component| positive (LAPACK's sign is arbitrary). */
int dc_estimate_basis(const float* pos, const int32_t* nbr, int32_t n, int32_t k, const float* orientation,
                      float* normal, float* x_basis, float* y_basis, void* stream);

/* ---- operator assembly ---------------------------------------------------------------------- */
/* build_grad_div (+ coords_projected, gaussian_weights, weighted_least_squares,
 * fit_vector_mapping) -- deltaconv/geometry/grad_div_mls.py:72-277.  Outputs the gradient and
 * divergence operators as G[Nt,k,2], D[Nt,k,2] over nbr (never COO/CSR). */
size_t dc_mls_workspace_bytes(int32_t num_clouds, int32_t num_points);
int dc_mls_assemble(const float* pos, const float* normal, const float* x_basis, const float* y_basis,
                    const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points,
                    int32_t max_cloud_size, int32_t k, float kernel_width, float regularizer, int32_t normalized,
                    float* G, float* D, void* workspace, size_t workspace_bytes, void* stream);
/* build_tangent_basis + build_grad_div in one call -- the model's path when normals are given (deltanet_base.py:59-61,69):
 * x_basis / y_basis [Nt,3] are OUTPUTS (written by the assembly's first launch); same values as dc_tangent_basis followed
 * by dc_mls_assemble, two launches fewer. */
int dc_mls_assemble_normals(const float* pos, const float* normal, const int32_t* nbr, const int32_t* cloud_ptr,
                            int32_t num_clouds, int32_t num_points, int32_t max_cloud_size, int32_t k, float kernel_width,
                            float regularizer, int32_t normalized, float* x_basis, float* y_basis, float* G, float* D,
                            void* workspace, size_t workspace_bytes, void* stream);
/* build_grad_div(..., shape_regularizer=s) -- grad_div_mls.py:241-244,266-267 (weighted_least_squares :146-150): the
 * gradient rows come from the fit regularised by `regularizer`, the surface coefficients behind the divergence rows
 * (fit_vector_mapping) from a second fit regularised by `shape_regularizer`. */
int dc_mls_assemble_shape(const float* pos, const float* normal, const float* x_basis, const float* y_basis,
                          const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds, int32_t num_points,
                          int32_t max_cloud_size, int32_t k, float kernel_width, float regularizer,
                          float shape_regularizer, int32_t normalized, float* G, float* D, void* workspace,
                          size_t workspace_bytes, void* stream);

/* The stages of build_grad_div on their own: the reference exports and tests them one by one
 * (deltaconv/geometry/__init__.py:3; test/geometry/test_grad_div_mls.py:58-275).  Same device functions as the
 * fused kernels behind dc_mls_assemble (csrc/point_math.h); fp32 in / fp32 out, fp64 inside.
 * Edges are centre-major in groups of k (edge e belongs to group e / k), as everywhere in the reference
 * (grad_div_mls.py:24-25,85,224). */
/* coords_projected -- grad_div_mls.py:72-97: coords[E,2] = ((p_col - p_row) minus its normal part) . (x, y) of frame e / k */
int dc_mls_coords(const float* pos, const float* normal, const float* x_basis, const float* y_basis,
                  const int32_t* row, const int32_t* col, int64_t num_edges, int32_t k, float* coords, void* stream);
/* gaussian_weights -- grad_div_mls.py:100-116: dist[Nt*k] -> weights[Nt*k]; cloud_ptr[num_clouds+1] delimits the
 * clouds whose mean edge length scales the kernel (batch=None: one cloud); workspace >= 128 * num_clouds bytes
 * (16 ordered fp64 partial sums per cloud) */
int dc_mls_gaussian_weights(const float* dist, const int32_t* cloud_ptr, int32_t num_clouds, int32_t max_cloud_size,
                            int32_t k, float kernel_width, float* weights, void* workspace, size_t workspace_bytes,
                            void* stream);
/* weighted_least_squares -- grad_div_mls.py:119-152: coords[Nt*k,2], weights[Nt*k] -> wls[Nt*k,6] =
 * ((B^T W B + regularizer I)^-1 B^T W)^T per point (the shape_regularizer variant is a second call) */
int dc_mls_wls(const float* coords, const float* weights, int32_t num_points, int32_t k, float regularizer, float* wls,
               void* stream);
/* fit_vector_mapping -- grad_div_mls.py:155-194: mapping[E,2,2] = g^-1 T per edge; row must be constant inside a
 * group of k edges (the scatter_add over row, :165, is the sum over the group in slot order) */
int dc_mls_vector_mapping(const float* pos, const float* normal, const float* x_basis, const float* y_basis,
                          const int32_t* row, const int32_t* col, int64_t num_edges, int32_t k, const float* wls,
                          const float* coords, float* mapping, void* stream);

/* ---- operator applies (SparseTensor @ dense: deltanet_base.py:78; deltaconv.py:57,66;
 *      operators.py:27,33,40,43) ------------------------------------------------------------- */
/* out[2Nt,C] = grad @ x[Nt,C] */
int dc_apply_grad(const float* G, const int32_t* nbr, int32_t n, int32_t k, const float* x, int32_t C, int64_t ldx,
                  float* out, int64_t ldo, void* stream);
/* out[Nt,C] = div @ v[2Nt,C] */
int dc_apply_div(const float* D, const int32_t* nbr, int32_t n, int32_t k, const float* v, int32_t C, int64_t ldv,
                 float* out, int64_t ldo, void* stream);
/* out[Nt,3C] = [div v | curl v | norm v]  (deltaconv.py:57 + operators.py:4-7,23-27), v read once */
int dc_apply_div_curl_norm(const float* D, const int32_t* nbr, int32_t n, int32_t k, const float* v, int32_t C,
                           int64_t ldv, float* out, int64_t ldo, void* stream);
/* out[2Nt,C] = hodge_laplacian(v) given dc[Nt,2C] = [div v | curl v]  (operators.py:35-46) */
int dc_apply_hodge(const float* G, const int32_t* nbr, int32_t n, int32_t k, const float* dc, int32_t C, int64_t lddc,
                   float* out, int64_t ldo, void* stream);

/* Transposed forms = backward of the four applies (operators carry no gradient).  The coefficient
 * argument is the operator in CSC order (dc_csc_permute_coef).  accumulate != 0 adds into the
 * destination (gradient accumulation without a separate add). */
int dc_apply_grad_T(const float* GT, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, const float* dy,
                    int32_t C, int64_t ldy, float* dx, int64_t ldx, int32_t accumulate, void* stream);
/* out[n, C] = a (+ b when non-null) + grad^T dy: dc_apply_grad_T with the accumulation of the other gradients of x'
 * (autograd's adds for a tensor with several consumers: the next layer and the concatenated embedding input,
 * deltanet_base.py:82-87, deltanet_classification.py:42) folded into the store; `out` is a fresh tensor. */
int dc_apply_grad_T_sum(const float* GT, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k,
                        const float* dy, int32_t C, int64_t ldy, const float* a, int64_t lda, const float* b,
                        int64_t ldb, float* out, int64_t ldo, void* stream);
int dc_apply_div_T(const float* DT, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, const float* dy,
                   int32_t C, int64_t ldy, float* dv, int64_t ldv, int32_t accumulate, void* stream);
int dc_apply_hodge_T(const float* GT, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, const float* dh,
                     int32_t C, int64_t ldh, float* ddc, int64_t lddc, int32_t accumulate, void* stream);
int dc_apply_div_curl_norm_T(const float* DT, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k,
                             const float* dout, int32_t C, int64_t ldo, const float* v, int64_t ldv, float* dv,
                             int64_t lddv, int32_t accumulate, void* stream);

/* ---- forward applies / max aggregation from a TILE PLAN --------------------------------------------------------
 * Same operations as dc_apply_{grad,div,div_curl_norm,hodge} / dc_knn_max[_affine] (same reference call sites:
 * deltanet_base.py:78; nn/deltaconv.py:52-57,66; geometry/operators.py:27-43), results identical bit for bit; the
 * neighbour rows come from LDS instead of the gather path.  The plan groups the points of every cloud into tiles of
 * P (32 or 64) points that are consecutive on a Morton curve through the positions and lists the unique neighbour
 * rows of each tile (layout: deltaconv_amd/csrc/tile_plan.h); it depends on positions + graph only and is built once
 * per batch, like the CSC.  The reference has no counterpart (torch_sparse / torch_scatter gather from global memory).
 * Restrictions: clouds of at most dc_tile_plan_max_cloud() points, k even, k <= 64, P * k <= 2048; the apply entry
 * points need C % 64 == 0 and 16-byte aligned rows and return DC_ERR_ARG otherwise (use the plain entry points).
 * The operator argument is the same G / D as for the plain entry points.  Occupied tile ids are a dense prefix of
 * [0, num_tiles): cloud b owns sum_{c<b} ceil(N_c / P) + [0, ceil(N_b / P)); for equal-sized clouds num_tiles is exact. */
int32_t dc_tile_plan_tiles(int32_t num_points, int32_t num_clouds, int32_t max_cloud, int32_t P);   /* number of tile ids T: the `num_tiles` of every call below */
size_t dc_tile_plan_words(int32_t num_tiles, int32_t k, int32_t P);   /* plan size in int32 words */
int32_t dc_tile_plan_max_cloud(void);
int dc_tile_plan_build(const float* pos, const int32_t* nbr, const int32_t* cloud_ptr, int32_t num_clouds,
                       int32_t num_points, int32_t max_cloud, int32_t k, int32_t P, int32_t* plan, void* stream);
int dc_apply_grad_tiled(const float* G, const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles,
                        int32_t k, int32_t P, const float* x, int32_t C, int64_t ldx, float* out, int64_t ldo,
                        void* stream);
int dc_apply_div_tiled(const float* D, const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles,
                       int32_t k, int32_t P, const float* v, int32_t C, int64_t ldv, float* out, int64_t ldo,
                       void* stream);
int dc_apply_div_curl_norm_tiled(const float* D, const int32_t* plan, const int32_t* nbr, int32_t n,
                                 int32_t num_tiles, int32_t k, int32_t P, const float* v, int32_t C, int64_t ldv,
                                 float* out, int64_t ldo, void* stream);
int dc_apply_hodge_tiled(const float* G, const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles,
                         int32_t k, int32_t P, const float* dc, int32_t C, int64_t lddc, float* out, int64_t ldo,
                         void* stream);
int dc_knn_max_tiled(const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles, int32_t k, int32_t P,
                     const float* h, int32_t C, int64_t ldh, float* out, int64_t ldo, uint8_t* arg, void* stream);
int dc_knn_max_affine_tiled(const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles, int32_t k,
                            int32_t P, const float* h, int32_t C, int64_t ldh, const float* scale, const float* shift,
                            float slope, float* out, int64_t ldo, uint8_t* arg, void* stream);
/* The same with the layer's last s_mlp block in its epilogue (round 6): out = act2(scale2 h2 + shift2) + max_j act(scale h_j + shift)
 * = `x = self.s_mlp(x) + x_max` of deltaconv/nn/deltaconv.py:54-59 with both BatchNorm + LeakyReLU pairs folded in; out2 (may be
 * NULL, row stride ldo2) receives a second copy (the layer's block of the heads' torch.cat buffer).  Same bits as
 * dc_knn_max_affine_tiled followed by dc_bn_act2 with the maximum as residual. */
int dc_knn_max_affine_residual_tiled(const int32_t* plan, const int32_t* nbr, int32_t n, int32_t num_tiles, int32_t k,
                                     int32_t P, const float* h, int32_t C, int64_t ldh, const float* scale,
                                     const float* shift, float slope, const float* h2, int64_t ldh2, const float* scale2,
                                     const float* shift2, float slope2, float* out, int64_t ldo, float* out2, int64_t ldo2,
                                     uint8_t* arg, void* stream);

/* ---- transposed applies / max-aggregation backward from the TRANSPOSED TILE PLAN (round 4) -----------------------
 * Same operations as dc_apply_{grad,grad_T_sum,div,hodge,div_curl_norm}_T and dc_knn_max_backward, i.e. the backward of
 * the reference's `SparseTensor @ dense` (torch_sparse autograd spmm with A^T: nn/deltaconv.py:57,66,
 * geometry/operators.py:27,33,40,43) and of torch_scatter.scatter(reduce='max') (nn/deltaconv.py:52,54); results
 * identical bit for bit (same sums, ascending edge id per target, no floating-point atomics).  The plan keeps the tiles
 * of the forward plan, orders the targets of a tile by in-degree (the four targets of a wavefront finish together),
 * lists the unique SOURCE rows of the tile (staged in LDS by LDS-DMA) and stores the tile's in-edge lists contiguously
 * (layout: deltaconv_amd/csrc/tile_plan.h, second half).  It is built once per batch from the forward plan + the CSC.
 * Operator argument: the coefficients in TILE order (dc_tile_plan_T_permute_coef: [dc_tile_plan_T_edges(...), 2] floats).  Same restrictions as the forward tiled entry points (C % 64 == 0, 16-byte rows). */
size_t dc_tile_plan_T_words(int32_t num_points, int32_t num_clouds, int32_t num_tiles, int32_t k, int32_t P);
int64_t dc_tile_plan_T_edges(int32_t num_points, int32_t num_clouds, int32_t num_tiles, int32_t k);
int64_t dc_tile_plan_T_edge_offset(int32_t num_points, int32_t num_clouds, int32_t num_tiles, int32_t k, int32_t P);
int dc_tile_plan_T_build(const int32_t* plan, const int32_t* tptr, const int32_t* tedge, const int32_t* cloud_ptr,
                         int32_t num_clouds, int32_t num_points, int32_t max_cloud, int32_t k, int32_t P, int32_t* planT,
                         void* stream);
int dc_tile_plan_T_permute_coef(const float* coef, const float* coefB, const int32_t* planT, int32_t num_points,
                                int32_t num_clouds, int32_t num_tiles, int32_t k, int32_t P, float* coefTt, float* coefBTt,
                                void* stream);   /* coefB / coefBTt (may be NULL): a second operator of the same graph in the same launch */
int dc_apply_grad_T_tiled(const float* GTt, const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles,
                          int32_t k, int32_t P, const float* dy, int32_t C, int64_t ldy, float* dx, int64_t ldx,
                          int32_t accumulate, void* stream);
int dc_apply_grad_T_sum_tiled(const float* GTt, const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles,
                              int32_t k, int32_t P, const float* dy, int32_t C, int64_t ldy, const float* a, int64_t lda,
                              const float* b, int64_t ldb, float* out, int64_t ldo, void* stream);
int dc_apply_div_T_tiled(const float* DTt, const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles,
                         int32_t k, int32_t P, const float* dy, int32_t C, int64_t ldy, float* dv, int64_t ldv,
                         int32_t accumulate, void* stream);
int dc_apply_hodge_T_tiled(const float* GTt, const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles,
                           int32_t k, int32_t P, const float* dh, int32_t C, int64_t ldh, float* ddc, int64_t lddc,
                           int32_t accumulate, void* stream);
int dc_apply_div_curl_norm_T_tiled(const float* DTt, const int32_t* planT, int32_t n, int32_t num_clouds,
                                   int32_t num_tiles, int32_t k, int32_t P, const float* dout, int32_t C, int64_t ldo,
                                   const float* v, int64_t ldv, float* dv, int64_t lddv, int32_t accumulate, void* stream);
int dc_knn_max_backward_tiled(const int32_t* planT, int32_t n, int32_t num_clouds, int32_t num_tiles, int32_t k,
                              int32_t P, const uint8_t* arg, const float* dout, int32_t C, int64_t ldo, float* dh,
                              int64_t ldh, int32_t accumulate, void* stream);

/* ---- max aggregation (torch_scatter.scatter(reduce='max'): deltaconv/nn/deltaconv.py:52,54) -- */
/* out[i,c] = max_s h[nbr[i,s],c]; arg[Nt,C] = first maximal slot (k <= 255) */
int dc_knn_max(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh, float* out,
               int64_t ldo, uint8_t* arg, void* stream);
/* out[i,c] = max_s leaky_slope(scale_c * h[nbr[i,s],c] + shift_c): the BatchNorm + activation of s_mlp_max
 * (nn/mlp.py:9) folded into the max-aggregation gather (nn/deltaconv.py:54); same values, ties and slots
 * as dc_bn_act followed by dc_knn_max.  Backward: dc_knn_max_backward then dc_bn_act_backward. */
int dc_knn_max_affine(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh,
                      const float* scale, const float* shift, float slope, float* out, int64_t ldo, uint8_t* arg,
                      void* stream);
/* The same with the layer's last s_mlp block in its epilogue (round 6; tiled twin: dc_knn_max_affine_residual_tiled):
 * out = act2(scale2 h2 + shift2) + max_s act(scale h[nbr[i,s]] + shift) = `x = self.s_mlp(x) + x_max` of nn/deltaconv.py:54-59 with both
 * BatchNorm + LeakyReLU pairs folded in; out2 (may be NULL): second copy.  Same bits as dc_knn_max_affine + dc_bn_act2(residual). */
int dc_knn_max_affine_residual(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh,
                               const float* scale, const float* shift, float slope, const float* h2, int64_t ldh2,
                               const float* scale2, const float* shift2, float slope2, float* out, int64_t ldo, float* out2,
                               int64_t ldo2, uint8_t* arg, void* stream);
int dc_knn_max_backward(const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, const uint8_t* arg,
                        const float* dout, int32_t C, int64_t ldo, float* dh, int64_t ldh, int32_t accumulate,
                        void* stream);
/* The other aggregations of DeltaConv(aggr=...) -- torch_scatter.scatter(reduce='sum' | 'add' | 'mean') at
 * nn/deltaconv.py:52,54: out[i,c] = scale * sum_s h[nbr[i,s],c] (scale = 1, or 1/k for the mean: every point has
 * exactly k neighbours incl. itself), slots in order; backward over the CSC in ascending edge order (no atomics).
 * ('min' runs as -max(-h) through dc_knn_max.) */
int dc_knn_sum(const int32_t* nbr, int32_t n, int32_t k, const float* h, int32_t C, int64_t ldh, float scale, float* out,
               int64_t ldo, void* stream);
int dc_knn_sum_backward(const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, const float* dout, int32_t C,
                        int64_t ldo, float scale, float* dh, int64_t ldh, int32_t accumulate, void* stream);

/* ---- scalar / vector MLP stream around the dense GEMM -------------------------------------------
 * Linear -> BatchNorm1d(over rows) -> LeakyReLU(0.2)   deltaconv/nn/mlp.py:7-11, nn/nonlin.py:11-35
 * Linear -> VectorNonLin(BatchNorm1d)                  deltaconv/nn/mlp.py:13-17, nn/nonlin.py:38-86
 * (ATen native_batch_norm / leaky_relu / linalg_vector_norm / mul / div in the reference).
 * Column reductions are ordered two-stage sums (bit-reproducible).  Workspace: dc_bn_workspace_bytes. */
size_t dc_bn_workspace_bytes(int64_t rows, int32_t C);
/* batch statistics of h[R,C] -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale;
 * running_mean/var (may be NULL) updated with momentum (unbiased variance), as nn.BatchNorm1d */
int dc_bn_stats(const float* h, int64_t R, int32_t C, int64_t ldh, const float* gamma, const float* beta, float eps,
                float momentum, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                float* shift, void* workspace, size_t workspace_bytes, void* stream);
/* inference coefficients from the running statistics */
int dc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int32_t C, float* mean, float* invstd, float* scale, float* shift, void* stream);
/* y = leaky_slope(scale*h + shift) (+ residual);  slope 1 = identity, 0 = ReLU */
int dc_bn_act(const float* h, int64_t R, int32_t C, int64_t ldh, const float* scale, const float* shift, float slope,
              const float* residual, int64_t ldr, float* y, int64_t ldy, void* stream);
/* dc_bn_act with a second destination y2 (may be NULL): a layer output that is also a column block of the
 * concatenated embedding input (`torch.cat(conv_out, dim=1)`, models/deltanet_classification.py:42) */
int dc_bn_act2(const float* h, int64_t R, int32_t C, int64_t ldh, const float* scale, const float* shift, float slope,
               const float* residual, int64_t ldr, float* y, int64_t ldy, float* y2, int64_t ldy2, void* stream);
/* backward of dc_bn_act (through the batch statistics when training != 0); dgamma/dbeta may be NULL */
int dc_bn_act_backward(const float* dy, int64_t lddy, const float* h, int64_t ldh, int64_t R, int32_t C,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       const float* gamma, float slope, int32_t training, float* dh, int64_t lddh, float* dgamma,
                       float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* vector block.  in: combine != 0 -> [2n, 2co] = [P | Q], the Linear applied to v_cat (NOT to
 * I_J(v_cat)) with the weight halves stacked, y_u = P_u - Q_v, y_v = P_v + Q_u
 * (deltaconv/geometry/operators.py:9-21 folded into the epilogue); combine == 0 -> [2n, co] = y;
 * combine == 2 -> P and Q interleaved (column 2c = P_c, 2c+1 = Q_c), i.e. v_cat @ W.view(2co, K)^T for the
 * reference's [co, 2K] weight of the first VectorMLP layer, no re-stacking of W. */
int dc_vn_stats(const float* in, int64_t n, int32_t co, int64_t ld, int32_t combine, const float* gamma,
                const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                float* invstd, float* scale, float* shift, void* workspace, size_t workspace_bytes, void* stream);
/* out[2n,co] = y * relu(scale*|y| + shift) / max(|y|, 1e-8) */
int dc_vn_apply(const float* in, int64_t n, int32_t co, int64_t ld, int32_t combine, const float* scale,
                const float* shift, float* out, int64_t ldo, void* stream);
int dc_vn_backward(const float* dout, int64_t lddo, const float* in, int64_t ld, int32_t combine, int64_t n,
                   int32_t co, const float* scale, const float* shift, const float* mean, const float* invstd,
                   const float* gamma, int32_t training, float* din, int64_t lddi, float* dgamma, float* dbeta,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- layer-0 ("centralized") edge MLP + max aggregation without the [E,C] tensor ------------------
 * scatter(s_mlp_max(x[col] - x[row]), row, reduce='max') -- deltaconv/nn/deltaconv.py:50-52 with a
 * depth-1 s_mlp_max (nn/mlp.py:7-11).  y = Linear(x) [Nt,C]; edge pre-activation a_e = y_j - y_i.
 * Derivation in deltaconv_amd/csrc/edge_math.h.  amax/amin/s1pt/dzs: [Nt,C] fp32 contiguous,
 * argmax/argmin/arg: uint8 [Nt,C].  Workspace: dc_bn_workspace_bytes(Nt, C). */
int dc_edge_gather_stats(const float* y, int64_t ldy, const int32_t* nbr, int32_t n, int32_t k, int32_t C,
                         int32_t compute_stats, const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, float* amax, float* amin, uint8_t* argmax,
                         uint8_t* argmin, float* s1pt, float* mean, float* invstd, float* scale, float* shift,
                         void* workspace, size_t workspace_bytes, void* stream);
int dc_edge_max_apply(const float* amax, const float* amin, const uint8_t* argmax, const uint8_t* argmin, int32_t n,
                      int32_t C, const float* scale, const float* shift, float slope, float* out, int64_t ldo,
                      uint8_t* arg, void* stream);
/* dc_edge_max_apply with the layer's last s_mlp block in its epilogue (round 6): out = act2(scale2 h2 + shift2) + x_max
 * (`x = self.s_mlp(x) + x_max`, deltaconv/nn/deltaconv.py:59), second copy in out2 (may be NULL).  Same bits as
 * dc_edge_max_apply followed by dc_bn_act2 with x_max as residual. */
int dc_edge_max_apply_residual(const float* amax, const float* amin, const uint8_t* argmax, const uint8_t* argmin, int32_t n,
                               int32_t C, const float* scale, const float* shift, float slope, const float* h2,
                               int64_t ldh2, const float* scale2, const float* shift2, float slope2, float* out,
                               int64_t ldo, float* out2, int64_t ldo2, uint8_t* arg, void* stream);
int dc_edge_max_backward(const float* dout, int64_t lddo, const float* y, int64_t ldy, const int32_t* tptr,
                         const int32_t* tedge, int32_t n, int32_t k, int32_t C, const float* amax, const float* amin,
                         const uint8_t* argmax, const uint8_t* argmin, const float* s1pt, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float slope, int32_t training,
                         float* dzs, float* dy, int64_t lddy, float* dgamma, float* dbeta, void* workspace,
                         size_t workspace_bytes, void* stream);
/* dc_edge_max_backward with its CSC pass running from the transposed tile plan (rows (y_i, dz*_i) of the in-edges' sources
 * and the selected slot bytes in LDS); argsel = the `arg` output of dc_edge_max_apply; y, dzs, amax, amin, s1pt contiguous
 * [Nt, C], C % 64 == 0.  Same results bit for bit (same reference lines: nn/deltaconv.py:50-52 backward). */
int dc_edge_max_backward_tiled(const float* dout, int64_t lddo, const float* y, const int32_t* planT, int32_t n,
                               int32_t num_clouds, int32_t num_tiles, int32_t k, int32_t P, int32_t C, const float* amax,
                               const float* amin, const uint8_t* argsel, const float* s1pt, const float* scale,
                               const float* shift, const float* mean, const float* invstd, float slope, int32_t training,
                               float* dzs, float* dy, int64_t lddy, float* dgamma, float* dbeta, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- general form of the centralised edge MLP: materialised edge tensor --------------------------------------------------
 * x_edge = x[col] - x[row] and scatter(h, row, reduce=aggr) of deltaconv/nn/deltaconv.py:50-52 for the shapes the two fused
 * forms do not cover (depth >= 3, other widths, aggr != 'max'): edges are centre-major with k contiguous slots
 * (grad_div_mls.py:24-25), so the scatter is a reduction over k consecutive rows.  mode: 0 max, 1 min, 2 sum, 3 mean;
 * max / min ties keep the first slot. */
int dc_edge_diff(const float* x, int64_t ldx, const int32_t* nbr, int32_t n, int32_t k, int32_t C, float* out, void* stream);
int dc_edge_diff_backward(const float* dE, const int32_t* tptr, const int32_t* tedge, int32_t n, int32_t k, int32_t C,
                          float* dx, int64_t lddx, void* stream);
int dc_seg_reduce(const float* h, int32_t n, int32_t k, int32_t C, int32_t mode, float* out, uint8_t* arg, void* stream);
int dc_seg_reduce_backward(const float* dout, int64_t lddo, const uint8_t* arg, int32_t n, int32_t k, int32_t C, int32_t mode,
                           float* dh, void* stream);

/* ---- depth-2 centralised edge MLP + max aggregation (first layer of the part-segmentation net) ------------------------
 * out[i] = max_s act2(bn2(W2 act1(bn1(W1 (x_j - x_i))))), BatchNorm statistics over all E = n k edges:
 * deltaconv/nn/deltaconv.py:50-52 with s_mlp_max = MLP([ci, 64, 64]) (experiments/train_shapenet.py:77-89, mlp_depth = 2),
 * i.e. index_select, two addmm, two native_batch_norm, two leaky_relu and scatter_max on [E, 64] tensors + their autograd.
 * Here (deltaconv_amd/csrc/edge2.hip): z = x W1^T on n rows (own GEMM), statistics of z_j - z_i by dc_edge_gather_stats, then
 * ONE pass over the edges on v_mfma_f32_16x16x4_f32 (exact fp32) that keeps, per (point, channel), the selected
 * pre-BatchNorm-2 value `ysel`, its first slot `arg`, and the BatchNorm-2 sums; out = act2(scale2 ysel + shift2) is a
 * dc_bn_act call.  Backward: ONE recompute pass (three chained MFMA products: y2, du1 = (dy2 W2) act1', dW2 += dy2^T h1)
 * + a CSC closing pass; bit-reproducible (ordered reductions, in-edges in ascending edge id).  64 channels in both blocks.
 * stats_mode: 1 = batch statistics -> mean2 / invstd2 / scale2 / shift2 (+ running statistics), 2 = fp64 sums only
 * (sums = double [2][2*64 + 1] = 258 doubles: two identical records [sum y2 | sum y2^2 | rows = n k], the layout of
 * dc_bn_sums; finished by dc_bn_coeffs_from_sums after the all-reduce: synchronised BatchNorm), 0 = none.  coef1 / coef2 = [mean | invstd | scale | shift]
 * rows (4 x 64).  x [n, ci] / W1 [64, ci] (may be NULL): for ci <= 3 the edge pre-activation is evaluated as W1 (x_j - x_i)
 * per edge, the reference's own order of operations, instead of z_j - z_i.  Workspace: dc_edge2_workspace_bytes(n, k, backward). */
size_t dc_edge2_workspace_bytes(int32_t n, int32_t k, int32_t backward);
/* BatchNorm-1 of that block for ci <= 3 WITHOUT z: y1 = W1 (x_j - x_i) is linear in the edge difference, so its batch statistics
 * follow from the ci + ci (ci + 1) / 2 first and second moments of the differences (ordered fp64 sums); sd [n, ci] = sum_s
 * (x_j - x_i) per point feeds the closed forms of the backward pass.  With it dc_edge2_forward / dc_edge2_backward take z = NULL
 * and `s1pt` = sd. */
int dc_edge2_bn1_stats(const float* x, int64_t ldx, int32_t ci, const float* W1, const int32_t* nbr, int32_t n, int32_t k,
                       const float* gamma1, const float* beta1, float eps, float momentum, float* running_mean,
                       float* running_var, float* sd, float* mean1, float* invstd1, float* scale1, float* shift1,
                       void* workspace, size_t workspace_bytes, void* stream);
int dc_edge2_forward(const float* z, const float* x, int64_t ldx, int32_t ci, const float* W1, const int32_t* nbr,
                     int32_t n, int32_t k, const float* W2, const float* scale1,
                     const float* shift1, float slope1, int32_t stats_mode, const float* gamma2, const float* beta2,
                     float eps, float momentum, float* running_mean, float* running_var, float* ysel, uint8_t* arg,
                     float* mean2, float* invstd2, float* scale2, float* shift2, double* sums, void* workspace,
                     size_t workspace_bytes, void* stream);
int dc_edge2_backward(const float* dout, int64_t lddo, const float* z, const float* x, int64_t ldx, int32_t ci,
                      const float* W1, const int32_t* nbr, const int32_t* tptr,
                      const int32_t* tedge, int32_t n, int32_t k, const float* W2, const float* coef1, const float* coef2,
                      const float* gamma2, float slope1, float slope2, int32_t training1, int32_t training2,
                      const float* ysel, const uint8_t* arg, const float* s1pt, float* dz, int64_t lddz, float* dW2,
                      float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- weight-gradient GEMM on the fp32 matrix cores ---------------------------------------------------
 * C[M,N] (+)= A^T B,  A [R,M], B [R,N], R >> M,N  (dW = dY^T X of every per-point Linear layer: ATen mm in the
 * autograd of deltaconv/nn/mlp.py:9,15).  v_mfma_f32_32x32x2_f32 through the LDS-staged kernel (any M, N, leading
 * dimension < 2^21), reduction split over row slabs (one resident wave of workgroups), ordered reduction kernel:
 * deterministic.  Workspace: dc_gemm_tn_workspace_bytes. */
size_t dc_gemm_tn_workspace_bytes(int64_t R, int32_t M, int32_t N);
int dc_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t R, int32_t M, int32_t N, float* C,
               int64_t ldc, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* The same product with the slab reduction deferred: an autograd node that forms several weight gradients (a DeltaConv layer:
 * v_mlp, s_mlp and max-aggregation Linear of deltaconv/nn/deltaconv.py:35-47) writes the partial tiles [slabs][M][N] of each into
 * its own workspace (dc_gemm_tn_slabs; *slabs <- their number, a HOST int) and sums them all with ONE dc_gemm_tn_reduce_many at
 * the end: C_i[rows_i, cols_i] (ldc_i) (+)= ordered sum of the slabs of entry i, i < count (host arrays of device addresses /
 * sizes) -- the bits dc_gemm_tn writes, in one launch instead of one per weight. */
int dc_gemm_tn_slabs(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t R, int32_t M, int32_t N,
                     void* workspace, size_t workspace_bytes, int32_t* slabs, void* stream);
int dc_gemm_tn_reduce_many(const int64_t* partials, const int64_t* outs, const int64_t* ldc, const int32_t* rows,
                           const int32_t* cols, const int32_t* slabs, const int32_t* accumulate, int32_t count,
                           void* stream);

/* ---- split statistics for synchronised BatchNorm (data parallel) ----------------------------------------
 * Same arithmetic as dc_bn_stats / dc_vn_stats / dc_bn_act_backward / dc_vn_backward (nn/nonlin.py:24-35, 63-79),
 * cut at the reduction so that the host can all-reduce the fp64 column sums across ranks in between
 * (deltaconv_amd/dp.py).  sums: double [2][2*C + 1] = two identical records
 * (sum_0[C] | sum_1[C] | rows of this rank): the host all-reduces the FIRST record in place, the second stays this rank's own
 * (dgamma / dbeta are local sums): no fill, no clone.  dc_sync_means: m1 = sum_0 / rows, m2 = sum_1 / rows from the reduced
 * record, dbeta / dgamma (may be NULL) from the local one.
 *   forward : dc_bn_sums | dc_vn_sums -> all-reduce -> dc_bn_coeffs_from_sums(count = rows of ALL ranks; count <= 0:
 *             the count is on the device at sums[2*C], all-reduced together with the sums -- no host sync)
 *   backward: dc_*_backward_sums -> all-reduce -> m1 = sum_0 / count, m2 = sum_1 / count -> dc_*_backward_apply;
 *             dbeta = local sum_0, dgamma = local sum_1 (averaged with the other gradients afterwards). */
int dc_bn_sums(const float* h, int64_t R, int32_t C, int64_t ldh, double* sums, void* workspace,
               size_t workspace_bytes, void* stream);
int dc_vn_sums(const float* in, int64_t n, int32_t co, int64_t ld, int32_t combine, double* sums, void* workspace,
               size_t workspace_bytes, void* stream);
int dc_bn_coeffs_from_sums(const double* sums, int64_t count, int32_t C, const float* gamma, const float* beta,
                           float eps, float momentum, float* running_mean, float* running_var, float* mean,
                           float* invstd, float* scale, float* shift, void* stream);
int dc_bn_act_backward_sums(const float* dy, int64_t lddy, const float* h, int64_t ldh, int64_t R, int32_t C,
                            const float* scale, const float* shift, const float* mean, const float* invstd,
                            float slope, double* sums, void* workspace, size_t workspace_bytes, void* stream);
int dc_sync_means(const double* sums, int32_t C, float* m1, float* m2, float* dgamma, float* dbeta, void* stream);
int dc_bn_act_backward_apply(const float* dy, int64_t lddy, const float* h, int64_t ldh, int64_t R, int32_t C,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             const float* gamma, float slope, int32_t training, const float* m1, const float* m2,
                             float* dh, int64_t lddh, void* stream);
int dc_vn_backward_sums(const float* dout, int64_t lddo, const float* in, int64_t ld, int32_t combine, int64_t n,
                        int32_t co, const float* scale, const float* shift, const float* mean, const float* invstd,
                        double* sums, void* workspace, size_t workspace_bytes, void* stream);
int dc_vn_backward_apply(const float* dout, int64_t lddo, const float* in, int64_t ld, int32_t combine, int64_t n,
                         int32_t co, const float* scale, const float* shift, const float* mean, const float* invstd,
                         const float* gamma, int32_t training, const float* m1, const float* m2, float* din,
                         int64_t lddi, void* stream);

/* ---- pre-split weight planes (round 4) -----------------------------------------------------------------------------
 * The split products cut every fp32 operand into three bfloat16 planes.  For the WEIGHT operand of the Linear layers
 * (nn/mlp.py:9,15) that work is the same in every workgroup of every product of a step: dc_presplit_weights cuts all
 * registered weight matrices once (one launch; table = device records {src, fwd planes | NULL, bwd planes | NULL, rows, cols,
 * row stride}, chunk_start = prefix of ceil(rows * cols / 1024) per record), dc_gemm_next_b_planes hands the planes of the
 * next dc_linear_* product of the calling thread to the library (transposed = planes of W^T, for dc_linear_backward_input).
 * `weight` is the fp32 matrix the planes were cut from: the hint is dropped unless the next product's weight operand is that
 * pointer (a hint left behind by a call that failed before its product cannot reach another product of the same shape).
 * Same bits as the in-loop split; the hint is ignored where it does not apply. */
int dc_presplit_weights(const int64_t* table, const int32_t* chunk_start, int32_t n_entries, int32_t total_chunks, void* stream);
int dc_gemm_next_b_planes(const void* weight, const void* planes, int64_t plane_elems, int64_t ld, int32_t transposed);

/* ---- forward / input-gradient GEMMs of the per-point Linear layers on the fp32 matrix cores -------------
 * Replace ATen addmm / mm behind every `Linear(bias=False)` of deltaconv/nn/mlp.py:9,15 (forward product and the
 * input-gradient product of its autograd).  v_mfma_f32_32x32x2_f32 (exact fp32), LDS-staged, any M, N, K and
 * leading dimensions (an unguarded fast path when everything is tile-aligned).  tile: 0 = automatic,
 * 1..4 = 128x128, 128x64, 64x64, 64x128 (measurement).
 *   dc_linear_forward           Y[M,N]  = X[M,K] W[N,K]^T
 *   dc_linear_backward_input    dX[M,K] (+)= dY[M,N] W[N,K]
 * and the forward product fused with the statistics of the layer that follows it (nn/nonlin.py:24-35, 63-79):
 *   dc_linear_bn_stats_forward  Y = X W^T  + BatchNorm batch statistics of Y (== dc_bn_stats on Y)
 *   dc_linear_vn_stats_forward  interleaved: PQ[2n,2co] = V[2n,K] Wst[2co,K]^T (columns interleaved (P_c, Q_c)) +
 *                               statistics of |y| over the n points (== dc_vn_stats(combine = 2) on PQ); otherwise
 *                               Y[2n,co] = V W[co,K]^T + statistics of |(Y_2i, Y_2i+1)| (== dc_vn_stats(combine = 0))
 * The statistics come out of the GEMM epilogue (fp64 tile sums, ordered final stage): no pass over Y.
 * Workspace: dc_linear_stats_workspace_bytes(M, N, K, tile)  (vn: M = 2n, N = 2co). */
int dc_linear_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int32_t N, int32_t K,
                      float* Y, int64_t ldy, int32_t tile, void* stream);
int dc_linear_backward_input(const float* dY, int64_t lddy, const float* W, int64_t ldw, int64_t M, int32_t N,
                             int32_t K, float* dX, int64_t lddx, int32_t accumulate, int32_t tile, void* stream);
size_t dc_linear_stats_workspace_bytes(int64_t M, int32_t N, int32_t K, int32_t tile);
int dc_linear_bn_stats_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int32_t N,
                               int32_t K, float* Y, int64_t ldy, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                               float* scale, float* shift, int32_t tile, void* workspace, size_t workspace_bytes,
                               void* stream);
int dc_linear_vn_stats_forward(const float* V, int64_t ldv, const float* Wst, int64_t ldw, int64_t n, int32_t co,
                               int32_t K, float* PQ, int64_t ldpq, int32_t interleaved, const float* gamma,
                               const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                               float* scale, float* shift, int32_t tile, void* workspace, size_t workspace_bytes,
                               void* stream);

/* the two statistics products with the reduction cut open (synchronised BatchNorm): sums = double [2][2*C + 1], the fp64 column sums of this
 * rank's rows (two records, as above), to be all-reduced and finished by dc_bn_coeffs_from_sums -- the fused layer nodes keep their GEMM epilogues
 * when BatchNorm statistics span the ranks of a data-parallel group. */
int dc_linear_bn_sums_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, int64_t M, int32_t N, int32_t K,
                              float* Y, int64_t ldy, double* sums, int32_t tile, void* workspace, size_t workspace_bytes,
                              void* stream);
int dc_linear_vn_sums_forward(const float* V, int64_t ldv, const float* Wst, int64_t ldw, int64_t n, int32_t co, int32_t K,
                              float* PQ, int64_t ldpq, int32_t interleaved, double* sums, int32_t tile, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- BatchNorm/activation backward folded into the GEMMs that consume it --------------------------------
 * Backward of one MLP block y = leaky(batch_norm(x W^T)) (nn/mlp.py:7-11, nn/nonlin.py:24-35): instead of
 * dc_bn_act_backward (reduce + a pass that writes dh) followed by two products that read dh,
 *   dc_bn_act_backward_reduce     dgamma, dbeta, coefs[5*C]  (c_sc | c_sh | c_g | c_a | c_b)
 *   dc_linear_bn_backward_input   dX[M,K] (+)= dh W,    dc_linear_bn_backward_weight   dW[N,K] (+)= dh^T X
 * rebuild dh = c_g * dy * act'(c_sc h + c_sh) + c_a h + c_b inside their operand loaders: dh is never materialised
 * (one launch and 3 passes over [M,N] less per block).  Workspaces: dc_bn_workspace_bytes(R, C) and
 * dc_gemm_tn_workspace_bytes(R, N, K). */
int dc_bn_act_backward_reduce(const float* dy, int64_t lddy, const float* h, int64_t ldh, int64_t R, int32_t C,
                              const float* scale, const float* shift, const float* mean, const float* invstd,
                              const float* gamma, float slope, int32_t training, float* dgamma, float* dbeta,
                              float* coefs, void* workspace, size_t workspace_bytes, void* stream);
/* the coefficient rows of that prologue from ALL-REDUCED sums (dc_bn_act_backward_sums on every rank -> all-reduce):
 * coefs[5 C] from global_sums and the global row count (count <= 0: on the device at global_sums[2 C]); dgamma / dbeta
 * from local_sums (this rank's share; gradients are averaged afterwards). */
int dc_bn_backward_coefs_from_sums(const double* global_sums, int64_t count, const double* local_sums, int32_t C,
                                   const float* gamma, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, int32_t training, float* dgamma, float* dbeta, float* coefs,
                                   void* stream);
int dc_linear_bn_backward_input(const float* dy, int64_t lddy, const float* h, int64_t ldh, const float* coefs,
                                float slope, const float* W, int64_t ldw, int64_t M, int32_t N, int32_t K, float* dX,
                                int64_t lddx, int32_t accumulate, int32_t tile, void* stream);
int dc_linear_bn_backward_weight(const float* dy, int64_t lddy, const float* h, int64_t ldh, const float* coefs,
                                 float slope, const float* X, int64_t ldx, int64_t R, int32_t N, int32_t K, float* dW,
                                 int64_t lddw, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* partial tiles [slabs][N][K] only; the ordered sum comes later from dc_gemm_tn_reduce_many (see dc_gemm_tn_slabs) */
int dc_linear_bn_backward_weight_slabs(const float* dy, int64_t lddy, const float* h, int64_t ldh, const float* coefs,
                                       float slope, const float* X, int64_t ldx, int64_t R, int32_t N, int32_t K,
                                       void* workspace, size_t workspace_bytes, int32_t* slabs, void* stream);

/* ---- per-cloud term of the segmentation head's first Linear ----------------------------------------------------------------
 * deltaconv/models/deltanet_segmentation.py:59-66: x_max[batch] (+ the category vector) is concatenated in front of the point
 * features and fed to Linear(E + S, 256).  Here Linear([x_max[batch] | conv]) = Linear_a(x_max)[batch] + Linear_b(conv): the
 * per-cloud half runs on B rows; dc_cloud_bias_add joins it to the per-point half (y[i] = h[i] + g[i / mx], equal-size clouds of
 * mx points, y may be h), dc_cloud_colsum is its backward (out[b] = sum of the rows of cloud b: the index_add of `[batch]`),
 * an ordered two-stage fp64 reduction (bit-reproducible).  Workspace: dc_cloud_colsum_workspace_bytes. */
int dc_cloud_bias_add(const float* h, int64_t ldh, const float* g, int64_t ldg, int64_t n, int32_t C, int64_t mx, float* y,
                      int64_t ldy, void* stream);
size_t dc_cloud_colsum_workspace_bytes(int32_t num_clouds, int64_t mx, int32_t C);
int dc_cloud_colsum(const float* x, int64_t ldx, int32_t num_clouds, int64_t mx, int32_t C, float* out, int64_t ldo,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- embedding head fused with the per-cloud pooling -------------------------------------------------
 * MLP([sum c, E]) -> global_max_pool | global_mean_pool  (deltaconv/models/deltanet_classification.py:42-49),
 * -> global_max_pool (deltanet_segmentation.py:58-61).  h = Linear output [B*N, C] (equal-size clouds);
 * the [B*N, C] activation is never written.  pooled[B, ldp] = [max | mean]; argmax[B, C] = row in cloud. */
int dc_bn_act_pool(const float* h, int64_t ldh, int32_t num_clouds, int32_t N, int32_t C, const float* scale,
                   const float* shift, float slope, int32_t with_mean, float* pooled, int64_t ldp, int32_t* argmax,
                   void* stream);
int dc_bn_act_pool_backward(const float* dpooled, int64_t ldp, const int32_t* argmax, const float* h, int64_t ldh,
                            int32_t num_clouds, int32_t N, int32_t C, const float* scale, const float* shift,
                            const float* mean, const float* invstd, const float* gamma, float slope,
                            int32_t with_mean, int32_t training, float* dh, int64_t lddh, float* dgamma, float* dbeta,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- training loss ------------------------------------------------------------------------------
 * Replaces experiments/utils.py:7-24 (`calc_loss`: label-smoothed cross entropy, eps = 0.2, or
 * `F.cross_entropy(..., reduction='mean')` when smoothing = 0) together with its autograd backward:
 *   loss    = mean_r( -sum_c q_rc * log_softmax(logits_r)_c ),  q = 1-smoothing on the label, smoothing/(C-1) elsewhere
 *   dlogits = (softmax(logits_r) - q_r) / num_rows            (multiply by the incoming gradient of the loss)
 * labels int64 in [0, num_classes) (a label outside poisons the loss with NaN).  Deterministic (ordered fp64 sum). */
size_t dc_ce_loss_workspace_bytes(int64_t num_rows);
int dc_ce_loss(const float* logits, int64_t ld_logits, const int64_t* labels, int64_t num_rows, int32_t num_classes,
               float smoothing, float* loss, float* dlogits, int64_t ld_dlogits, void* workspace,
               size_t workspace_bytes, void* stream);

/* ---- MLP blocks on a handful of rows: the classification head, one row per cloud --------------------------------
 * (/root/reference/deltaconv/models/deltanet_classification.py:34-36,51: MLP([2048,512]) -> Dropout -> MLP([512,256]) ->
 * Dropout -> Linear(256, num_classes); blocks = nn/mlp.py:7-11).  One kernel per block and direction: product +
 * BatchNorm statistics over the rows + finalisation + running statistics + activation forward; BatchNorm / activation
 * backward + d gamma / d beta + weight gradient backward (the input gradient dX = dH W is dc_linear_backward_input).
 * mode 0 = Linear (+ bias) only, 1 = BatchNorm with batch statistics, 2 = with running statistics.
 * M <= dc_rowblock_max_rows() (64), K % 4 == 0, rows of X / W / dW 16-byte aligned.  coef[4,N] = mean, invstd, scale,
 * shift (written forward, read backward); H = the Linear output (saved for backward; mode 0: Y may alias H). */
int32_t dc_rowblock_max_rows(void);
int dc_rowblock_forward(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, int32_t M, int32_t N,
                        int32_t K, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                        float* running_var, int32_t mode, float slope, float* H, int64_t ldh, float* coef, float* Y,
                        int64_t ldy, void* stream);
int dc_rowblock_backward(const float* dY, int64_t lddy, const float* H, int64_t ldh, const float* coef, const float* gamma,
                         float slope, int32_t mode, const float* X, int64_t ldx, int32_t M, int32_t N, int32_t K, float* dW,
                         int64_t lddw, float* dbias, float* dgamma, float* dbeta, float* dH, int64_t lddh, void* stream);
/* The block FOLLOWED BY torch.nn.Dropout(p) (deltaconv/models/deltanet_classification.py:34-36: MLP -> Dropout(0.5) -> MLP ->
 * Dropout(0.5) -> Linear) in the block's own launches: forward Y = dropout(act(bn(X W^T))) and mask[M, N] (1 = kept);
 * backward takes the gradient behind the dropout.  Draws: Philox-4x32-10 keyed by `seed`, counter (element, salt, *step);
 * `step` points at the block's BatchNorm num_batches_tracked on the device (a new mask in every training step, also in
 * replays of a captured graph), `salt` tells the dropout layers of a model apart.  mode 1 | 2 as above. */
int dc_rowblock_forward_dropout(const float* X, int64_t ldx, const float* W, int64_t ldw, int32_t M, int32_t N, int32_t K,
                                const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                float* running_var, int32_t mode, float slope, float* H, int64_t ldh, float* coef, float* Y,
                                int64_t ldy, float p, int32_t seed, const int64_t* step, int32_t salt, uint8_t* mask,
                                void* stream);
int dc_rowblock_backward_dropout(const float* dY, int64_t lddy, const float* H, int64_t ldh, const float* coef,
                                 const float* gamma, float slope, int32_t mode, const float* X, int64_t ldx, int32_t M,
                                 int32_t N, int32_t K, float* dW, int64_t lddw, float* dgamma, float* dbeta, float* dH,
                                 int64_t lddh, const uint8_t* mask, float p, void* stream);

/* ---- optimizer step (step glue: experiments/train_modelnet.py:67, train_scanobjectnn.py:77, train_shapenet.py:95) -------- */
/* torch.optim.SGD(lr, momentum, weight_decay) (dampening 0, no Nesterov) over ALL parameters in one launch (96 tensors per
 * launch):  g' = g + weight_decay p;  buf = momentum buf + g';  p -= lr buf  (buf starts at zero).  params / grads / bufs /
 * numel: HOST arrays of `count` device addresses / element counts (fp32, contiguous); lr: DEVICE scalar (a scheduler writes
 * it between replays of a captured step). */
int dc_sgd_step(const int64_t* params, const int64_t* grads, const int64_t* bufs, const int64_t* numel, int32_t count,
                const float* lr, float momentum, float weight_decay, void* stream);
/* torch.optim.Adam(lr, betas, eps, weight_decay) (no amsgrad; experiments/train_shapeseg.py:82) over ALL parameters in one
 * launch (80 tensors per launch), torch's single-tensor op order:  t = step + 1;  g' = g + weight_decay p;
 * m += (1 - beta1)(g' - m);  v = beta2 v + (1 - beta2) g'^2;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps).
 * Host arrays as in dc_sgd_step; lr: DEVICE scalar; step: DEVICE fp32 scalar shared by all parameters, incremented by one (by
 * the last workgroup to finish); ticket: DEVICE int32 holding zero (left zero). */
int dc_adam_step(const int64_t* params, const int64_t* grads, const int64_t* exp_avgs, const int64_t* exp_avg_sqs,
                 const int64_t* numel, int32_t count, const float* lr, float* step, int32_t* ticket, double beta1, double beta2,
                 float eps, float weight_decay, void* stream);
/* Several small (strided) copies in one launch: the batch load into the inputs of a captured step (`data.to(device)`,
 * experiments/train_modelnet.py:99) and the first layer's operand blocks (the `torch.cat([x, ...])` of deltaconv/nn/deltaconv.py:57,65).
 * srcs / dsts: HOST arrays of `count` device addresses (4-byte aligned); ld_src / ld_dst / rows / cols in 4-byte words. */
int dc_copy_many(const int64_t* srcs, const int64_t* dsts, const int64_t* ld_src, const int64_t* ld_dst, const int32_t* rows,
                 const int32_t* cols, int32_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DELTACONV_HIP_H */
