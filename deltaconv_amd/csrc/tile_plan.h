// Tile plan of a kNN graph: the per-batch structure behind the LDS-deduplicated forward applies (ell_tile.h).
//
// Why.  A forward apply gathers k neighbour rows per point: k = 20 times the compulsory read volume, all of it
// through the texture addresser / L1 path, which -- not HBM -- bounds the staged kernels (profiles/r02n, r03b).
// Points that are close in space share most of their neighbours, so the points of a cloud are cut into TILES of P
// consecutive points of a Morton (Z-curve) order of their positions; the UNIQUE neighbour rows of a tile (~170 for
// P = 64 instead of 1280) are brought into LDS once by LDS-DMA and the k-loop reads LDS only (conflict-free: 16
// lanes x 16 bytes per 256-byte row).  Tensors keep their point order: a tile is a LIST of point ids.
//
// The plan depends only on positions + graph, is built once per batch beside the CSC (tileplan.hip) and is one
// int32 device blob whose section offsets follow from (num_points, num_clouds, k, P):
//   pts  [T][P]     int32   point ids of the tile in Morton order, -1 = padding (last tile of a cloud / unused tile)
//   nu   [T]        int32   number of unique rows U of the tile (0 = empty tile; may exceed P*k when points are not their own
//                           neighbours: uniq then lists the first P*k and the kernels fetch the other rows by neighbour id)
//   uniq [T][P*k]   int32   the unique row ids, ascending (the tile's own points are members); tail = last id
//   loc  [T][P*k]   uint16  tile-local index of neighbour (p, s) in uniq
//   self [T][P]     uint16  tile-local index of the point itself
// T (dc_tile_plan_num_tiles) is known on the host without a sync: exact for equal-sized clouds, else an upper bound; cloud b owns the tiles
// sum_{c < b} ceil(N_c / P) + [0, ceil(N_b / P)): the occupied ids are a dense prefix, the unused ids trail and stay empty.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DC_TP_HD __host__ __device__ __forceinline__
#else
#define DC_TP_HD inline
#endif

struct DcTilePlan {
    int T, P, k, PK;
    long o_pts, o_nu, o_uniq, o_loc, o_self, words;   // section offsets in int32 words (multiples of 4: 16-byte aligned)
};

DC_TP_HD long dc_tp_round4(long w) { return (w + 3) & ~3L; }

// number of tile ids T: exact when every cloud has max_cloud points (num_points == num_clouds * max_cloud: the usual
// fixed-size batches -- the launch then has no empty workgroups at all), else the bound ceil(num_points / P) + num_clouds
DC_TP_HD int dc_tile_plan_num_tiles(int num_points, int num_clouds, int max_cloud, int P) {
    if ((long)num_clouds * max_cloud == (long)num_points) return num_clouds * ((max_cloud + P - 1) / P);
    return (num_points + P - 1) / P + num_clouds;
}

DC_TP_HD DcTilePlan dc_tile_plan_layout(int num_tiles, int k, int P) {
    DcTilePlan p;
    p.P = P;
    p.k = k;
    p.PK = P * k;
    p.T = num_tiles;
    long w = 0;
    p.o_pts = w;   w = dc_tp_round4(w + (long)p.T * P);
    p.o_nu = w;    w = dc_tp_round4(w + p.T);
    p.o_uniq = w;  w = dc_tp_round4(w + (long)p.T * p.PK);
    p.o_loc = w;   w = dc_tp_round4(w + ((long)p.T * p.PK + 1) / 2);
    p.o_self = w;  w = dc_tp_round4(w + ((long)p.T * P + 1) / 2);
    p.words = w + 64;                                 // slack: clamped tail chunks of the LDS-DMA copies stay inside
    return p;
}


// ---- Transposed tile plan (round 4): the same tiles, seen from the TARGET side -----------------------------------------
// The transposed applies / max-aggregation backward walk the IN-edges of every point (variable degree, mean k).  The
// gather kernels (ell_stage.h: ell_T_kernel) give a wavefront four targets and make it walk to the largest of their four
// degrees with a serial chain of dependent global gathers per target: 0.13 - 0.25 of the HBM roofline (r03).  The
// transposed plan keeps the Morton tiles of the forward plan (pts), and per tile
//   * ORDERS THE TARGETS BY IN-DEGREE (descending; ties by position): lane group g of the workgroup owns target tg[g], so
//     the four targets of a wavefront have (nearly) equal degrees and finish together -- the wave-level loop count is the
//     sum of the degrees / 4 instead of the sum of the maxima of random quadruples (+30 %);
//   * lists the UNIQUE SOURCE rows of the tile (ascending id): they come into LDS by LDS-DMA exactly as the neighbour rows of
//     the forward kernels do;
//   * stores the in-edge lists of the tile CONTIGUOUSLY in (lane group, ascending edge id) order -- the per-target order of
//     the CSC, so every sum runs in the same order as in the gather kernels: bit-identical results, no fp atomics --: one
//     record per in-edge (tile-local source index | slot << 16) and the global edge id (= the permutation that brings an
//     operator's coefficients into tile order: dc_tile_plan_T_permute_coef).
// One int32 blob:
//   tg   [T][P]  int4    {target point id or -1, offset of its list inside the tile's range, in-degree, 0}
//   hdr  [T]     int4    {U = unique sources, start of the tile's range in rec / edge (multiple of 4), Et = entries, 0}
//   uniq [T][UQ] int32   unique source ids, ascending; tail = last id (UQ = 256 >= the LDS capacity of the kernels)
//   rec  [EP]    uint32  tile-local source index | slot << 16
//   edge [EP]    int32   edge id e = i * k + s (padding: a valid edge id of the cloud)
// The range of tile t starts at a multiple of 4 entries (16-byte aligned LDS-DMA sources): cloud b starts at
// round4(cloud_ptr[b] * k) + 4 * (first_tile(b) + b), tiles follow each other with their lengths rounded up to 4;
// EP = round4(E) + 4 * (T + B) + 8 covers every batch.
struct DcTilePlanT {
    int T, P, k, UQ;
    long EP;
    long o_tg, o_hdr, o_uniq, o_rec, o_edge, words;
};
enum { DC_TPT_UQ = 256 };

DC_TP_HD long dc_tile_plan_T_edges(long num_points, int k, int num_tiles, int num_clouds) {
    return dc_tp_round4(num_points * k) + 4L * (num_tiles + num_clouds) + 8;
}
DC_TP_HD DcTilePlanT dc_tile_plan_T_layout(long num_points, int num_clouds, int num_tiles, int k, int P) {
    DcTilePlanT p;
    p.T = num_tiles;
    p.P = P;
    p.k = k;
    p.UQ = DC_TPT_UQ;
    p.EP = dc_tile_plan_T_edges(num_points, k, num_tiles, num_clouds);
    long w = 0;
    p.o_tg = w;    w += 4L * p.T * P;
    p.o_hdr = w;   w += 4L * p.T;
    p.o_uniq = w;  w += (long)p.T * p.UQ;
    p.o_rec = w;   w += p.EP;
    p.o_edge = w;  w += p.EP;
    p.words = w + 64;
    return p;
}
