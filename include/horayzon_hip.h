/*
 * horayzon_hip.h -- C ABI of libhorayzon_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the ray-casting core of HORAYZON.  Each entry point
 * names the reference interface it replaces (paths relative to the reference
 * tree).  The reference exposes C++-linkage functions to Cython
 * (horizon.pyx:13-27, shadow.pyx:8-15); a maintainer binds these C symbols
 * instead (see INTEGRATION.md for the Cython / ctypes stubs).
 *
 * Conventions
 *   - plain pointers and sizes only; every array is caller-owned and is never
 *     retained after the call returns (the reference keeps raw pointers in
 *     CppTerrain, shadow_comp.cpp:332-346; this library copies to HBM).
 *   - every data pointer may be HOST memory (NumPy) or DEVICE memory (HBM) of
 *     the selected GPU; the library detects which (hipPointerGetAttributes).
 *   - every function returns 0 on success and a non-zero status otherwise;
 *     hz_last_error() returns the message (thread local).  The reference
 *     returns void and only prints (horizon_comp.cpp:74-76).
 *   - units and meaning of the scalar arguments are exactly the reference's
 *     (degrees, kilometres, ... as in horizon_comp.h:8-20, shadow_comp.h:22-38).
 */
#ifndef HORAYZON_HIP_H
#define HORAYZON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HZ_OK 0
#define HZ_ERR_ARG 1      /* invalid argument                                  */
#define HZ_ERR_HIP 2      /* HIP runtime failure (message has the HIP error)   */
#define HZ_ERR_NODEV 3    /* no usable gfx950 device                           */
#define HZ_ERR_DEPTH 4    /* BVH deeper than the traversal stack               */

/* Optional controls; pass NULL for the reference's behaviour on device 0.     */
typedef struct hz_opts {
    int32_t device;        /* HIP device ordinal                               */
    int32_t verbose;       /* 1: print the reference's stdout report           */
    int32_t row_begin;     /* inner-domain row slab [row_begin, row_end) to compute.  {0, 0} (a zeroed struct) = the   */
    int32_t row_end;       /*   whole inner domain; row_end == -1 = dim_in_0 (explicit sentinel); every other pair    */
                           /*   must satisfy 0 <= row_begin <= row_end <= dim_in_0, else HZ_ERR_ARG (negative or        */
                           /*   out-of-range slabs are rejected, not clamped).  row_begin == row_end (> 0) is an empty  */
                           /*   slab: the call succeeds and computes nothing (a rank without rows)                      */
    int32_t top_nodes;     /* > 0: stage that many top-of-tree BVH nodes in LDS (guess_constant;   */
                           /*   measured 2 % slower than L1 reads, so <= 0 means none)            */
    int32_t regroup;       /* wave regroup threshold in lanes (<= 0: default)  */
    int32_t count_work;    /* 1: also count BVH nodes / triangle tests (slow)  */
    int32_t no_hit_cache;  /* 0 (default): rays expected to be blocked first walk the subtree  */
                           /*   that blocked the cell's previous ray; 1: always start at the root */
    float  *svf;           /* optional fused sky view factor out, f32[y][x]    */
    const float *vec_tilt; /* tilted normals f32[y][x][3] for svf              */
    int32_t skip_hori;     /* 1: hori_buffer may be NULL, only svf is written  */
    int32_t chunk_rows;    /* rows per launch when hori is host memory or skipped (chunks are      */
                           /*   double buffered and copied out while the next one is traced);      */
                           /*   <= 0: as many rows as fit 4 GiB                                     */
    int32_t level_stack;   /* 1: use the one-entry-per-tree-level traversal stack from the start (cannot overflow, +10 % */
                           /*   VALU); 0: the fast discipline first, the other one only for launches whose stack overflowed; */
                           /*   < 0 (tests): the fast discipline with -level_stack entries                                 */
    int32_t hori_is_slab;  /* 0: hori_buffer (and svf) address inner-domain row 0 (reference layout, [dim_in_0][..]); */
                           /*   1: they address row_begin, i.e. hold only the slab [row_end - row_begin][dim_in_1].. */
                           /*   -- the form for resident HBM slab buffers (the caller never forms an address        */
                           /*   outside its allocation).  For the inputs see inputs_are_slab                         */
    int32_t no_near_skip;  /* 1: do not compute / use the near-field certificates (hz_near.hip): every ray starts at   */
                           /*   parameter 0.  Results are the same either way; the default (0) is faster and is only   */
                           /*   active for a DEM mesh that IS a height field over the world (x, y) plane (checked by   */
                           /*   the scene build, hz_stats.height_field); -1 (tests): certificates even if it is not    */
    int32_t verify_near;   /* N >= 1: check the near-field certificates while computing; disagreeing hit decisions are   */
                           /*   counted in hz_stats.near_violations (must stay 0), the rays checked in near_verified.    */
                           /*   With count_work: the counting instantiation traces one of every N shortened rays (N       */
                           /*   rounded up to a power of two; 1: every one) a second time over its full length.  Without: */
                           /*   the production launch is untouched and a second, counting launch re-traces EVERY          */
                           /*   shortened ray of one of every N 8 x 8 blocks of cells, next to it on a stream of its own --  */
                           /*   N = 256 monitors real inputs for 2 - 3 % of the run time (profiles/r04/)                  */
    int32_t inputs_are_slab; /* 0: vec_norm, vec_north, mask (and opts.vec_tilt) address inner-domain row 0 (reference   */
                           /*   layout; only the slab's rows are read or uploaded); 1: they address row_begin, i.e. the */
                           /*   caller holds only its slab [row_end - row_begin][dim_in_1] of each -- the form for a    */
                           /*   rank of a row-sharded job                                                                */
    int32_t no_host_pin;   /* 0 (default): a HOST hori_buffer is page-locked chunk by chunk on a helper thread while the   */
                           /*   first chunks are traced (hipHostRegister, released before the call returns), so that the  */
                           /*   device-to-host copies are DMA instead of staged pageable copies; 1: leave it pageable     */
    int32_t left_min;      /* leftover cells (hz_horizon.hip): an 8 x 8 block of cells ends when at most this many of its  */
                           /*   cells are unfinished; they are finished 64 at a time by follow-up launches, which hand     */
                           /*   over again.  Byte l = the threshold of level l (0 = production launch; a zero byte: that   */
                           /*   level runs to its end); 0: the default; < 0: off.  Results do not depend on it             */
    int32_t persist_grid;  /* 0 (default): persistent waves -- a launch has as many workgroups as are resident at once and */
                           /*   every wave pulls 8 x 8 blocks from its XCD's queue; < 0: one 16 x 16 tile per workgroup;   */
                           /*   n > 0 (tests): persistent with n workgroups, so that small grids run the block loop too    */
    int32_t left_cap_test; /* tests: > 0 caps every region of the leftover records at this many (rounded up to 64), so       */
                           /*   that the out-of-room path runs on small grids                                               */
    int32_t left_tune;     /* tuning of the follow-up launches (0: defaults): byte 0 = their compaction threshold in lanes   */
                           /*   (opts.regroup of those launches), byte 1 = 1 + log2 of the width of the azimuths-left        */
                           /*   classes their cells are sorted into (1: one class per azimuth)                               */
} hz_opts;

/* Run-time self report (the quantities the reference prints,                  */
/* horizon_comp.cpp:225-227, 805-810, 816-818).                                */
typedef struct hz_stats {
    uint64_t num_rays;     /* occlusion queries, counted as the reference does */
    uint64_t num_cells;    /* cells with mask == 1 in the computed slab        */
    uint64_t guard_events; /* searches stopped where the reference never ends  */
    uint64_t nodes_visited;/* only with count_work                             */
    uint64_t tris_tested;  /* only with count_work                             */
    double t_bvh_s;        /* "BVH build time"                                 */
    double t_h2d_s;        /* host -> HBM copies                               */
    double t_kernel_s;     /* "Ray tracing time" (traversal kernels only)      */
    double t_d2h_s;        /* HBM -> host copies                               */
    double t_total_s;      /* "Total run time"                                 */
    int32_t bvh_height;
    int32_t elev_num;
    uint64_t scene_bytes;  /* HBM bytes of vertices + LBVH                     */
    uint64_t wave_node_iters; /* count_work: wave-level executions of the node  */
    uint64_t wave_leaf_iters; /*   step / leaf step / ray refill section (SIMT  */
    uint64_t wave_refills;    /*   efficiency = lane count / (64 x wave count)) */
    double t_svf_s;        /* sky-view-factor kernel (when opts.svf is set)    */
    uint64_t stack_fallbacks; /* launches after which blocks (or the whole launch) were repeated with the one-entry-per- */
                              /*   level stack because a ray ran out of entries of the fast one                          */
    uint64_t rays_shortened;  /* count_work: rays that started beyond the cell's neighbourhood (near-field certificate) */
    uint64_t near_violations; /* count_work + verify_near: shortened rays whose full-length re-trace disagreed (0)      */
    double t_near_s;       /* certificate pre-pass (hz_near.hip)                */
    uint64_t stack_redo_blocks; /* 8 x 8 blocks repeated one by one after such an overflow (few: deep trees overflow in    */
                               /*   a few places; many overflows repeat the launch and switch the scene for good)        */
    uint64_t guard_cells;  /* cells with at least one guard event (the reference's search would not terminate there,     */
                           /*   horizon_comp.cpp:474-488)                                                                 */
    int32_t height_field;  /* 1: the scene's DEM mesh is a height field over the world (x, y) plane                      */
    int32_t near_used;     /* 1: the near-field certificates were active in this call                                    */
    uint64_t near_verified;/* opts.verify_near: shortened rays that were traced a second time over their full length     */
    double t_left_s;       /* leftover launches: part of t_kernel_s spent finishing the cells that blocks of the production   */
    uint64_t left_cells;   /*   launch handed over when <= opts.left_min of their 64 cells were unfinished (hz_horizon.hip)   */
    uint64_t left_again;   /* hand-overs by the leftover launches themselves (levels >= 1)                                    */
    uint64_t scratch_bytes;/* HBM scratch this call used besides the scene and the caller's buffers: near-field certificates, */
                           /*   leftover records, the horizon chunk buffers of a host / skipped hori_buffer                   */
    uint64_t left_redo_groups; /* groups of 64 leftover cells computed again with the one-entry-per-level stack after a ray of */
                           /*   theirs ran out of entries of the fast one (counted in stack_redo_blocks as well)               */
} hz_stats;

const char *hz_last_error(void);
/* sizeof(hz_opts), sizeof(hz_stats) as compiled: lets a binding verify its mirror */
int hz_abi_struct_sizes(int *opts_bytes, int *stats_bytes);
/* ABI revision.  6 (round 6): opts.left_min, persist_grid, left_cap_test, left_tune and hz_stats.left_again, scratch_bytes, left_redo_groups appended.  5 (round 5): hz_stats.t_left_s, left_cells appended.  4 (round 4): hz_stats.near_verified appended; opts.verify_near is a sampling period (1 = every ray, as   */
/* before).  3 (round 3): {row_begin > 0, row_end = 0} is rejected (was "to the end": use row_end = -1); opts.regroup <= 0 */
/* means the default threshold (was: 0 = ray compaction off; 64 | bias << 8 still disables the early exit in effect)     */
int hz_abi_version(void);
int hz_device_count(int *count);
/* name[0..cap) <- device name, *cu <- compute units, *hbm_bytes <- total HBM  */
int hz_device_info(int device, char *name, int cap, int *cu, uint64_t *hbm_bytes);

/* ------------------------------------------------------------------------- */
/* Scene = vertices + flat LBVH, resident in HBM as ONE contiguous blob        */
/* (so that a multi-GPU job can broadcast it with a single collective).        */
/* Replaces initializeDevice/initializeScene, horizon_comp.cpp:74-231,         */
/* shadow_comp.cpp:171-298 (Embree rtcNewScene ... rtcCommitScene).            */
/* ------------------------------------------------------------------------- */
typedef struct hz_scene hz_scene;

int hz_scene_create(const float *vert_grid, int dem_dim_0, int dem_dim_1,
                    const char *geom_type,
                    const float *vert_simp, int num_vert_simp,
                    const int32_t *tri_ind_simp, int num_tri_simp,
                    int device, hz_scene **scene, hz_stats *stats);
/* the blob: position independent, valid on any gfx950 device                  */
int hz_scene_blob(const hz_scene *scene, void **device_ptr, size_t *nbytes);
/* the vertex array inside the blob (f32[dem_dim_0 * dem_dim_1][3], the caller's vert_grid without padding): lets a   */
/* rank that received the blob derive per-cell inputs of its slab on the device without the host arrays               */
int hz_scene_vertices(const hz_scene *scene, const float **device_ptr, int *dem_dim_0, int *dem_dim_1,
                      int *height_field);
/* wrap a blob that already sits in HBM of `device` (e.g. received by an RCCL  */
/* broadcast into caller-owned memory); the scene does not own the memory      */
int hz_scene_adopt(void *device_ptr, size_t nbytes, int device, hz_scene **scene);
int hz_scene_destroy(hz_scene *scene);

/* ------------------------------------------------------------------------- */
/* Horizon                                                                     */
/* ------------------------------------------------------------------------- */

/* One-shot call; argument list mirrors horizon_gridded_comp                   */
/* (horizon_comp.h:8-20, horizon_comp.cpp:629-822).  Builds the scene,         */
/* computes, releases (the reference also rebuilds per call, :813-814).        */
int hz_horizon_gridded(const float *vert_grid, int dem_dim_0, int dem_dim_1,
                       const float *vec_norm, const float *vec_north,
                       int offset_0, int offset_1,
                       float *hori_buffer, int dim_in_0, int dim_in_1,
                       int azim_num, float dist_search, float hori_acc,
                       const char *ray_algorithm, const char *geom_type,
                       const float *vert_simp, int num_vert_simp,
                       const int32_t *tri_ind_simp, int num_tri_simp,
                       float elev_ang_low_lim, const uint8_t *mask,
                       float hori_fill, float ray_org_elev,
                       const hz_opts *opts, hz_stats *stats);

/* Same computation on an existing scene (persistent BVH; additive API).       */
int hz_horizon_gridded_scene(const hz_scene *scene,
                             const float *vec_norm, const float *vec_north,
                             int offset_0, int offset_1,
                             float *hori_buffer, int dim_in_0, int dim_in_1,
                             int azim_num, float dist_search, float hori_acc,
                             const char *ray_algorithm,
                             float elev_ang_low_lim, const uint8_t *mask,
                             float hori_fill, float ray_org_elev,
                             const hz_opts *opts, hz_stats *stats);

/* Horizon (and optionally distance to the horizon) for arbitrary locations; argument list     */
/* mirrors horizon_locations_comp (horizon_comp.h:22-34, horizon_comp.cpp:828-1094).           */
/* coords f32[num_loc][3], vec_norm / vec_north f32[num_loc][3], ray_org_elev f32[num_loc],    */
/* hori_buffer f32[num_loc][azim_num]; hori_dist_buffer f32[num_loc][azim_num] is only written */
/* when hori_dist_out != 0 (may be NULL otherwise).  Locations whose normal never meets the    */
/* mesh keep the caller's values (the reference pre-fills NaN, horizon.pyx:331-341).           */
int hz_horizon_locations(const float *vert_grid, int dem_dim_0, int dem_dim_1,
                         const float *coords, const float *vec_norm, const float *vec_north,
                         float *hori_buffer, float *hori_dist_buffer, int num_loc,
                         int azim_num, float dist_search, float hori_acc,
                         const char *ray_algorithm, const char *geom_type,
                         float elev_ang_low_lim, const float *ray_org_elev, int hori_dist_out,
                         const hz_opts *opts, hz_stats *stats);
int hz_horizon_locations_scene(const hz_scene *scene,
                               const float *coords, const float *vec_norm, const float *vec_north,
                               float *hori_buffer, float *hori_dist_buffer, int num_loc,
                               int azim_num, float dist_search, float hori_acc,
                               const char *ray_algorithm, float elev_ang_low_lim,
                               const float *ray_org_elev, int hori_dist_out,
                               const hz_opts *opts, hz_stats *stats);

/* Trig tables exactly as horizon_comp.cpp:711-731 builds them (host side;     */
/* exported so tests can compare them bit for bit with the oracle).            */
/* Returns elev_num through *elev_num; arrays may be NULL to query the size.   */
int hz_horizon_tables(int azim_num, float hori_acc, float elev_ang_low_lim,
                      float *azim_sin, float *azim_cos, int elev_cap,
                      float *elev_ang, float *elev_sin, float *elev_cos,
                      int *elev_num);

/* Sky view factor from a horizon array: _sky_view_factor_cy,                  */
/* topo_param.pyx:412-460.                                                     */
int hz_sky_view_factor(const float *azim, const float *hori, const float *vec_tilt,
                       int len_0, int len_1, int len_2, float *svf, int device);
/* _visible_sky_fraction_cy (topo_param.pyx:499-543) and _topographic_openness_cy (:577-603):   */
/* the other two reductions of the same horizon array.                                          */
int hz_visible_sky_fraction(const float *azim, const float *hori, const float *vec_tilt,
                            int len_0, int len_1, int len_2, float *vsf, int device);
int hz_topographic_openness(const float *azim, const float *hori, int len_0, int len_1, int len_2,
                            float *top, int device);

/* Test hooks for the hand-written build primitives (stable radix sort of uint32 pairs by key,   */
/* exclusive prefix sum); host arrays, in place / in -> out.  Not needed by a binding.            */
int hz_debug_sort_pairs(uint32_t *keys, uint32_t *vals, size_t n, int device);
int hz_debug_exclusive_scan(const uint32_t *in, uint32_t *out, size_t n, int device);
/* Machine calibration for the roofline (bench.py, untimed section).  hz_debug_valu_peak: chains of      */
/* independent v_fma_f32 (packed = 1: v_pk_fma_f32) at 8 waves per SIMD, `waves_per_simd` rounds of      */
/* 3.5 ms (the name is historical: it only sets the length of the run; long pure-FMA runs are power-     */
/* limited) -> wave-level VALU instructions per second and SIMD, the engine clock [GHz], the SIMD count. */
/* hz_debug_copy_peak: float4 device-to-device copy of `bytes` -> read + write GB/s.                     */
int hz_debug_valu_peak(int device, int packed, int waves_per_simd, double *winst_per_s_per_simd,
                       double *clock_ghz, int *simds);
int hz_debug_copy_peak(int device, size_t bytes, double *gbs);
/* issue rate of one VALU instruction kind (selector list: hz_bench.hip) in cycles per wave64 instruction per SIMD */
int hz_debug_inst_rate(int device, int op, double *cycles_per_inst);
/* test knobs (process wide; results never depend on them): "shadow_fast_cap" = entries of the shadow kernel's fast stack (< 0: */
/* the default; small values make the in-kernel retry with the level stack the common case), "topo_wide" = 1: the reductions    */
/* over the azimuth axis use the one-lane-per-cell fallback kernel                                                               */
int hz_debug_set(const char *key, int value);

/* ------------------------------------------------------------------------- */
/* Steps next to the path: slope and input preparation */
/* ------------------------------------------------------------------------- */

/* _slope_plane_meth_cy / _slope_vector_meth_cy, topo_param.pyx:84-225, :284-372.               */
/* x, y, z f32[len_0][len_1]; rot_mat f32[len_0][len_1][3][3] or NULL; vec_tilt f32[..][3]      */
/* (border cells NaN).                                                                          */
int hz_slope_plane_meth(const float *x, const float *y, const float *z, int len_0, int len_1,
                        const float *rot_mat, int output_rot, float *vec_tilt, int device);
int hz_slope_vector_meth(const float *x, const float *y, const float *z, int len_0, int len_1,
                         const float *rot_mat, int output_rot, float *vec_tilt, int device);
/* ellps: 0 "sphere", 1 "GRS80", 2 "WGS84".                                                     */
/* _lonlat2ecef_1d, transform.pyx:60-103: lon, lat f64[n] [degree], h f32[n] -> f64[n] x 3      */
int hz_lonlat2ecef(const double *lon, const double *lat, const float *h, size_t n, int ellps,
                   double *x_ecef, double *y_ecef, double *z_ecef, int device);
/* -- NOT a row of the scope table (SURVEY.md section 8(f)4 names transform.pyx:60-103, 152-189, 231-261 only; section 2   */
/*    lists coordinate transforms as out of scope): the two Swiss-grid routines below were added in round 2, are kept   */
/*    because the reference's swissALTI3D examples feed the path through them, and are not part of any parity claim.    */
/* _wgs2swiss_1d, transform.pyx:306-345: lon, lat f64[n] [degree], h_wgs f32[n] -> LV95 e, n f64[n] [m], */
/* h_ch f32[n]; _swiss2wgs_1d, transform.pyx:390-432: the inverse (swisstopo's approximate formulas)  */
int hz_wgs2swiss(const double *lon, const double *lat, const float *h_wgs, size_t n,
                 double *e, double *n_out, float *h_ch, int device);
int hz_swiss2wgs(const double *e, const double *n_in, const float *h_ch, size_t n,
                 double *lon, double *lat, float *h_wgs, int device);
/* _ecef2enu_1d with TransformerEcef2enu(lon_or, lat_or, ellps), transform.pyx:152-189, 438-487 */
int hz_ecef2enu(const double *x_ecef, const double *y_ecef, const double *z_ecef, size_t n,
                double lon_or, double lat_or, int ellps, float *x_enu, float *y_enu, float *z_enu,
                int device);
/* _ecef2enu_vector_1d, transform.pyx:231-261: f32[n][3] -> f32[n][3]                           */
int hz_ecef2enu_vector(const float *vec_ecef, size_t n, double lon_or, double lat_or, int ellps,
                       float *vec_enu, int device);
/* _surf_norm_1d, direction.pyx:48-70                                                           */
int hz_surf_norm(const double *lon, const double *lat, size_t n, float *vec_norm_ecef, int device);
/* _north_dir_1d, direction.pyx:125-178                                                         */
int hz_north_dir(const double *x_ecef, const double *y_ecef, const double *z_ecef,
                 const float *vec_norm_ecef, size_t n, int ellps, float *vec_north_ecef, int device);

/* rearrange_pad_buffer / pad_buffer, auxiliary.py:49-95, :99-133: coordinate planes x, y, z f32[num_vertices]  */
/* (row-major DEM) -> the interleaved xyz vertex buffer `vert_grid` every entry point above takes, with the    */
/* reference's zero padding.  hz_vert_grid_len gives the padded length in floats.  With device pointers the     */
/* whole chain lon/lat/elevation -> ENU -> vert_grid -> scene -> horizon / SVF / shadow never leaves HBM.       */
size_t hz_vert_grid_len(size_t num_vertices);
int hz_pack_vertices(const float *x, const float *y, const float *z, size_t num_vertices, float *vert_grid,
                     size_t vert_grid_len, int device);

/* ------------------------------------------------------------------------- */
/* Shadow: handle API mirroring class CppTerrain (shadow_comp.h:4-39)          */
/* ------------------------------------------------------------------------- */
typedef struct hz_terrain hz_terrain;

/* CppTerrain::CppTerrain, shadow_comp.cpp:304-308 */
int hz_terrain_create(int device, hz_terrain **terrain);
/* CppTerrain::initialise, shadow_comp.cpp:318-380 (argument order as there)   */
int hz_terrain_initialise(hz_terrain *terrain, const float *vert_grid,
                          int dem_dim_0, int dem_dim_1, int offset_0, int offset_1,
                          const float *vec_tilt, const float *vec_norm,
                          int dim_in_0, int dim_in_1,
                          const float *surf_enl_fac, const float *elevation,
                          const uint8_t *mask, const char *geom_type,
                          float sw_dir_cor_fill, float ang_max, int refrac_cor,
                          hz_stats *stats);
/* additive: initialise on an existing scene (scene must outlive the terrain)  */
int hz_terrain_initialise_scene(hz_terrain *terrain, const hz_scene *scene,
                                int offset_0, int offset_1,
                                const float *vec_tilt, const float *vec_norm,
                                int dim_in_0, int dim_in_1,
                                const float *surf_enl_fac, const float *elevation,
                                const uint8_t *mask,
                                float sw_dir_cor_fill, float ang_max, int refrac_cor);
/* CppTerrain::shadow, shadow_comp.cpp:386-491: u8[y][x], 0/1/2/3              */
int hz_terrain_shadow(hz_terrain *terrain, const float *sun_position,
                      uint8_t *shadow_buffer, hz_stats *stats);
/* CppTerrain::sw_dir_cor, shadow_comp.cpp:495-605: f32[y][x]                  */
int hz_terrain_sw_dir_cor(hz_terrain *terrain, const float *sun_position,
                          float *sw_dir_cor_buffer, hz_stats *stats);
/* additive batch forms: num_sun positions f32[num_sun][3] ->                  */
/* buffers [num_sun][y][x]; all positions in ONE launch (grid.y = position)    */
int hz_terrain_shadow_batch(hz_terrain *terrain, const float *sun_positions,
                            int num_sun, uint8_t *shadow_buffers, hz_stats *stats);
int hz_terrain_sw_dir_cor_batch(hz_terrain *terrain, const float *sun_positions,
                                int num_sun, float *sw_dir_cor_buffers, hz_stats *stats);
/* additive: 1 = later shadow / sw_dir_cor calls run the counting instantiation and also fill   */
/* hz_stats.nodes_visited / tris_tested / wave_*_iters (slower; for the roofline's B_trav)      */
int hz_terrain_count_work(hz_terrain *terrain, int on);
/* CppTerrain::~CppTerrain, shadow_comp.cpp:310-316 */
int hz_terrain_destroy(hz_terrain *terrain);

#ifdef __cplusplus
}
#endif
#endif /* HORAYZON_HIP_H */
