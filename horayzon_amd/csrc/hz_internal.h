// hz_internal.h -- host-side internals of libhorayzon_hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "../../include/horayzon_hip.h"
#include "hz_common.h"
#include <atomic>
#include <mutex>

#define HZ_MAX_TOP_NODES 2047   // upper bound of BFS-ordered top nodes (LDS staging)
#define HZ_MAX_STACK 40         // tree levels (= LDS stack entries per lane) the traversal kernels accept
// hit cache: levels between a leaf and its cached ancestor.  Measured 1..9 on the 3601^2 tile in round 1 (5-6 best) and again in
// round 5 after a cache walk that finds nothing stopped costing a trip through the caller (hz_horizon.hip: the root waits below the
// cached subtree): 3 and 4 tie, 0.7 % ahead of 5, 6 loses 1.6 %; discrete_sampling +7 % with 4 (profiles/r05/sweep_regroup_anc_after_cache_fix.log)
#ifndef HZ_ANC_LEVELS
#define HZ_ANC_LEVELS 4
#endif

namespace hz {

int set_error(int code, const char *fmt, ...);

#define HZ_SHADOW_FAST_CAP_DEFAULT 19   // entries of k_shadow_refill's fast stack (hz_shadow.hip)
// test knobs (hz_debug_set, include/horayzon_hip.h): process wide, read at every launch; results never depend on them
extern std::atomic<int> g_shadow_fast_cap;   // entries of k_shadow_refill's fast stack (default HZ_SHADOW_FAST_CAP_DEFAULT; 0: level stack only)
extern std::atomic<int> g_topo_wide;         // 1: the reductions over the azimuth axis use the fallback kernel k_topo_wide

#define HZ_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return ::hz::set_error(HZ_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                                   hipGetErrorString(e_), __FILE__, __LINE__);            \
    } while (0)

struct Timer {
    std::chrono::high_resolution_clock::time_point t0;
    void start() { t0 = std::chrono::high_resolution_clock::now(); }
    double stop() const {
        return std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    }
};

struct Scene {
    int device = 0;
    hipStream_t stream = nullptr;
    void *blob = nullptr;
    size_t blob_bytes = 0;
    bool owns_blob = false;
    // 1 once a launch with the fast stack discipline overflowed: later launches use the one-entry-per-level kernel
    mutable std::atomic<int> level_stack{0};
    // Calls that launch on the scene's stream (horizon, locations, terrain set-up and shadow) hold `run_mu` from their
    // first enqueue to their last synchronisation: concurrent calls on one scene are allowed through the C ABI
    // (`const hz_scene *`, ctypes releases the GIL) and simply run one after the other -- they share the stream, so
    // the GPU would serialise them anyway.  This is what makes the scene-owned scratch below safe.
    mutable std::mutex run_mu;
    // scratch of the near-field certificates, grown on demand and kept with the scene (hz_api.hip; guarded by run_mu)
    mutable void *near_buf = nullptr;
    mutable size_t near_bytes = 0;
    // records of the cells a production launch of k_horizon left unfinished (hz_horizon.hip: leftover cells; guarded by run_mu)
    mutable void *left_buf = nullptr;
    mutable size_t left_bytes = 0;
    BlobHeader hdr;
    const float *verts() const { return (const float *)((const char *)blob + hdr.off_verts); }
    const Node *nodes() const { return (const Node *)((const char *)blob + hdr.off_nodes); }
    const Prim *prims() const { return (const Prim *)((const char *)blob + hdr.off_prims); }
    const int *anc() const { return (const int *)((const char *)blob + hdr.off_anc); }
};

// hz_scene.hip
int scene_build(Scene *sc, const float *vert_grid, int d0, int d1, const float *vert_simp, int nvs,
                const int32_t *tri_simp, int nts, hz_stats *stats);

// device-side view handed to the traversal kernels
struct SceneView {
    const float *verts;
    const Node *nodes;
    const Prim *prims;
    const int *anc;  // hit-cache ancestors, one per leaf
    int d1;          // DEM row length (vertices)
    int n_top;       // BFS-ordered top nodes available for LDS staging
    float cx, cy, cz;
    float tau;       // box tests start at -tau and end at tfar + tau (hz_common.h: HZ_BOX_START_PADS)
};

inline SceneView scene_view(const Scene *sc) {
    SceneView v;
    v.verts = sc->verts(); v.nodes = sc->nodes(); v.prims = sc->prims(); v.anc = sc->anc();
    v.d1 = sc->hdr.d1; v.n_top = sc->hdr.n_top;
    v.cx = sc->hdr.center[0]; v.cy = sc->hdr.center[1]; v.cz = sc->hdr.center[2];
    v.tau = HZ_BOX_START_PADS * sc->hdr.pad;
    return v;
}

inline TileMap make_tile_map(int tiles_i, int tiles_j, int gw = 8) {
    TileMap m;
    m.tiles_i = tiles_i; m.tiles_j = tiles_j; m.gw = gw;
    m.rj = std::max(1, (tiles_j + 7) / 8);
    // patch rows: near-square patches, but at least 16 rows when the grid is high enough and at most 64.  Measured on
    // the 3601^2 tile (224 x 224 tiles; kernel ms per tile): 1 row 2016 (= one region per XCD, rounds 1-2: 2009 - 2014),
    // 2: 1957, 4: 1980, 8: 1945, 16: 1932, 32: 1931, 56: 1930, 112: 1928, 224: 1934; shadow kernel on its 224 x 7 grid of
    // super tiles (ms per sun position): 1: 1.455, 8: 1.265, 32: 1.317, 56: 1.260, 112: 1.417
    int pi = (tiles_i + m.rj - 1) / m.rj;
    pi = std::min(std::max(pi, std::min(16, tiles_i / 4)), 64);
    pi = std::max(1, std::min(pi, tiles_i));
    m.pi = pi;
    m.ri = (tiles_i + pi - 1) / pi;
    m.pi = (tiles_i + m.ri - 1) / m.ri;                  // drop patch rows that would be empty
    m.per_xcd = m.pi * m.ri * m.rj;
    return m;
}

// true if p points to device memory (HBM); false for host memory
bool is_device_ptr(const void *p);

// hz_horizon.hip
#define HZ_LEFT_LEVELS 4                 // leftover regions (hand-over levels)
struct HorizonArgs {
    const float *vec_norm, *vec_north;   // device
    const uint8_t *mask;                 // device
    float *hori;                         // device, may be null (skip_hori)
    int offset_0, offset_1, dim_in_0, dim_in_1;
    int row_begin, row_end;
    int azim_num, elev_num, alg;
    float hori_acc, low, up, dist;       // radians / metres
    float hori_fill, ray_org_elev;
    const float *azim_sin, *azim_cos, *elev_ang, *elev_sin, *elev_cos;  // device tables
    const int *mid_idx;
    int top_nodes, regroup, count_work, hit_cache;
    int level_stack;                     // 1: one-entry-per-level traversal stack (cannot overflow); 0: fast discipline
    const int *tile_list;                // null: every tile of the rows; else the n_list 8 x 8 blocks (workgroup number of
    int n_list;                          //   the full launch * 4 + wave) to compute: the blocks whose fast stack overflowed
    const unsigned short *near_idx;      // near-field certificates of rows [row_begin, row_end) (hz_near.hip) or null
    const float *near_r;
    int verify_near;                     // counting instantiation: N >= 1 re-traces one of every N shortened rays from parameter 0 (1: all)
    float *scratch_row;                  // counting instantiation only: null, or a device row of azim_num floats that takes EVERY store of the launch instead of
                                         // `hori` (the certificate monitor runs next to the production launch and must not write its rows)
    // Leftover cells (hz_horizon.hip): a block ends when at most left_min of its cells are unfinished and appends them to region
    // `left_mode` of left_rec; a launch with left_mode = l >= 1 finishes the records of region l - 1, in the order left_sort(l - 1) left
    // in left_sort (and hands over again when its left_min > 0 and a region l exists).  Region r: left_cap[r] records from record left_base[r].
    int left_min = 0;
    unsigned *left_rec = nullptr;        // HZ_LEFT_WORDS words per record; null: off
    int left_mode = 0;
    unsigned left_base[HZ_LEFT_LEVELS] = {0, 0, 0, 0}, left_cap[HZ_LEFT_LEVELS] = {0, 0, 0, 0};
    int left_regroup = 0, left_key_shift = 0;      // follow-up launches: compaction threshold (0: as `regroup`), log2 of the class width of the sort key
    uint32_t *left_sort = nullptr;       // sort scratch: 4 arrays of left_cap_max words (keys / values, in / out) + sort_temp_elems(left_cap_max)
    unsigned left_cap_max = 0;
    int no_persist = 0;                  // 1: one tile per workgroup instead of persistent waves (same-box A/Bs, tests)
    int persist_grid = 0;                // > 0 (tests): that many workgroups in a persistent launch, so that small grids run the block loop
    unsigned long long *counters;        // device u64[HZ_CNT_N] + int[HZ_REDO_CAP] (tiles to redo, count in [8]): [0] rays, [1] guards, [2] nodes, [3] tris, [4] cells,
                                         // [5..7] wave iterations, [8] waves whose fast-discipline stack overflowed,
                                         // [9] rays shortened by a certificate, [10] certificate violations (verify), [11] cells with a guard event,
                                         // [21] shortened rays that were re-traced (verify);
                                         // [24..27] = unsigned[8]: the per-XCD block queues of a persistent launch (hz_horizon.hip; zeroed by horizon_launch);
                                         // [HZ_CNT_LEFT ..): unsigned[HZ_LEFT_LEVELS][16], the control words of the leftover regions: [r][0] = slots
                                         // allocated in region r, [r][1] = valid records (k_left_keys), [r][8 + x] = groups of 64 handed to XCD x
};
// default hand-over thresholds: byte l = level l (hz_opts.left_min)
#define HZ_LEFT_DEFAULT 0x00000024u
// follow-up launches (hz_opts.left_tune; swept in round 6, profiles/r06/ab_leftover_tune.log: 73 ms against 76 with classes of one
// azimuth and the production launch's threshold of 36; classes of 32 azimuths: 95 ms)
#define HZ_LEFT_KEY_SHIFT 3              // log2 of the width of the azimuths-left classes of the sort key: 8 azimuths
#define HZ_LEFT_REGROUP 24               // compaction threshold in lanes
#define HZ_CNT_LEFT 32                   // first u64 word of the leftover control words
#define HZ_CNT_N 64                      // u64 words in front of the redo list
#define HZ_LEFT_WORDS 16                 // 32-bit words per leftover record
#define HZ_REDO_CAP 16384                // 8 x 8 blocks of one launch that can be repeated one by one after a stack overflow
int horizon_launch(const Scene *sc, const HorizonArgs &a, hipStream_t st, int *used_level_stack = nullptr);
int left_sort(const HorizonArgs &a, int region, hipStream_t st);
int horizon_num_blocks(const HorizonArgs &a);
int topo_launch(int kind, const float *azim, const float *hori, const float *vec_tilt, int len_0, int len_1,
                int len_2, float *out, hipStream_t st);
int svf_launch(const float *azim, const float *hori, const float *vec_tilt, int len_0, int len_1,
               int len_2, float *svf, hipStream_t st);

// hz_near.hip: near-field certificates (pre-pass of the horizon kernel)
struct NearArgs {
    const float *vec_norm, *vec_north;   // device
    const uint8_t *mask;
    const float *azim_sin, *azim_cos;    // device tables
    int offset_0, offset_1, dim_in_1, row_begin, row_end, azim_num, elev_num;
    float ray_org_elev, hori_acc, low, up;   // radians / metres
    unsigned short *near_idx;            // out [rows * dim_in_1][azim_num]
    float *near_r;                       // out [rows * dim_in_1]
    unsigned *reasons = nullptr;         // debug: device u32[20] histogram of refusals (hz_near.hip), or null
    int ignore_bad_map = 0;              // tests only (opts.no_near_skip < 0): certificates without the scene's per-cell guard
};
int near_launch(const Scene *sc, const NearArgs &a, hipStream_t st);
int near_max_azim();

// hz_locations.hip
struct LocationsArgs {
    const float *coords, *vec_norm, *vec_north, *ray_org_elev;   // device
    float *hori, *dist;                                           // device; dist may be null
    int num_loc, azim_num, elev_num, alg, hori_dist_out;
    float hori_acc, low, up, dist_m;
    const float *azim_sin, *azim_cos, *elev_ang, *elev_sin, *elev_cos;
    const int *mid_idx;
    unsigned long long *counters;
};
int locations_launch(const Scene *sc, const LocationsArgs &a, hipStream_t st);

// hz_shadow.hip
struct ShadowArgs {
    const float *vec_tilt, *vec_norm, *surf_enl_fac, *elevation;   // device
    const uint8_t *mask;
    int offset_0, offset_1, dim_in_0, dim_in_1;
    const float *suns;                   // device f32[num_sun][3]
    int num_sun;                         // positions computed by ONE launch (grid.y); outputs [num_sun][dim_in_0][dim_in_1]
    float sw_dir_cor_fill, dot_prod_min;
    int refrac_cor;
    const double *refrac_fac;            // refrac_cor: device f64[cells], shadow_refrac_factor()
    int which;                           // 0 shadow (u8), 1 sw_dir_cor (f32)
    uint8_t *out_u8; float *out_f32;
    int top_nodes;
    int count_work;                      // 1: the counting instantiation
    unsigned long long *counters;        // device u64[16]: [0] rays; count_work: [1] node visits, [2] triangle tests,
                                         // [3] / [4] wave-level node / leaf steps
};
int shadow_launch(const Scene *sc, const ShadowArgs &a, hipStream_t st);
int shadow_refrac_factor(const float *elevation, size_t n, double *out, hipStream_t st);

// hz_sort.hip: hand-written stable LSD radix sort (pairs) and exclusive scan, uint32
size_t sort_temp_elems(size_t n);
size_t scan_temp_elems(size_t n);
int radix_sort_pairs_u32(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, size_t n,
                         uint32_t *temp, hipStream_t st, int passes = 4);
int exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *temp, hipStream_t st);

// hz_bench.hip: machine calibration kernels (current device)
int bench_valu_peak(int packed, int waves_per_simd, double *winst_per_s_per_simd, double *clock_ghz, int *simds);
int bench_copy_peak(size_t bytes, double *gbs);
int bench_inst_rate(int op, double *cycles_per_inst);

// hz_prep.hip (device pointers)
int prep_slope(int which, const float *x, const float *y, const float *z, int len_0, int len_1,
               const float *rot_mat, int output_rot, float *vec_tilt, hipStream_t st);
int prep_lonlat2ecef(int ellps, const double *lon, const double *lat, const float *h, size_t n, double *X,
                     double *Y, double *Z, hipStream_t st);
int prep_ecef2enu(int ellps, double lon_or, double lat_or, const double *X, const double *Y, const double *Z,
                  size_t n, float *xe, float *ye, float *ze, hipStream_t st);
int prep_ecef2enu_vector(int ellps, double lon_or, double lat_or, const float *v, size_t n, float *o, hipStream_t st);
int prep_surf_norm(const double *lon, const double *lat, size_t n, float *o, hipStream_t st);
int prep_wgs2swiss(const double *lon, const double *lat, const float *h_wgs, size_t n, double *e, double *no, float *h_ch,
                   hipStream_t st);
int prep_swiss2wgs(const double *e, const double *no, const float *h_ch, size_t n, double *lon, double *lat, float *h_wgs,
                   hipStream_t st);
int prep_pack_vertices(const float *x, const float *y, const float *z, size_t n, size_t n_total, float *out,
                       hipStream_t st);
int prep_north_dir(int ellps, const double *X, const double *Y, const double *Z, const float *vn, size_t n,
                   float *o, hipStream_t st);

}  // namespace hz
