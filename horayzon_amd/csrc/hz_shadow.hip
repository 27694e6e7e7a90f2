// hz_shadow.hip -- shadow mask / direct-shortwave correction kernels for gfx950.
//
// Replaces CppTerrain::shadow and CppTerrain::sw_dir_cor (shadow_comp.cpp:386-491,
// :495-605): one lane per inner-domain cell, at most one any-hit ray with
// tfar = infinity towards the sun (optionally bent by atmospheric refraction,
// shadow_comp.cpp:430-446).  Same tile / XCD mapping and LDS staging as the horizon
// kernel; the BVH is persistent in the Terrain handle (built once, :318-380).
#include "hz_internal.h"
#include "hz_crmath.h"
#include <cstdlib>

namespace hz {

#define HZ_TPB 256
#ifndef HZ_SHADOW_LEAF_BIAS
#define HZ_SHADOW_LEAF_BIAS 24   // node step when 16 * n_node >= bias * n_leaf (hz_trace); 20 until round 4 (profiles/r04/shadow_leaf_bias.log)
#endif

struct ShadowParams {
    SceneView sv;
    const float *vec_tilt, *vec_norm, *surf_enl_fac, *elevation;
    const uint8_t *mask;
    int offset_0, offset_1, dim_in_0, dim_in_1;
    TileMap tm;
    const float *suns;       // device f32[num_sun][3]; blockIdx.y selects the position
    size_t out_stride;       // cells per sun position (outputs of position s start at s * out_stride)
    float fill, dot_prod_min;
    int refrac, which;
    const double *refrac_fac;   // refrac: per cell ((double)pressure / 101.0) * (283.0 / (273.0 + (double)temperature_degC)), k_refrac_factor
    uint8_t *out_u8; float *out_f32;
    int top_nodes, stack_bytes;
    int stack_cap;                 // FAST: entries of the fast stack (hz_trace, !LEVELSTACK), sentinel included
    int nb;                  // k_shadow_refill: 8 x 8 blocks per wave
    unsigned long long *counters;
};

// The float libm calls of the reference's refraction branch (acos / tan / pow / cos / sin on float arguments
// resolve to the float overloads, shadow_comp.cpp:19): self-contained correctly rounded versions shared with
// the CPU oracle (hz_crmath.h): with refraction, shadow codes and sw_dir_cor are bit-identical TO THAT CONTRACT (the
// correctly rounded float), not to the reference's glibc calls -- tests/test_gpu_parity.py::
// test_refraction_against_platform_libm measures the distance to the platform libm (a tolerance, not equality).
__device__ __forceinline__ float f_acos(float x) { return hz_crm_acosf(x); }
__device__ __forceinline__ float f_tan(float x) { return hz_crm_tanf(x); }
__device__ __forceinline__ float f_cos(float x) { return hz_crm_cosf(x); }
__device__ __forceinline__ float f_sin(float x) { return hz_crm_sinf(x); }
__device__ __forceinline__ float f_pow(float x, float y) { return hz_crm_powf(x, y); }

// shadow_comp.cpp:43-62: float in, double arithmetic, float out.  The divisions by the two constants are the correctly rounded
// quotients in four instructions each (hz_crmath.h: hz_crm_div_const)
__device__ __forceinline__ float deg2rad_f(float a) {
    return (float)(hz_crm_div_const((double)a, 180.0, 1.0 / 180.0) * 3.14159265358979323846);
}
__device__ __forceinline__ float rad2deg_f(float a) {
    return (float)(hz_crm_div_const((double)a, 3.14159265358979323846, 1.0 / 3.14159265358979323846) * 180.0);
}

// shadow_comp.cpp:96-106
__device__ __forceinline__ void vec_unit(float &x, float &y, float &z) {
    const float mag = __builtin_sqrtf((x * x + y * y) + z * z);
    x = x / mag; y = y / mag; z = z / mag;
}

// shadow_comp.cpp:135-159 (Saemundsson), float/double promotions as there.  `fac` = the factor that depends on the cell's
// pressure and temperature only: ((double)pressure / 101.0) * (283.0 / (273.0 + (double)temp)) (:156), formed once per cell
// at Terrain::initialise (k_refrac_factor) -- the same float64 value the expression yields here
__device__ __forceinline__ float atmos_refrac(float elev_ang_true, double fac) {
    elev_ang_true = __builtin_fmaxf(-1.0f, __builtin_fminf(elev_ang_true, 90.0f));
    float refrac_cor = (float)(1.02 / (double)f_tan(deg2rad_f(
        (float)((double)elev_ang_true + 10.3 / ((double)elev_ang_true + 5.11)))));
    refrac_cor = (float)((double)refrac_cor + 0.0019279);
    refrac_cor = (float)((double)refrac_cor * fac);
    return (float)((double)refrac_cor * (1.0 / 60.0));
}

// :438-441 and the pressure / temperature factor of :156, per cell
__global__ __launch_bounds__(256) void k_refrac_factor(const float *__restrict__ elevation, size_t n, double *__restrict__ out) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const float temperature_ref = 283.15f, pressure_ref = 101.0f, lapse_rate = 0.0065f;
    const float g = 9.81f, R_d = 287.0f;
    const float expo = g / (R_d * lapse_rate);                 // :353-354
    const float temperature = temperature_ref - (lapse_rate * elevation[c]);
    const float pressure = pressure_ref * f_pow(temperature / temperature_ref, expo);
    const float temp = (float)((double)temperature - 273.15);  // K2degC, :146-148
    out[c] = ((double)pressure / 101.0) * (283.0 / (273.0 + (double)temp));
}

int shadow_refrac_factor(const float *elevation, size_t n, double *out, hipStream_t st) {
    if (n == 0) return HZ_OK;
    hipLaunchKernelGGL(k_refrac_factor, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, elevation, n, out);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

// any-hit traversal to completion (regroup = 0: never suspends; no LDS nodelet: top = null)
template <bool COUNT>
__device__ __forceinline__ void shadow_counters(const ShadowParams &p, unsigned rays, const TravCounters &tc, int lane) {
    unsigned long long r = rays;
    for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off);
    if (lane == 0 && r) atomicAdd(&p.counters[0], r);
    if (COUNT) {
        unsigned long long nc = tc.nodes, tn = tc.tris, wn = tc.w_nodes, wl = tc.w_leaves;
        for (int off = 32; off > 0; off >>= 1) {
            nc += __shfl_xor(nc, off); tn += __shfl_xor(tn, off); wn += __shfl_xor(wn, off); wl += __shfl_xor(wl, off);
        }
        if (lane == 0) {
            atomicAdd(&p.counters[1], nc); atomicAdd(&p.counters[2], tn);
            atomicAdd(&p.counters[3], wn); atomicAdd(&p.counters[4], wl);
        }
    }
}

// Per-cell set-up shared by both kernels: classification without a ray (masked / self-shaded / outside ang_max) is
// written at once; otherwise the ray (origin, direction) and the two dot products come back and 1 is returned.
struct ShadowRay { float ox, oy, oz, dx, dy, dz, dot_ts, dot_ns; };

__device__ __forceinline__ int shadow_setup(const ShadowParams &p, int i, int j, float p_sun_x, float p_sun_y, float p_sun_z,
                                            uint8_t *out_u8, float *out_f32, ShadowRay &r) {
    const size_t cell = (size_t)i * p.dim_in_1 + j;
    if (p.mask[cell] != 1) {                                       // shadow_comp.cpp:480-484 / :594-598
        if (p.which == 0) out_u8[cell] = 3; else out_f32[cell] = p.fill;
        return 0;
    }
    const float tilt_x = p.vec_tilt[3 * cell], tilt_y = p.vec_tilt[3 * cell + 1], tilt_z = p.vec_tilt[3 * cell + 2];
    const float norm_x = p.vec_norm[3 * cell], norm_y = p.vec_norm[3 * cell + 1], norm_z = p.vec_norm[3 * cell + 2];
    const float ray_org_elev = 0.05f;                              // :388, :497
    const float *v = p.sv.verts + 3 * ((size_t)(i + p.offset_0) * p.sv.d1 + (size_t)(j + p.offset_1));
    const float ox = v[0] + norm_x * ray_org_elev;
    const float oy = v[1] + norm_y * ray_org_elev;
    const float oz = v[2] + norm_z * ray_org_elev;
    float sun_x = p_sun_x - ox, sun_y = p_sun_y - oy, sun_z = p_sun_z - oz;   // :422-425
    vec_unit(sun_x, sun_y, sun_z);
    float dot_prod_ns = (norm_x * sun_x + norm_y * sun_y) + norm_z * sun_z;
    if (p.refrac == 1) {                                           // :430-446
        const double fac = p.refrac_fac[cell];
        // The refraction turns the sun direction about an axis normal to it by theta <= 38.81' * fac (the formula's value at
        // the clamp -1 deg, where it is largest) = 0.01129 * fac rad: tilt . sun changes by less than |tilt| * theta.  A cell
        // whose tilted surface faces away from the unrefracted sun by more than that is self-shaded (`!(dot > 0)` below,
        // and `!(dot > dot_prod_min)` with dot_prod_min > 0) whatever the refraction is: night positions and back slopes
        // skip the five libm calls.  (NaN factor: the comparison is false, the full path runs.)
        {
            const float dot0 = (tilt_x * sun_x + tilt_y * sun_y) + tilt_z * sun_z;
            const float tl = __builtin_fmaxf((tilt_x * tilt_x + tilt_y * tilt_y) + tilt_z * tilt_z, 1.0f);     // >= |tilt|
            const float bound = (0.0114f * __builtin_fabsf((float)fac)) * tl * 1.01f + 1.0e-5f;
            if (dot0 < -bound) {
                if (p.which == 0) out_u8[cell] = 1; else out_f32[cell] = 0.0f;
                return 0;
            }
        }
        const float elev_ang_true = (float)(90.0 - (double)rad2deg_f(f_acos(dot_prod_ns)));
        const float refrac_cor = atmos_refrac(elev_ang_true, fac);
        float k_x = sun_y * norm_z - sun_z * norm_y;
        float k_y = sun_z * norm_x - sun_x * norm_z;
        float k_z = sun_x * norm_y - sun_y * norm_x;
        vec_unit(k_x, k_y, k_z);
        const float theta = deg2rad_f(refrac_cor);                // vec_rot, :109-132
        const float ct = f_cos(theta), st = f_sin(theta);
        const float part = (float)((double)((k_x * sun_x + k_y * sun_y) + k_z * sun_z) * (1.0 - (double)ct));
        const float rx = (sun_x * ct + (k_y * sun_z - k_z * sun_y) * st) + k_x * part;
        const float ry = (sun_y * ct + (k_z * sun_x - k_x * sun_z) * st) + k_y * part;
        const float rz = (sun_z * ct + (k_x * sun_y - k_y * sun_x) * st) + k_z * part;
        sun_x = rx; sun_y = ry; sun_z = rz;
        dot_prod_ns = (norm_x * sun_x + norm_y * sun_y) + norm_z * sun_z;
    }
    const float dot_prod_ts = (tilt_x * sun_x + tilt_y * sun_y) + tilt_z * sun_z;
    if (p.which == 0) {                                            // :451-478
        if (!(dot_prod_ts > 0.0f)) { out_u8[cell] = 1; return 0; }
    } else {                                                       // :561-592
        if (!(dot_prod_ts > p.dot_prod_min)) { out_f32[cell] = 0.0f; return 0; }
    }
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = sun_x; r.dy = sun_y; r.dz = sun_z;
    r.dot_ts = dot_prod_ts; r.dot_ns = dot_prod_ns;
    return 1;
}

__device__ __forceinline__ void shadow_result(const ShadowParams &p, size_t cell, bool hit, const ShadowRay &r,
                                              uint8_t *out_u8, float *out_f32) {
    if (p.which == 0) { out_u8[cell] = hit ? 2 : 0; return; }
    if (hit) { out_f32[cell] = 0.0f; return; }
    float dot_prod_ns = r.dot_ns;
    if (dot_prod_ns < p.dot_prod_min) dot_prod_ns = p.dot_prod_min;
    out_f32[cell] = (r.dot_ts / dot_prod_ns) * p.surf_enl_fac[cell];
}


// k_shadow_refill: a wave owns p.nb consecutive 8 x 8 blocks (the same quadrant of nb neighbouring 16 x 16
// tiles) and hands their cells to its lanes as they become free.  The rays of one sun position are parallel, so with one
// ray per lane traced to completion (the round-2 kernel k_shadow, removed in round 6) a wave lasted as long as its longest
// ray while most lanes idled (45 % of the lanes active per VALU instruction, profiles/r02/pmc_shadow_summary.json); here a
// lane whose ray is finished takes the next cell once fewer than `regroup` lanes are still traversing (the ray compaction of
// the horizon kernel).  Cells are handed out in block order, so the rays in flight stay neighbours.
#ifndef HZ_SHADOW_FAST_CAP_DEFAULT
#define HZ_SHADOW_FAST_CAP_DEFAULT 19   // 21 KiB of LDS per workgroup: 7 resident; round 4: level stack 1.074 -> fast stack 0.964 ms per position at 5 workgroups (profiles/r04/ab_shadow_fast_stack.log)
#endif
#ifndef HZ_SHADOW_REGROUP
#define HZ_SHADOW_REGROUP 16      // refill when fewer lanes than this are still traversing (40 until round 5; re-swept 40 ... 8 after the loop rewrite:
                                  // 16 - 20 is the flat optimum, -0.5 % without and -7.7 % with refraction, profiles/r05/ab_shadow_and_locations_thresholds.log)
#endif
// FAST: the fast stack discipline of hz_trace (every pending sibling its own entry, written for the fewest instructions;
// `stack_cap` entries behind two padding rows).  A ray that runs out of entries is traced again, to completion, with the
// one-entry-per-level discipline in the same LDS column of its lane (the launcher makes sure the tree's height fits).
// resident workgroups per CU the register allocation is held to (profiles/r04/ab_shadow_fast_stack.log, ms per sun position:
// unconstrained = 86 VGPRs = 5 workgroups 0.955; 6 (80 VGPRs, no spills) 0.884; 7 (72 VGPRs, 6 spilled) 0.857; 8 (64 VGPRs, 21
// spilled) 0.957).  The counting instantiation carries ~10 more live values and is left at 5.
#ifndef HZ_SHADOW_WG
#define HZ_SHADOW_WG 7
#endif
// COUNT: also count node visits / triangle tests / wave-level steps (Terrain count_work; the roofline's B_trav)
template <bool COUNT, bool FAST>
__global__ __launch_bounds__(HZ_TPB, COUNT ? 5 : HZ_SHADOW_WG) void k_shadow_refill(ShadowParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *stack = reinterpret_cast<int *>(smem + (FAST ? 2 * HZ_TPB * 4 : 0));
    const int tid = threadIdx.x;
    int ti = 0, tj = 0;
    const bool has_tile = hz_tile_of_block(p.tm, blockIdx.x, &ti, &tj);      // tile map over super tiles (16 x 16 nb cells)
    const int wave = tid >> 6, lane = tid & 63;
    const int sun_idx = blockIdx.y;
    const float p_sun_x = p.suns[3 * sun_idx], p_sun_y = p.suns[3 * sun_idx + 1], p_sun_z = p.suns[3 * sun_idx + 2];
    uint8_t *const out_u8 = p.out_u8 ? p.out_u8 + (size_t)sun_idx * p.out_stride : nullptr;
    float *const out_f32 = p.out_f32 ? p.out_f32 + (size_t)sun_idx * p.out_stride : nullptr;
    unsigned rays = 0;
    TravCounters tc; tc.nodes = 0; tc.tris = 0; tc.w_nodes = 0; tc.w_leaves = 0;
    const int i_base = ti * 16 + (wave >> 1) * 8, j_base = tj * (16 * p.nb) + (wave & 1) * 8;
    const int total = has_tile ? 64 * p.nb : 0;
    int next = 0;                                   // cells handed out so far (wave uniform)
    bool ray_active = false;
    ShadowRay r; r.ox = r.oy = r.oz = 0.0f; r.dx = r.dy = 0.0f; r.dz = 1.0f; r.dot_ts = r.dot_ns = 0.0f;
    size_t cell = 0;
    RayBox rb = hz_raybox(0, 0, 0, 0, 0, 1);
    TravState ts; hz_trav_reset(ts);
    bool overflow = false;
    for (;;) {
        if (next < total) {
            const unsigned long long need = __ballot(!ray_active);
            if (need != 0ull) {
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
                const int my = next + rank;
                next += __popcll(need);
                if (!ray_active && my < total) {
                    const int i = i_base + ((my & 63) >> 3), j = j_base + (my >> 6) * 16 + (my & 7);
                    if (i < p.dim_in_0 && j < p.dim_in_1 && shadow_setup(p, i, j, p_sun_x, p_sun_y, p_sun_z, out_u8, out_f32, r)) {
                        cell = (size_t)i * p.dim_in_1 + j;
                        // (box tests start at -tau: hz_common.h; tfar is infinite here)
                        rb = hz_raybox((r.ox - p.sv.cx) - p.sv.tau * r.dx, (r.oy - p.sv.cy) - p.sv.tau * r.dy, (r.oz - p.sv.cz) - p.sv.tau * r.dz,
                                       r.dx, r.dy, r.dz);
                        hz_trav_reset(ts);
                        overflow = false;
                        ray_active = true;
                        rays++;
                    }
                }
            }
        }
        if (__ballot(ray_active) == 0ull) {
            if (next >= total) break;
            continue;
        }
        if (ray_active) {
            // while cells are left the traversal returns when fewer than 40 lanes are busy (and one finished)
            int res = hz_trace<HZ_TPB, COUNT, 2, false, !FAST>(p.sv.nodes, p.sv.prims, nullptr, 0, stack, tid, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz,
                                                    __builtin_inff(), __builtin_inff(), rb, ts, (next < total) ? HZ_SHADOW_REGROUP : 0, HZ_SHADOW_LEAF_BIAS, tc, p.stack_cap, overflow);
            if (FAST && res != 2 && overflow) {
                bool unused = false;
                hz_trav_reset(ts);
                res = hz_trace<HZ_TPB, COUNT, 2, false, true>(p.sv.nodes, p.sv.prims, nullptr, 0, stack, tid, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz,
                                                                    __builtin_inff(), __builtin_inff(), rb, ts, 0, HZ_SHADOW_LEAF_BIAS, tc, 0, unused);
                if (COUNT && p.counters) atomicAdd(&p.counters[8], 1ull);     // rays traced twice
            }
            if (res != 2) {
                shadow_result(p, cell, res == 1, r, out_u8, out_f32);
                ray_active = false;
            }
        }
    }
    shadow_counters<COUNT>(p, rays, tc, lane);
}

std::atomic<int> g_shadow_fast_cap{HZ_SHADOW_FAST_CAP_DEFAULT};      // hz_debug_set (hz_internal.h)
std::atomic<int> g_topo_wide{0};

int shadow_launch(const Scene *sc, const ShadowArgs &a, hipStream_t st) {
    ShadowParams p;
    p.sv = scene_view(sc);
    p.vec_tilt = a.vec_tilt; p.vec_norm = a.vec_norm; p.surf_enl_fac = a.surf_enl_fac; p.elevation = a.elevation;
    p.mask = a.mask;
    p.offset_0 = a.offset_0; p.offset_1 = a.offset_1; p.dim_in_0 = a.dim_in_0; p.dim_in_1 = a.dim_in_1;
    if (a.dim_in_0 <= 0 || a.dim_in_1 <= 0) return HZ_OK;
    const int tiles_i = (a.dim_in_0 + 15) / 16;
    p.tm = make_tile_map(tiles_i, (a.dim_in_1 + 15) / 16);
    p.suns = a.suns; p.out_stride = (size_t)a.dim_in_0 * (size_t)a.dim_in_1;
    if (a.num_sun <= 0) return HZ_OK;
    p.fill = a.sw_dir_cor_fill; p.dot_prod_min = a.dot_prod_min;
    p.refrac = (a.refrac_cor && a.refrac_fac != nullptr) ? 1 : 0; p.which = a.which; p.refrac_fac = a.refrac_fac;
    p.out_u8 = a.out_u8; p.out_f32 = a.out_f32;
    // LDS stack: one entry per tree level (hz_common.h): 12 - 14 KB per workgroup, so the VGPRs decide the residency
    p.stack_bytes = std::max(sc->hdr.height, 1) * HZ_TPB * 4;
    p.top_nodes = 0;
    p.counters = a.counters;
    // fast stack (k_shadow_refill<.., true>): HZ_SHADOW_FAST_CAP entries (0: off) behind two padding rows, if the level
    // stack of the in-kernel retry fits into them
    const int fast_cap_env = g_shadow_fast_cap.load(std::memory_order_relaxed);      // (hz_debug_set("shadow_fast_cap", n): tests)
    const int height = std::max(sc->hdr.height, 1);
    const int fast_cap = std::min(fast_cap_env, 3 * height + 1);
    const bool fast = fast_cap >= 5 && fast_cap >= height;
    p.stack_cap = fast ? fast_cap : 0;
    const size_t lds = fast ? (size_t)(fast_cap + 2) * HZ_TPB * 4 : (size_t)p.stack_bytes;
    // blocks per wave: as many as leave >= ~48 workgroups per CU over the whole launch (measured on the 3601^2 tile, 144
    // positions per launch: 2 / 4 / 8 / 16 / 32 / 64 blocks -> 1.64 / 1.49 / 1.42 / 1.36 / 1.29 / 1.51 ms per position; a single
    // position: 1 / 2 / 4 / 8 / 16 blocks -> 2.49 / 1.77 / 1.67 / 1.65 / 1.79 ms, k_shadow 2.52 ms)
    {
        const int tiles_j = (a.dim_in_1 + 15) / 16;
        const double wgs = (double)tiles_i * tiles_j * (double)a.num_sun;
        int nb = (int)(wgs / (48.0 * 256.0));
        nb = std::max(1, std::min(std::min(nb, 32), tiles_j));
        // a count that cuts the tile row without a ragged last super tile if there is one nearby (224 tiles: 24 blocks per
        // wave measured 1.41 ms, 32 blocks 1.29 ms per position)
        for (int c = nb; c >= std::max(1, (nb * 3) / 4); c--)
            if (tiles_j % c == 0) { nb = c; break; }
        p.nb = nb;
        p.tm = make_tile_map(tiles_i, (tiles_j + nb - 1) / nb);
    }
    // grid.x is a multiple of 8, so the workgroup -> XCD assignment (flat id % 8) is the same for every grid.y row
    const dim3 grid((unsigned)(p.tm.per_xcd * 8), (unsigned)a.num_sun);
#define HZ_LAUNCH_SHADOW(K) do { \
        HZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(K, grid, dim3(HZ_TPB), lds, st, p); } while (0)
    if (fast) { if (a.count_work) HZ_LAUNCH_SHADOW((k_shadow_refill<true, true>)); else HZ_LAUNCH_SHADOW((k_shadow_refill<false, true>)); }
    else { if (a.count_work) HZ_LAUNCH_SHADOW((k_shadow_refill<true, false>)); else HZ_LAUNCH_SHADOW((k_shadow_refill<false, false>)); }
#undef HZ_LAUNCH_SHADOW
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

}  // namespace hz
