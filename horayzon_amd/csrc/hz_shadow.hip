// hz_shadow.hip -- shadow mask / direct-shortwave correction kernels for gfx950.
//
// Replaces CppTerrain::shadow and CppTerrain::sw_dir_cor (shadow_comp.cpp:386-491,
// :495-605): one lane per inner-domain cell, at most one any-hit ray with
// tfar = infinity towards the sun (optionally bent by atmospheric refraction,
// shadow_comp.cpp:430-446).  Same tile / XCD mapping and LDS staging as the horizon
// kernel; the BVH is persistent in the Terrain handle (built once, :318-380).
#include "hz_internal.h"
#include "hz_crmath.h"

namespace hz {

#define HZ_TPB 256
#ifndef HZ_SHADOW_LEAF_BIAS
#define HZ_SHADOW_LEAF_BIAS 20   // node step when 16 * n_node >= bias * n_leaf (hz_trace)
#endif

struct ShadowParams {
    SceneView sv;
    const float *vec_tilt, *vec_norm, *surf_enl_fac, *elevation;
    const uint8_t *mask;
    int offset_0, offset_1, dim_in_0, dim_in_1;
    TileMap tm;
    float sun_x, sun_y, sun_z;
    float fill, dot_prod_min;
    int refrac, which;
    uint8_t *out_u8; float *out_f32;
    int top_nodes, stack_bytes;
    unsigned long long *counters;
};

// The float libm calls of the reference's refraction branch (acos / tan / pow / cos / sin on float arguments
// resolve to the float overloads, shadow_comp.cpp:19): self-contained correctly rounded versions shared with
// the CPU oracle (hz_crmath.h), so shadow codes and sw_dir_cor are bit-identical with refraction too.
__device__ __forceinline__ float f_acos(float x) { return hz_crm_acosf(x); }
__device__ __forceinline__ float f_tan(float x) { return hz_crm_tanf(x); }
__device__ __forceinline__ float f_cos(float x) { return hz_crm_cosf(x); }
__device__ __forceinline__ float f_sin(float x) { return hz_crm_sinf(x); }
__device__ __forceinline__ float f_pow(float x, float y) { return hz_crm_powf(x, y); }

__device__ __forceinline__ float deg2rad_f(float a) { return (float)(((double)a / 180.0) * 3.14159265358979323846); }
__device__ __forceinline__ float rad2deg_f(float a) { return (float)(((double)a / 3.14159265358979323846) * 180.0); }

// shadow_comp.cpp:96-106
__device__ __forceinline__ void vec_unit(float &x, float &y, float &z) {
    const float mag = __builtin_sqrtf((x * x + y * y) + z * z);
    x = x / mag; y = y / mag; z = z / mag;
}

// shadow_comp.cpp:135-159 (Saemundsson), float/double promotions as there
__device__ __forceinline__ float atmos_refrac(float elev_ang_true, float temp, float pressure) {
    elev_ang_true = __builtin_fmaxf(-1.0f, __builtin_fminf(elev_ang_true, 90.0f));
    float refrac_cor = (float)(1.02 / (double)f_tan(deg2rad_f(
        (float)((double)elev_ang_true + 10.3 / ((double)elev_ang_true + 5.11)))));
    refrac_cor = (float)((double)refrac_cor + 0.0019279);
    refrac_cor = (float)((double)refrac_cor * (((double)pressure / 101.0) * (283.0 / (273.0 + (double)temp))));
    return (float)((double)refrac_cor * (1.0 / 60.0));
}

// any-hit traversal to completion (regroup = 0: never suspends; no LDS nodelet: top = null)
template <bool COUNT>
__device__ __forceinline__ bool occluded(const SceneView &sv, int *stack, int tid,
                                         float ox, float oy, float oz, float dx, float dy, float dz,
                                         float tfar, TravCounters &tc) {
    const RayBox rb = hz_raybox(ox - sv.cx, oy - sv.cy, oz - sv.cz, dx, dy, dz);
    TravState ts; hz_trav_reset(ts);
    unsigned overflow = 0;      // unused: the one-entry-per-level stack cannot overflow
    return hz_trace<HZ_TPB, COUNT>(sv.nodes, sv.prims, nullptr, 0, stack, tid, ox, oy, oz, dx, dy, dz, tfar, rb,
                                   ts, 0, HZ_SHADOW_LEAF_BIAS, tc, 0, overflow) == 1;
}

// COUNT: also count node visits / triangle tests / wave-level steps (Terrain count_work; the roofline's B_trav)
template <bool COUNT>
__global__ __launch_bounds__(HZ_TPB) void k_shadow(ShadowParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *stack = reinterpret_cast<int *>(smem);
    const int tid = threadIdx.x;
    int ti = 0, tj = 0;
    const bool has_tile = hz_tile_of_block(p.tm, blockIdx.x, &ti, &tj);
    const int wave = tid >> 6, lane = tid & 63;
    const int i = ti * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int j = tj * 16 + (wave & 1) * 8 + (lane & 7);
    const bool in_dom = has_tile && (i < p.dim_in_0) && (j < p.dim_in_1);
    const size_t cell = in_dom ? ((size_t)i * p.dim_in_1 + j) : 0;
    unsigned rays = 0;
    TravCounters tc; tc.nodes = 0; tc.tris = 0; tc.w_nodes = 0; tc.w_leaves = 0;
    if (in_dom) {
        if (p.mask[cell] != 1) {                                   // shadow_comp.cpp:480-484 / :594-598
            if (p.which == 0) p.out_u8[cell] = 3; else p.out_f32[cell] = p.fill;
        } else {
            const float tilt_x = p.vec_tilt[3 * cell], tilt_y = p.vec_tilt[3 * cell + 1], tilt_z = p.vec_tilt[3 * cell + 2];
            const float norm_x = p.vec_norm[3 * cell], norm_y = p.vec_norm[3 * cell + 1], norm_z = p.vec_norm[3 * cell + 2];
            const float ray_org_elev = 0.05f;                      // :388, :497
            const float *v = p.sv.verts + 3 * ((size_t)(i + p.offset_0) * p.sv.d1 + (size_t)(j + p.offset_1));
            const float ox = v[0] + norm_x * ray_org_elev;
            const float oy = v[1] + norm_y * ray_org_elev;
            const float oz = v[2] + norm_z * ray_org_elev;
            float sun_x = p.sun_x - ox, sun_y = p.sun_y - oy, sun_z = p.sun_z - oz;   // :422-425
            vec_unit(sun_x, sun_y, sun_z);
            float dot_prod_ns = (norm_x * sun_x + norm_y * sun_y) + norm_z * sun_z;
            if (p.refrac == 1) {                                   // :430-446
                const float elev_ang_true = (float)(90.0 - (double)rad2deg_f(f_acos(dot_prod_ns)));
                const float temperature_ref = 283.15f, pressure_ref = 101.0f, lapse_rate = 0.0065f;
                const float g = 9.81f, R_d = 287.0f;
                const float expo = g / (R_d * lapse_rate);         // :353-354
                const float temperature = temperature_ref - (lapse_rate * p.elevation[cell]);
                const float pressure = pressure_ref * f_pow(temperature / temperature_ref, expo);
                const float refrac_cor = atmos_refrac(elev_ang_true, (float)((double)temperature - 273.15), pressure);
                float k_x = sun_y * norm_z - sun_z * norm_y;
                float k_y = sun_z * norm_x - sun_x * norm_z;
                float k_z = sun_x * norm_y - sun_y * norm_x;
                vec_unit(k_x, k_y, k_z);
                const float theta = deg2rad_f(refrac_cor);        // vec_rot, :109-132
                const float ct = f_cos(theta), st = f_sin(theta);
                const float part = (float)((double)((k_x * sun_x + k_y * sun_y) + k_z * sun_z) * (1.0 - (double)ct));
                const float rx = (sun_x * ct + (k_y * sun_z - k_z * sun_y) * st) + k_x * part;
                const float ry = (sun_y * ct + (k_z * sun_x - k_x * sun_z) * st) + k_y * part;
                const float rz = (sun_z * ct + (k_x * sun_y - k_y * sun_x) * st) + k_z * part;
                sun_x = rx; sun_y = ry; sun_z = rz;
                dot_prod_ns = (norm_x * sun_x + norm_y * sun_y) + norm_z * sun_z;
            }
            const float dot_prod_ts = (tilt_x * sun_x + tilt_y * sun_y) + tilt_z * sun_z;
            const float inf = __builtin_inff();
            if (p.which == 0) {                                    // :451-478
                if (dot_prod_ts > 0.0f) {
                    rays = 1;
                    const bool h = occluded<COUNT>(p.sv, stack, tid, ox, oy, oz, sun_x, sun_y, sun_z, inf, tc);
                    p.out_u8[cell] = h ? 2 : 0;
                } else {
                    p.out_u8[cell] = 1;
                }
            } else {                                               // :561-592
                if (dot_prod_ts > p.dot_prod_min) {
                    rays = 1;
                    const bool h = occluded<COUNT>(p.sv, stack, tid, ox, oy, oz, sun_x, sun_y, sun_z, inf, tc);
                    if (h) p.out_f32[cell] = 0.0f;
                    else {
                        if (dot_prod_ns < p.dot_prod_min) dot_prod_ns = p.dot_prod_min;
                        p.out_f32[cell] = (dot_prod_ts / dot_prod_ns) * p.surf_enl_fac[cell];
                    }
                } else {
                    p.out_f32[cell] = 0.0f;
                }
            }
        }
    }
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

int shadow_launch(const Scene *sc, const ShadowArgs &a, hipStream_t st) {
    ShadowParams p;
    p.sv = scene_view(sc);
    p.vec_tilt = a.vec_tilt; p.vec_norm = a.vec_norm; p.surf_enl_fac = a.surf_enl_fac; p.elevation = a.elevation;
    p.mask = a.mask;
    p.offset_0 = a.offset_0; p.offset_1 = a.offset_1; p.dim_in_0 = a.dim_in_0; p.dim_in_1 = a.dim_in_1;
    if (a.dim_in_0 <= 0 || a.dim_in_1 <= 0) return HZ_OK;
    const int tiles_i = (a.dim_in_0 + 15) / 16;
    p.tm = make_tile_map(tiles_i, (a.dim_in_1 + 15) / 16);
    p.sun_x = a.sun[0]; p.sun_y = a.sun[1]; p.sun_z = a.sun[2];
    p.fill = a.sw_dir_cor_fill; p.dot_prod_min = a.dot_prod_min;
    p.refrac = a.refrac_cor; p.which = a.which;
    p.out_u8 = a.out_u8; p.out_f32 = a.out_f32;
    // LDS stack: one entry per tree level (hz_common.h): 12 - 14 KB per workgroup, so the kernel's 59 VGPRs decide
    // the residency (8 waves per SIMD)
    p.stack_bytes = std::max(sc->hdr.height, 1) * HZ_TPB * 4;
    p.top_nodes = 0;
    p.counters = a.counters;
    const size_t lds = (size_t)p.stack_bytes;
    const int grid = p.tm.per_xcd * 8;
    if (a.count_work) {
        HZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_shadow<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_shadow<true>, dim3(grid), dim3(HZ_TPB), lds, st, p);
    } else {
        HZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_shadow<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_shadow<false>, dim3(grid), dim3(HZ_TPB), lds, st, p);
    }
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

}  // namespace hz
