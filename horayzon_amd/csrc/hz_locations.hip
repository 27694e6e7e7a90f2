// hz_locations.hip -- horizon for arbitrary locations (gfx950).
//
// Replaces horizon_locations_comp (horizon_comp.cpp:828-1094): per location the point is first
// put onto the mesh with a closest-hit query along +normal, then -normal (max 100 km, :951-957);
// the search then runs exactly as in the gridded case (any-hit), or with closest-hit queries that
// also return the distance to the horizon (the *_hori_dist variants, :519-612).
// One lane per location; locations are few, so this kernel shares the traversal code of the
// gridded path but is not a throughput target.
#include "hz_search.h"

namespace hz {

#define HZ_TPB 256

struct LocParams {
    SceneView sv;
    Tables tb;
    const float *coords, *vec_norm, *vec_north, *ray_org_elev;
    float *hori, *dist;
    int num_loc, stack_bytes;
    float tfar;
    unsigned long long *counters;   // [0] rays, [1] guards, [4] locations on the mesh
};

template <int ALG, bool DIST>
__global__ __launch_bounds__(HZ_TPB) void k_locations(LocParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *stack = reinterpret_cast<int *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = blockIdx.x * HZ_TPB + tid;
    const Tables &t = p.tb;
    bool done = i >= p.num_loc;
    float ox = 0, oy = 0, oz = 0;
    float r00 = 0, r01 = 0, r02 = 0, r10 = 0, r11 = 0, r12 = 0, r20 = 0, r21 = 0, r22 = 0;
    unsigned found = 0;
    if (!done) {
        const float norm_x = p.vec_norm[3 * i], norm_y = p.vec_norm[3 * i + 1], norm_z = p.vec_norm[3 * i + 2];
        const float north_x = p.vec_north[3 * i], north_y = p.vec_north[3 * i + 1], north_z = p.vec_north[3 * i + 2];
        const float ini_x = p.coords[3 * i], ini_y = p.coords[3 * i + 1], ini_z = p.coords[3 * i + 2];
        // put the location onto the mesh: +normal, then -normal (horizon_comp.cpp:947-957)
        float dist = 0.0f;
        // (every box test starts at -tau: the frame of a RayBox is the origin shifted back by tau, hz_common.h)
        const float tau = p.sv.tau;
        const float icx = ini_x - p.sv.cx, icy = ini_y - p.sv.cy, icz = ini_z - p.sv.cz;
        RayBox rbn = hz_raybox(icx - tau * norm_x, icy - tau * norm_y, icz - tau * norm_z, norm_x, norm_y, norm_z);
        bool hit = hz_closest<HZ_TPB>(p.sv.nodes, p.sv.prims, stack, tid, ini_x, ini_y, ini_z, norm_x, norm_y, norm_z,
                                      100000.0f, tau, rbn, &dist);
        if (!hit) {
            rbn = hz_raybox(icx + tau * norm_x, icy + tau * norm_y, icz + tau * norm_z, -norm_x, -norm_y, -norm_z);
            hit = hz_closest<HZ_TPB>(p.sv.nodes, p.sv.prims, stack, tid, ini_x, ini_y, ini_z, -norm_x, -norm_y,
                                     -norm_z, 100000.0f, tau, rbn, &dist);
            dist = (float)((double)dist * -1.0);
        }
        if (!hit) {
            done = true;                                            // outputs keep the caller's NaN
        } else {
            found = 1;
            const float lift = dist + p.ray_org_elev[i];             // :961-963
            ox = ini_x + norm_x * lift; oy = ini_y + norm_y * lift; oz = ini_z + norm_z * lift;
            const float east_x = north_y * norm_z - north_z * norm_y;
            const float east_y = north_z * norm_x - north_x * norm_z;
            const float east_z = north_x * norm_y - north_y * norm_x;
            r00 = east_x; r01 = north_x; r02 = norm_x;
            r10 = east_y; r11 = north_y; r12 = norm_y;
            r20 = east_z; r21 = north_z; r22 = norm_z;
        }
    }
    const float ocx = ox - p.sv.cx, ocy = oy - p.sv.cy, ocz = oz - p.sv.cz;
    Sink out;
    const size_t ii = done ? 0 : (size_t)i;
    out.hori = p.hori + ii * (size_t)t.azim_num;
    out.dist = DIST ? p.dist + ii * (size_t)t.azim_num : nullptr;
    out.dist_hit = 0.0f;
    out.stage = nullptr; out.stride = 0;
    Search s;
    s.k = 0; s.phase = PH_NEWAZ; s.ind = 0; s.prev = 0; s.pazim = 0; s.count = 0;
    s.lim_up = 0; s.lim_low = 0; s.elev_samp = 0; s.ev = 0;
    unsigned rays = 0, guards = 0;
    bool last_hit = false;
    TravCounters tc; tc.nodes = 0; tc.tris = 0; tc.w_nodes = 0; tc.w_leaves = 0;
    while (__ballot(!done) != 0ull) {
        bool have_ray = false;
        float dx = 0, dy = 0, dz = 1;
        if (!done) {
            if (advance<ALG, false>(s, last_hit, t, out, guards)) {
                const float ec = t.elev_cos[s.ind], es = t.elev_sin[s.ind];
                const float rx = ec * t.azim_sin[s.k], ry = ec * t.azim_cos[s.k], rz = es;
                dx = (r00 * rx + r01 * ry) + r02 * rz;
                dy = (r10 * rx + r11 * ry) + r12 * rz;
                dz = (r20 * rx + r21 * ry) + r22 * rz;
                have_ray = true;
                rays++;
            } else {
                done = true;
            }
        }
        if (have_ray) {
            const RayBox rb = hz_raybox(ocx - p.sv.tau * dx, ocy - p.sv.tau * dy, ocz - p.sv.tau * dz, dx, dy, dz);
            if (DIST) {                                             // castRay_intersect1, :268-292
                float d = 0.0f;
                last_hit = hz_closest<HZ_TPB>(p.sv.nodes, p.sv.prims, stack, tid, ox, oy, oz, dx, dy, dz, p.tfar, p.sv.tau, rb, &d);
                if (last_hit) out.dist_hit = d;                     // :545-547 / :589-591
            } else {                                                // castRay_occluded1, :241-262
                TravState ts; hz_trav_reset(ts);
                bool overflow = false;       // unused: the one-entry-per-level stack cannot overflow
                last_hit = hz_trace<HZ_TPB, false>(p.sv.nodes, p.sv.prims, nullptr, 0, stack, tid, ox, oy, oz, dx, dy,
                                                   dz, p.tfar, p.tfar + 2.0f * p.sv.tau, rb, ts, 0, 16, tc, 0, overflow) == 1;
            }
        }
    }
    unsigned long long r = rays, g = guards, f = found;
    for (int off = 32; off > 0; off >>= 1) { r += __shfl_xor(r, off); g += __shfl_xor(g, off); f += __shfl_xor(f, off); }
    if (lane == 0) {
        if (r) atomicAdd(&p.counters[0], r);
        if (g) atomicAdd(&p.counters[1], g);
        if (f) atomicAdd(&p.counters[4], f);
    }
}

template <int ALG, bool DIST>
static int launch(const LocParams &p, int grid, size_t lds, hipStream_t st) {
    HZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_locations<ALG, DIST>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_locations<ALG, DIST>), dim3(grid), dim3(HZ_TPB), lds, st, p);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

int locations_launch(const Scene *sc, const LocationsArgs &a, hipStream_t st) {
    if (a.num_loc <= 0) return HZ_OK;
    LocParams p;
    p.sv = scene_view(sc);
    p.tb.azim_sin = a.azim_sin; p.tb.azim_cos = a.azim_cos;
    p.tb.elev_ang = a.elev_ang; p.tb.elev_sin = a.elev_sin; p.tb.elev_cos = a.elev_cos; p.tb.mid_idx = a.mid_idx;
    p.tb.azim_num = a.azim_num; p.tb.elev_num = a.elev_num;
    p.tb.hori_acc = a.hori_acc; p.tb.low = a.low; p.tb.up = a.up;
    p.tb.step = (double)a.hori_acc / 5.0;
    p.coords = a.coords; p.vec_norm = a.vec_norm; p.vec_north = a.vec_north; p.ray_org_elev = a.ray_org_elev;
    p.hori = a.hori; p.dist = a.dist; p.num_loc = a.num_loc; p.tfar = a.dist_m;
    p.counters = a.counters;
    const int depth = 3 * std::max(sc->hdr.height, 1);
    p.stack_bytes = depth * HZ_TPB * 4;
    const size_t lds = (size_t)p.stack_bytes;
    const int grid = (a.num_loc + HZ_TPB - 1) / HZ_TPB;
    const bool d = a.hori_dist_out != 0;
    switch (a.alg) {
        case ALG_DISCRETE: return d ? launch<ALG_DISCRETE, true>(p, grid, lds, st) : launch<ALG_DISCRETE, false>(p, grid, lds, st);
        case ALG_BINARY: return d ? launch<ALG_BINARY, true>(p, grid, lds, st) : launch<ALG_BINARY, false>(p, grid, lds, st);
        default: return launch<ALG_GUESS, false>(p, grid, lds, st);
    }
}

}  // namespace hz
