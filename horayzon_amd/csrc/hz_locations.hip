// hz_locations.hip -- horizon for arbitrary locations (gfx950).
//
// Replaces horizon_locations_comp (horizon_comp.cpp:828-1094): per location the point is first
// put onto the mesh with a closest-hit query along +normal, then -normal (max 100 km, :951-957);
// the search then runs exactly as in the gridded case (any-hit), or with closest-hit queries that
// also return the distance to the horizon (the *_hori_dist variants, :519-612).
// One lane per location.  Round 5: the any-hit searches run like the gridded kernel's -- a lane whose ray is decided takes
// the next sample of ITS search as soon as fewer than 40 lanes of the wave are still traversing (ray compaction), on the
// fast stack discipline of hz_trace with an in-kernel retry on the one-entry-per-level stack for a ray that runs out of
// entries (the pattern of k_shadow_refill).  Rounds 1-4 traced every ray of a wave to completion before any lane got its
// next one.  The closest-hit variants (distance output, snap onto the mesh) keep their plain per-lane loop.
#include "hz_search.h"

namespace hz {

#define HZ_TPB 256
#ifndef HZ_LOC_REGROUP
#define HZ_LOC_REGROUP 16      // refill when fewer lanes than this are still traversing (40, the gridded kernel's value, until round 5; swept 40 ... 16:
                               // monotonic, -6 % at 16 -- a refill here starts a whole binary search, profiles/r05/ab_shadow_and_locations_thresholds.log)
#endif
#ifndef HZ_LOC_FAST_CAP
#define HZ_LOC_FAST_CAP 27     // entries of the fast stack: (27 + 2) x 1 KiB of LDS per workgroup, 5 resident
#endif

struct LocParams {
    SceneView sv;
    Tables tb;
    const float *coords, *vec_norm, *vec_north, *ray_org_elev;
    float *hori, *dist;
    int num_loc, stack_bytes;
    int stack_cap;                  // entries of the fast stack (0: the one-entry-per-level stack only)
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
    if (DIST) {
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
            if (have_ray) {                                             // castRay_intersect1, :268-292
                const RayBox rb = hz_raybox(ocx - p.sv.tau * dx, ocy - p.sv.tau * dy, ocz - p.sv.tau * dz, dx, dy, dz);
                float d = 0.0f;
                last_hit = hz_closest<HZ_TPB>(p.sv.nodes, p.sv.prims, stack, tid, ox, oy, oz, dx, dy, dz, p.tfar, p.sv.tau, rb, &d);
                if (last_hit) out.dist_hit = d;                         // :545-547 / :589-591
            }
        }
    } else {                                                            // castRay_occluded1, :241-262
        const bool fast = p.stack_cap > 0;
        int *fstack = reinterpret_cast<int *>(smem + (fast ? 2 * HZ_TPB * 4 : 0));     // two padding rows below the fast stack (hz_trace)
        bool ray_active = false, overflow = false;
        float dx = 0, dy = 0, dz = 1;
        RayBox rb = hz_raybox(0, 0, 0, 0, 0, 1);
        TravState ts; hz_trav_reset(ts);
        const float tfar_box = p.tfar + 2.0f * p.sv.tau;
        while (__ballot(!done) != 0ull) {
            if (!done && !ray_active) {
                if (advance<ALG, false>(s, last_hit, t, out, guards)) {
                    const float ec = t.elev_cos[s.ind], es = t.elev_sin[s.ind];
                    const float rx = ec * t.azim_sin[s.k], ry = ec * t.azim_cos[s.k], rz = es;
                    dx = (r00 * rx + r01 * ry) + r02 * rz;
                    dy = (r10 * rx + r11 * ry) + r12 * rz;
                    dz = (r20 * rx + r21 * ry) + r22 * rz;
                    rb = hz_raybox(ocx - p.sv.tau * dx, ocy - p.sv.tau * dy, ocz - p.sv.tau * dz, dx, dy, dz);
                    hz_trav_reset(ts);
                    overflow = false;
                    ray_active = true;
                    rays++;
                } else {
                    done = true;
                }
            }
            if (ray_active) {
                int res;
                if (fast) {
                    res = hz_trace<HZ_TPB, false, 2, false, false>(p.sv.nodes, p.sv.prims, nullptr, 0, fstack, tid, ox, oy, oz, dx, dy, dz,
                                                                   p.tfar, tfar_box, rb, ts, HZ_LOC_REGROUP, 24, tc, p.stack_cap, overflow);
                    if (res != 2 && overflow) {        // out of entries: this ray again, to completion, one entry per tree level
                        bool unused = false;
                        hz_trav_reset(ts);
                        res = hz_trace<HZ_TPB, false, 2, false, true>(p.sv.nodes, p.sv.prims, nullptr, 0, fstack, tid, ox, oy, oz, dx, dy, dz,
                                                                      p.tfar, tfar_box, rb, ts, 0, 24, tc, 0, unused);
                    }
                } else {
                    res = hz_trace<HZ_TPB, false, 2, false, true>(p.sv.nodes, p.sv.prims, nullptr, 0, stack, tid, ox, oy, oz, dx, dy, dz,
                                                                  p.tfar, tfar_box, rb, ts, HZ_LOC_REGROUP, 24, tc, 0, overflow);
                }
                if (res != 2) { ray_active = false; last_hit = (res == 1); }
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
    const int height = std::max(sc->hdr.height, 1);
    const bool d = a.hori_dist_out != 0;
    // closest hit: individual child links, up to 3 per level; any-hit: the fast stack (+ 2 padding rows) if the level stack
    // of the in-kernel retry fits into it, else the level stack alone
    const int fast_cap = std::min(HZ_LOC_FAST_CAP, 3 * height + 1);
    const bool fast = !d && fast_cap >= 5 && fast_cap >= height;
    p.stack_cap = fast ? fast_cap : 0;
    // (the snap onto the mesh is a closest-hit query in every variant: its 3 x height entries must fit too)
    const int depth = std::max(3 * height, fast ? fast_cap + 2 : 0);
    p.stack_bytes = depth * HZ_TPB * 4;
    const size_t lds = (size_t)p.stack_bytes;
    const int grid = (a.num_loc + HZ_TPB - 1) / HZ_TPB;
    switch (a.alg) {
        case ALG_DISCRETE: return d ? launch<ALG_DISCRETE, true>(p, grid, lds, st) : launch<ALG_DISCRETE, false>(p, grid, lds, st);
        case ALG_BINARY: return d ? launch<ALG_BINARY, true>(p, grid, lds, st) : launch<ALG_BINARY, false>(p, grid, lds, st);
        default: return launch<ALG_GUESS, false>(p, grid, lds, st);
    }
}

}  // namespace hz
