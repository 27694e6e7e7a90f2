// hz_scene.hip -- on-device LBVH build for DEM meshes (gfx950).
//
// Replaces the Embree scene build of the reference (initializeScene,
// horizon_comp.cpp:101-231 / shadow_comp.cpp:198-298, rtcCommitScene :223).
//
// Pipeline (all on one stream, no host round trip except the 6-float bounds):
//   1. k_bounds      min/max of all vertices            (HBM streaming, 12 B/vertex)
//   2. k_morton      one key per primitive: 2 x 16 bit Morton code of the
//                    centroid (x, y); primitive = DEM quad (2 triangles) or TIN triangle
//   3. rocprim radix sort of (key, primitive id)
//   4. k_karras      binary radix tree over the sorted keys (Karras 2012)
//   5. k_refit       leaf AABBs + bottom-up union with one atomic counter per node
//   6. k_top         breadth-first relabel of the top of the tree (for LDS staging)
//   7. k_emit        final 64 B nodes (both child AABBs inline) and 48 B leaf records
#include <cstring>
#include <cstdlib>
#include "hz_internal.h"
#include <rocprim/rocprim.hpp>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace hz {

// --- order-preserving float <-> uint encoding for atomic min/max -------------
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}

// bounds[0..2] = min (encoded), bounds[3..5] = max (encoded)
__global__ __launch_bounds__(256) void k_bounds(const float *__restrict__ v, size_t nvert,
                                               uint32_t *__restrict__ bounds) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvert;
         i += (size_t)gridDim.x * blockDim.x) {
        const float x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
        lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
        lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
        lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
    }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; k++) {
            atomicMin(&bounds[k], f2ord(lo[k]));
            atomicMax(&bounds[3 + k], f2ord(hi[k]));
        }
    }
}

struct BuildParams {
    const float *verts;      // grid vertices
    const float *vs;         // TIN vertices (may be null)
    const int32_t *ts;       // TIN indices
    int d0, d1, nq1;         // nq1 = d1 - 1 (quads per row)
    int n_quads, n_tin, n_prims;
    float cx, cy, cz, pad;   // frame centre, AABB padding
    float sx, sy, ox, oy;    // Morton quantisation: q = (x - ox) * sx
};

__device__ __forceinline__ uint32_t spread16(uint32_t v) {
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// corner vertices of primitive p; returns false for a TIN triangle (d unused)
__device__ __forceinline__ bool prim_vertices(const BuildParams &b, int p, float (&a)[3],
                                              float (&bb)[3], float (&c)[3], float (&d)[3]) {
    if (p < b.n_quads) {
        const int i = p / b.nq1, j = p - i * b.nq1;
        const float *r0 = b.verts + 3 * ((size_t)i * b.d1 + j);
        const float *r1 = r0 + 3 * (size_t)b.d1;
#pragma unroll
        for (int k = 0; k < 3; k++) { a[k] = r0[k]; bb[k] = r0[3 + k]; c[k] = r1[k]; d[k] = r1[3 + k]; }
        return true;
    }
    const int t = p - b.n_quads;
    const float *p0 = b.vs + 3 * (size_t)b.ts[3 * t], *p1 = b.vs + 3 * (size_t)b.ts[3 * t + 1];
    const float *p2 = b.vs + 3 * (size_t)b.ts[3 * t + 2];
#pragma unroll
    for (int k = 0; k < 3; k++) { a[k] = p0[k]; bb[k] = p1[k]; c[k] = p2[k]; d[k] = p2[k]; }
    return false;
}

__global__ __launch_bounds__(256) void k_morton(BuildParams b, uint32_t *__restrict__ keys,
                                               uint32_t *__restrict__ vals) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.n_prims) return;
    float a[3], bb[3], c[3], d[3];
    const bool quad = prim_vertices(b, p, a, bb, c, d);
    float mx, my;
    if (quad) { mx = 0.25f * (a[0] + bb[0] + c[0] + d[0]); my = 0.25f * (a[1] + bb[1] + c[1] + d[1]); }
    else { mx = (a[0] + bb[0] + c[0]) * (1.0f / 3.0f); my = (a[1] + bb[1] + c[1]) * (1.0f / 3.0f); }
    const float qx = fminf(fmaxf((mx - b.ox) * b.sx, 0.0f), 65535.0f);
    const float qy = fminf(fmaxf((my - b.oy) * b.sy, 0.0f), 65535.0f);
    keys[p] = (spread16((uint32_t)qy) << 1) | spread16((uint32_t)qx);
    vals[p] = (uint32_t)p;
}

// --- Karras 2012: binary radix tree ------------------------------------------
__device__ __forceinline__ int delta(const uint32_t *__restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t ki = keys[i], kj = keys[j];
    if (ki == kj) return 32 + __clz((uint32_t)(i ^ j));
    return __clz(ki ^ kj);
}

// children: >= 0 internal, < 0 leaf (~sorted position)
__global__ __launch_bounds__(256) void k_karras(const uint32_t *__restrict__ keys, int n,
                                               int2 *__restrict__ child, int *__restrict__ parent_int,
                                               int *__restrict__ parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int left = (lo == gamma) ? ~gamma : gamma;
    const int right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    child[i] = make_int2(left, right);
    if (left >= 0) parent_int[left] = i; else parent_leaf[~left] = i;
    if (right >= 0) parent_int[right] = i; else parent_leaf[~right] = i;
    if (i == 0) parent_int[0] = -1;
}

// --- leaf boxes + bottom-up refit ---------------------------------------------
// box arrays: lo/hi as float4 (w of lo = height as int bits for internal nodes)
__global__ __launch_bounds__(256) void k_refit(BuildParams b, const uint32_t *__restrict__ vals,
                                              const int2 *__restrict__ child,
                                              const int *__restrict__ parent_int,
                                              const int *__restrict__ parent_leaf,
                                              float4 *leaf_lo, float4 *leaf_hi,
                                              float4 *node_lo, float4 *node_hi, int *counter) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= b.n_prims) return;
    float a[3], bb[3], c[3], d[3];
    prim_vertices(b, (int)vals[s], a, bb, c, d);
    float lo[3], hi[3];
    const float ctr[3] = {b.cx, b.cy, b.cz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        lo[k] = (fminf(fminf(a[k], bb[k]), fminf(c[k], d[k])) - ctr[k]) - b.pad;
        hi[k] = (fmaxf(fmaxf(a[k], bb[k]), fmaxf(c[k], d[k])) - ctr[k]) + b.pad;
    }
    leaf_lo[s] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    leaf_hi[s] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    if (b.n_prims == 1) return;
    int height = 0;
    int node = parent_leaf[s];
    __threadfence();
    while (node >= 0) {
        if (atomicAdd(&counter[node], 1) == 0) return;   // first arrival: sibling not ready
        __threadfence();
        const int2 ch = child[node];
        // children boxes were published (store -> fence -> counter) before the first arrival
        const float4 l0 = (ch.x >= 0) ? node_lo[ch.x] : leaf_lo[~ch.x];
        const float4 h0 = (ch.x >= 0) ? node_hi[ch.x] : leaf_hi[~ch.x];
        const float4 l1 = (ch.y >= 0) ? node_lo[ch.y] : leaf_lo[~ch.y];
        const float4 h1 = (ch.y >= 0) ? node_hi[ch.y] : leaf_hi[~ch.y];
        const int hgt0 = (ch.x >= 0) ? __float_as_int(l0.w) : 0;
        const int hgt1 = (ch.y >= 0) ? __float_as_int(l1.w) : 0;
        height = max(hgt0, hgt1) + 1;
        float4 nl = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), __int_as_float(height));
        float4 nh = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
        node_lo[node] = nl;
        node_hi[node] = nh;
        __threadfence();
        node = parent_int[node];
    }
}

// --- breadth-first relabel of the top of the tree -------------------------------
// perm[old] = new.  Single workgroup, single lane: n_top <= a few thousand.
__global__ void k_top(const int2 *__restrict__ child, int n_nodes, int n_top,
                      int *__restrict__ perm, int *__restrict__ top, uint8_t *__restrict__ in_top) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // BFS
    int head = 0, tail = 0;
    top[tail++] = 0;
    while (head < tail && tail < n_top) {
        const int2 ch = child[top[head++]];
        if (ch.x >= 0 && tail < n_top) top[tail++] = ch.x;
        if (ch.y >= 0 && tail < n_top) top[tail++] = ch.y;
    }
    const int k = tail;  // actual number of relabelled nodes (<= n_top, <= n_nodes)
    for (int r = 0; r < k; r++) if (top[r] < k) in_top[top[r]] = 1;
    for (int r = 0; r < k; r++) perm[top[r]] = r;
    // displaced low-index nodes take the vacated high indices
    int x = 0;
    for (int r = 0; r < k; r++) {
        if (top[r] >= k) {
            while (in_top[x]) x++;
            perm[x++] = top[r];
        }
    }
    (void)n_nodes;
}

__global__ __launch_bounds__(256) void k_iota(int *__restrict__ perm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = i;
}

__global__ __launch_bounds__(256) void k_emit_nodes(int n_nodes, const int2 *__restrict__ child,
                                                   const int *__restrict__ perm,
                                                   const float4 *__restrict__ leaf_lo,
                                                   const float4 *__restrict__ leaf_hi,
                                                   const float4 *__restrict__ node_lo,
                                                   const float4 *__restrict__ node_hi,
                                                   Node *__restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int2 ch = child[i];
    const float4 l0 = (ch.x >= 0) ? node_lo[ch.x] : leaf_lo[~ch.x];
    const float4 h0 = (ch.x >= 0) ? node_hi[ch.x] : leaf_hi[~ch.x];
    const float4 l1 = (ch.y >= 0) ? node_lo[ch.y] : leaf_lo[~ch.y];
    const float4 h1 = (ch.y >= 0) ? node_hi[ch.y] : leaf_hi[~ch.y];
    Node n;
    n.lo0[0] = l0.x; n.lo0[1] = l0.y; n.lo0[2] = l0.z; n.hi0[0] = h0.x; n.hi0[1] = h0.y; n.hi0[2] = h0.z;
    n.lo1[0] = l1.x; n.lo1[1] = l1.y; n.lo1[2] = l1.z; n.hi1[0] = h1.x; n.hi1[1] = h1.y; n.hi1[2] = h1.z;
    n.c0 = (ch.x >= 0) ? perm[ch.x] : ch.x;
    n.c1 = (ch.y >= 0) ? perm[ch.y] : ch.y;
    n.pad_[0] = 0; n.pad_[1] = 0;
    nodes[perm[i]] = n;
}

__global__ __launch_bounds__(256) void k_emit_prims(BuildParams b, const uint32_t *__restrict__ vals,
                                                   Prim *__restrict__ prims) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= b.n_prims) return;
    float a[3], bb[3], c[3], d[3];
    const bool quad = prim_vertices(b, (int)vals[s], a, bb, c, d);
    Prim p;
#pragma unroll
    for (int k = 0; k < 3; k++) { p.a[k] = a[k]; p.b[k] = bb[k]; p.c[k] = c[k]; p.d[k] = d[k]; }
    if (!quad) p.d[0] = __int_as_float(0x7fc00000);   // NaN: TIN triangle, single test
    prims[s] = p;
}

// single primitive: a root node whose second child is an empty box
__global__ void k_single_node(const float4 *leaf_lo, const float4 *leaf_hi, Node *nodes) {
    Node n;
    n.lo0[0] = leaf_lo[0].x; n.lo0[1] = leaf_lo[0].y; n.lo0[2] = leaf_lo[0].z;
    n.hi0[0] = leaf_hi[0].x; n.hi0[1] = leaf_hi[0].y; n.hi0[2] = leaf_hi[0].z;
    for (int k = 0; k < 3; k++) { n.lo1[k] = INFINITY; n.hi1[k] = -INFINITY; }
    n.c0 = ~0; n.c1 = ~0; n.pad_[0] = n.pad_[1] = 0;
    nodes[0] = n;
}

// ------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct TempBuf {
    void *p = nullptr;
    ~TempBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
};

int scene_build(Scene *sc, const float *vert_grid, int d0, int d1,
                const float *vert_simp, int nvs, const int32_t *tri_simp, int nts,
                hz_stats *stats) {
    if (d0 < 2 || d1 < 2) return set_error(HZ_ERR_ARG, "dem_dim_0 and dem_dim_1 must be >= 2");
    if (d0 > 32767 || d1 > 32767)
        return set_error(HZ_ERR_ARG, "maximal allowed input length for dem_dim_0 and dem_dim_1 is 32'767");
    const bool has_tin = (nvs >= 3) && (nts >= 1);   // horizon_comp.cpp:199
    const size_t nvert = (size_t)d0 * d1;
    const int n_quads = (d0 - 1) * (d1 - 1);
    const int n_tin = has_tin ? nts : 0;
    const int n_prims = n_quads + n_tin;
    const int n_nodes = (n_prims > 1) ? n_prims - 1 : 1;
    hipStream_t st = sc->stream;
    Timer t_total; t_total.start();

    // ---- blob allocation -------------------------------------------------------
    BlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = HZ_BLOB_MAGIC; h.version = HZ_BLOB_VERSION;
    h.d0 = d0; h.d1 = d1; h.n_quads = n_quads; h.n_tin = n_tin; h.n_prims = n_prims; h.n_nodes = n_nodes;
    h.off_verts = sizeof(BlobHeader);
    h.off_nodes = align_up(h.off_verts + nvert * 12, 256);
    h.off_prims = align_up(h.off_nodes + (size_t)n_nodes * sizeof(Node), 256);
    h.total_bytes = align_up(h.off_prims + (size_t)n_prims * sizeof(Prim), 256);
    HZ_HIP(hipMalloc(&sc->blob, h.total_bytes));
    sc->owns_blob = true;
    sc->blob_bytes = h.total_bytes;
    char *blob = (char *)sc->blob;
    float *d_verts = (float *)(blob + h.off_verts);
    Node *d_nodes = (Node *)(blob + h.off_nodes);
    Prim *d_prims = (Prim *)(blob + h.off_prims);

    // ---- upload vertices (host or device source) ---------------------------------
    Timer t_h2d; t_h2d.start();
    HZ_HIP(hipMemcpyAsync(d_verts, vert_grid, nvert * 12, hipMemcpyDefault, st));
    TempBuf b_vs, b_ts;
    if (has_tin) {
        HZ_HIP(b_vs.alloc((size_t)nvs * 12));
        HZ_HIP(b_ts.alloc((size_t)nts * 12));
        HZ_HIP(hipMemcpyAsync(b_vs.p, vert_simp, (size_t)nvs * 12, hipMemcpyDefault, st));
        HZ_HIP(hipMemcpyAsync(b_ts.p, tri_simp, (size_t)nts * 12, hipMemcpyDefault, st));
    }
    HZ_HIP(hipStreamSynchronize(st));
    const double h2d_s = t_h2d.stop();

    Timer t_bvh; t_bvh.start();
    // ---- 1. bounds ----------------------------------------------------------------
    TempBuf b_bounds;
    HZ_HIP(b_bounds.alloc(6 * 4));
    uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HZ_HIP(hipMemcpyAsync(b_bounds.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    {
        const int grid = (int)std::min<size_t>((nvert + 255) / 256, 2048);
        hipLaunchKernelGGL(k_bounds, dim3(grid), dim3(256), 0, st, d_verts, nvert, (uint32_t *)b_bounds.p);
        if (has_tin)
            hipLaunchKernelGGL(k_bounds, dim3(std::min((nvs + 255) / 256, 2048)), dim3(256), 0, st,
                               (const float *)b_vs.p, (size_t)nvs, (uint32_t *)b_bounds.p);
    }
    uint32_t enc[6];
    HZ_HIP(hipMemcpyAsync(enc, b_bounds.p, sizeof(enc), hipMemcpyDeviceToHost, st));
    HZ_HIP(hipStreamSynchronize(st));
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) { lo[k] = ord2f(enc[k]); hi[k] = ord2f(enc[3 + k]); }
    for (int k = 0; k < 3; k++)
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k]))
            return set_error(HZ_ERR_ARG, "vertex buffer contains non-finite coordinates");
    float maxabs = 0.0f;
    double diag2 = 0.0;
    for (int k = 0; k < 3; k++) {
        h.lo[k] = lo[k]; h.hi[k] = hi[k];
        h.center[k] = 0.5f * lo[k] + 0.5f * hi[k];
        maxabs = std::max(maxabs, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
        diag2 += ((double)hi[k] - lo[k]) * ((double)hi[k] - lo[k]);
    }
    // conservative padding: see DESIGN.md section 4 ("why the tree cannot change a result")
    h.pad = (float)(1.0e-6 * std::sqrt(diag2) + 4.0 * FLT_EPSILON * (double)maxabs) + FLT_MIN;

    BuildParams bp;
    bp.verts = d_verts; bp.vs = (const float *)b_vs.p; bp.ts = (const int32_t *)b_ts.p;
    bp.d0 = d0; bp.d1 = d1; bp.nq1 = d1 - 1;
    bp.n_quads = n_quads; bp.n_tin = n_tin; bp.n_prims = n_prims;
    bp.cx = h.center[0]; bp.cy = h.center[1]; bp.cz = h.center[2]; bp.pad = h.pad;
    bp.ox = lo[0]; bp.oy = lo[1];
    bp.sx = (hi[0] > lo[0]) ? 65535.0f / (hi[0] - lo[0]) : 0.0f;
    bp.sy = (hi[1] > lo[1]) ? 65535.0f / (hi[1] - lo[1]) : 0.0f;

    // ---- 2./3. Morton keys + radix sort ---------------------------------------------
    TempBuf b_k0, b_k1, b_v0, b_v1, b_sort;
    HZ_HIP(b_k0.alloc((size_t)n_prims * 4)); HZ_HIP(b_k1.alloc((size_t)n_prims * 4));
    HZ_HIP(b_v0.alloc((size_t)n_prims * 4)); HZ_HIP(b_v1.alloc((size_t)n_prims * 4));
    const int gp = (n_prims + 255) / 256;
    hipLaunchKernelGGL(k_morton, dim3(gp), dim3(256), 0, st, bp, (uint32_t *)b_k0.p, (uint32_t *)b_v0.p);
    size_t sort_bytes = 0;
    HZ_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint32_t *)b_k0.p, (uint32_t *)b_k1.p,
                                     (uint32_t *)b_v0.p, (uint32_t *)b_v1.p, (size_t)n_prims, 0, 32, st));
    HZ_HIP(b_sort.alloc(sort_bytes));
    HZ_HIP(rocprim::radix_sort_pairs(b_sort.p, sort_bytes, (uint32_t *)b_k0.p, (uint32_t *)b_k1.p,
                                     (uint32_t *)b_v0.p, (uint32_t *)b_v1.p, (size_t)n_prims, 0, 32, st));
    const uint32_t *keys = (const uint32_t *)b_k1.p;
    const uint32_t *vals = (const uint32_t *)b_v1.p;

    // ---- 4./5. hierarchy + refit ------------------------------------------------------
    TempBuf b_child, b_pint, b_pleaf, b_llo, b_lhi, b_nlo, b_nhi, b_cnt, b_perm, b_top, b_intop;
    HZ_HIP(b_child.alloc((size_t)n_nodes * 8));
    HZ_HIP(b_pint.alloc((size_t)n_nodes * 4));
    HZ_HIP(b_pleaf.alloc((size_t)n_prims * 4));
    HZ_HIP(b_llo.alloc((size_t)n_prims * 16)); HZ_HIP(b_lhi.alloc((size_t)n_prims * 16));
    HZ_HIP(b_nlo.alloc((size_t)n_nodes * 16)); HZ_HIP(b_nhi.alloc((size_t)n_nodes * 16));
    HZ_HIP(b_cnt.alloc((size_t)n_nodes * 4));
    HZ_HIP(hipMemsetAsync(b_cnt.p, 0, (size_t)n_nodes * 4, st));
    HZ_HIP(hipMemsetAsync(b_nlo.p, 0, (size_t)n_nodes * 16, st));
    if (n_prims > 1)
        hipLaunchKernelGGL(k_karras, dim3((n_prims - 1 + 255) / 256), dim3(256), 0, st, keys, n_prims,
                           (int2 *)b_child.p, (int *)b_pint.p, (int *)b_pleaf.p);
    hipLaunchKernelGGL(k_refit, dim3(gp), dim3(256), 0, st, bp, vals, (const int2 *)b_child.p,
                       (const int *)b_pint.p, (const int *)b_pleaf.p, (float4 *)b_llo.p, (float4 *)b_lhi.p,
                       (float4 *)b_nlo.p, (float4 *)b_nhi.p, (int *)b_cnt.p);

    // ---- 6./7. relabel + emit ------------------------------------------------------------
    int n_top = 0;
    if (n_prims > 1) {
        n_top = std::min(n_nodes, HZ_MAX_TOP_NODES);
        HZ_HIP(b_perm.alloc((size_t)n_nodes * 4));
        HZ_HIP(b_top.alloc((size_t)n_top * 4));
        HZ_HIP(b_intop.alloc((size_t)n_top));
        HZ_HIP(hipMemsetAsync(b_intop.p, 0, (size_t)n_top, st));
        hipLaunchKernelGGL(k_iota, dim3((n_nodes + 255) / 256), dim3(256), 0, st, (int *)b_perm.p, n_nodes);
        hipLaunchKernelGGL(k_top, dim3(1), dim3(64), 0, st, (const int2 *)b_child.p, n_nodes, n_top,
                           (int *)b_perm.p, (int *)b_top.p, (uint8_t *)b_intop.p);
        hipLaunchKernelGGL(k_emit_nodes, dim3((n_nodes + 255) / 256), dim3(256), 0, st, n_nodes,
                           (const int2 *)b_child.p, (const int *)b_perm.p, (const float4 *)b_llo.p,
                           (const float4 *)b_lhi.p, (const float4 *)b_nlo.p, (const float4 *)b_nhi.p, d_nodes);
    } else {
        hipLaunchKernelGGL(k_single_node, dim3(1), dim3(1), 0, st, (const float4 *)b_llo.p,
                           (const float4 *)b_lhi.p, d_nodes);
        n_top = 1;
    }
    hipLaunchKernelGGL(k_emit_prims, dim3(gp), dim3(256), 0, st, bp, vals, d_prims);
    // tree height = height stored with the root box
    float4 root_lo = make_float4(0, 0, 0, 0);
    if (n_prims > 1) HZ_HIP(hipMemcpyAsync(&root_lo, b_nlo.p, 16, hipMemcpyDeviceToHost, st));
    HZ_HIP(hipStreamSynchronize(st));
    HZ_HIP(hipGetLastError());
    int height = 1;
    if (n_prims > 1) memcpy(&height, &root_lo.w, 4);
    h.height = height;
    h.n_top = n_top;
    if (height > HZ_MAX_STACK)
        return set_error(HZ_ERR_DEPTH, "BVH height %d exceeds the traversal stack (%d)", height, HZ_MAX_STACK);
    HZ_HIP(hipMemcpyAsync(blob, &h, sizeof(h), hipMemcpyHostToDevice, st));
    HZ_HIP(hipStreamSynchronize(st));
    sc->hdr = h;
    const double bvh_s = t_bvh.stop();
    if (stats) {
        stats->t_bvh_s += bvh_s; stats->t_h2d_s += h2d_s;
        stats->bvh_height = height; stats->scene_bytes = h.total_bytes;
    }
    (void)t_total;
    return HZ_OK;
}

}  // namespace hz
