// hz_scene.hip -- on-device LBVH build for DEM meshes (gfx950).
//
// Replaces the Embree scene build of the reference (initializeScene,
// horizon_comp.cpp:101-231 / shadow_comp.cpp:198-298, rtcCommitScene :223).
//
// Pipeline (all on one stream, no host round trip except the 6-float bounds):
//   1. k_bounds      min/max of all vertices            (HBM streaming, 12 B/vertex)
//   2. k_morton      one key per primitive: Morton code of the quad's grid index (DEM quad = 2 triangles) or of
//                    the centroid position (TIN triangle, own key range)
//   3. radix sort of (key, primitive id): hz_sort.hip (stable LSD, 8 bit digits, hand written)
//   4. k_karras      binary radix tree over the sorted keys (Karras 2012)
//   5. k_leaf_boxes / k_refit_pass  leaf AABBs; bottom-up union, level synchronous with work lists
//   6. k_roots/scan  collapse along the 2-bit Morton digits: a binary node starts a 4-wide
//                    node when its common-prefix length enters a new digit (quadtree level)
//   7. k_emit4       temp 4-wide nodes with conservatively quantised child bounds (8 bit; one x range per column half, one y range per row half, one z range per child),
//                    children sorted tallest first
//   8. k_bfs_*       breadth-first numbering of ALL nodes, level by level, with the children of every node in one
//                    contiguous block of 4 slots (nodes or 48 B leaf records); emits the final nodes and leaves
//      k_anc_bfs     per leaf the node HZ_ANC_LEVELS above it (hit cache)
#include <cstring>
#include <cstdlib>
#include "hz_internal.h"
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

// Keys.  DEM quads: the Morton code of the quad's GRID INDEX (i, j), 15 bits each -- the 2-bit digits of the key are
// then the levels of a perfect quadtree over the grid, every 4-wide node has its four children (full, contiguous
// child blocks; boxes that follow the quads) wherever the grid is not cut off.  (Keys from quantised centroid
// POSITIONS cut quads at arbitrary code boundaries: 40 % more node slots and 16 % unused leaf slots on the 3601^2
// tile.)  TIN triangles: position Morton code of the centroid (15 bits per axis) in a key range of their own
// (bit 30), i.e. the outer-domain mesh forms a subtree of its own under the root.  Any partition is a valid BVH.
__global__ __launch_bounds__(256) void k_morton(BuildParams b, uint32_t *__restrict__ keys,
                                               uint32_t *__restrict__ vals) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.n_prims) return;
    if (p < b.n_quads) {
        const uint32_t i = (uint32_t)(p / b.nq1), j = (uint32_t)(p - (int)i * b.nq1);
        keys[p] = (spread16(i) << 1) | spread16(j);        // dem_dim <= 32767: 15 bits each
    } else {
        float a[3], bb[3], c[3], d[3];
        prim_vertices(b, p, a, bb, c, d);
        const float mx = (a[0] + bb[0] + c[0]) * (1.0f / 3.0f), my = (a[1] + bb[1] + c[1]) * (1.0f / 3.0f);
        const float qx = fminf(fmaxf((mx - b.ox) * b.sx * 0.5f, 0.0f), 32767.0f);
        const float qy = fminf(fmaxf((my - b.oy) * b.sy * 0.5f, 0.0f), 32767.0f);
        keys[p] = 0x40000000u | (spread16((uint32_t)qy) << 1) | spread16((uint32_t)qx);
    }
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
                                               int *__restrict__ parent_leaf, uint8_t *__restrict__ plen,
                                               int *__restrict__ first) {
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
    plen[i] = (uint8_t)dnode;      // common prefix length of the node's key range (0..63)
    first[i] = lo;                 // first sorted leaf of the range
    if (left >= 0) parent_int[left] = i; else parent_leaf[~left] = i;
    if (right >= 0) parent_int[right] = i; else parent_leaf[~right] = i;
    if (i == 0) parent_int[0] = -1;
}

// --- leaf boxes + bottom-up refit ---------------------------------------------
// box arrays: lo/hi as float4 (w of lo = 4-wide levels below the node's digit group, as int bits)
//
// Refit is level synchronous with work lists: a finished child bumps its parent's arrival counter
// (relaxed device atomic); the second arrival appends the parent to the NEXT launch's work list.
// Boxes are only read in a launch after the one that wrote them, so kernel boundaries give the
// ordering -- no fences on the data path.  (An in-kernel atomic-counter refit spent 97 % of the
// build in L2 write-backs on the 8-XCD part; dense per-level passes over all nodes cost
// O(nodes x height).)  Total work O(nodes), `height` launches.
__device__ __forceinline__ void refit_notify(int parent, int *__restrict__ arrivals, int *__restrict__ next_list,
                                             unsigned int *__restrict__ next_count) {
    if (parent < 0) return;
    if (atomicAdd(&arrivals[parent], 1) == 1) next_list[atomicAdd(next_count, 1u)] = parent;
}

// Orientation of a triangle's projection onto the world (x, y) plane: +1 counter-clockwise, -1 clockwise, 0 when the 2-D
// cross product is not larger than its own rounding (collapsed or edge-on projection: which way it faces is not known).
// Only the PROJECTION matters for the near-field certificates' distance bound (hz_common.h: HZ_BLOB_HEIGHT_FIELD) -- a
// triangle may be arbitrarily steep.  (Rounds 3-4 also counted |n_z| <= 1e-3 |n| as degenerate: one NoData sample of
// -32768 m in a 30 m grid switched the certificates off for the whole scene.)
__device__ __forceinline__ int tri_orient_xy(const float (&p0)[3], const float (&p1)[3], const float (&p2)[3]) {
    const float ux = p1[0] - p0[0], uy = p1[1] - p0[1], vx = p2[0] - p0[0], vy = p2[1] - p0[1];
    const float m0 = ux * vy, m1 = uy * vx, nz = m0 - m1;
    // the differences carry <= 2^-24 relative each (plus the vertices' own), the products and the difference one more
    // each: |error of nz| <= 4 * 2^-24 (|m0| + |m1|) to first order -- 1e-5 leaves a factor 40
    if (!(__builtin_fabsf(nz) > 1.0e-5f * (__builtin_fabsf(m0) + __builtin_fabsf(m1)))) return 0;
    return nz > 0.0f ? 1 : -1;
}

// HZ_BLOB_BAD_MAP: every DEM quad with a degenerate or minority-oriented triangle, and every TIN triangle, marks the
// bitmap cells its (x, y) footprint touches (bounding box + one cell all round: the look-up side computes its cell
// indices with the same expression, a float product that can land one cell off at a cell border; a TIN triangle, which
// can be kilometres long, only the cells of that box that no edge of the triangle separates from it).  A footprint of
// more than HZ_BAD_MAX_SPAN cells per axis raises `overflow` instead (no certificates for the scene).
#define HZ_BAD_MAX_SPAN 512
struct BadMap { uint32_t *bits; int nb; float x0, y0, sx, sy; };
__device__ __forceinline__ int bad_cell(float v, float v0, float s, int nb) {
    return min(max((int)__builtin_floorf((v - v0) * s), 0), nb - 1);
}
__global__ __launch_bounds__(256) void k_mark_bad(BuildParams b, int majority, BadMap m, unsigned int *__restrict__ overflow) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.n_prims) return;
    float a[3], bb[3], c[3], d[3];
    const bool quad = prim_vertices(b, p, a, bb, c, d);
    if (quad && tri_orient_xy(a, bb, c) == majority && tri_orient_xy(bb, d, c) == majority) return;
    const float xl = fminf(fminf(a[0], bb[0]), fminf(c[0], d[0])), xh = fmaxf(fmaxf(a[0], bb[0]), fmaxf(c[0], d[0]));
    const float yl = fminf(fminf(a[1], bb[1]), fminf(c[1], d[1])), yh = fmaxf(fmaxf(a[1], bb[1]), fmaxf(c[1], d[1]));
    const int i0 = max(bad_cell(xl, m.x0, m.sx, m.nb) - 1, 0), i1 = min(bad_cell(xh, m.x0, m.sx, m.nb) + 1, m.nb - 1);
    const int j0 = max(bad_cell(yl, m.y0, m.sy, m.nb) - 1, 0), j1 = min(bad_cell(yh, m.y0, m.sy, m.nb) + 1, m.nb - 1);
    if (i1 - i0 >= HZ_BAD_MAX_SPAN || j1 - j0 >= HZ_BAD_MAX_SPAN) { atomicAdd(overflow, 1u); return; }
    // TIN triangle: skip bitmap cells that lie strictly on the outer side of one of its edges (with two cells of margin);
    // a collapsed triangle has no usable edge normals: its whole box is marked
    float ex[3] = {0, 0, 0}, ey[3] = {0, 0, 0}, ec[3] = {0, 0, 0};
    bool cull = false;
    if (!quad) {
        const int o = tri_orient_xy(a, bb, c);
        if (o != 0) {
            const float px[3] = {a[0], bb[0], c[0]}, py[3] = {a[1], bb[1], c[1]};
            const float wx = 1.0f / m.sx, wy = 1.0f / m.sy;
            for (int k = 0; k < 3; k++) {
                const int k1 = (k + 1) % 3;
                // outward normal of edge k -> k1 (for a counter-clockwise triangle: (dy, -dx)); a point q is outside by
                // more than the margin when n . (q - p_k) > margin
                const float dx_ = px[k1] - px[k], dy_ = py[k1] - py[k];
                ex[k] = (float)o * dy_; ey[k] = -(float)o * dx_;
                ec[k] = ex[k] * px[k] + ey[k] * py[k] + 3.0f * (__builtin_fabsf(ex[k]) * wx + __builtin_fabsf(ey[k]) * wy);
            }
            cull = true;
        }
    }
    for (int j = j0; j <= j1; j++) {
        const float cy = m.y0 + ((float)j + 0.5f) / m.sy;
        for (int i = i0; i <= i1; i++) {
            if (cull) {
                const float cx = m.x0 + ((float)i + 0.5f) / m.sx;
                if (ex[0] * cx + ey[0] * cy > ec[0] || ex[1] * cx + ey[1] * cy > ec[1] || ex[2] * cx + ey[2] * cy > ec[2]) continue;
            }
            atomicOr(&m.bits[((size_t)j * m.nb + i) >> 5], 1u << (((size_t)j * m.nb + i) & 31));      // bit (j * nb + i) of the bitmap
        }
    }
}

__global__ __launch_bounds__(256) void k_leaf_boxes(BuildParams b, const uint32_t *__restrict__ vals,
                                                   const int *__restrict__ parent_leaf,
                                                   float4 *__restrict__ leaf_lo, float4 *__restrict__ leaf_hi,
                                                   int *__restrict__ arrivals, int *__restrict__ next_list,
                                                   unsigned int *__restrict__ next_count,
                                                   unsigned int *__restrict__ orient) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = s < b.n_prims;
    float a[3] = {0, 0, 0}, bb[3] = {0, 0, 0}, c[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    const bool quad = have && prim_vertices(b, (int)vals[s], a, bb, c, d);
    {   // height-field check (HZ_BLOB_HEIGHT_FIELD): orientation of the two DEM triangles (a, b, c) / (b, d, c) in the
        // world (x, y) plane; orient[0] counter-clockwise, [1] clockwise, [2] degenerate projection (tri_orient_xy)
        unsigned pos = 0, neg = 0, deg = 0;
        if (quad) {
            const int o0 = tri_orient_xy(a, bb, c), o1 = tri_orient_xy(bb, d, c);
            pos = (o0 > 0) + (o1 > 0); neg = (o0 < 0) + (o1 < 0); deg = (o0 == 0) + (o1 == 0);
        }
        for (int off = 32; off > 0; off >>= 1) { pos += __shfl_xor(pos, off); neg += __shfl_xor(neg, off); deg += __shfl_xor(deg, off); }
        if ((threadIdx.x & 63) == 0) {
            if (pos) atomicAdd(&orient[0], pos);
            if (neg) atomicAdd(&orient[1], neg);
            if (deg) atomicAdd(&orient[2], deg);
        }
    }
    if (!have) return;
    float lo[3], hi[3];
    const float ctr[3] = {b.cx, b.cy, b.cz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        lo[k] = (fminf(fminf(a[k], bb[k]), fminf(c[k], d[k])) - ctr[k]) - b.pad;
        hi[k] = (fmaxf(fmaxf(a[k], bb[k]), fmaxf(c[k], d[k])) - ctr[k]) + b.pad;
    }
    leaf_lo[s] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    leaf_hi[s] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    if (b.n_prims > 1) refit_notify(parent_leaf[s], arrivals, next_list, next_count);
}

__global__ __launch_bounds__(256) void k_refit_pass(const int *__restrict__ list, unsigned int n_list,
                                                   const int2 *__restrict__ child,
                                                   const int *__restrict__ parent_int,
                                                   const uint8_t *__restrict__ plen,
                                                   const float4 *__restrict__ leaf_lo,
                                                   const float4 *__restrict__ leaf_hi, float4 *node_lo,
                                                   float4 *node_hi, int *__restrict__ arrivals,
                                                   int *__restrict__ next_list, unsigned int *__restrict__ next_count) {
    const unsigned int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_list) return;
    const int i = list[q];
    const int2 ch = child[i];
    const float4 l0 = (ch.x >= 0) ? node_lo[ch.x] : leaf_lo[~ch.x];
    const float4 h0 = (ch.x >= 0) ? node_hi[ch.x] : leaf_hi[~ch.x];
    const float4 l1 = (ch.y >= 0) ? node_lo[ch.y] : leaf_lo[~ch.y];
    const float4 h1 = (ch.y >= 0) ? node_hi[ch.y] : leaf_hi[~ch.y];
    // 4-wide levels strictly below this node's digit group (see k_roots)
    const int dg = plen[i] >> 1;
    const int hgt0 = (ch.x >= 0) ? __float_as_int(l0.w) + (((plen[ch.x] >> 1) != dg) ? 1 : 0) : 0;
    const int hgt1 = (ch.y >= 0) ? __float_as_int(l1.w) + (((plen[ch.y] >> 1) != dg) ? 1 : 0) : 0;
    node_lo[i] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), __int_as_float(max(hgt0, hgt1)));
    node_hi[i] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
    refit_notify(parent_int[i], arrivals, next_list, next_count);
}

// --- collapse to 4-wide nodes --------------------------------------------------------
// A binary node opens a 4-wide node when it is the root or when its prefix length lies in
// another 2-bit digit than its parent's.  Inside one digit a subtree has at most 3 binary
// nodes and 4 exits; exit k goes to child slot = the digit value of its keys (2*ybit + xbit).
__global__ __launch_bounds__(256) void k_roots(int n_nodes, const int *__restrict__ parent_int,
                                              const uint8_t *__restrict__ plen, uint32_t *__restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int par = parent_int[i];
    flag[i] = (par < 0 || (plen[par] >> 1) != (plen[i] >> 1)) ? 1u : 0u;
}

// 4-wide node as k_emit4 produces it (children addressed by individual links, numbered in compaction order);
// the breadth-first pass below turns these into the final `Node`s with contiguous children
#define HZ_TMP_EMPTY ((int)0x80000000)   // "no child" among the temp links (>= 0 temp node, < 0 leaf = ~sorted position)
struct NodeTmp {
    float org[3];        // quantisation origins with the step exponents in their low mantissa bits (hz_common.h: Node)
    uint32_t pad_;
    uint32_t qx, qy, qz[2];   // as in Node
    int32_t link[4];     // per quadrant slot: >= 0 temp node, < 0 leaf (~sorted position), HZ_TMP_EMPTY none
};

struct Emit4 {
    const int2 *child; const uint8_t *plen; const int *first;
    const uint32_t *keys; const uint32_t *flag; const uint32_t *idx;
    const float4 *leaf_lo, *leaf_hi, *node_lo, *node_hi;
    int n_nodes;
};

__device__ __forceinline__ int digit_slot(uint32_t key, uint32_t pos, int dg) {
    const unsigned long long k64 = ((unsigned long long)key << 32) | pos;
    return (int)((k64 >> (62 - 2 * dg)) & 3ull);
}

// conservative quantisation of [lo, hi] against origin o and step s into [0, qmax]
__device__ __forceinline__ uint32_t quant_lo(float lo, float o, float s, float qmax) {
    float q = fminf(fmaxf(floorf((lo - o) / s), 0.0f), qmax);
    while (q > 0.0f && __builtin_fmaf(q, s, o) > lo) q -= 1.0f;
    return (uint32_t)q;
}
__device__ __forceinline__ uint32_t quant_hi(float hi, float o, float s, float qmax) {
    float q = fminf(fmaxf(ceilf((hi - o) / s), 0.0f), qmax);
    while (q < qmax && __builtin_fmaf(q, s, o) < hi) q += 1.0f;
    return (uint32_t)q;
}
// smallest power-of-two step with qmax * step >= extent; returns the biased exponent
__device__ __forceinline__ uint32_t step_exponent(float extent, float qmax) {
    const float need = fmaxf(extent / qmax, 1.0e-30f);
    int e;
    frexpf(need, &e);                 // need = m * 2^e, m in [0.5, 1)  ->  2^e >= need
    int biased = e + 127;
    biased = min(max(biased, 1), 254);
    while (biased < 254 && __uint_as_float((uint32_t)biased << 23) * qmax < extent) biased++;
    return (uint32_t)biased;
}

// Bounds (hz_common.h: hz_node_hits).  The traversal decodes code q in [0, 255] as the half float 1024 + q and
// evaluates t = (1024 + q) * (s rd) + (o' rd - oc rd) with the STORED origin o' ~ lo_node - 1024 s.  In exact
// arithmetic that is the plane X(q) = o' + (1024 + q) s; the build guarantees X(q_lo) <= lo - m and X(q_hi) >= hi + m,
// checked in float64 (exact for these operands).  m = s 2^-12 pays for what the 1024 s offset adds to the rounding
// of the two ray constants (b = fma(o', rd, -oc rd): |o'| grows by <= 1024 s, so its rounding by <= 1024 s 2^-24 =
// s 2^-14 in space units; the origin's own rounding is below that) -- everything else is the error
// structure the leaf padding has covered since round 1.
struct AxisQ { uint32_t e; float s, o, m; };
__device__ __forceinline__ double axis_plane(const AxisQ &a, float q) { return (double)a.o + (double)(1024.0f + q) * (double)a.s; }
// The step exponents travel in the low 9 mantissa bits of the node's x and z origins (bit 8 zero, bits 7..0 the biased
// exponent: `bits << 23` is the step).  embed_down: the largest float <= target with those low bits; embed_below: the next
// such float below f.  (Moves an origin down by at most 511 ulp: 4 m at |o| = 1e5 -- the quantisation below is done
// against the float that is actually stored, so this only costs a few of the 256 codes at the finest levels.)
__device__ __forceinline__ float embed_down(float target, uint32_t e) {
    if (!(__builtin_fabsf(target) > 1.0e-30f)) target = -1.0e-30f;
    const uint32_t b = __float_as_uint(target);
    if ((b >> 31) == 0u) {
        uint32_t c = (b & ~0x1ffu) | e;
        if (c > b) {
            if (c < 0x200u + 0x00800000u) return embed_down(-1.0e-30f, e);      // would leave the normal positives
            c -= 0x200u;
        }
        return __uint_as_float(c);
    }
    const uint32_t mag = b & 0x7fffffffu;
    uint32_t c = (mag & ~0x1ffu) | e;
    if (c < mag) c += 0x200u;
    return __uint_as_float(0x80000000u | c);
}
__device__ __forceinline__ float embed_below(float f, uint32_t e) {
    const uint32_t b = __float_as_uint(f);
    if ((b >> 31) == 0u) return (b >= 0x200u + 0x00800000u) ? __uint_as_float(b - 0x200u) : embed_down(-1.0e-30f, e);
    return __uint_as_float(b + 0x200u);
}
// x and y share one step.  embed = true (x): the origin carries the exponent bits.
__device__ __forceinline__ bool axis_try(AxisQ &a, float nl, float nh, uint32_t e, bool embed) {
    a.e = e;
    a.s = __uint_as_float(e << 23);
    a.m = a.s * (1.0f / 4096.0f);
    a.o = nl - 1024.0f * a.s;
    if (embed) a.o = embed_down(a.o, e);
    for (int g = 0; g < 16 && axis_plane(a, 0.0f) > (double)nl - (double)a.m; g++)
        a.o = embed ? embed_below(a.o, e) : nextafterf(a.o, -INFINITY);
    return axis_plane(a, 0.0f) <= (double)nl - (double)a.m && axis_plane(a, 255.0f) >= (double)nh + (double)a.m;
}
__device__ __forceinline__ void axes_setup(float xl, float xh, float yl, float yh, AxisQ &ax, AxisQ &ay) {
    uint32_t e = max(step_exponent(xh - xl, 250.0f), step_exponent(yh - yl, 250.0f));
    for (;;) {
        const bool okx = axis_try(ax, xl, xh, e, true), oky = axis_try(ay, yl, yh, e, false);
        if ((okx && oky) || e >= 254u) break;
        e++;
    }
}
// z: a step of its own, the exponent bits in the origin
__device__ __forceinline__ void axis_setup_z(float zl, float zh, AxisQ &az) {
    uint32_t e = step_exponent(zh - zl, 250.0f);
    for (;;) {
        if (axis_try(az, zl, zh, e, true) || e >= 254u) break;
        e++;
    }
}
__device__ __forceinline__ uint32_t axis_lo(const AxisQ &a, float lo) {
    const double want = (double)lo - (double)a.m;
    float q = fminf(fmaxf(floorf((float)((want - (double)a.o) / (double)a.s) - 1024.0f), 0.0f), 255.0f);
    while (q > 0.0f && axis_plane(a, q) > want) q -= 1.0f;
    while (q < 255.0f && axis_plane(a, q + 1.0f) <= want) q += 1.0f;
    return (uint32_t)q;
}
__device__ __forceinline__ uint32_t axis_hi(const AxisQ &a, float hi) {
    const double want = (double)hi + (double)a.m;
    float q = fminf(fmaxf(ceilf((float)((want - (double)a.o) / (double)a.s) - 1024.0f), 0.0f), 255.0f);
    while (q < 255.0f && axis_plane(a, q) < want) q += 1.0f;
    while (q > 0.0f && axis_plane(a, q - 1.0f) >= want) q -= 1.0f;
    return (uint32_t)q;
}
#define HZ_PAIR_EMPTY 0x00ffu         // (lo = 255, hi = 0): a range nothing can hit
// one (lo, hi) byte pair
__device__ __forceinline__ uint32_t axis_pair(const AxisQ &a, float lo, float hi) {
    return (lo <= hi) ? (axis_lo(a, lo) | (axis_hi(a, hi) << 8)) : HZ_PAIR_EMPTY;
}

__global__ __launch_bounds__(256) void k_emit4(Emit4 e, NodeTmp *__restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n_nodes || !e.flag[i]) return;
    const int dg = e.plen[i] >> 1;
    // gather the (<= 4) exits of the digit group rooted at i
    int link[4]; float lo[4][3], hi[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        link[k] = HZ_TMP_EMPTY;
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[k][a] = 0.0f; hi[k][a] = 0.0f; }
    }
    int pend[4]; int np = 0;
    { const int2 ch = e.child[i]; pend[np++] = ch.x; pend[np++] = ch.y; }
    while (np > 0) {
        const int c = pend[--np];
        if (c >= 0 && (e.plen[c] >> 1) == dg) {       // same digit: expand (its children leave the digit)
            const int2 ch = e.child[c];
            pend[np++] = ch.x; pend[np++] = ch.y;
            continue;
        }
        int slot; float4 bl, bh; int lk;
        if (c < 0) {
            const int sidx = ~c;
            slot = digit_slot(e.keys[sidx], (uint32_t)sidx, dg);
            bl = e.leaf_lo[sidx]; bh = e.leaf_hi[sidx]; lk = c;
        } else {
            const int f = e.first[c];
            slot = digit_slot(e.keys[f], (uint32_t)f, dg);
            bl = e.node_lo[c]; bh = e.node_hi[c]; lk = (int)e.idx[c];
        }
        // slots are distinct by construction; dynamic indexing is fine here (build-time kernel)
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k == slot) { link[k] = lk; lo[k][0] = bl.x; lo[k][1] = bl.y; lo[k][2] = bl.z;
                             hi[k][0] = bh.x; hi[k][1] = bh.y; hi[k][2] = bh.z; }
    }
    // The children stay in quadrant order (slot = 2 * row bit + column bit of this node's Morton digit): the two children
    // of a column half share ONE x range, the two of a row half ONE y range -- what the 32 B node holds (hz_common.h).  For
    // a DEM whose grid is aligned with the axes these are the children's own ranges; for rotated / curved grids the union
    // over the half.  Measured against per-child ranges + 11-bit z + tallest-first order (48 B nodes): +0.9 % node visits,
    // +1.8 % leaf visits (profiles/r04/probe_node32_bounds_and_wg6.log), for one 16 B load less per visit.
    float hl[2][2], hh[2][2];      // [axis][half]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int h = 0; h < 2; h++) { hl[a][h] = INFINITY; hh[a][h] = -INFINITY; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (link[k] == HZ_TMP_EMPTY) continue;
        const int hx = k & 1, hy = k >> 1;
        hl[0][hx] = fminf(hl[0][hx], lo[k][0]); hh[0][hx] = fmaxf(hh[0][hx], hi[k][0]);
        hl[1][hy] = fminf(hl[1][hy], lo[k][1]); hh[1][hy] = fmaxf(hh[1][hy], hi[k][1]);
    }
    const float4 nl = e.node_lo[i], nh = e.node_hi[i];
    AxisQ ax, ay, az;
    axes_setup(nl.x, nh.x, nl.y, nh.y, ax, ay);
    axis_setup_z(nl.z, nh.z, az);
    NodeTmp n;
    n.org[0] = ax.o; n.org[1] = ay.o; n.org[2] = az.o;
    n.pad_ = 0u;
    n.qx = axis_pair(ax, hl[0][0], hh[0][0]) | (axis_pair(ax, hl[0][1], hh[0][1]) << 16);
    n.qy = axis_pair(ay, hl[1][0], hh[1][0]) | (axis_pair(ay, hl[1][1], hh[1][1]) << 16);
    uint32_t zq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        n.link[k] = link[k];
        zq[k] = (link[k] == HZ_TMP_EMPTY) ? HZ_PAIR_EMPTY : axis_pair(az, lo[k][2], hi[k][2]);
    }
    n.qz[0] = zq[0] | (zq[1] << 16);
    n.qz[1] = zq[2] | (zq[3] << 16);
    nodes[e.idx[i]] = n;
}

// --- breadth-first numbering with contiguous children ---------------------------------------------
// The temp nodes are renumbered level by level.  A frontier holds, for every node slot of a level, what sits
// there: >= 0 a temp node, <= -2 a single leaf that needs a wrapper node (sorted position -2 - v), -1 nothing.
// Every entry asks for one block of 4 child slots: a NODE block when the temp node has internal children (its
// leaf children are then wrapped: they go to the next frontier as <= -2 entries), else a LEAF block.  Blocks are
// handed out by an exclusive scan over the frontier, so children are contiguous, siblings keep their slot order
// (tallest first) and the numbering is breadth first.
__device__ __forceinline__ void bfs_kinds(const NodeTmp &t, int &n_int, int &n_leaf) {
    n_int = 0; n_leaf = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (t.link[k] >= 0) n_int++;
        else if (t.link[k] != HZ_TMP_EMPTY) n_leaf++;
    }
}

// exact sizes of the final arrays: node blocks = temp nodes with an internal child (+ the root's own block),
// leaf blocks = temp nodes without one + one per wrapped leaf
__global__ __launch_bounds__(256) void k_bfs_sizes(int n4, const NodeTmp *__restrict__ tmp, unsigned int *__restrict__ sizes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nb = 0, lb = 0;
    if (i < n4) {
        int ni, nl;
        bfs_kinds(tmp[i], ni, nl);
        if (ni > 0) { nb = 1; lb = (unsigned)nl; } else lb = 1;
    }
    for (int off = 32; off > 0; off >>= 1) { nb += __shfl_xor(nb, off); lb += __shfl_xor(lb, off); }
    if ((threadIdx.x & 63) == 0) { if (nb) atomicAdd(&sizes[0], nb); if (lb) atomicAdd(&sizes[1], lb); }
}

__global__ __launch_bounds__(256) void k_bfs_need(int cnt, const int *__restrict__ frontier, const NodeTmp *__restrict__ tmp,
                                                 uint32_t *__restrict__ need_node, uint32_t *__restrict__ need_leaf) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= cnt) return;
    const int v = frontier[pos];
    uint32_t nn = 0, nl = 0;
    if (v >= 0) { int ni, nlf; bfs_kinds(tmp[v], ni, nlf); if (ni > 0) nn = 1; else nl = 1; }
    else if (v <= -2) nl = 1;
    need_node[pos] = nn; need_leaf[pos] = nl;
}

struct BfsEmit {
    const int *frontier; int cnt, level_start;         // node slots [level_start, level_start + cnt)
    const NodeTmp *tmp;
    const uint32_t *scan_node, *scan_leaf;             // exclusive scans of the needs over this frontier
    int node_blocks_before, leaf_blocks_before;        // blocks handed out on the levels above
    int *next_frontier;                                // 4 entries per node block of this level
    Node *nodes; Prim *prims; int *parent; int *leaf_parent;   // parent[node slot], leaf_parent[leaf block]
    const uint32_t *vals;                              // sorted position -> primitive id
};

__device__ __forceinline__ void write_prim(const BuildParams &b, const uint32_t *__restrict__ vals, int sorted_pos, Prim *dst) {
    float a[3], bb[3], c[3], d[3];
    const bool quad = prim_vertices(b, (int)vals[sorted_pos], a, bb, c, d);
    Prim p;
#pragma unroll
    for (int k = 0; k < 3; k++) { p.a[k] = a[k]; p.b[k] = bb[k]; p.c[k] = c[k]; p.d[k] = d[k]; }
    if (!quad) p.d[0] = __int_as_float(0x7fc00000);   // NaN: TIN triangle, single test
    *dst = p;
}

__global__ __launch_bounds__(256) void k_bfs_emit(BfsEmit e, BuildParams b) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= e.cnt) return;
    const int v = e.frontier[pos];
    if (v == -1) return;                                  // unused slot of a block (a dead node: k_fill_dead_nodes)
    const int self = e.level_start + pos;
    if (v <= -2) {                                        // wrapper node (header written by its parent): its one leaf
        const int lb = e.leaf_blocks_before + (int)e.scan_leaf[pos];
        e.nodes[self].first = (int)(HZ_LEAF_BIT | (unsigned)(4 * lb));
        write_prim(b, e.vals, -2 - v, &e.prims[(size_t)4 * lb]);
        e.leaf_parent[lb] = self;
        return;
    }
    const NodeTmp t = e.tmp[v];
    int ni, nl;
    bfs_kinds(t, ni, nl);
    Node n;
    n.org[0] = t.org[0]; n.org[1] = t.org[1]; n.org[2] = t.org[2];
    n.qx = t.qx; n.qy = t.qy; n.qz[0] = t.qz[0]; n.qz[1] = t.qz[1];
    if (ni == 0) {                                        // all children are leaves: one leaf block
        const int lb = e.leaf_blocks_before + (int)e.scan_leaf[pos];
        n.first = (int)(HZ_LEAF_BIT | (unsigned)(4 * lb));
#pragma unroll
        for (int k = 0; k < 4; k++) if (t.link[k] != HZ_TMP_EMPTY) write_prim(b, e.vals, ~t.link[k], &e.prims[(size_t)4 * lb + k]);
        e.leaf_parent[lb] = self;
    } else {                                              // a node block; leaf children get wrapper nodes
        const int rel = (int)e.scan_node[pos];
        const int first = 4 * (e.node_blocks_before + rel);
        n.first = first;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int f = -1;
            if (t.link[k] >= 0) f = t.link[k];
            else if (t.link[k] != HZ_TMP_EMPTY) {
                f = -2 - (~t.link[k]);
                Node w;                                    // single-child node: this slot's box in this node's frame,
                w.org[0] = n.org[0]; w.org[1] = n.org[1]; w.org[2] = n.org[2];     // moved to slot 0 of the wrapper
                w.first = 0;
                w.qx = ((n.qx >> (16 * (k & 1))) & 0xffffu) | (HZ_PAIR_EMPTY << 16);
                w.qy = ((n.qy >> (16 * (k >> 1))) & 0xffffu) | (HZ_PAIR_EMPTY << 16);
                w.qz[0] = ((n.qz[k >> 1] >> (16 * (k & 1))) & 0xffffu) | (HZ_PAIR_EMPTY << 16);
                w.qz[1] = HZ_PAIR_EMPTY | (HZ_PAIR_EMPTY << 16);
                e.nodes[first + k] = w;
            }
            e.next_frontier[4 * rel + k] = f;
            if (f != -1) e.parent[first + k] = self;
        }
    }
    e.nodes[self] = n;
}

// Unused node slots (the three spare slots of the root's block, absent children of node blocks) are never the target of
// a link whose box can be hit -- the parent marks an absent child with an inverted z pair.  The box test has a relative
// slack (tmin <= tmax * (1 + 1e-6)), though, so for a ray that starts several scene diagonals outside the scene
// (hz_horizon_locations accepts any coordinates) an inverted pair of a very flat node CAN pass.  Round 4 left these slots
// zero: a zero node is a point box at the scene centre whose children are block 0 -- the root -- so a ray through that
// point would walk the tree again and again.  Every slot now starts as a DEAD node: all ranges inverted by 255 steps of
// 1 m on every axis (passes only for |t| > 2.5e8) and a link to leaf block 0 -- real triangles, tested with the real test:
// harmless for any-hit and closest-hit alike.  The breadth-first pass overwrites the slots that are in use.
__global__ __launch_bounds__(256) void k_fill_dead_nodes(Node *__restrict__ nodes, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Node d;
    d.org[0] = __uint_as_float(127u); d.org[1] = 0.0f; d.org[2] = __uint_as_float(127u);    // steps 2^0 in the low bits
    d.first = (int)HZ_LEAF_BIT;
    d.qx = d.qy = d.qz[0] = d.qz[1] = HZ_PAIR_EMPTY | (HZ_PAIR_EMPTY << 16);
    nodes[i] = d;
}

// hit-cache ancestors: for every leaf slot the node `levels` levels above it (the leaf block's parent counts as 1)
__global__ __launch_bounds__(256) void k_anc_bfs(int n_leaf_blocks, int levels, const int *__restrict__ parent,
                                                const int *__restrict__ leaf_parent, int *__restrict__ anc) {
    const int lb = blockIdx.x * blockDim.x + threadIdx.x;
    if (lb >= n_leaf_blocks) return;
    int n = leaf_parent[lb];
    for (int r = 1; r < levels; r++) {
        const int p = parent[n];
        if (p < 0) break;
        n = p;
    }
    reinterpret_cast<int4 *>(anc)[lb] = make_int4(n, n, n, n);
}

// single primitive: a temp root whose slot 0 is the leaf, quantised against its own box
__global__ void k_single_tmp(const float4 *leaf_lo, const float4 *leaf_hi, NodeTmp *nodes) {
    const float4 l = leaf_lo[0], h = leaf_hi[0];
    AxisQ ax, ay, az;
    axes_setup(l.x, h.x, l.y, h.y, ax, ay);
    axis_setup_z(l.z, h.z, az);
    NodeTmp n;
    n.org[0] = ax.o; n.org[1] = ay.o; n.org[2] = az.o;
    n.pad_ = 0u;
    for (int k = 0; k < 4; k++) n.link[k] = HZ_TMP_EMPTY;
    n.link[0] = ~0;
    n.qx = axis_pair(ax, l.x, h.x) | (HZ_PAIR_EMPTY << 16);
    n.qy = axis_pair(ay, l.y, h.y) | (HZ_PAIR_EMPTY << 16);
    n.qz[0] = axis_pair(az, l.z, h.z) | (HZ_PAIR_EMPTY << 16);
    n.qz[1] = HZ_PAIR_EMPTY | (HZ_PAIR_EMPTY << 16);
    nodes[0] = n;
}

// ------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Build temporaries come out of (at most) two device arenas: ~25 separate multi-GB hipMalloc /
// hipFree pairs made the build time of large scenes erratic (0.1 ... 1.9 s for the same input).
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    ~Arena() { if (base) (void)hipFree(base); }
    hipError_t reserve(size_t bytes) { cap = bytes; off = 0; return hipMalloc((void **)&base, bytes ? bytes : 256); }
    static size_t pad(size_t n) { return (n + 255) & ~(size_t)255; }
};
static thread_local Arena *g_arena = nullptr;

struct TempBuf {
    void *p = nullptr;
    bool owned = false;
    ~TempBuf() { if (p && owned) (void)hipFree(p); }
    hipError_t alloc(size_t n) {
        n = n ? n : 16;
        if (g_arena && g_arena->off + Arena::pad(n) <= g_arena->cap) {
            p = g_arena->base + g_arena->off; g_arena->off += Arena::pad(n); owned = false;
            return hipSuccess;
        }
        owned = true;
        return hipMalloc(&p, n);
    }
};
struct ArenaScope {   // the arena serves TempBuf::alloc while in scope
    explicit ArenaScope(Arena *a) { g_arena = a; }
    ~ArenaScope() { g_arena = nullptr; }
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
    const int n_bin = (n_prims > 1) ? n_prims - 1 : 1;      // binary radix tree nodes
    hipStream_t st = sc->stream;

    // ---- one arena for all temporaries whose size is known up front ------------------------
    Arena arena1, arena2, arena3;
    {
        const size_t P = (size_t)n_prims, B = (size_t)n_bin;
        const size_t list_cap0 = B / 2 + 2;
        const size_t sizes[] = {is_device_ptr(vert_grid) ? 0 : nvert * 12, has_tin ? (size_t)nvs * 12 : 0,
                                has_tin ? (size_t)nts * 12 : 0, 24, P * 4, P * 4, P * 4, P * 4,
                                sort_temp_elems(P) * 4, B * 8, B * 4, P * 4, B, B * 4, P * 16, P * 16, B * 16, B * 16,
                                (B + 2 * list_cap0 + 8) * 4, B * 4, B * 4, scan_temp_elems(B) * 4};
        size_t total = 0;
        for (size_t x : sizes) total += Arena::pad(x ? x : 16);
        HZ_HIP(arena1.reserve(total + 4096));
    }
    ArenaScope scope1(&arena1);

    // ---- vertices on the device (host or device source) -------------------------------
    Timer t_h2d; t_h2d.start();
    TempBuf b_verts, b_vs, b_ts;
    const float *d_verts_src = vert_grid;
    if (!is_device_ptr(vert_grid)) {
        HZ_HIP(b_verts.alloc(nvert * 12));
        HZ_HIP(hipMemcpyAsync(b_verts.p, vert_grid, nvert * 12, hipMemcpyHostToDevice, st));
        d_verts_src = (const float *)b_verts.p;
    }
    if (has_tin) {
        HZ_HIP(b_vs.alloc((size_t)nvs * 12));
        HZ_HIP(b_ts.alloc((size_t)nts * 12));
        HZ_HIP(hipMemcpyAsync(b_vs.p, vert_simp, (size_t)nvs * 12, hipMemcpyDefault, st));
        HZ_HIP(hipMemcpyAsync(b_ts.p, tri_simp, (size_t)nts * 12, hipMemcpyDefault, st));
    }
    HZ_HIP(hipStreamSynchronize(st));
    const double h2d_s = t_h2d.stop();

    Timer t_bvh; t_bvh.start();
    BlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = HZ_BLOB_MAGIC; h.version = HZ_BLOB_VERSION;
    h.d0 = d0; h.d1 = d1; h.n_quads = n_quads; h.n_tin = n_tin; h.n_prims = n_prims;

    // ---- 1. bounds ----------------------------------------------------------------
    TempBuf b_bounds;
    HZ_HIP(b_bounds.alloc(6 * 4));
    uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HZ_HIP(hipMemcpyAsync(b_bounds.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    {
        const int grid = (int)std::min<size_t>((nvert + 255) / 256, 2048);
        hipLaunchKernelGGL(k_bounds, dim3(grid), dim3(256), 0, st, d_verts_src, nvert, (uint32_t *)b_bounds.p);
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
    bp.verts = d_verts_src; bp.vs = (const float *)b_vs.p; bp.ts = (const int32_t *)b_ts.p;
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
    const int gn = (n_bin + 255) / 256;
    hipLaunchKernelGGL(k_morton, dim3(gp), dim3(256), 0, st, bp, (uint32_t *)b_k0.p, (uint32_t *)b_v0.p);
    HZ_HIP(b_sort.alloc(sort_temp_elems((size_t)n_prims) * 4));
    {   // sorted pairs come back in (k0, v0); (k1, v1) are scratch
        const int rc = radix_sort_pairs_u32((uint32_t *)b_k0.p, (uint32_t *)b_v0.p, (uint32_t *)b_k1.p,
                                            (uint32_t *)b_v1.p, (size_t)n_prims, (uint32_t *)b_sort.p, st);
        if (rc) return rc;
    }
    const uint32_t *keys = (const uint32_t *)b_k0.p;
    const uint32_t *vals = (const uint32_t *)b_v0.p;

    // ---- 4./5. binary hierarchy + refit -------------------------------------------------
    TempBuf b_child, b_pint, b_pleaf, b_plen, b_first, b_llo, b_lhi, b_nlo, b_nhi, b_cnt;
    HZ_HIP(b_child.alloc((size_t)n_bin * 8));
    HZ_HIP(b_pint.alloc((size_t)n_bin * 4));
    HZ_HIP(b_pleaf.alloc((size_t)n_prims * 4));
    HZ_HIP(b_plen.alloc((size_t)n_bin));
    HZ_HIP(b_first.alloc((size_t)n_bin * 4));
    HZ_HIP(b_llo.alloc((size_t)n_prims * 16)); HZ_HIP(b_lhi.alloc((size_t)n_prims * 16));
    HZ_HIP(b_nlo.alloc((size_t)n_bin * 16)); HZ_HIP(b_nhi.alloc((size_t)n_bin * 16));
    // arrivals[n_bin] | two work lists [n_bin/2 + 1 each, a node enters a list once] | 2 list counters
    const size_t list_cap = (size_t)n_bin / 2 + 2;
    HZ_HIP(b_cnt.alloc(((size_t)n_bin + 2 * list_cap + 8) * 4));
    int *arrivals = (int *)b_cnt.p;
    int *lists[2] = {arrivals + n_bin, arrivals + n_bin + list_cap};
    unsigned int *counts = (unsigned int *)(arrivals + n_bin + 2 * list_cap);
    unsigned int *orient = counts + 4;                   // height-field check of k_leaf_boxes
    HZ_HIP(hipMemsetAsync(arrivals, 0, (size_t)n_bin * 4, st));
    HZ_HIP(hipMemsetAsync(counts, 0, 32, st));
    HZ_HIP(hipMemsetAsync(b_nlo.p, 0, (size_t)n_bin * 16, st));
    if (n_prims > 1)
        hipLaunchKernelGGL(k_karras, dim3((n_prims - 1 + 255) / 256), dim3(256), 0, st, keys, n_prims,
                           (int2 *)b_child.p, (int *)b_pint.p, (int *)b_pleaf.p, (uint8_t *)b_plen.p,
                           (int *)b_first.p);
    hipLaunchKernelGGL(k_leaf_boxes, dim3(gp), dim3(256), 0, st, bp, vals, (const int *)b_pleaf.p,
                       (float4 *)b_llo.p, (float4 *)b_lhi.p, arrivals, lists[0], &counts[0], orient);
    if (n_prims > 1) {
        size_t finished = 0;
        int cur = 0;
        for (int pass = 0; pass < 200; pass++) {
            unsigned int n_list = 0;
            HZ_HIP(hipMemcpyAsync(&n_list, &counts[cur], 4, hipMemcpyDeviceToHost, st));
            HZ_HIP(hipStreamSynchronize(st));
            if (n_list == 0) break;
            if (n_list > list_cap) return set_error(HZ_ERR_HIP, "BVH refit work list overflow");
            HZ_HIP(hipMemsetAsync(&counts[cur ^ 1], 0, 4, st));
            hipLaunchKernelGGL(k_refit_pass, dim3((n_list + 255) / 256), dim3(256), 0, st, (const int *)lists[cur],
                               n_list, (const int2 *)b_child.p, (const int *)b_pint.p, (const uint8_t *)b_plen.p,
                               (const float4 *)b_llo.p, (const float4 *)b_lhi.p, (float4 *)b_nlo.p,
                               (float4 *)b_nhi.p, arrivals, lists[cur ^ 1], &counts[cur ^ 1]);
            finished += n_list;
            cur ^= 1;
        }
        if (finished != (size_t)n_bin)
            return set_error(HZ_ERR_DEPTH, "BVH refit did not converge (%zu of %d nodes)", finished, n_bin);
    }

    int majority = 1;
    bool want_bad_map = false;
    {   // height field over the world (x, y) plane?  (hz_common.h: HZ_BLOB_HEIGHT_FIELD / HZ_BLOB_BAD_MAP)
        unsigned int oc[3] = {0, 0, 0};
        HZ_HIP(hipMemcpyAsync(oc, orient, sizeof(oc), hipMemcpyDeviceToHost, st));
        HZ_HIP(hipStreamSynchronize(st));
        const unsigned int minority = std::min(oc[0], oc[1]);
        majority = (oc[0] >= oc[1]) ? 1 : -1;
        h.n_flipped = minority + oc[2];
        h.flags = (h.n_flipped == 0 && n_quads > 0 && n_tin == 0) ? HZ_BLOB_HEIGHT_FIELD : 0u;
        want_bad_map = h.flags == 0u && n_quads > 0;
    }

    // ---- 6. which binary nodes open a 4-wide node; compact indices ------------------------
    int n4 = 1;
    TempBuf b_flag, b_idx, b_scan;
    if (n_prims > 1) {
        HZ_HIP(b_flag.alloc((size_t)n_bin * 4));
        HZ_HIP(b_idx.alloc((size_t)n_bin * 4));
        hipLaunchKernelGGL(k_roots, dim3(gn), dim3(256), 0, st, n_bin, (const int *)b_pint.p,
                           (const uint8_t *)b_plen.p, (uint32_t *)b_flag.p);
        HZ_HIP(b_scan.alloc(scan_temp_elems((size_t)n_bin) * 4));
        {
            const int rc = exclusive_scan_u32((const uint32_t *)b_flag.p, (uint32_t *)b_idx.p, (size_t)n_bin,
                                              (uint32_t *)b_scan.p, st);
            if (rc) return rc;
        }
        uint32_t last_idx = 0, last_flag = 0;
        HZ_HIP(hipMemcpyAsync(&last_idx, (uint32_t *)b_idx.p + (n_bin - 1), 4, hipMemcpyDeviceToHost, st));
        HZ_HIP(hipMemcpyAsync(&last_flag, (uint32_t *)b_flag.p + (n_bin - 1), 4, hipMemcpyDeviceToHost, st));
        HZ_HIP(hipStreamSynchronize(st));
        n4 = (int)(last_idx + last_flag);
    }

    // ---- 7. 4-wide temp nodes (compaction order, individual child links) -------------------------------------
    TempBuf b_tmp4, b_sizes;
    g_arena = nullptr;                                   // what follows is sized by n4: own allocations / arena2
    HZ_HIP(arena2.reserve(Arena::pad((size_t)n4 * sizeof(NodeTmp)) + 64 * 4096));
    g_arena = &arena2;
    HZ_HIP(b_tmp4.alloc((size_t)n4 * sizeof(NodeTmp)));
    if (n_prims > 1) {
        Emit4 e;
        e.child = (const int2 *)b_child.p; e.plen = (const uint8_t *)b_plen.p; e.first = (const int *)b_first.p;
        e.keys = keys; e.flag = (const uint32_t *)b_flag.p; e.idx = (const uint32_t *)b_idx.p;
        e.leaf_lo = (const float4 *)b_llo.p; e.leaf_hi = (const float4 *)b_lhi.p;
        e.node_lo = (const float4 *)b_nlo.p; e.node_hi = (const float4 *)b_nhi.p;
        e.n_nodes = n_bin;
        hipLaunchKernelGGL(k_emit4, dim3(gn), dim3(256), 0, st, e, (NodeTmp *)b_tmp4.p);
    } else {
        hipLaunchKernelGGL(k_single_tmp, dim3(1), dim3(1), 0, st, (const float4 *)b_llo.p, (const float4 *)b_lhi.p,
                           (NodeTmp *)b_tmp4.p);
    }
    // exact sizes of the final arrays
    HZ_HIP(b_sizes.alloc(16));
    HZ_HIP(hipMemsetAsync(b_sizes.p, 0, 16, st));
    hipLaunchKernelGGL(k_bfs_sizes, dim3((n4 + 255) / 256), dim3(256), 0, st, n4, (const NodeTmp *)b_tmp4.p,
                       (unsigned int *)b_sizes.p);
    unsigned int sizes[2] = {0, 0};
    HZ_HIP(hipMemcpyAsync(sizes, b_sizes.p, 8, hipMemcpyDeviceToHost, st));
    HZ_HIP(hipStreamSynchronize(st));
    const size_t n_node_blocks = (size_t)sizes[0] + 1;                 // + the root's own block
    const size_t n_leaf_blocks = sizes[1];
    const size_t n_nodes = 4 * n_node_blocks, n_prim_slots = 4 * n_leaf_blocks;
    if (n_nodes > 0x3ffffff0u || n_prim_slots > 0x3ffffff0u)     // links use 30 bits (hz_common.h: stack entries)
        return set_error(HZ_ERR_DEPTH, "scene too large for the 32-bit traversal links");

    // ---- blob allocation (exact size now known) ------------------------------------------------
    h.n_nodes = (int32_t)n_nodes;
    h.n_prim_slots = (int32_t)n_prim_slots;
    h.off_verts = sizeof(BlobHeader);
    h.off_nodes = align_up(h.off_verts + nvert * 12, 256);
    h.off_prims = align_up(h.off_nodes + n_nodes * sizeof(Node), 256);
    h.off_anc = align_up(h.off_prims + n_prim_slots * sizeof(Prim), 256);
    h.anc_levels = HZ_ANC_LEVELS;
    h.total_bytes = align_up(h.off_anc + n_prim_slots * 4, 256);
    if (want_bad_map) {
        // bitmap cells of about two DEM cells: a bad quad then costs the certificates of a handful of cells around it
        int nb = 8;
        while (nb < 2048 && 2 * nb < std::max(d0, d1)) nb *= 2;
        h.bad_nb = nb;
        h.bad_x0 = lo[0]; h.bad_y0 = lo[1];
        h.bad_sx = (hi[0] > lo[0]) ? (float)nb / (hi[0] - lo[0]) : 0.0f;
        h.bad_sy = (hi[1] > lo[1]) ? (float)nb / (hi[1] - lo[1]) : 0.0f;
        h.off_bad = h.total_bytes;
        h.total_bytes = align_up(h.off_bad + (size_t)nb * nb / 8, 256);
    }
    HZ_HIP(hipMalloc(&sc->blob, h.total_bytes));
    sc->owns_blob = true;
    sc->blob_bytes = h.total_bytes;
    char *blob = (char *)sc->blob;
    float *d_verts = (float *)(blob + h.off_verts);
    Node *d_nodes = (Node *)(blob + h.off_nodes);
    Prim *d_prims = (Prim *)(blob + h.off_prims);
    int *d_anc = (int *)(blob + h.off_anc);
    HZ_HIP(hipMemcpyAsync(d_verts, d_verts_src, nvert * 12, hipMemcpyDeviceToDevice, st));
    // unused leaf slots stay zero (a collapsed triangle: den = 0, never a hit); unused node slots: k_fill_dead_nodes
    HZ_HIP(hipMemsetAsync(blob + h.off_nodes, 0, h.total_bytes - h.off_nodes, st));
    hipLaunchKernelGGL(k_fill_dead_nodes, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, st, d_nodes, n_nodes);
    if (want_bad_map && h.bad_sx > 0.0f && h.bad_sy > 0.0f) {
        TempBuf b_ovf;
        HZ_HIP(b_ovf.alloc(16));
        HZ_HIP(hipMemsetAsync(b_ovf.p, 0, 4, st));
        BadMap bm;
        bm.bits = (uint32_t *)(blob + h.off_bad); bm.nb = h.bad_nb;
        bm.x0 = h.bad_x0; bm.y0 = h.bad_y0; bm.sx = h.bad_sx; bm.sy = h.bad_sy;
        hipLaunchKernelGGL(k_mark_bad, dim3(gp), dim3(256), 0, st, bp, majority, bm, (unsigned int *)b_ovf.p);
        unsigned int ovf = 0;
        HZ_HIP(hipMemcpyAsync(&ovf, b_ovf.p, 4, hipMemcpyDeviceToHost, st));
        HZ_HIP(hipStreamSynchronize(st));
        if (ovf == 0) h.flags |= HZ_BLOB_BAD_MAP;
    }

    // ---- 8. breadth-first numbering, level by level ---------------------------------------------------------
    TempBuf b_fr0, b_fr1, b_need_n, b_need_l, b_scan_n, b_scan_l, b_scan_tmp, b_parent, b_lparent;
    HZ_HIP(arena3.reserve(7 * Arena::pad(n_nodes * 4) + Arena::pad(scan_temp_elems(n_nodes) * 4 + 16) +
                          Arena::pad((n_leaf_blocks ? n_leaf_blocks : 1) * 4) + 4096));
    g_arena = &arena3;
    HZ_HIP(b_fr0.alloc(n_nodes * 4)); HZ_HIP(b_fr1.alloc(n_nodes * 4));
    HZ_HIP(b_need_n.alloc(n_nodes * 4)); HZ_HIP(b_need_l.alloc(n_nodes * 4));
    HZ_HIP(b_scan_n.alloc(n_nodes * 4)); HZ_HIP(b_scan_l.alloc(n_nodes * 4));
    HZ_HIP(b_scan_tmp.alloc(scan_temp_elems(n_nodes) * 4 + 16));
    HZ_HIP(b_parent.alloc(n_nodes * 4)); HZ_HIP(b_lparent.alloc((n_leaf_blocks ? n_leaf_blocks : 1) * 4));
    HZ_HIP(hipMemsetAsync(b_parent.p, 0xff, n_nodes * 4, st));       // -1: the root has no parent
    int *fr[2] = {(int *)b_fr0.p, (int *)b_fr1.p};
    {   // (static: the source of an asynchronous copy must outlive the enqueue)
        static const int first_frontier[4] = {0, -1, -1, -1};          // block 0: the root and three unused slots
        HZ_HIP(hipMemcpyAsync(fr[0], first_frontier, sizeof(first_frontier), hipMemcpyHostToDevice, st));
    }
    int cur = 0, cnt = 4, level_start = 0, levels = 0;
    size_t node_blocks_done = 1, leaf_blocks_done = 0;
    while (cnt > 0) {
        levels++;
        if (levels > HZ_MAX_STACK) return set_error(HZ_ERR_DEPTH, "BVH deeper than %d levels", HZ_MAX_STACK);
        const int g = (cnt + 255) / 256;
        hipLaunchKernelGGL(k_bfs_need, dim3(g), dim3(256), 0, st, cnt, (const int *)fr[cur], (const NodeTmp *)b_tmp4.p,
                           (uint32_t *)b_need_n.p, (uint32_t *)b_need_l.p);
        int rc = exclusive_scan_u32((const uint32_t *)b_need_n.p, (uint32_t *)b_scan_n.p, (size_t)cnt, (uint32_t *)b_scan_tmp.p, st);
        if (rc) return rc;
        rc = exclusive_scan_u32((const uint32_t *)b_need_l.p, (uint32_t *)b_scan_l.p, (size_t)cnt, (uint32_t *)b_scan_tmp.p, st);
        if (rc) return rc;
        uint32_t last[4] = {0, 0, 0, 0};                               // scan and need of the last entry, both kinds
        HZ_HIP(hipMemcpyAsync(&last[0], (uint32_t *)b_scan_n.p + (cnt - 1), 4, hipMemcpyDeviceToHost, st));
        HZ_HIP(hipMemcpyAsync(&last[1], (uint32_t *)b_need_n.p + (cnt - 1), 4, hipMemcpyDeviceToHost, st));
        HZ_HIP(hipMemcpyAsync(&last[2], (uint32_t *)b_scan_l.p + (cnt - 1), 4, hipMemcpyDeviceToHost, st));
        HZ_HIP(hipMemcpyAsync(&last[3], (uint32_t *)b_need_l.p + (cnt - 1), 4, hipMemcpyDeviceToHost, st));
        HZ_HIP(hipStreamSynchronize(st));
        const size_t nb = (size_t)last[0] + last[1], lb = (size_t)last[2] + last[3];
        if (node_blocks_done + nb > n_node_blocks || leaf_blocks_done + lb > n_leaf_blocks)
            return set_error(HZ_ERR_HIP, "BVH numbering ran past its arrays (internal error)");
        BfsEmit e;
        e.frontier = fr[cur]; e.cnt = cnt; e.level_start = level_start;
        e.tmp = (const NodeTmp *)b_tmp4.p;
        e.scan_node = (const uint32_t *)b_scan_n.p; e.scan_leaf = (const uint32_t *)b_scan_l.p;
        e.node_blocks_before = (int)node_blocks_done; e.leaf_blocks_before = (int)leaf_blocks_done;
        e.next_frontier = fr[cur ^ 1];
        e.nodes = d_nodes; e.prims = d_prims; e.parent = (int *)b_parent.p; e.leaf_parent = (int *)b_lparent.p;
        e.vals = vals;
        hipLaunchKernelGGL(k_bfs_emit, dim3(g), dim3(256), 0, st, e, bp);
        level_start = (int)(4 * node_blocks_done);
        node_blocks_done += nb; leaf_blocks_done += lb;
        cnt = (int)(4 * nb);
        cur ^= 1;
    }
    if (node_blocks_done != n_node_blocks || leaf_blocks_done != n_leaf_blocks)
        return set_error(HZ_ERR_HIP, "BVH numbering did not fill its arrays (internal error)");
    // ---- hit-cache ancestors (from the final numbering) ------------------------------------------------------
    if (n_leaf_blocks)
        hipLaunchKernelGGL(k_anc_bfs, dim3((unsigned)((n_leaf_blocks + 255) / 256)), dim3(256), 0, st, (int)n_leaf_blocks,
                           std::max(1, h.anc_levels), (const int *)b_parent.p, (const int *)b_lparent.p, d_anc);
    HZ_HIP(hipStreamSynchronize(st));
    HZ_HIP(hipGetLastError());
    // height = node levels of the numbering (wrapper levels included): the traversal stack holds one entry per level
    const int height = levels;
    const int n_top = (int)std::min<size_t>(n_nodes, HZ_MAX_TOP_NODES);
    h.height = height;
    h.n_top = n_top;
    HZ_HIP(hipMemcpyAsync(blob, &h, sizeof(h), hipMemcpyHostToDevice, st));
    HZ_HIP(hipStreamSynchronize(st));
    sc->hdr = h;
    const double bvh_s = t_bvh.stop();
    if (stats) {
        stats->t_bvh_s += bvh_s; stats->t_h2d_s += h2d_s;
        stats->bvh_height = height; stats->scene_bytes = h.total_bytes;
    }
    return HZ_OK;
}

}  // namespace hz
