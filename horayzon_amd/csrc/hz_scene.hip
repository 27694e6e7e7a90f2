// hz_scene.hip -- on-device LBVH build for DEM meshes (gfx950).
//
// Replaces the Embree scene build of the reference (initializeScene,
// horizon_comp.cpp:101-231 / shadow_comp.cpp:198-298, rtcCommitScene :223).
//
// Pipeline (all on one stream, no host round trip except the 6-float bounds):
//   1. k_bounds      min/max of all vertices            (HBM streaming, 12 B/vertex)
//   2. k_morton      one key per primitive: 2 x 16 bit Morton code of the
//                    centroid (x, y); primitive = DEM quad (2 triangles) or TIN triangle
//   3. radix sort of (key, primitive id): hz_sort.hip (stable LSD, 8 bit digits, hand written)
//   4. k_karras      binary radix tree over the sorted keys (Karras 2012)
//   5. k_leaf_boxes / k_refit_pass  leaf AABBs; bottom-up union, level synchronous with work lists
//   6. k_roots/scan  collapse along the 2-bit Morton digits: a binary node starts a 4-wide
//                    node when its common-prefix length enters a new digit (quadtree level)
//   7. k_emit4       64 B nodes with conservatively quantised child AABBs (8 bit x/y, 16 bit z),
//                    children sorted tallest first
//   8. k_top/k_permute  breadth-first relabel of the top of the tree (hot levels contiguous)
//      k_parents4/k_ancestors  per leaf the node HZ_ANC_LEVELS above it (hit cache)
//   9. k_emit_prims  48 B leaf records in Morton order
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

__global__ __launch_bounds__(256) void k_leaf_boxes(BuildParams b, const uint32_t *__restrict__ vals,
                                                   const int *__restrict__ parent_leaf,
                                                   float4 *__restrict__ leaf_lo, float4 *__restrict__ leaf_hi,
                                                   int *__restrict__ arrivals, int *__restrict__ next_list,
                                                   unsigned int *__restrict__ next_count) {
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

__global__ __launch_bounds__(256) void k_emit4(Emit4 e, Node *__restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n_nodes || !e.flag[i]) return;
    const int dg = e.plen[i] >> 1;
    // gather the (<= 4) exits of the digit group rooted at i
    int link[4]; float lo[4][3], hi[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        link[k] = HZ_EMPTY;
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
    // the traversal visits slot 0 first: put the tallest child there (a blocked ray is most likely
    // blocked by the child that reaches highest), empty slots last
    {
        auto key = [&](int k) { return link[k] == HZ_EMPTY ? -INFINITY : hi[k][2]; };
        auto cswap = [&](int x, int y) {
            if (key(x) < key(y)) {
                const int t = link[x]; link[x] = link[y]; link[y] = t;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    float f = lo[x][q]; lo[x][q] = lo[y][q]; lo[y][q] = f;
                    f = hi[x][q]; hi[x][q] = hi[y][q]; hi[y][q] = f;
                }
            }
        };
        cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
    }
    const float4 nl = e.node_lo[i], nh = e.node_hi[i];
    const float org[3] = {nl.x, nl.y, nl.z};
    const float ext[3] = {nh.x - nl.x, nh.y - nl.y, nh.z - nl.z};
    const uint32_t ex = step_exponent(ext[0], 255.0f), ey = step_exponent(ext[1], 255.0f);
    const uint32_t ez = step_exponent(ext[2], 65535.0f);
    const float sx = __uint_as_float(ex << 23), sy = __uint_as_float(ey << 23), sz = __uint_as_float(ez << 23);
    Node n;
    n.org[0] = org[0]; n.org[1] = org[1]; n.org[2] = org[2];
    n.scale = ex | (ey << 8) | (ez << 16);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        n.link[k] = link[k];
        if (link[k] == HZ_EMPTY) { n.qxy[k] = 0x00ff00ffu; n.qz[k] = 0x0000ffffu; continue; }   // lo > hi: never hit
        const uint32_t xl = quant_lo(lo[k][0], org[0], sx, 255.0f), xh = quant_hi(hi[k][0], org[0], sx, 255.0f);
        const uint32_t yl = quant_lo(lo[k][1], org[1], sy, 255.0f), yh = quant_hi(hi[k][1], org[1], sy, 255.0f);
        const uint32_t zl = quant_lo(lo[k][2], org[2], sz, 65535.0f), zh = quant_hi(hi[k][2], org[2], sz, 65535.0f);
        n.qxy[k] = xl | (xh << 8) | (yl << 16) | (yh << 24);
        n.qz[k] = zl | (zh << 16);
    }
    nodes[e.idx[i]] = n;
}

// --- breadth-first relabel of the top of the tree -----------------------------------------
// perm[old] = new.  Single lane: n_top <= a few thousand nodes.
__global__ void k_top(const Node *__restrict__ nodes, int n_nodes, int n_top,
                      int *__restrict__ perm, int *__restrict__ top, uint8_t *__restrict__ in_top) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int head = 0, tail = 0;
    top[tail++] = 0;
    while (head < tail && tail < n_top) {
        const Node &nd = nodes[top[head++]];
        for (int k = 0; k < 4; k++)
            if (nd.link[k] >= 0 && tail < n_top) top[tail++] = nd.link[k];
    }
    const int k = tail;
    for (int r = 0; r < k; r++) if (top[r] < k) in_top[top[r]] = 1;
    for (int r = 0; r < k; r++) perm[top[r]] = r;
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

__global__ __launch_bounds__(256) void k_permute(int n_nodes, const int *__restrict__ perm,
                                                const Node *__restrict__ src, Node *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    Node n = src[i];
#pragma unroll
    for (int k = 0; k < 4; k++) if (n.link[k] >= 0) n.link[k] = perm[n.link[k]];
    dst[perm[i]] = n;
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

// --- hit-cache ancestors: for every leaf the node `levels` levels above it --------------------------
__global__ __launch_bounds__(256) void k_parents4(int n4, const Node *__restrict__ nodes, int *__restrict__ parent4,
                                                 int *__restrict__ leaf_parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    if (i == 0) parent4[0] = -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int l = nodes[i].link[k];
        if (l >= 0) parent4[l] = i;
        else if (l != HZ_EMPTY) leaf_parent[~l] = i;
    }
}

__global__ __launch_bounds__(256) void k_ancestors(int n_prims, int levels, const int *__restrict__ parent4,
                                                  const int *__restrict__ leaf_parent, int *__restrict__ anc) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_prims) return;
    int n = leaf_parent[s];
    for (int r = 1; r < levels; r++) {
        const int p = parent4[n];
        if (p < 0) break;
        n = p;
    }
    anc[s] = n;
}

// single primitive: a root whose slot 0 is the leaf, quantised against its own box
__global__ void k_single_node(const float4 *leaf_lo, const float4 *leaf_hi, Node *nodes) {
    const float4 l = leaf_lo[0], h = leaf_hi[0];
    const uint32_t ex = step_exponent(h.x - l.x, 255.0f), ey = step_exponent(h.y - l.y, 255.0f);
    const uint32_t ez = step_exponent(h.z - l.z, 65535.0f);
    Node n;
    n.org[0] = l.x; n.org[1] = l.y; n.org[2] = l.z;
    n.scale = ex | (ey << 8) | (ez << 16);
    for (int k = 0; k < 4; k++) { n.link[k] = HZ_EMPTY; n.qxy[k] = 0x00ff00ffu; n.qz[k] = 0x0000ffffu; }
    n.link[0] = ~0;
    n.qxy[0] = 0u | (quant_hi(h.x, l.x, __uint_as_float(ex << 23), 255.0f) << 8)
             | (quant_hi(h.y, l.y, __uint_as_float(ey << 23), 255.0f) << 24);
    n.qz[0] = quant_hi(h.z, l.z, __uint_as_float(ez << 23), 65535.0f) << 16;
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
    Arena arena1, arena2;
    {
        const size_t P = (size_t)n_prims, B = (size_t)n_bin;
        const size_t list_cap0 = B / 2 + 2;
        const size_t sizes[] = {is_device_ptr(vert_grid) ? 0 : nvert * 12, has_tin ? (size_t)nvs * 12 : 0,
                                has_tin ? (size_t)nts * 12 : 0, 24, P * 4, P * 4, P * 4, P * 4,
                                sort_temp_elems(P) * 4, B * 8, B * 4, P * 4, B, B * 4, P * 16, P * 16, B * 16, B * 16,
                                (B + 2 * list_cap0 + 4) * 4, B * 4, B * 4, scan_temp_elems(B) * 4};
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
    HZ_HIP(b_cnt.alloc(((size_t)n_bin + 2 * list_cap + 4) * 4));
    int *arrivals = (int *)b_cnt.p;
    int *lists[2] = {arrivals + n_bin, arrivals + n_bin + list_cap};
    unsigned int *counts = (unsigned int *)(arrivals + n_bin + 2 * list_cap);
    HZ_HIP(hipMemsetAsync(arrivals, 0, (size_t)n_bin * 4, st));
    HZ_HIP(hipMemsetAsync(counts, 0, 16, st));
    HZ_HIP(hipMemsetAsync(b_nlo.p, 0, (size_t)n_bin * 16, st));
    if (n_prims > 1)
        hipLaunchKernelGGL(k_karras, dim3((n_prims - 1 + 255) / 256), dim3(256), 0, st, keys, n_prims,
                           (int2 *)b_child.p, (int *)b_pint.p, (int *)b_pleaf.p, (uint8_t *)b_plen.p,
                           (int *)b_first.p);
    hipLaunchKernelGGL(k_leaf_boxes, dim3(gp), dim3(256), 0, st, bp, vals, (const int *)b_pleaf.p,
                       (float4 *)b_llo.p, (float4 *)b_lhi.p, arrivals, lists[0], &counts[0]);
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

    // ---- blob allocation (exact size now known) ------------------------------------------------
    h.n_nodes = n4;
    h.off_verts = sizeof(BlobHeader);
    h.off_nodes = align_up(h.off_verts + nvert * 12, 256);
    h.off_prims = align_up(h.off_nodes + (size_t)n4 * sizeof(Node), 256);
    h.off_anc = align_up(h.off_prims + (size_t)n_prims * sizeof(Prim), 256);
    h.anc_levels = HZ_ANC_LEVELS;
    h.total_bytes = align_up(h.off_anc + (size_t)n_prims * 4, 256);
    HZ_HIP(hipMalloc(&sc->blob, h.total_bytes));
    sc->owns_blob = true;
    sc->blob_bytes = h.total_bytes;
    char *blob = (char *)sc->blob;
    float *d_verts = (float *)(blob + h.off_verts);
    Node *d_nodes = (Node *)(blob + h.off_nodes);
    Prim *d_prims = (Prim *)(blob + h.off_prims);
    int *d_anc = (int *)(blob + h.off_anc);
    HZ_HIP(hipMemcpyAsync(d_verts, d_verts_src, nvert * 12, hipMemcpyDeviceToDevice, st));

    // ---- 7./8. emit 4-wide nodes, relabel the top breadth first ----------------------------------
    int n_top = 1;
    TempBuf b_tmp4, b_perm, b_top, b_intop;
    if (n_prims > 1) {
        HZ_HIP(arena2.reserve(Arena::pad((size_t)n4 * sizeof(Node)) + 2 * Arena::pad((size_t)n4 * 4) + 4 * 4096 +
                              Arena::pad((size_t)HZ_MAX_TOP_NODES * 8) + Arena::pad((size_t)n_prims * 4)));
        g_arena = &arena2;
        HZ_HIP(b_tmp4.alloc((size_t)n4 * sizeof(Node)));
        Emit4 e;
        e.child = (const int2 *)b_child.p; e.plen = (const uint8_t *)b_plen.p; e.first = (const int *)b_first.p;
        e.keys = keys; e.flag = (const uint32_t *)b_flag.p; e.idx = (const uint32_t *)b_idx.p;
        e.leaf_lo = (const float4 *)b_llo.p; e.leaf_hi = (const float4 *)b_lhi.p;
        e.node_lo = (const float4 *)b_nlo.p; e.node_hi = (const float4 *)b_nhi.p;
        e.n_nodes = n_bin;
        hipLaunchKernelGGL(k_emit4, dim3(gn), dim3(256), 0, st, e, (Node *)b_tmp4.p);
        n_top = std::min(n4, HZ_MAX_TOP_NODES);
        HZ_HIP(b_perm.alloc((size_t)n4 * 4));
        HZ_HIP(b_top.alloc((size_t)n_top * 4));
        HZ_HIP(b_intop.alloc((size_t)n_top));
        HZ_HIP(hipMemsetAsync(b_intop.p, 0, (size_t)n_top, st));
        const int g4 = (n4 + 255) / 256;
        hipLaunchKernelGGL(k_iota, dim3(g4), dim3(256), 0, st, (int *)b_perm.p, n4);
        hipLaunchKernelGGL(k_top, dim3(1), dim3(64), 0, st, (const Node *)b_tmp4.p, n4, n_top,
                           (int *)b_perm.p, (int *)b_top.p, (uint8_t *)b_intop.p);
        hipLaunchKernelGGL(k_permute, dim3(g4), dim3(256), 0, st, n4, (const int *)b_perm.p,
                           (const Node *)b_tmp4.p, d_nodes);
    } else {
        hipLaunchKernelGGL(k_single_node, dim3(1), dim3(1), 0, st, (const float4 *)b_llo.p,
                           (const float4 *)b_lhi.p, d_nodes);
    }
    // ---- hit-cache ancestors (from the final node numbering) ---------------------------------------
    if (n_prims > 1) {
        TempBuf b_par4, b_lpar;
        HZ_HIP(b_par4.alloc((size_t)n4 * 4));
        HZ_HIP(b_lpar.alloc((size_t)n_prims * 4));
        hipLaunchKernelGGL(k_parents4, dim3((n4 + 255) / 256), dim3(256), 0, st, n4, (const Node *)d_nodes,
                           (int *)b_par4.p, (int *)b_lpar.p);
        hipLaunchKernelGGL(k_ancestors, dim3(gp), dim3(256), 0, st, n_prims, std::max(1, h.anc_levels),
                           (const int *)b_par4.p, (const int *)b_lpar.p, d_anc);
        HZ_HIP(hipStreamSynchronize(st));      // b_par4 / b_lpar leave scope (arena memory stays valid anyway)
    } else {
        HZ_HIP(hipMemsetAsync(d_anc, 0, 4, st));
    }
    // ---- 9. leaf records ------------------------------------------------------------------------
    hipLaunchKernelGGL(k_emit_prims, dim3(gp), dim3(256), 0, st, bp, vals, d_prims);
    // height in 4-wide levels = 1 + levels below the root's digit group (stored with the root box)
    float4 root_lo = make_float4(0, 0, 0, 0);
    if (n_prims > 1) HZ_HIP(hipMemcpyAsync(&root_lo, b_nlo.p, 16, hipMemcpyDeviceToHost, st));
    HZ_HIP(hipStreamSynchronize(st));
    HZ_HIP(hipGetLastError());
    int below = 0;
    if (n_prims > 1) memcpy(&below, &root_lo.w, 4);
    const int height = below + 1;
    h.height = height;
    h.n_top = n_top;
    if (3 * height > HZ_MAX_STACK)
        return set_error(HZ_ERR_DEPTH, "BVH height %d (4-wide levels) exceeds the traversal stack (%d entries)",
                         height, HZ_MAX_STACK);
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
