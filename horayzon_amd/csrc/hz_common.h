// hz_common.h -- shared host/device definitions for libhorayzon_hip (gfx950).
//
// Scene blob layout in HBM (one contiguous, position-independent allocation):
//
//   [ BlobHeader (256 B) | vertices f32[3*V] | nodes Node[~P/3] | prims Prim[P] | anc i32[P] ]
//
// vertices : the caller's vert_grid, untouched (ray origins read them;
//            reference: shared vertex buffer, horizon_comp.cpp:126-127).
// nodes    : flat LBVH collapsed to 4-wide nodes along the 2-bit digits of the Morton key (= a quadtree over the
//            (x, y) centroids).  One node is 32 B and holds the conservatively quantised bounds of its children (8 bit,
//            relative to the node's own box) of up to 4 children in quadrant order: one x range per column half, one y
//            range per row half, one z range per child.  ALL nodes
//            are numbered breadth first and the children of a node are CONTIGUOUS: one index (`first`) addresses
//            the block of 4 child slots -- 4 consecutive nodes, or 4 consecutive leaf records (a node whose children
//            were mixed got its leaves wrapped into single-child nodes, so a block is of one kind).  A traversal
//            therefore keeps ONE stack entry per tree level (block + 3-bit mask of the siblings still to visit): the
//            stack is `height` entries deep, with no overflow case.  AABBs live in a frame centred on the scene
//            (`center`) and are padded by `pad`, which makes the box test conservative with respect to the float32
//            triangle test: hit decisions depend on the triangle test only, never on the tree.
// anc      : per leaf the index of the node a few levels above it.  A ray that is expected to be
//            blocked (it points below the horizon found for the previous azimuth) first walks the
//            subtree above the leaf that blocked the previous ray of this cell -- the blocking
//            ridge moves little between neighbouring azimuths -- and only falls back to the root
//            when that finds nothing.  Any-hit results cannot change (section 4 of DESIGN.md).
// prims    : 48 B leaf records in blocks of 4 (the leaf children of one node, in that node's slot order; unused
//            slots are zero) = the 4 corner vertices of a DEM quad (two triangles a,b,c / b,d,c -- the split of
//            horizon_comp.cpp:139-151) or the 3 vertices of a TIN triangle (d.x = NaN), raw coordinates.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define HZ_BLOB_MAGIC 0x485a4c42u /* "HZLB" */
#define HZ_BLOB_VERSION 10u
#define HZ_WAVE 64

struct BlobHeader {
    uint32_t magic, version;
    int32_t d0, d1;
    int32_t n_quads, n_tin, n_prims, n_nodes;
    int32_t height, n_top;
    float center[3];
    float pad;
    float lo[3], hi[3];
    uint32_t reserved0[2];
    uint64_t off_verts, off_nodes, off_prims, total_bytes;
    uint64_t off_anc;        // int32[P]: for every leaf the node `anc_levels` levels above it (hit cache)
    int32_t anc_levels;
    int32_t n_prim_slots;    // leaf records incl. the unused slots of partly filled blocks (prims, anc are this long)
    uint32_t flags;          // HZ_BLOB_HEIGHT_FIELD / HZ_BLOB_BAD_MAP: see below
    uint32_t n_flipped;      // DEM triangles whose (x, y) projection is degenerate or oriented against the majority
    uint64_t off_bad;        // HZ_BLOB_BAD_MAP: bitmap of bad_nb x bad_nb bits (row-major in y, uint32 words) over the scene's (x, y) box
    int32_t bad_nb;
    float bad_x0, bad_y0, bad_sx, bad_sy;   // bitmap cell of a point: (floor((x - bad_x0) * bad_sx), floor((y - bad_y0) * bad_sy)), clamped
    uint8_t reserved[256 - 172];
};
static_assert(sizeof(BlobHeader) == 256, "BlobHeader must be 256 bytes");
// flags bit 0 (HZ_BLOB_HEIGHT_FIELD): the DEM mesh is a height field over the world (x, y) plane -- every DEM triangle
// projects onto that plane with the same orientation and a non-degenerate area (the 2-D cross product of two projected
// edges is larger than its own rounding: |n_z| > 1e-5 (|ux vy| + |uy vx|); how STEEP a triangle is does not matter -- a
// NoData hole or a 30 km step on a regular (x, y) grid is a height field), and there is no outer TIN: then the
// projection of the grid is injective and "outside a window of quads" implies "horizontally outside its boundary
// polygon".  The near-field certificates (hz_near.hip) rely on exactly that; the reference accepts any vertex buffer
// (horizon_comp.cpp:126-127), e.g. a frame whose z axis is not "up" or a mesh folded over itself.
// flags bit 1 (HZ_BLOB_BAD_MAP; round 5): some DEM quads are NOT like that (or there is an outer TIN), and the blob
// carries a coarse bitmap over the scene's (x, y) box in which every such quad and every TIN triangle has marked the
// cells its (x, y) footprint touches.  A cell whose window's (x, y) bounding box touches no marked bitmap cell keeps
// its certificate (hz_near.hip; DESIGN.md section 4.3 row R has the argument), every other cell is refused.  Neither
// flag: a bad primitive's footprint was too large to rasterise -- no certificates for this scene.
#define HZ_BLOB_HEIGHT_FIELD 1u
#define HZ_BLOB_BAD_MAP 2u

// traversal links: >= 0 node index; < 0 leaf, record index = link & 0x7fffffff (sign + magnitude, so that child k of
// a block is `first + k` for both kinds); HZ_EMPTY: nothing (the largest positive value: "is a node" is one unsigned
// compare, "is a leaf" a sign test)
#define HZ_EMPTY ((int)0x7fffffff)
#define HZ_IS_NODE(link) ((unsigned)(link) < 0x7fffffffu)
#define HZ_LEAF_BIT 0x80000000u
#define HZ_LEAF_ID(link) ((int)((unsigned)(link) & 0x7fffffffu))
struct __attribute__((aligned(16))) Node {
    float org[3];        // origin of the node's quantisation frame (centred scene frame): a bound with code q lies at
                         // org[axis] + (1024 + q) * step (the form hz_node_hits decodes: the byte q becomes the half float
                         // 0x64qq = 1024 + q).  The steps are powers of two and live IN these floats: the low 9 mantissa bits
                         // of org[0] hold the biased exponent of the x / y step (one step for both), those of org[2] that of
                         // the z step (bit 8 is zero, so `bits << 23` IS the step).  The build chooses the origins as floats
                         // with those low bits and quantises against exactly these values (hz_scene.hip: axis_try).
    int32_t first;       // link of child slot 0; slots 1..3 are first + 1 .. first + 3: 4 consecutive nodes or 4
                         // consecutive leaf records (blocks are 4-aligned).  Slot k is the quadrant 2 * row bit + column bit
                         // of the Morton digit this node splits.
    uint32_t qx;         // x ranges, 8 bit each: (lo, hi) of the two children with column bit 0, (lo, hi) of those with bit 1
    uint32_t qy;         // y ranges: (lo, hi) of the children with row bit 0, (lo, hi) of those with row bit 1
    uint32_t qz[2];      // z ranges per child slot: qz[0] = (lo0, hi0, lo1, hi1), qz[1] = (lo2, hi2, lo3, hi3);
                         // an unused slot has lo = 255 > hi = 0
};
static_assert(sizeof(Node) == 32, "Node must be 32 bytes");

struct __attribute__((aligned(16))) Prim {
    float a[3], b[3], c[3], d[3];
};
static_assert(sizeof(Prim) == 48, "Prim must be 48 bytes");

// XCD-aware block -> tile mapping (device + host).  Workgroup b runs on XCD b % 8 (observed dispatch order; used for
// L2 affinity only), and the workgroups of one XCD are dispatched in the order of their number.  The tile grid is cut
// into 8 columns x `pi` rows of compact PATCHES; XCD x owns one patch of every patch row (in row k the patch of column
// (x + 5 k) % 8, so its patches are spread over the grid) and walks them one after the other, each in column groups of
// `gw` tiles, so that the workgroups resident at the same time on one XCD cover a compact piece of the DEM (their rays
// share BVH nodes in that XCD's L2).  Rounds 1-2 gave every XCD ONE region (an eighth of the grid): the XCDs then finish
// when their own region is done, and on the 3601^2 tile the slowest one needed 5.9 % longer than the mean
// (HZ_XCD_TRACE, 2091 ... 2326 ms) -- the launch waits for it.  Several patches per XCD average the terrain out.
struct TileMap {
    int tiles_i, tiles_j;   // tile grid
    int pi;                 // patch rows (8 patch columns): every XCD owns pi patches
    int ri, rj;             // tiles per patch (rows, columns; the last patches of a row / column may be cut)
    int gw;                 // column-group width inside a patch
    int per_xcd;            // pi * ri * rj: workgroups launched per XCD
};

#ifdef __HIPCC__
__device__ __forceinline__ bool hz_tile_of_block(const TileMap &m, int b, int *ti, int *tj) {
    const int x = b & 7, t = b >> 3;
    const int per_patch = m.ri * m.rj;
    const int k = t / per_patch, r0 = t - k * per_patch;
    if (k >= m.pi) return false;
    const int i0 = k * m.ri, j0 = ((x + 5 * k) & 7) * m.rj;
    const int rows = min(m.ri, m.tiles_i - i0), cols = min(m.rj, m.tiles_j - j0);
    if (rows <= 0 || cols <= 0 || r0 >= rows * cols) return false;
    const int per_group = m.gw * rows;
    const int g = r0 / per_group, r = r0 - g * per_group;
    const int w = min(m.gw, cols - g * m.gw);          // width of this (possibly last, narrower) group
    *ti = i0 + r / w;
    *tj = j0 + g * m.gw + (r - (r / w) * w);
    return true;
}
#endif

#ifdef __HIPCC__

// "den != 0" of the triangle test with the rounding of den taken into account (round 5): a ray
// within 2^-20 (relative to |nx dx| + |ny dy| + |nz dz|) of a triangle's plane counts as parallel to it.  With a den that is
// rounding noise all three edge functions are noise too (coplanar lines meet somewhere), so the exact `den != 0` of rounds 1-4
// accepted "hits at t = 0" of triangles the ray passes kilometres beside whenever the origin lay in their plane as well
// (integer terrace heights with an integer ray_org_elev) -- hits that depend on which boxes a traversal happens to open.
#define HZ_DEN_NOISE 9.5367431640625e-07f      // 2^-20

// ---------------------------------------------------------------------------
// ray / triangle: Embree-robust style Pluecker edge test in float32.
// This translation unit is compiled with -ffp-contract=off: every multiply
// and add below rounds separately, in exactly this association order.  The
// hit decision is the product's numerical contract.
// ---------------------------------------------------------------------------
// Build-time switch -DHZ_TRI_FMA (NOT the contract; scripts/build_variant.sh fma -DHZ_TRI_FMA): the cross and dot products of
// the triangle test with fused multiply-adds, the way Embree's vector code evaluates them on an FMA machine (common/math/vec3.h:
// cross(a, b).x = msub(a.y, b.z, a.z * b.y), dot(a, b) = madd(a.x, b.x, madd(a.y, b.y, a.z * b.z))); everything else unchanged.
// The oracle's triangle mode "plain_fma" is the same arithmetic on the CPU (tests/test_gpu_tri_fma.py: bit-identical).  37 fewer
// VALU instructions per leaf step; should the Embree pin (README.md) show that its FMA evaluation decides differently from
// the unfused one, adopting it is this flag and a re-validation, not a rewrite.
#ifdef HZ_TRI_FMA
#define HZ_CROSS1(ay, bz, az, by) __builtin_fmaf((ay), (bz), -((az) * (by)))
#define HZ_DOT3(ax, bx, ay, by, az, bz) __builtin_fmaf((ax), (bx), __builtin_fmaf((ay), (by), (az) * (bz)))
#define HZ_DEN_SUM(nx, dx, ny, dy, nz, dz, pnx, pny, pnz) \
    __builtin_fmaf(__builtin_fabsf(nx), __builtin_fabsf(dx), __builtin_fmaf(__builtin_fabsf(ny), __builtin_fabsf(dy), __builtin_fabsf(nz) * __builtin_fabsf(dz)))
#define HZ_DEN(nx, dx, ny, dy, nz, dz, pnx, pny, pnz) HZ_DOT3(nx, dx, ny, dy, nz, dz)
#else
#define HZ_CROSS1(ay, bz, az, by) ((ay) * (bz) - (az) * (by))
#define HZ_DOT3(ax, bx, ay, by, az, bz) (((ax) * (bx) + (ay) * (by)) + (az) * (bz))
#define HZ_DEN_SUM(nx, dx, ny, dy, nz, dz, pnx, pny, pnz) ((__builtin_fabsf(pnx) + __builtin_fabsf(pny)) + __builtin_fabsf(pnz))
#define HZ_DEN(nx, dx, ny, dy, nz, dz, pnx, pny, pnz) (((pnx) + (pny)) + (pnz))
#endif
__device__ __forceinline__ bool hz_tri_hit(float ox, float oy, float oz,
                                           float dx, float dy, float dz, float tfar,
                                           float p0x, float p0y, float p0z,
                                           float p1x, float p1y, float p1z,
                                           float p2x, float p2y, float p2z) {
    const float v0x = p0x - ox, v0y = p0y - oy, v0z = p0z - oz;
    const float v1x = p1x - ox, v1y = p1y - oy, v1z = p1z - oz;
    const float v2x = p2x - ox, v2y = p2y - oy, v2z = p2z - oz;
    const float e0x = v2x - v0x, e0y = v2y - v0y, e0z = v2z - v0z;
    const float e1x = v0x - v1x, e1y = v0y - v1y, e1z = v0z - v1z;
    const float e2x = v1x - v2x, e2y = v1y - v2y, e2z = v1z - v2z;
    const float s0x = v2x + v0x, s0y = v2y + v0y, s0z = v2z + v0z;
    const float s1x = v0x + v1x, s1y = v0y + v1y, s1z = v0z + v1z;
    const float s2x = v1x + v2x, s2y = v1y + v2y, s2z = v1z + v2z;
    const float c0x = HZ_CROSS1(e0y, s0z, e0z, s0y);
    const float c0y = HZ_CROSS1(e0z, s0x, e0x, s0z);
    const float c0z = HZ_CROSS1(e0x, s0y, e0y, s0x);
    const float c1x = HZ_CROSS1(e1y, s1z, e1z, s1y);
    const float c1y = HZ_CROSS1(e1z, s1x, e1x, s1z);
    const float c1z = HZ_CROSS1(e1x, s1y, e1y, s1x);
    const float c2x = HZ_CROSS1(e2y, s2z, e2z, s2y);
    const float c2y = HZ_CROSS1(e2z, s2x, e2x, s2z);
    const float c2z = HZ_CROSS1(e2x, s2y, e2y, s2x);
    const float U = HZ_DOT3(c0x, dx, c0y, dy, c0z, dz);
    const float V = HZ_DOT3(c1x, dx, c1y, dy, c1z, dz);
    const float W = HZ_DOT3(c2x, dx, c2y, dy, c2z, dz);
    const float UVW = (U + V) + W;
    const float eps = 1.1920928955078125e-7f * __builtin_fabsf(UVW);
    const float mn = __builtin_fminf(U, __builtin_fminf(V, W));
    const float mx = __builtin_fmaxf(U, __builtin_fmaxf(V, W));
    if (!((mn >= -eps) || (mx <= eps))) return false;
    const float nx = HZ_CROSS1(e1y, e0z, e1z, e0y);
    const float ny = HZ_CROSS1(e1z, e0x, e1x, e0z);
    const float nz = HZ_CROSS1(e1x, e0y, e1y, e0x);
    const float pnx = nx * dx, pny = ny * dy, pnz = nz * dz;      // (HZ_TRI_FMA: unused, removed by the compiler)
    const float den = HZ_DEN(nx, dx, ny, dy, nz, dz, pnx, pny, pnz);
    const float T = HZ_DOT3(v0x, nx, v0y, ny, v0z, nz);
    if (!(__builtin_fabsf(den) > HZ_DEN_NOISE * HZ_DEN_SUM(nx, dx, ny, dy, nz, dz, pnx, pny, pnz))) return false;   // (parallel within rounding)
    const float Ts = (den < 0.0f) ? -T : T;
    const float ad = __builtin_fabsf(den);
    if (!(Ts >= 0.0f)) return false;
    if (!(Ts <= tfar * ad)) return false;
    return true;
}

// The two triangles (a, b, c) and (b, d, c) of a DEM quad in one go.  Bitwise the same decisions as
// two hz_tri_hit calls: the second triangle's first edge function runs over the shared diagonal
// c - b = -(b - c), and every step of that edge function is an exact negation of the first
// triangle's third one (IEEE rounding is sign symmetric), so U1 = -W0 is reused instead of
// recomputed.  `second` = false for a TIN triangle (only a, b, c are tested).
__device__ __forceinline__ bool hz_quad_hit(float ox, float oy, float oz, float dx, float dy, float dz, float tfar,
                                            float ax, float ay, float az, float bx, float by, float bz,
                                            float cx, float cy, float cz, float qx, float qy, float qz,
                                            bool second) {
    const float v0x = ax - ox, v0y = ay - oy, v0z = az - oz;      // a
    const float v1x = bx - ox, v1y = by - oy, v1z = bz - oz;      // b
    const float v2x = cx - ox, v2y = cy - oy, v2z = cz - oz;      // c
    const float e0x = v2x - v0x, e0y = v2y - v0y, e0z = v2z - v0z;
    const float e1x = v0x - v1x, e1y = v0y - v1y, e1z = v0z - v1z;
    const float e2x = v1x - v2x, e2y = v1y - v2y, e2z = v1z - v2z;
    const float s0x = v2x + v0x, s0y = v2y + v0y, s0z = v2z + v0z;
    const float s1x = v0x + v1x, s1y = v0y + v1y, s1z = v0z + v1z;
    const float s2x = v1x + v2x, s2y = v1y + v2y, s2z = v1z + v2z;
    const float U = HZ_DOT3(HZ_CROSS1(e0y, s0z, e0z, s0y), dx, HZ_CROSS1(e0z, s0x, e0x, s0z), dy, HZ_CROSS1(e0x, s0y, e0y, s0x), dz);
    const float V = HZ_DOT3(HZ_CROSS1(e1y, s1z, e1z, s1y), dx, HZ_CROSS1(e1z, s1x, e1x, s1z), dy, HZ_CROSS1(e1x, s1y, e1y, s1x), dz);
    const float W = HZ_DOT3(HZ_CROSS1(e2y, s2z, e2z, s2y), dx, HZ_CROSS1(e2z, s2x, e2x, s2z), dy, HZ_CROSS1(e2x, s2y, e2y, s2x), dz);
    // No early outs (round 4): with ~25 rays of a wave in a leaf step some lane passes every partial test, so the skipped
    // blocks ran anyway and every `if` was an exec-mask save / branch / restore on top.  The decisions are the same
    // comparisons on the same values, combined with non-short-circuit & and |.
    bool hit0;
    {
        const float UVW = (U + V) + W;
        const float eps = 1.1920928955078125e-7f * __builtin_fabsf(UVW);
        const float mn = __builtin_fminf(U, __builtin_fminf(V, W));
        const float mx = __builtin_fmaxf(U, __builtin_fmaxf(V, W));
        const float nx = HZ_CROSS1(e1y, e0z, e1z, e0y), ny = HZ_CROSS1(e1z, e0x, e1x, e0z), nz = HZ_CROSS1(e1x, e0y, e1y, e0x);
        const float pnx = nx * dx, pny = ny * dy, pnz = nz * dz;
        const float den = HZ_DEN(nx, dx, ny, dy, nz, dz, pnx, pny, pnz);
        const float T = HZ_DOT3(v0x, nx, v0y, ny, v0z, nz);
        const float Ts = (den < 0.0f) ? -T : T;
        const float ad = __builtin_fabsf(den);
        hit0 = ((mn >= -eps) | (mx <= eps)) & (ad > HZ_DEN_NOISE * HZ_DEN_SUM(nx, dx, ny, dy, nz, dz, pnx, pny, pnz)) &
               (Ts >= 0.0f) & (Ts <= tfar * ad);
    }
    // triangle (b, d, c): v0' = b, v1' = d, v2' = c
    const float w1x = qx - ox, w1y = qy - oy, w1z = qz - oz;      // d
    const float f0x = -e2x, f0y = -e2y, f0z = -e2z;               // e0' = c - b
    const float f1x = v1x - w1x, f1y = v1y - w1y, f1z = v1z - w1z;   // e1' = b - d
    const float f2x = w1x - v2x, f2y = w1y - v2y, f2z = w1z - v2z;   // e2' = d - c
    const float t1x = v1x + w1x, t1y = v1y + w1y, t1z = v1z + w1z;
    const float t2x = w1x + v2x, t2y = w1y + v2y, t2z = w1z + v2z;
    const float U1 = -W;
    const float V1 = HZ_DOT3(HZ_CROSS1(f1y, t1z, f1z, t1y), dx, HZ_CROSS1(f1z, t1x, f1x, t1z), dy, HZ_CROSS1(f1x, t1y, f1y, t1x), dz);
    const float W1 = HZ_DOT3(HZ_CROSS1(f2y, t2z, f2z, t2y), dx, HZ_CROSS1(f2z, t2x, f2x, t2z), dy, HZ_CROSS1(f2x, t2y, f2y, t2x), dz);
    const float UVW = (U1 + V1) + W1;
    const float eps = 1.1920928955078125e-7f * __builtin_fabsf(UVW);
    const float mn = __builtin_fminf(U1, __builtin_fminf(V1, W1));
    const float mx = __builtin_fmaxf(U1, __builtin_fmaxf(V1, W1));
    const float nx = HZ_CROSS1(f1y, f0z, f1z, f0y), ny = HZ_CROSS1(f1z, f0x, f1x, f0z), nz = HZ_CROSS1(f1x, f0y, f1y, f0x);
    const float pnx = nx * dx, pny = ny * dy, pnz = nz * dz;
    const float den = HZ_DEN(nx, dx, ny, dy, nz, dz, pnx, pny, pnz);
    const float T = HZ_DOT3(v1x, nx, v1y, ny, v1z, nz);
    const float Ts = (den < 0.0f) ? -T : T;
    const float ad = __builtin_fabsf(den);
    // (a TIN triangle's record holds NaNs in d: every comparison below is false for it, `second` only states it)
    const bool hit1 = second & ((mn >= -eps) | (mx <= eps)) & (ad > HZ_DEN_NOISE * HZ_DEN_SUM(nx, dx, ny, dy, nz, dz, pnx, pny, pnz)) &
                      (Ts >= 0.0f) & (Ts <= tfar * ad);
    return hit0 | hit1;
}

// closest-hit variant (rtcIntersect1): same acceptance test; t = T / den as one IEEE division
__device__ __forceinline__ bool hz_tri_hit_t(float ox, float oy, float oz, float dx, float dy, float dz,
                                             float tfar, float p0x, float p0y, float p0z, float p1x,
                                             float p1y, float p1z, float p2x, float p2y, float p2z, float *t) {
    if (!hz_tri_hit(ox, oy, oz, dx, dy, dz, tfar, p0x, p0y, p0z, p1x, p1y, p1z, p2x, p2y, p2z)) return false;
    const float v0x = p0x - ox, v0y = p0y - oy, v0z = p0z - oz;
    const float v1x = p1x - ox, v1y = p1y - oy, v1z = p1z - oz;
    const float v2x = p2x - ox, v2y = p2y - oy, v2z = p2z - oz;
    const float e0x = v2x - v0x, e0y = v2y - v0y, e0z = v2z - v0z;
    const float e1x = v0x - v1x, e1y = v0y - v1y, e1z = v0z - v1z;
    const float nx = HZ_CROSS1(e1y, e0z, e1z, e0y);
    const float ny = HZ_CROSS1(e1z, e0x, e1x, e0z);
    const float nz = HZ_CROSS1(e1x, e0y, e1y, e0x);
    const float den = HZ_DOT3(nx, dx, ny, dy, nz, dz);
    const float T = HZ_DOT3(v0x, nx, v0y, ny, v0z, nz);
    *t = T / den;
    return true;
}

// ---------------------------------------------------------------------------
// per-ray constants for the conservative slab test (centred frame)
//
// Where the box tests START (round 5).  The triangle test accepts 0 <= T sgn(den) in float32.
// For a ray whose origin lies within rounding of a triangle's plane (|h| <~ 16 u |v0| / sin(phi), u = 2^-24, phi the
// triangle's angle at v0) the computed T can have the wrong sign: the test then accepts a crossing that in exact
// arithmetic lies BEHIND the origin, at t* = h / sin(theta) (theta = angle between ray and plane) -- millimetres for a
// grazing ray, while the origin may sit just outside that triangle's padded box.  A tree whose box tests start at exactly
// 0 culls that box; brute force reports the hit (found by the adversarial sweep at the end of round 4: seed 48001, #2536).
// So box tests run over [-tau, tfar + tau] of the ray, tau = HZ_BOX_START_PADS * pad (hz_scene.hip: pad = 1e-6 diag +
// 4 eps max|coord|): every traversal kernel shifts the box-test origin BACK by tau along the ray (the same mechanism the
// near-field certificates use to shift it forward) and passes tfar_box = tfar + 2 tau (the far end seen from the shifted
// origin) -- no instruction in the node step changes.  Triangles are always tested with the true origin and the true tfar.
// ---------------------------------------------------------------------------
#ifndef HZ_BOX_START_PADS
#define HZ_BOX_START_PADS 16.0f
#endif
struct RayBox {
    float rdx, rdy, rdz;     // 1 / d (clamped away from inf)
    float ordx, ordy, ordz;  // (o - center) * rd
    uint32_t sel_x, sel_y, sel_z;   // v_perm selectors that turn the byte pair (lo, hi) in bytes 0 / 1 of a bounds word into
                                    // two half floats, the NEAR bound in the low half word and the FAR bound in the high one
                                    // (near = lo when 1/d > 0, hi otherwise); + HZ_SEL_PAIR1 addresses the pair in bytes 2 / 3
};
#define HZ_SEL_PAIR1 0x00020002u

__device__ __forceinline__ float hz_safe_rcp(float d) {
    // v_rcp_f32 (1 ulp) is enough: these constants only feed the conservative box test
    return (__builtin_fabsf(d) > 1e-30f) ? __builtin_amdgcn_rcpf(d) : __builtin_copysignf(1e30f, d);
}

__device__ __forceinline__ RayBox hz_raybox(float ocx, float ocy, float ocz,
                                            float dx, float dy, float dz) {
    RayBox r;
    r.rdx = hz_safe_rcp(dx); r.rdy = hz_safe_rcp(dy); r.rdz = hz_safe_rcp(dz);
    r.ordx = ocx * r.rdx; r.ordy = ocy * r.rdy; r.ordz = ocz * r.rdz;
    // selector bytes 0..3 pick bytes 0..3 of the second v_perm operand (the packed bounds), 4 picks byte 0 of the
    // first one (HZ_HALF_MAGIC): every bound is a single byte that becomes the mantissa of a half float
    r.sel_x = (r.rdx < 0.0f) ? 0x04000401u : 0x04010400u;
    r.sel_y = (r.rdy < 0.0f) ? 0x04000401u : 0x04010400u;
    r.sel_z = (r.rdz < 0.0f) ? 0x04000401u : 0x04010400u;
    return r;
}

// per-node constants: t = q * a + b maps a quantised coordinate to a ray parameter
struct NodeRay { float ax, bx, ay, by, az, bz; };

__device__ __forceinline__ NodeRay hz_node_ray(const RayBox &r, float ox, float oy, float oz) {
    // the quantisation steps: the low 9 bits of org[0] / org[2], shifted into the exponent field (Node)
    const float sxy = __uint_as_float(__float_as_uint(ox) << 23), sz = __uint_as_float(__float_as_uint(oz) << 23);
    NodeRay n;
    n.ax = sxy * r.rdx; n.bx = __builtin_fmaf(ox, r.rdx, -r.ordx);
    n.ay = sxy * r.rdy; n.by = __builtin_fmaf(oy, r.rdy, -r.ordy);
    n.az = sz * r.rdz; n.bz = __builtin_fmaf(oz, r.rdz, -r.ordz);
    return n;
}

// Bounds are decoded WITHOUT integer -> float conversions (24 v_cvt + 24 v_fma per node step in rounds 1-3: the conversions
// are slow-class VALU instructions and the three-source FMAs collided in the VGPR banks): every bound is an 8-bit integer q;
// one v_perm_b32 turns a (lo, hi) byte pair into the two half floats 0x64qq = 1024 + q (near bound in the low half word,
// far bound in the high one; the byte 0x64 comes from the constant operand), and v_fma_mix_f32 reads a half-float source
// directly: t = (1024 + q) * a + b.  The node stores its origins shifted by -1024 steps (Node::org).
// Round 4, 32 B nodes: the four children are the quadrants of one split, so x has ONE range per column half and y one per
// row half (2 v_perm + 4 v_fma_mix per axis and node), z one range per child (4 + 8): 8 v_perm + 16 v_fma_mix per node
// instead of 12 + 24, and two 16 B loads per visit instead of three.
#define HZ_HALF_MAGIC 0x64646464u
typedef _Float16 hz_half2 __attribute__((ext_vector_type(2)));
typedef int hz_int2 __attribute__((ext_vector_type(2)));

// (float)half * a + b as ONE v_fma_mix_f32 (the compiler folds the conversion into the FMA's operand modifier; written in
// C++ rather than inline asm so that it knows the results are arithmetic values: no canonicalising v_max_f32 before min / max)
__device__ __forceinline__ float hz_fma_mix_lo(uint32_t h2, float a, float b) {
    return __builtin_fmaf((float)__builtin_bit_cast(hz_half2, h2).x, a, b);
}
__device__ __forceinline__ float hz_fma_mix_hi(uint32_t h2, float a, float b) {
    return __builtin_fmaf((float)__builtin_bit_cast(hz_half2, h2).y, a, b);
}

// which of the four child boxes of a node does [0, tfar] overlap?  `q` = the node's second 16 bytes (qx, qy, qz[0], qz[1]).
// The ray's direction signs say which bound of each slab is entered first, so every (lo, hi) pair is byte-permuted into
// (near, far) once (v_perm_b32) instead of being sorted with min / max.
__device__ __forceinline__ void hz_node_hits(const NodeRay &n, const RayBox &r, float tfar, const uint4 &q,
                                             bool &h0, bool &h1, bool &h2, bool &h3) {
    const uint32_t px0 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.x, r.sel_x), px1 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.x, r.sel_x + HZ_SEL_PAIR1);
    const uint32_t py0 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.y, r.sel_y), py1 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.y, r.sel_y + HZ_SEL_PAIR1);
    const uint32_t pz0 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.z, r.sel_z), pz1 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.z, r.sel_z + HZ_SEL_PAIR1);
    const uint32_t pz2 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.w, r.sel_z), pz3 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.w, r.sel_z + HZ_SEL_PAIR1);
    const float nx0 = hz_fma_mix_lo(px0, n.ax, n.bx), fx0 = hz_fma_mix_hi(px0, n.ax, n.bx);
    const float nx1 = hz_fma_mix_lo(px1, n.ax, n.bx), fx1 = hz_fma_mix_hi(px1, n.ax, n.bx);
    const float ny0 = hz_fma_mix_lo(py0, n.ay, n.by), fy0 = hz_fma_mix_hi(py0, n.ay, n.by);
    const float ny1 = hz_fma_mix_lo(py1, n.ay, n.by), fy1 = hz_fma_mix_hi(py1, n.ay, n.by);
    // the ray's own interval [0, tfar] is folded into the two x terms, which every child shares with one other child:
    // 2 + 2 instead of 4 + 4 clamps, and one max3 / min3 per child
    const float cx0 = __builtin_fmaxf(nx0, 0.0f), cx1 = __builtin_fmaxf(nx1, 0.0f);
    const float gx0 = __builtin_fminf(fx0, tfar), gx1 = __builtin_fminf(fx1, tfar);
#define HZ_CHILD(pz, nx, fx, ny, fy, out) do { \
        const float nz_ = hz_fma_mix_lo(pz, n.az, n.bz), fz_ = hz_fma_mix_hi(pz, n.az, n.bz); \
        const float tmin_ = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz_); \
        const float tmax_ = __builtin_fminf(__builtin_fminf(fx, fy), fz_); \
        out = tmin_ <= tmax_ * 1.000001f; } while (0)
    HZ_CHILD(pz0, cx0, gx0, ny0, fy0, h0);      // slot k: column half k & 1, row half k >> 1
    HZ_CHILD(pz1, cx1, gx1, ny0, fy0, h1);
    HZ_CHILD(pz2, cx0, gx0, ny1, fy1, h2);
    HZ_CHILD(pz3, cx1, gx1, ny1, fy1, h3);
#undef HZ_CHILD
}

// The same test with the four results as lane MASKS (all ones: the child box is hit; zero: it is not) formed from the SIGN of
// tmin - tmax * slack, for the fast stack discipline: a compare plus the selects that consume it are slow-class VALU
// instructions (v_cmp_le_f32 4.2 cycles, v_cndmask_b32 with a scalar mask 4.1: profiles/r04/inst_rates.json), a fused
// multiply-add, an arithmetic shift and a logic operation fast-class ones (2.3 - 2.5).  One rounding instead of two, and a
// box whose interval has shrunk to exactly a point (tmin == tmax * slack) now counts as missed: both are far inside the slack.
__device__ __forceinline__ void hz_node_hit_masks(const NodeRay &n, const RayBox &r, float tfar, const uint4 &q,
                                                  int &m0, int &m1, int &m2, int &m3) {
    const uint32_t px0 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.x, r.sel_x), px1 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.x, r.sel_x + HZ_SEL_PAIR1);
    const uint32_t py0 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.y, r.sel_y), py1 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.y, r.sel_y + HZ_SEL_PAIR1);
    const uint32_t pz0 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.z, r.sel_z), pz1 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.z, r.sel_z + HZ_SEL_PAIR1);
    const uint32_t pz2 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.w, r.sel_z), pz3 = __builtin_amdgcn_perm(HZ_HALF_MAGIC, q.w, r.sel_z + HZ_SEL_PAIR1);
    const float nx0 = hz_fma_mix_lo(px0, n.ax, n.bx), fx0 = hz_fma_mix_hi(px0, n.ax, n.bx);
    const float nx1 = hz_fma_mix_lo(px1, n.ax, n.bx), fx1 = hz_fma_mix_hi(px1, n.ax, n.bx);
    const float ny0 = hz_fma_mix_lo(py0, n.ay, n.by), fy0 = hz_fma_mix_hi(py0, n.ay, n.by);
    const float ny1 = hz_fma_mix_lo(py1, n.ay, n.by), fy1 = hz_fma_mix_hi(py1, n.ay, n.by);
    const float cx0 = __builtin_fmaxf(nx0, 0.0f), cx1 = __builtin_fmaxf(nx1, 0.0f);
    const float gx0 = __builtin_fminf(fx0, tfar), gx1 = __builtin_fminf(fx1, tfar);
#define HZ_CHILD_M(pz, nx, fx, ny, fy, out) do { \
        const float nz_ = hz_fma_mix_lo(pz, n.az, n.bz), fz_ = hz_fma_mix_hi(pz, n.az, n.bz); \
        const float tmin_ = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz_); \
        const float tmax_ = __builtin_fminf(__builtin_fminf(fx, fy), fz_); \
        out = __float_as_int(__builtin_fmaf(tmax_, -1.000001f, tmin_)) >> 31; \
        asm volatile("" : "+v"(out));     /* (the compiler must not know that this is 0 / -1: it would turn every use back into a compare + select) */ \
        } while (0)
    HZ_CHILD_M(pz0, cx0, gx0, ny0, fy0, m0);      // slot k: column half k & 1, row half k >> 1
    HZ_CHILD_M(pz1, cx1, gx1, ny0, fy0, m1);
    HZ_CHILD_M(pz2, cx0, gx0, ny1, fy1, m2);
    HZ_CHILD_M(pz3, cx1, gx1, ny1, fy1, m3);
#undef HZ_CHILD_M
}

// One 32 B node = 2 x 16 B global loads issued back to back and waited for once.
// (hipcc was seen to put an s_waitcnt between the halves of the plain C++ form.)  Rounds 1-3 had 64 B nodes (four loads),
// the first half of round 4 48 B ones (three): the vector-memory pipe is the kernel's co-limit -- every load instruction
// per visit is worth ~4 % (+1 load: -3 %, round 3; 4 -> 3 loads: +4.2 %; a fire-and-forget prefetch load: -6 %).
__device__ __forceinline__ void hz_load_node(const Node *n, float4 &n0, uint4 &n1) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(n0), "=&v"(n1)
                 : "v"(n)
                 : "memory");
}

// The same node from the LDS nodelet (2 x ds_read_b128, one wait).
__device__ __forceinline__ void hz_load_node_lds(const float4 *q, float4 &n0, uint4 &n1) {
    const unsigned addr = (unsigned)(size_t)reinterpret_cast<const __attribute__((address_space(3))) char *>(
        (const __attribute__((address_space(3))) float4 *)q);
    asm volatile("ds_read_b128 %0, %2\n\t"
                 "ds_read_b128 %1, %2 offset:16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(n0), "=&v"(n1)
                 : "v"(addr)
                 : "memory");
}

// One 48 B leaf record = 3 x 16 B global loads, one wait.
__device__ __forceinline__ void hz_load_prim(const Prim *q, float4 &q0, float4 &q1, float4 &q2) {
    asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2)
                 : "v"(q)
                 : "memory");
}

// ---------------------------------------------------------------------------
// any-hit traversal of one ray, resumable.
//   TravState : node (current link), sp (LDS stack pointer), pf / pm (the pending siblings of the level being
//               descended, kept in registers: link of slot 0 of their block and the mask of slots still to visit;
//               pm = 0: none), up to 2 queued leaves
//   stack     : per-lane LDS stack, entry k of lane tid at stack[k * TPB + tid]; one entry per tree level = the
//               (pf, pm) pair of that level packed into 32 bits (blocks are 4-aligned and links use 30 bits, so the
//               three mask bits of slots 1..3 -- slot 0 is never pending: the first hit child is entered at once --
//               live in bits 0, 1 and 30).  At most one entry per level is alive: `height` entries cannot overflow.
//   top       : LDS copy of the first ntop nodes, read only by the NODELET instantiation (else null / 0)
//   regroup   : leave when fewer than `regroup` lanes of the wave are still traversing (and at least one
//               lane finished in this call, so the caller can refill it)
//   leaf_bias : the wave takes the node step when 16 * (lanes with a node) >= leaf_bias * (lanes with a leaf)
// Scheduling inside the wave: every lane sets leaves aside (up to QLEN) and keeps descending;
// each iteration the wave executes ONE kind of step -- the node step or the leaf step --
// whichever more lanes are ready for (ballot + popcount vote).  This keeps the 64 lanes
// busy although neighbouring rays reach their leaves at different times.
// returns 0 = miss, 1 = hit (t.lq0 is the blocking leaf), 2 = suspended (state is valid, call again)
// ---------------------------------------------------------------------------
struct TravCounters { unsigned nodes, tris, w_nodes, w_leaves;
};
struct TravState { int node, sp, pf, pm, lq0, lq1; };

__device__ __forceinline__ void hz_trav_reset(TravState &t) {
    t.node = 0; t.sp = 0; t.pf = 0; t.pm = 0; t.lq0 = HZ_EMPTY; t.lq1 = HZ_EMPTY;
}

// (pf, pm) <-> one LDS word
__device__ __forceinline__ int hz_entry_pack(int pf, int pm) {
    return (int)((unsigned)pf | (((unsigned)pm >> 1) & 3u) | (((unsigned)pm & 8u) << 27));
}
__device__ __forceinline__ void hz_entry_unpack(int e, int &pf, int &pm) {
    pm = (int)((((unsigned)e & 3u) << 1) | (((unsigned)e >> 27) & 8u));
    pf = (int)((unsigned)e & 0xbffffffcu);
}

#define HZ_WAVE_TICK(c, lane_) do { const unsigned long long m_ = __ballot(1); if ((lane_) == __ffsll((long long)m_) - 1) (c)++; } while (0)

// Two stack disciplines (LEVELSTACK), same results:
//   false: every pending sibling is its own LDS entry (up to 3 per level).  Cheapest in VALU instructions -- the kernel
//          is VALU-issue bound -- but its worst case is 3 x height entries, more LDS than five resident workgroups per
//          CU can have; the stack gets `stack_cap` entries (rays need far fewer in practice), a node step that would
//          not find 3 free entries drops the excess instead of writing out of bounds and raises `overflow`, and the
//          host repeats that launch with the other discipline.
//   true:  one entry per level as described above: `height` entries, no overflow case (+10 % VALU instructions on
//          the 3601^2 tile).
template <int TPB, bool COUNT, int QLEN = 2, bool NODELET = false, bool LEVELSTACK = true>
__device__ __forceinline__ int hz_trace(const Node *__restrict__ nodes, const Prim *__restrict__ prims,
                                        const float4 *top, int ntop, int *stack, int tid,
                                        float ox, float oy, float oz, float dx, float dy, float dz, float tfar,
                                        float tfar_box, const RayBox &rb, TravState &t, int regroup, int leaf_bias,
                                        TravCounters &cnt, int stack_cap, bool &overflow) {
    const int lane = tid & 63;
    int node = t.node, sp = t.sp, pf = t.pf, pm = t.pm, lq0 = t.lq0, lq1 = t.lq1;
    const int n_entry = __popcll(__ballot(1));   // lanes that entered with a ray
// next link.  LEVELSTACK: a pending sibling of the current level, else of the closest level above that has one (LDS);
// else: the top LDS entry -- entry 0 of every lane holds HZ_EMPTY, so a pop needs no "is the stack empty" test
#define HZ_POP() do { \
        if (LEVELSTACK) { \
            if (pm == 0) { const bool ne = sp > 0; sp = ne ? sp - 1 : 0; const int pv = stack[sp * TPB + tid]; \
                           hz_entry_unpack(ne ? pv : 0, pf, pm); } \
            const int slot_ = __builtin_ctz((unsigned)pm | 16u); \
            node = (pm != 0) ? pf + slot_ : HZ_EMPTY; pm &= pm - 1; \
        } else { \
            node = HZ_STACK_AT(sa + (unsigned)(TPB * 4)); sa -= (unsigned)(TPB * 4); \
        } } while (0)
// The fast discipline (round 4: the wave's instruction count is what the node step pays for -- any instruction, scalar ones
// included, profiles/r04/sensitivity_pads.log -- so everything below is written for the fewest instructions):
//   * `sa` is the LDS BYTE ADDRESS of the entry BELOW this lane's top entry (sa = sa0 + (sp - 1) * TPB * 4 with the LDS base
//     folded into sa0): the top two entries, the pop value of a node step and its three push slots are all `sa` + an
//     immediate offset -- a stack access is one ds instruction, no address arithmetic.  Entry 0 is a sentinel holding
//     HZ_EMPTY (sp = 0: only the sentinel, sa one row below the stack -- that row is read with the top, never used).
//     A finished ray pops the sentinel (sp = -1); its lane then holds HZ_EMPTY and never pops or pushes again, but it
//     keeps reading "the top two entries" while its queued leaves are tested: the caller keeps two rows (2 x TPB x 4
//     bytes) of LDS in front of the stack.
//   * the two top entries are read once per iteration (one ds_read2st64_b32); the two queue fills below are selects;
//   * a node step stores its three push candidates unconditionally above the top and advances `sa` by a select per
//     candidate (garbage above the top is harmless: `sa_cap` keeps three entries free) -- no exec-mask branch per push;
//   * `overflow` is a lane mask in scalar registers (what the caller needs is "any lane of the wave").
typedef __attribute__((address_space(3))) int hz_lds_int;
#define HZ_STACK_AT(addr) (*reinterpret_cast<hz_lds_int *>((size_t)(addr)))
#define HZ_SAVE() do { if (!LEVELSTACK) sp = (int)(sa - sa0) / (TPB * 4) + 1; \
                       t.node = node; t.sp = sp; t.pf = pf; t.pm = pm; t.lq0 = lq0; t.lq1 = lq1; } while (0)
    const unsigned sa0 = (unsigned)(size_t)reinterpret_cast<hz_lds_int *>(
                             (__attribute__((address_space(3))) char *)reinterpret_cast<char *>(stack)) + (unsigned)tid * 4u;
    unsigned sa = sa0 + (unsigned)(sp - 1) * (unsigned)(TPB * 4);
    const unsigned sa_cap = sa0 + (unsigned)(stack_cap - 5) * (unsigned)(TPB * 4);
    if (!LEVELSTACK) HZ_STACK_AT(sa0) = HZ_EMPTY;
    unsigned sa_hi = 0u;          // highest stack pointer a NODE STEP of this call started with (overflow is decided once, at the
                                  // exit; a lane may enter with more: leaf links above the three free entries are fine)
    // The loop is left by the WAVE: when fewer than n_leave lanes are still traversing (`regroup`, and only if some lane
    // finished its ray in this call -- it can refill, so the caller always makes progress) or when none is.  A lane whose
    // ray is decided idles until then: a miss holds HZ_EMPTY and an empty queue; a hit holds HZ_EMPTY and the NUMBER of
    // the blocking leaf instead of its link in lq0 (non-negative: "no leaf queued"; never HZ_EMPTY).  No per-lane exit, no result register in the loop.
    const int n_leave = max(min(regroup, n_entry), 1);
    // The loop is rotated by hand: [queue fills + votes] once in front of it and again at the end of its body, so that the
    // wave's exit test is the loop condition itself.
    bool can_node, can_leaf;
    unsigned long long m_node, m_leaf;
    int n_all;
    auto top_of_iteration = [&]() __attribute__((always_inline)) {
        // set leaves aside while the leaf queue (QLEN entries, filled front to back; a queued leaf is negative, an empty
        // place HZ_EMPTY) has room
        if (LEVELSTACK) {
            if (node < 0 && lq0 >= 0) { lq0 = node; HZ_POP(); }
            if (node < 0 && lq1 >= 0) { lq1 = node; HZ_POP(); }
        } else {
            // top entry and the one below it (the row below the sentinel belongs to the caller's staging buffer or padding:
            // read with the sentinel, never used)
            hz_int2 t10;
            // (second element = one stack row above: TPB * 4 bytes = TPB / 64 units of 64 dwords)
            asm volatile("ds_read2st64_b32 %0, %1 offset1:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(t10) : "v"(sa), "n"(TPB / 64) : "memory");
            int t0 = t10.y;
            const int t1 = t10.x;
            // "node is a leaf link (negative) and the place is free (HZ_EMPTY: positive)" as ONE vector compare of
            // node & ~place: scalar logic on two compare results waits for both (round 4: a scalar instruction that
            // consumes a vector compare costs about 1 % of the kernel, profiles/r04/ab_scalar_mask_logic.log)
            // (round 5: the sign of node & ~place is smeared into a lane mask and the moves are bit selects -- v_ashrrev_i32 +
            //  v_bitop3_b32, fast-class instructions, instead of a compare and v_cndmask_b32s, slow-class ones)
#define HZ_SELM(m, x, y) (((m) & (x)) | (~(m) & (y)))
            int m1 = (node & ~lq0) >> 31;
            asm volatile("" : "+v"(m1));          // (the compiler must not know that this is 0 / -1)
            lq0 = HZ_SELM(m1, node, lq0); node = HZ_SELM(m1, t0, node); t0 = HZ_SELM(m1, t1, t0); sa -= (unsigned)m1 & (unsigned)(TPB * 4);
            int m2 = (node & ~lq1) >> 31;
            asm volatile("" : "+v"(m2));
            lq1 = HZ_SELM(m2, node, lq1); node = HZ_SELM(m2, t0, node); sa -= (unsigned)m2 & (unsigned)(TPB * 4);
#undef HZ_SELM
        }
        can_node = HZ_IS_NODE(node);
        can_leaf = lq0 < 0;
        // votes taken before any lane leaves: a lane that is finished contributes to neither mask, so the
        // masks equal those of the lanes that stay (and stay plain scalar compares)
        m_node = __ballot(can_node); m_leaf = __ballot(can_leaf);
        n_all = __popcll(m_node | m_leaf);
    };
    top_of_iteration();
    while (n_all >= n_leave) {
        const int n_node = __popcll(m_node);
        const int n_leaf = __popcll(m_leaf);
        if (n_node * 16 >= n_leaf * leaf_bias) {
            // ---------------- node step ------------------------------------------------------
            if (can_node) {
                float4 n0; uint4 n1;
                // NODELET: top-of-tree nodes from LDS, the rest from global memory (two separate asm paths:
                // a per-lane pointer select would be compiled into slow flat loads).  Measured 2 % slower
                // than plain global loads -- the top of the tree is L1 resident -- so it is opt-in.
                if (NODELET && node < ntop) hz_load_node_lds(top + 2 * node, n0, n1);
                else hz_load_node(nodes + node, n0, n1);
                if (COUNT) { cnt.nodes++; HZ_WAVE_TICK(cnt.w_nodes, lane); }
                const NodeRay nr = hz_node_ray(rb, n0.x, n0.y, n0.z);
                // children are visited in slot order (quadrant order).  A front-to-back order (slot r ^ direction
                // signs, ~45 VALU per step) was measured to be a net loss: these are any-hit rays, a blocked ray
                // is blocked over a long stretch behind the first ridge, and near-horizon rays that only
                // nick a crest are served by the hit cache.  (Rounds 2-3 stored the tallest child first: +1 %; the
                // quadrant order is what lets x and y share one range per half, i.e. the 32 B node.)
                const int first = __float_as_int(n0.w);
                if (LEVELSTACK) {
                    bool h0, h1, h2, h3;
                    hz_node_hits(nr, rb, tfar_box, n1, h0, h1, h2, h3);
                    const int h = (h0 ? 1 : 0) | (h1 ? 2 : 0) | (h2 ? 4 : 0) | (h3 ? 8 : 0);
                    if (h != 0) {                    // the hit children become the pending set of this level ...
                        if (pm != 0) { stack[sp * TPB + tid] = hz_entry_pack(pf, pm); sp++; }
                        pf = first; pm = h;
                    }
                    HZ_POP();                        // ... and the first of them (or of a level above) is entered
                } else {
                    int m0, m1, m2, m3;              // lane masks: all ones = child hit (hz_node_hit_masks)
                    hz_node_hit_masks(nr, rb, tfar_box, n1, m0, m1, m2, m3);
                    // out of entries: remember the highest pointer (checked once, at the exit) and clamp
                    sa_hi = max(sa_hi, sa);
                    sa = min(sa, sa_cap);
                    // Pushes without a branch, without scalar logic and (round 5) without compares: the three candidates are
                    // always stored above the top and only kept (pointer advanced) when they were real links, i.e. when a
                    // higher slot was hit too.  The hit masks become 0 / one-entry steps (a_k = m_k & S), "a higher slot was
                    // hit" is an OR of those, "advance" an AND; the link that continues is picked with bit selects (m ? x : y
                    // = v_bfi_b32); the lane without a hit child continues with the top entry pv and steps down.
                    const unsigned S = (unsigned)(TPB * 4);
                    const int pv = HZ_STACK_AT(sa + S);
                    const unsigned a0 = (unsigned)m0 & S, a1 = (unsigned)m1 & S, a2 = (unsigned)m2 & S, a3 = (unsigned)m3 & S;
                    const unsigned o32 = a3 | a2, o321 = o32 | a1, o3210 = o321 | a0;
#define HZ_SEL(m, x, y) (((m) & (x)) | (~(m) & (y)))
                    int next = HZ_SEL(m3, first + 3, pv);
                    HZ_STACK_AT(sa + 2u * S) = next; sa += a2 & a3;   next = HZ_SEL(m2, first + 2, next);
                    HZ_STACK_AT(sa + 2u * S) = next; sa += a1 & o32;  next = HZ_SEL(m1, first + 1, next);
                    HZ_STACK_AT(sa + 2u * S) = next; sa += a0 & o321; next = HZ_SEL(m0, first, next);
#undef HZ_SEL
                    node = next; sa = sa + o3210 - S;
                }
            }
        } else {
            // ---------------- leaf step: the two triangles of a DEM quad (or one TIN triangle) ---
            if (can_leaf) {
                float4 q0, q1, q2;
                hz_load_prim(prims + HZ_LEAF_ID(lq0), q0, q1, q2);
                // a = (q0.x q0.y q0.z) b = (q0.w q1.x q1.y) c = (q1.z q1.w q2.x) d = (q2.y q2.z q2.w)
                if (COUNT) { cnt.tris += (q2.y == q2.y) ? 2 : 1; HZ_WAVE_TICK(cnt.w_leaves, lane); }
                const bool hit = hz_quad_hit(ox, oy, oz, dx, dy, dz, tfar, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                                             q2.x, q2.y, q2.z, q2.w, q2.y == q2.y);
                if (hit) { lq0 = HZ_LEAF_ID(lq0); node = HZ_EMPTY; }     // decided: blocked by this leaf
                else { lq0 = lq1; lq1 = HZ_EMPTY; }
            }
        }
        top_of_iteration();
    }
    if (!LEVELSTACK && sa_hi > sa_cap) overflow = true;
    const bool blocked = lq0 >= 0 && lq0 != HZ_EMPTY;
    const int res = blocked ? 1 : ((HZ_IS_NODE(node) || lq0 < 0) ? 2 : 0);
    if (blocked) lq0 = (int)((unsigned)lq0 | 0x80000000u);      // the caller reads the blocking leaf (as a link) from the state
    HZ_SAVE();
    return res;
#undef HZ_POP
#undef HZ_SAVE
#undef HZ_STACK_AT
}

// ---------------------------------------------------------------------------
// closest hit of one ray (rtcIntersect1 semantics): the minimum t over ALL triangles accepted
// with the caller's tfar, so the result does not depend on the visiting order; boxes are pruned
// conservatively against the best t so far.  Plain per-lane loop (used for a handful of rays
// per location, not the throughput path).  Returns true and *dist when anything was hit.
// The stack holds individual child links here: up to 3 per level.
// ---------------------------------------------------------------------------
template <int TPB>
__device__ __forceinline__ bool hz_closest(const Node *__restrict__ nodes, const Prim *__restrict__ prims,
                                           int *stack, int tid, float ox, float oy, float oz, float dx,
                                           float dy, float dz, float tfar, float tau, const RayBox &rb, float *dist) {
    int node = 0, sp = 0;
    float best = __builtin_inff();
    bool any = false;
    while (node != HZ_EMPTY) {
        if (node >= 0) {
            float4 n0; uint4 n1;
            hz_load_node(nodes + node, n0, n1);
            // (`rb` is the frame of the origin shifted back by tau: the box tests run over [-tau, tf + tau] of the ray)
            const float tf = (any ? __builtin_fminf(tfar, best * 1.0001f) : tfar) + 2.0f * tau;
            const NodeRay nr = hz_node_ray(rb, n0.x, n0.y, n0.z);
            bool h0, h1, h2, h3;
            hz_node_hits(nr, rb, tf, n1, h0, h1, h2, h3);
            const int first = __float_as_int(n0.w);
            const int c0 = first, c1 = first + 1, c2 = first + 2, c3 = first + 3;
            int next = HZ_EMPTY;
            if (h3) next = c3;
            if (h2) { if (next != HZ_EMPTY) { stack[sp * TPB + tid] = next; sp++; } next = c2; }
            if (h1) { if (next != HZ_EMPTY) { stack[sp * TPB + tid] = next; sp++; } next = c1; }
            if (h0) { if (next != HZ_EMPTY) { stack[sp * TPB + tid] = next; sp++; } next = c0; }
            if (next != HZ_EMPTY) { node = next; continue; }
        } else {
            float4 q0, q1, q2;
            hz_load_prim(prims + HZ_LEAF_ID(node), q0, q1, q2);
            float t;
            if (hz_tri_hit_t(ox, oy, oz, dx, dy, dz, tfar, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, &t)) {
                any = true; best = __builtin_fminf(best, t);
            }
            if ((q2.y == q2.y) &&
                hz_tri_hit_t(ox, oy, oz, dx, dy, dz, tfar, q0.w, q1.x, q1.y, q2.y, q2.z, q2.w, q1.z, q1.w, q2.x, &t)) {
                any = true; best = __builtin_fminf(best, t);
            }
        }
        if (sp > 0) { sp--; node = stack[sp * TPB + tid]; } else node = HZ_EMPTY;
    }
    if (any) *dist = best;
    return any;
}

#endif  // __HIPCC__
