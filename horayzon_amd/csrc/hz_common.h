// hz_common.h -- shared host/device definitions for libhorayzon_hip (gfx950).
//
// Scene blob layout in HBM (one contiguous, position-independent allocation):
//
//   [ BlobHeader (256 B) | vertices f32[3*V] | nodes Node[max(P-1,1)] | prims Prim[P] ]
//
// vertices : the caller's vert_grid, untouched (ray origins read them;
//            reference: shared vertex buffer, horizon_comp.cpp:126-127).
// nodes    : flat LBVH, 64 B per internal node holding BOTH child AABBs, so one
//            fetch decides both children.  The first `n_top` nodes are the top
//            of the tree in breadth-first order (staged in LDS by the kernels).
//            AABBs live in a frame centred on the scene (`center`) and are
//            padded by `pad`, which makes the box test conservative with
//            respect to the float32 triangle test: hit decisions depend on the
//            triangle test only, never on the tree.
// prims    : one 48 B record per leaf in Morton order = the 4 corner vertices
//            of a DEM quad (two triangles a,b,c / b,d,c -- the split of
//            horizon_comp.cpp:139-151) or the 3 vertices of a TIN triangle
//            (d.x = NaN), raw coordinates.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define HZ_BLOB_MAGIC 0x485a4c42u /* "HZLB" */
#define HZ_BLOB_VERSION 2u
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
    uint8_t reserved[256 - 120];
};
static_assert(sizeof(BlobHeader) == 256, "BlobHeader must be 256 bytes");

// child >= 0: internal node index; child < 0: leaf, prim index = ~child
struct __attribute__((aligned(64))) Node {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t c0, c1;
    int32_t pad_[2];
};
static_assert(sizeof(Node) == 64, "Node must be 64 bytes");

struct __attribute__((aligned(16))) Prim {
    float a[3], b[3], c[3], d[3];
};
static_assert(sizeof(Prim) == 48, "Prim must be 48 bytes");

#ifdef __HIPCC__

// ---------------------------------------------------------------------------
// ray / triangle: Embree-robust style Pluecker edge test in float32.
// This translation unit is compiled with -ffp-contract=off: every multiply
// and add below rounds separately, in exactly this association order.  The
// hit decision is the product's numerical contract.
// ---------------------------------------------------------------------------
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
    const float c0x = e0y * s0z - e0z * s0y;
    const float c0y = e0z * s0x - e0x * s0z;
    const float c0z = e0x * s0y - e0y * s0x;
    const float c1x = e1y * s1z - e1z * s1y;
    const float c1y = e1z * s1x - e1x * s1z;
    const float c1z = e1x * s1y - e1y * s1x;
    const float c2x = e2y * s2z - e2z * s2y;
    const float c2y = e2z * s2x - e2x * s2z;
    const float c2z = e2x * s2y - e2y * s2x;
    const float U = (c0x * dx + c0y * dy) + c0z * dz;
    const float V = (c1x * dx + c1y * dy) + c1z * dz;
    const float W = (c2x * dx + c2y * dy) + c2z * dz;
    const float UVW = (U + V) + W;
    const float eps = 1.1920928955078125e-7f * __builtin_fabsf(UVW);
    const float mn = __builtin_fminf(U, __builtin_fminf(V, W));
    const float mx = __builtin_fmaxf(U, __builtin_fmaxf(V, W));
    if (!((mn >= -eps) || (mx <= eps))) return false;
    const float nx = e1y * e0z - e1z * e0y;
    const float ny = e1z * e0x - e1x * e0z;
    const float nz = e1x * e0y - e1y * e0x;
    const float den = (nx * dx + ny * dy) + nz * dz;
    const float T = (v0x * nx + v0y * ny) + v0z * nz;
    if (den == 0.0f) return false;
    const float Ts = (den < 0.0f) ? -T : T;
    const float ad = __builtin_fabsf(den);
    if (!(Ts >= 0.0f)) return false;
    if (!(Ts <= tfar * ad)) return false;
    return true;
}

// ---------------------------------------------------------------------------
// per-ray constants for the conservative slab test (centred frame)
// ---------------------------------------------------------------------------
struct RayBox {
    float rdx, rdy, rdz;     // 1 / d (clamped away from inf)
    float ordx, ordy, ordz;  // (o - center) * rd
};

__device__ __forceinline__ float hz_safe_rcp(float d) {
    return (__builtin_fabsf(d) > 1e-30f) ? 1.0f / d : __builtin_copysignf(1e30f, d);
}

__device__ __forceinline__ RayBox hz_raybox(float ocx, float ocy, float ocz,
                                            float dx, float dy, float dz) {
    RayBox r;
    r.rdx = hz_safe_rcp(dx); r.rdy = hz_safe_rcp(dy); r.rdz = hz_safe_rcp(dz);
    r.ordx = ocx * r.rdx; r.ordy = ocy * r.rdy; r.ordz = ocz * r.rdz;
    return r;
}

// entry distance in *tn; returns whether [0, tfar] overlaps the box
__device__ __forceinline__ bool hz_box_hit(const RayBox &r, float tfar,
                                           float lox, float loy, float loz,
                                           float hix, float hiy, float hiz, float *tn) {
    const float t0x = __builtin_fmaf(lox, r.rdx, -r.ordx), t1x = __builtin_fmaf(hix, r.rdx, -r.ordx);
    const float t0y = __builtin_fmaf(loy, r.rdy, -r.ordy), t1y = __builtin_fmaf(hiy, r.rdy, -r.ordy);
    const float t0z = __builtin_fmaf(loz, r.rdz, -r.ordz), t1z = __builtin_fmaf(hiz, r.rdz, -r.ordz);
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                       __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                                       __builtin_fminf(__builtin_fmaxf(t0z, t1z), tfar));
    *tn = tmin;
    return tmin <= tmax * 1.000001f;
}

// One 64 B node = 4 x 16 B global loads issued back to back and waited for once.
// (hipcc was seen to put an s_waitcnt between the halves of the plain C++ form.)
__device__ __forceinline__ void hz_load_node(const Node *n, float4 &n0, float4 &n1, float4 &n2, int2 &ch) {
    float4 t3;
    asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(t3)
                 : "v"(n)
                 : "memory");
    ch.x = __float_as_int(t3.x); ch.y = __float_as_int(t3.y);
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

#endif  // __HIPCC__
