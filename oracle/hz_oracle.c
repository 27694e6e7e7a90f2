/*
 * hz_oracle.c -- CPU restatement of HORAYZON's terrain-horizon / shadow ray casting.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (horayzon_amd/, the
 * C-ABI library) may include, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference delegates every hit/no-hit decision to Intel
 * Embree 4 (conda-forge `embree`, version not pinned by the reference;
 * horizon_comp.cpp:5, shadow_comp.cpp:6), which is not present in
 * /root/reference and not installable here; the reference ships no tests or
 * golden vectors for this path.  This file restates
 *   - the reference's OWN code statement by statement, including its
 *     float/double promotion pattern:
 *       helpers            horizon_comp.cpp:26-62, shadow_comp.cpp:41-159
 *       mesh topology      horizon_comp.cpp:139-151 (2 triangles per quad)
 *       query semantics    horizon_comp.cpp:241-262 (any-hit, tnear=0, tfar)
 *       search algorithms  horizon_comp.cpp:302-333, 339-381, 387-498
 *       gridded driver     horizon_comp.cpp:629-822
 *       shadow / sw_dir_cor shadow_comp.cpp:318-380, 386-491, 495-605
 *   - Embree's published ROBUST-mode triangle test (Pluecker edge functions
 *     with an ulp-relative tolerance, two-sided, depth test on the plane hit)
 *     in plain IEEE float32 without FMA contraction.  Embree evaluates the
 *     same algebra with FMA/rcp on SIMD lanes, so individual borderline rays
 *     may differ from Embree; that cannot be checked here.
 * What IS pinned: analytic known answers (tests/), a double-precision
 * brute-force intersector (mode 2 below), and three independent acceleration
 * paths (brute force, this file's index-rectangle BVH, the GPU's Morton LBVH)
 * that must agree on every hit decision.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
/* The five float libm calls of the refraction branch (shadow_comp.cpp:135-159, :430-446) go through
 * hz_crmath.h: self-contained, correctly rounded, shared with the HIP kernels (the platform's float routines
 * are not a fixed target -- see that header).  orc_set_libm(1) switches this file to the platform's
 * acosf / tanf / powf / cosf / sinf so that a test can measure how much of the output depends on them. */
#include "../horayzon_amd/csrc/hz_crmath.h"
static int g_platform_libm = 0;
void orc_set_libm(int platform) { g_platform_libm = platform; }
static int g_quad_order = 0;      /* 1: second triangle of a quad as Embree's quad / grid intersector orders it */
void orc_set_quad_order(int embree_quad) { g_quad_order = embree_quad; }
/* Where the tree's box tests start and end (round 5): over [-tau, tfar + tau] of the ray, tau =
 * ORC_BOX_START_PADS * pad, NOT over [0, tfar].  The float triangle test can accept a crossing that in exact arithmetic lies
 * a little behind the origin (or beyond tfar) when the ray grazes the triangle's plane (T cancels to rounding noise); a tree
 * that starts its box tests at exactly 0 then disagrees with brute force (adversarial sweep seed 48001, configuration 2536).
 * The HIP kernels do the same with the same tau (hz_common.h: HZ_BOX_START_PADS).  orc_set_box_start(0) restores the
 * round-4 behaviour for the test that pins the counter-example. */
#define ORC_BOX_START_PADS 16.0f
static float g_box_start_pads = ORC_BOX_START_PADS;
void orc_set_box_start(float tau_pads) { g_box_start_pads = tau_pads; }
/* mode 0: hz_crmath.h (shared with the HIP kernels: GPU = oracle by construction); 1: the platform's float routines (what the
 * reference calls); 2: THE ORACLE'S OWN evaluation of the same contract -- the correctly rounded float -- through the x87
 * long double libm (64-bit mantissa, < 1 ulp of that: the rounding to float is correct unless the exact value lies within
 * ~1e-19 relative of a rounding boundary).  Mode 2 shares no code with the kernels: tests/test_gpu_parity.py::
 * test_refraction_against_the_oracles_own_functions compares the GPU with it. */
static inline float r_acosf(float x) { return g_platform_libm == 2 ? (float)acosl((long double)x) : g_platform_libm ? acosf(x) : hz_crm_acosf(x); }
static inline float r_tanf(float x) { return g_platform_libm == 2 ? (float)tanl((long double)x) : g_platform_libm ? tanf(x) : hz_crm_tanf(x); }
static inline float r_cosf(float x) { return g_platform_libm == 2 ? (float)cosl((long double)x) : g_platform_libm ? cosf(x) : hz_crm_cosf(x); }
static inline float r_sinf(float x) { return g_platform_libm == 2 ? (float)sinl((long double)x) : g_platform_libm ? sinf(x) : hz_crm_sinf(x); }
static inline float r_powf(float x, float y) { return g_platform_libm == 2 ? (float)powl((long double)x, (long double)y) : g_platform_libm ? powf(x, y) : hz_crm_powf(x, y); }
/* exhaustive comparison of hz_crmath.h with (float) of the platform's float64 routine over the float range
 * [lo, hi] (which: 0 acos, 1 tan, 2 cos, 3 sin, 4 pow(x, y)); out[0] = values, out[1] = differing from the
 * rounded float64 result, out[2] = differing from the platform's float routine */
void orc_crmath_sweep(int which, float lo, float hi, float y, uint64_t *out) {
    uint32_t a, b;
    memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
    uint64_t n = 0, bad_d = 0, bad_f = 0;
#pragma omp parallel for reduction(+ : n, bad_d, bad_f)
    for (int64_t u = (int64_t)a; u <= (int64_t)b; u++) {
        const uint32_t uu = (uint32_t)u;
        float x, rc, rd, rf;
        memcpy(&x, &uu, 4);
        switch (which) {
            case 0: rc = hz_crm_acosf(x); rd = (float)acos((double)x); rf = acosf(x); break;
            case 1: rc = hz_crm_tanf(x); rd = (float)tan((double)x); rf = tanf(x); break;
            case 2: rc = hz_crm_cosf(x); rd = (float)cos((double)x); rf = cosf(x); break;
            case 3: rc = hz_crm_sinf(x); rd = (float)sin((double)x); rf = sinf(x); break;
            default: rc = hz_crm_powf(x, y); rd = (float)pow((double)x, (double)y); rf = powf(x, y); break;
        }
        n++;
        if (memcmp(&rc, &rd, 4) != 0 && !(rc != rc && rd != rd)) bad_d++;
        if (memcmp(&rc, &rf, 4) != 0 && !(rc != rc && rf != rf)) bad_f++;
    }
    out[0] = n; out[1] = bad_d; out[2] = bad_f;
}
/* hz_crm_div_const(x, c, RN(1 / c)) (hz_crmath.h: what the HIP kernels evaluate for `/ 180.0` and `/ M_PI`) against the IEEE
 * division, for every float x whose bit pattern lies in [lo_bits, hi_bits] promoted to double; out[0] = values, out[1] = results
 * that differ (NaN = NaN) */
void orc_div_const_sweep(double c, uint32_t lo_bits, uint32_t hi_bits, uint64_t *out) {
    const double rc = 1.0 / c;
    uint64_t n = 0, bad = 0;
#pragma omp parallel for reduction(+ : n, bad)
    for (int64_t u = (int64_t)lo_bits; u <= (int64_t)hi_bits; u++) {
        const uint32_t uu = (uint32_t)u;
        float xf;
        memcpy(&xf, &uu, 4);
        const double x = (double)xf;
        const double a = hz_crm_div_const(x, c, rc), b = x / c;
        n++;
        if (memcmp(&a, &b, 8) != 0 && !(a != a && b != b)) bad++;
    }
    out[0] = n; out[1] = bad;
}
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------- */
/* helpers (horizon_comp.cpp:26-62)                                           */
/* ------------------------------------------------------------------------- */

/* horizon_comp.cpp:37-39: float in, double arithmetic, float out */
static inline float deg2rad_f(float ang) {
    return (float)(((double)ang / 180.0) * M_PI);
}
/* shadow_comp.cpp:53-62 */
static inline float rad2deg_f(float ang) {
    return (float)(((double)ang / M_PI) * 180.0);
}
/* shadow_comp.cpp:65-74 */
static inline float K2degC_f(float temp) {
    return (float)((double)temp - 273.15);
}

/* ------------------------------------------------------------------------- */
/* scene                                                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    float lo[3], hi[3];
    int left, right;      /* internal: child node indices; leaf: left = -1   */
    int i0, j0;           /* leaf: first quad row / column                   */
    int ni, nj;           /* leaf: quad rows / columns (<= 2 x 2)            */
} gnode;

typedef struct {
    float lo[3], hi[3];
    int left, right;      /* internal; leaf: left = -1                       */
    int first, count;     /* leaf: range in tri_order                        */
} tnode;

typedef struct {
    int d0, d1;
    float *vert;          /* 3 * d0 * d1, private copy                       */
    int nvs, nts;
    float *vs;            /* simplified outer TIN vertices                   */
    int *ts;              /* TIN indices                                     */
    gnode *gn; int n_gn;
    tnode *tn; int n_tn; int *tri_order;
    float pad;            /* conservative AABB padding                       */
    /* counters (not thread safe; only filled by the *_counted entry points) */
} orc_scene;

typedef struct {
    uint64_t nodes;       /* AABB pairs / nodes popped                       */
    uint64_t tris;        /* triangle tests                                  */
} orc_counters;

static inline const float *vtx(const orc_scene *s, int i, int j) {
    return s->vert + 3 * ((size_t)i * (size_t)s->d1 + (size_t)j);
}

/* ------------------------------------------------------------------------- */
/* ray / triangle: Embree-robust style Pluecker test, float32, no FMA         */
/* ------------------------------------------------------------------------- */

/* returns 1 if the ray (o, d, [0, tfar]) hits triangle (p0, p1, p2)          */
/* threshold of the parallel test of tri_hit_plain, in units of the sum of the absolute products (hz_common.h: HZ_DEN_NOISE);
 * orc_set_den_noise(0) restores the exact `den != 0` of rounds 1-4 for the test that pins the counter-example */
#define ORC_DEN_NOISE 9.5367431640625e-07f      /* 2^-20 */
static float g_den_noise = ORC_DEN_NOISE;
void orc_set_den_noise(float k) { g_den_noise = k; }
static inline int tri_hit_plain(const float *o, const float *d, float tfar,
                                const float *p0, const float *p1, const float *p2) {
    const float v0x = p0[0] - o[0], v0y = p0[1] - o[1], v0z = p0[2] - o[2];
    const float v1x = p1[0] - o[0], v1y = p1[1] - o[1], v1z = p1[2] - o[2];
    const float v2x = p2[0] - o[0], v2y = p2[1] - o[1], v2z = p2[2] - o[2];
    const float e0x = v2x - v0x, e0y = v2y - v0y, e0z = v2z - v0z;
    const float e1x = v0x - v1x, e1y = v0y - v1y, e1z = v0z - v1z;
    const float e2x = v1x - v2x, e2y = v1y - v2y, e2z = v1z - v2z;
    /* U = dot(cross(e0, v2 + v0), d) etc. */
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
    const float U = (c0x * d[0] + c0y * d[1]) + c0z * d[2];
    const float V = (c1x * d[0] + c1y * d[1]) + c1z * d[2];
    const float W = (c2x * d[0] + c2y * d[1]) + c2z * d[2];
    const float UVW = (U + V) + W;
    const float eps = FLT_EPSILON * fabsf(UVW);
    const float mn = fminf(U, fminf(V, W));
    const float mx = fmaxf(U, fmaxf(V, W));
    if (!((mn >= -eps) || (mx <= eps))) return 0;
    /* geometric normal Ng = e1 x e0, plane hit t = dot(v0, Ng) / dot(d, Ng) */
    const float nx = e1y * e0z - e1z * e0y;
    const float ny = e1z * e0x - e1x * e0z;
    const float nz = e1x * e0y - e1y * e0x;
    const float pnx = nx * d[0], pny = ny * d[1], pnz = nz * d[2];
    const float den = (pnx + pny) + pnz;
    const float T = (v0x * nx + v0y * ny) + v0z * nz;
    /* "den != 0" with the rounding of den taken into account (round 5): a ray that lies IN the triangle's plane to within the
     * rounding of this sum (|den| <= 2^-20 (|nx dx| + |ny dy| + |nz dz|)) is parallel to it.  With such a den all three edge
     * functions are rounding noise as well -- coplanar lines meet somewhere -- and the test would accept a triangle the ray
     * passes kilometres beside (CPU sweep seed 51001, adversarial configuration 580: integer terrace heights, ray_org_elev
     * 2.0 m; a "hit at t = 0" of a triangle 12.6 km away that no tree can reproduce).  DESIGN.md section 4 item 3. */
    const float den_sum = (fabsf(pnx) + fabsf(pny)) + fabsf(pnz);
    if (!(fabsf(den) > g_den_noise * den_sum)) return 0;
    /* 0 <= T / den <= tfar without a division */
    const float Ts = (den < 0.0f) ? -T : T;
    const float ad = fabsf(den);
    if (!(Ts >= 0.0f)) return 0;
    if (!(Ts <= tfar * ad)) return 0;
    return 1;
}

/* ------------------------------------------------------------------------- */
/* SENSITIVITY VARIANTS of the triangle test (diagnostics only, never the contract).                       */
/* The reference's hit decisions are Embree's (horizon_comp.cpp:106 ROBUST flag, :258 rtcOccluded1;         */
/* shadow_comp.cpp:466, :576); Embree is not available here, so how far its evaluation can sit from        */
/* tri_hit_plain is BOUNDED by running the same workloads with the evaluations Embree plausibly uses:       */
/*   mode 1 "embree_fma_rcp": the Pluecker test of Embree's robust intersector as its SIMD code evaluates  */
/*           it on an FMA machine -- cross(a, b) = (msub(a.y, b.z, a.z b.y), ...), dot(a, b) = madd(a.x,   */
/*           b.x, madd(a.y, b.y, a.z b.z)), the geometric normal by `stable_triangle_normal` (per          */
/*           component the cross product of the edge pair with the smaller products), den and T doubled,   */
/*           depth test 0 <= rcp(den) T <= tfar with rcp = hardware reciprocal estimate + one Newton step  */
/*           (restated from the published Embree 3.13 / 4 sources, kernels/geometry/                        */
/*           triangle_intersector_pluecker.h and common/math/vec3.h, from memory: that tree is not in the  */
/*           image);                                                                                        */
/*   mode 2 "moeller_trumbore": the classic Moeller-Trumbore test north_star names, two-sided, float32, no */
/*           FMA, u / v / t through one reciprocal of the determinant, no epsilon other than det == 0.     */
/* orc_set_tri_mode selects the test every query uses; orc_set_tri_compare(m) keeps the shipped decisions   */
/* but ALSO evaluates every ray with mode m and counts the rays whose decision differs.                     */
/* ------------------------------------------------------------------------- */
#include <xmmintrin.h>
static inline float rcp_nr(float a) {          /* Embree rcp(): rcpss estimate, one Newton-Raphson step (AVX2 form) */
    const float r = _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(a)));
    return __builtin_fmaf(r, __builtin_fmaf(-a, r, 1.0f), r);
}
#define MSUB(a, b, c) __builtin_fmaf((a), (b), -(c))
#define MADD(a, b, c) __builtin_fmaf((a), (b), (c))
static inline int tri_hit_embree_fma(const float *o, const float *d, float tfar,
                                     const float *p0, const float *p1, const float *p2) {
    const float v0[3] = {p0[0] - o[0], p0[1] - o[1], p0[2] - o[2]};
    const float v1[3] = {p1[0] - o[0], p1[1] - o[1], p1[2] - o[2]};
    const float v2[3] = {p2[0] - o[0], p2[1] - o[1], p2[2] - o[2]};
    float e0[3], e1[3], e2[3], s0[3], s1[3], s2[3];
    for (int k = 0; k < 3; k++) {
        e0[k] = v2[k] - v0[k]; e1[k] = v0[k] - v1[k]; e2[k] = v1[k] - v2[k];
        s0[k] = v2[k] + v0[k]; s1[k] = v0[k] + v1[k]; s2[k] = v1[k] + v2[k];
    }
#define CROSS(r, a, b) do { (r)[0] = MSUB((a)[1], (b)[2], (a)[2] * (b)[1]); (r)[1] = MSUB((a)[2], (b)[0], (a)[0] * (b)[2]); \
                            (r)[2] = MSUB((a)[0], (b)[1], (a)[1] * (b)[0]); } while (0)
#define DOT(a, b) MADD((a)[0], (b)[0], MADD((a)[1], (b)[1], (a)[2] * (b)[2]))
    float c0[3], c1[3], c2[3];
    CROSS(c0, e0, s0); CROSS(c1, e1, s1); CROSS(c2, e2, s2);
    const float U = DOT(c0, d), V = DOT(c1, d), W = DOT(c2, d);
    const float UVW = (U + V) + W;
    const float eps = FLT_EPSILON * fabsf(UVW);
    const float mn = fminf(U, fminf(V, W)), mx = fmaxf(U, fmaxf(V, W));
    if (!((mn >= -eps) || (mx <= eps))) return 0;
    /* stable_triangle_normal(e0, e1, e2): per component cross(e0, e1) or cross(e1, e2), whichever has the smaller
     * subtracted product (both are the same vector in exact arithmetic: -(e1 x e0)) */
    float ng[3];
    {
        const float ab_x = e0[2] * e1[1], ab_y = e0[0] * e1[2], ab_z = e0[1] * e1[0];
        const float bc_x = e1[2] * e2[1], bc_y = e1[0] * e2[2], bc_z = e1[1] * e2[0];
        const float cab[3] = {MSUB(e0[1], e1[2], ab_x), MSUB(e0[2], e1[0], ab_y), MSUB(e0[0], e1[1], ab_z)};
        const float cbc[3] = {MSUB(e1[1], e2[2], bc_x), MSUB(e1[2], e2[0], bc_y), MSUB(e1[0], e2[1], bc_z)};
        ng[0] = (fabsf(ab_x) < fabsf(bc_x)) ? cab[0] : cbc[0];
        ng[1] = (fabsf(ab_y) < fabsf(bc_y)) ? cab[1] : cbc[1];
        ng[2] = (fabsf(ab_z) < fabsf(bc_z)) ? cab[2] : cbc[2];
    }
    const float dn = DOT(ng, d), tn = DOT(v0, ng);
    const float den = dn + dn, T = tn + tn;            /* twice() */
    if (den == 0.0f) return 0;
    const float t = rcp_nr(den) * T;
    return (0.0f <= t) && (t <= tfar);
#undef CROSS
#undef DOT
}
#undef MSUB
#undef MADD

static inline int tri_hit_mt(const float *o, const float *d, float tfar,
                             const float *p0, const float *p1, const float *p2) {
    const float e1x = p1[0] - p0[0], e1y = p1[1] - p0[1], e1z = p1[2] - p0[2];
    const float e2x = p2[0] - p0[0], e2y = p2[1] - p0[1], e2z = p2[2] - p0[2];
    const float px = d[1] * e2z - d[2] * e2y, py = d[2] * e2x - d[0] * e2z, pz = d[0] * e2y - d[1] * e2x;
    const float det = (e1x * px + e1y * py) + e1z * pz;
    if (det == 0.0f) return 0;
    const float inv = 1.0f / det;
    const float tx = o[0] - p0[0], ty = o[1] - p0[1], tz = o[2] - p0[2];
    const float u = ((tx * px + ty * py) + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    const float v = ((d[0] * qx + d[1] * qy) + d[2] * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return 0;
    const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
    return (t >= 0.0f) && (t <= tfar);
}

/*   mode 3 "plain_fma": tri_hit_plain with the cross and dot products evaluated the way Embree's vector code does on an
 *           FMA machine (cross(a, b).x = msub(a.y, b.z, a.z b.y); dot(a, b) = madd(a.x, b.x, madd(a.y, b.y, a.z b.z))) and
 *           everything else (normal e1 x e0, robust parallel test, division-free depth test) as in tri_hit_plain.  This is the
 *           CPU side of the product's build-time switch -DHZ_TRI_FMA (hz_common.h): a library built with it must agree with
 *           this mode bit for bit (tests/test_gpu_tri_fma.py).  Not the contract: 37 fewer VALU instructions per leaf step,
 *           priced in DESIGN.md section 5; adopting it is a flag and a re-validation should the Embree pin ask for it.  */
static inline int tri_hit_plain_fma(const float *o, const float *d, float tfar,
                                    const float *p0, const float *p1, const float *p2) {
#define MSUB(a, b, c) __builtin_fmaf((a), (b), -(c))
#define MADD(a, b, c) __builtin_fmaf((a), (b), (c))
    const float v0x = p0[0] - o[0], v0y = p0[1] - o[1], v0z = p0[2] - o[2];
    const float v1x = p1[0] - o[0], v1y = p1[1] - o[1], v1z = p1[2] - o[2];
    const float v2x = p2[0] - o[0], v2y = p2[1] - o[1], v2z = p2[2] - o[2];
    const float e0x = v2x - v0x, e0y = v2y - v0y, e0z = v2z - v0z;
    const float e1x = v0x - v1x, e1y = v0y - v1y, e1z = v0z - v1z;
    const float e2x = v1x - v2x, e2y = v1y - v2y, e2z = v1z - v2z;
    const float s0x = v2x + v0x, s0y = v2y + v0y, s0z = v2z + v0z;
    const float s1x = v0x + v1x, s1y = v0y + v1y, s1z = v0z + v1z;
    const float s2x = v1x + v2x, s2y = v1y + v2y, s2z = v1z + v2z;
    const float c0x = MSUB(e0y, s0z, e0z * s0y), c0y = MSUB(e0z, s0x, e0x * s0z), c0z = MSUB(e0x, s0y, e0y * s0x);
    const float c1x = MSUB(e1y, s1z, e1z * s1y), c1y = MSUB(e1z, s1x, e1x * s1z), c1z = MSUB(e1x, s1y, e1y * s1x);
    const float c2x = MSUB(e2y, s2z, e2z * s2y), c2y = MSUB(e2z, s2x, e2x * s2z), c2z = MSUB(e2x, s2y, e2y * s2x);
    const float U = MADD(c0x, d[0], MADD(c0y, d[1], c0z * d[2]));
    const float V = MADD(c1x, d[0], MADD(c1y, d[1], c1z * d[2]));
    const float W = MADD(c2x, d[0], MADD(c2y, d[1], c2z * d[2]));
    const float UVW = (U + V) + W;
    const float eps = FLT_EPSILON * fabsf(UVW);
    const float mn = fminf(U, fminf(V, W));
    const float mx = fmaxf(U, fmaxf(V, W));
    if (!((mn >= -eps) || (mx <= eps))) return 0;
    const float nx = MSUB(e1y, e0z, e1z * e0y), ny = MSUB(e1z, e0x, e1x * e0z), nz = MSUB(e1x, e0y, e1y * e0x);
    const float den = MADD(nx, d[0], MADD(ny, d[1], nz * d[2]));
    const float T = MADD(v0x, nx, MADD(v0y, ny, v0z * nz));
    const float den_sum = MADD(fabsf(nx), fabsf(d[0]), MADD(fabsf(ny), fabsf(d[1]), fabsf(nz) * fabsf(d[2])));
    if (!(fabsf(den) > g_den_noise * den_sum)) return 0;
    const float Ts = (den < 0.0f) ? -T : T;
    const float ad = fabsf(den);
    if (!(Ts >= 0.0f)) return 0;
    if (!(Ts <= tfar * ad)) return 0;
    return 1;
#undef MSUB
#undef MADD
}

static int g_tri_mode = 0, g_tri_compare = -1;
static _Thread_local int tl_tri_mode = 0;
static uint64_t g_cmp_rays = 0, g_cmp_flips = 0;
void orc_set_tri_mode(int mode) { g_tri_mode = mode; }
void orc_set_tri_compare(int mode) { g_tri_compare = mode; g_cmp_rays = 0; g_cmp_flips = 0; }
void orc_tri_compare_counts(uint64_t *out) { out[0] = g_cmp_rays; out[1] = g_cmp_flips; }

static inline int tri_hit_f(const float *o, const float *d, float tfar,
                            const float *p0, const float *p1, const float *p2) {
    if (tl_tri_mode == 1) return tri_hit_embree_fma(o, d, tfar, p0, p1, p2);
    if (tl_tri_mode == 2) return tri_hit_mt(o, d, tfar, p0, p1, p2);
    if (tl_tri_mode == 3) return tri_hit_plain_fma(o, d, tfar, p0, p1, p2);
    return tri_hit_plain(o, d, tfar, p0, p1, p2);
}

/* closest-hit variant (rtcIntersect1, horizon_comp.cpp:268-292): same acceptance test, and the
 * hit distance t = T / den as one IEEE float division.  Returns 1 and *t when accepted.           */
static inline int tri_hit_t_f(const float *o, const float *d, float tfar,
                              const float *p0, const float *p1, const float *p2, float *t) {
    const float v0x = p0[0] - o[0], v0y = p0[1] - o[1], v0z = p0[2] - o[2];
    const float v1x = p1[0] - o[0], v1y = p1[1] - o[1], v1z = p1[2] - o[2];
    const float v2x = p2[0] - o[0], v2y = p2[1] - o[1], v2z = p2[2] - o[2];
    const float e0x = v2x - v0x, e0y = v2y - v0y, e0z = v2z - v0z;
    const float e1x = v0x - v1x, e1y = v0y - v1y, e1z = v0z - v1z;
    const float nx = e1y * e0z - e1z * e0y;
    const float ny = e1z * e0x - e1x * e0z;
    const float nz = e1x * e0y - e1y * e0x;
    if (!tri_hit_f(o, d, tfar, p0, p1, p2)) return 0;
    if (tl_tri_mode == 3) {      /* "plain_fma": the same normal, den and T as tri_hit_plain_fma */
        const float fx = __builtin_fmaf(e1y, e0z, -(e1z * e0y)), fy = __builtin_fmaf(e1z, e0x, -(e1x * e0z)), fz = __builtin_fmaf(e1x, e0y, -(e1y * e0x));
        const float fden = __builtin_fmaf(fx, d[0], __builtin_fmaf(fy, d[1], fz * d[2]));
        const float fT = __builtin_fmaf(v0x, fx, __builtin_fmaf(v0y, fy, v0z * fz));
        *t = fT / fden;
        return 1;
    }
    const float den = (nx * d[0] + ny * d[1]) + nz * d[2];
    const float T = (v0x * nx + v0y * ny) + v0z * nz;
    *t = T / den;
    return 1;
}

/* same algebra in double, zero tolerance: geometric reference for diagnostics */
static inline int tri_hit_d(const float *of, const float *df, float tfarf,
                            const float *p0, const float *p1, const float *p2) {
    const double o[3] = {of[0], of[1], of[2]}, d[3] = {df[0], df[1], df[2]};
    double v0[3], v1[3], v2[3], e0[3], e1[3], e2[3];
    for (int k = 0; k < 3; k++) {
        v0[k] = (double)p0[k] - o[k];
        v1[k] = (double)p1[k] - o[k];
        v2[k] = (double)p2[k] - o[k];
    }
    for (int k = 0; k < 3; k++) {
        e0[k] = v2[k] - v0[k]; e1[k] = v0[k] - v1[k]; e2[k] = v1[k] - v2[k];
    }
#define CROSSDOT(e, a, b) \
    (((e)[1] * ((a)[2] + (b)[2]) - (e)[2] * ((a)[1] + (b)[1])) * d[0] + \
     ((e)[2] * ((a)[0] + (b)[0]) - (e)[0] * ((a)[2] + (b)[2])) * d[1] + \
     ((e)[0] * ((a)[1] + (b)[1]) - (e)[1] * ((a)[0] + (b)[0])) * d[2])
    const double U = CROSSDOT(e0, v2, v0);
    const double V = CROSSDOT(e1, v0, v1);
    const double W = CROSSDOT(e2, v1, v2);
#undef CROSSDOT
    const double mn = fmin(U, fmin(V, W)), mx = fmax(U, fmax(V, W));
    if (!((mn >= 0.0) || (mx <= 0.0))) return 0;
    const double nx = e1[1] * e0[2] - e1[2] * e0[1];
    const double ny = e1[2] * e0[0] - e1[0] * e0[2];
    const double nz = e1[0] * e0[1] - e1[1] * e0[0];
    const double den = nx * d[0] + ny * d[1] + nz * d[2];
    const double T = v0[0] * nx + v0[1] * ny + v0[2] * nz;
    if (den == 0.0) return 0;
    const double t = T / den;
    return (t >= 0.0) && (t <= (double)tfarf);
}

/* the two triangles of grid quad (i, j): horizon_comp.cpp:139-151 */
static inline int quad_hit(const orc_scene *s, int i, int j, const float *o,
                           const float *d, float tfar, int dbl,
                           orc_counters *cnt) {
    const float *a = vtx(s, i, j), *b = vtx(s, i, j + 1);
    const float *c = vtx(s, i + 1, j), *e = vtx(s, i + 1, j + 1);
    if (cnt) cnt->tris += 2;
    if (dbl) return tri_hit_d(o, d, tfar, a, b, c) || tri_hit_d(o, d, tfar, b, e, c);
    /* geom_type "quad" / "grid" hand Embree the quad (v0, v1, v2, v3) = (a, b, e, c) (horizon_comp.cpp:165-172); its
     * quad intersector tests the triangles (v0, v1, v3) and (v2, v3, v1) = (a, b, c) and (e, c, b): the same two
     * triangles as the explicit "triangle" topology (:142-148), the second one with its vertices rotated.  The
     * rotation permutes the edge functions and changes which edge pair forms the normal, i.e. only the rounding.
     * orc_set_quad_order(1) evaluates that order (tests measure how many results it moves: none so far). */
    if (g_quad_order) return tri_hit_f(o, d, tfar, a, b, c) || tri_hit_f(o, d, tfar, e, c, b);
    return tri_hit_f(o, d, tfar, a, b, c) || tri_hit_f(o, d, tfar, b, e, c);
}

static inline int tin_hit(const orc_scene *s, int t, const float *o,
                          const float *d, float tfar, int dbl,
                          orc_counters *cnt) {
    const float *p0 = s->vs + 3 * (size_t)s->ts[3 * t + 0];
    const float *p1 = s->vs + 3 * (size_t)s->ts[3 * t + 1];
    const float *p2 = s->vs + 3 * (size_t)s->ts[3 * t + 2];
    if (cnt) cnt->tris += 1;
    return dbl ? tri_hit_d(o, d, tfar, p0, p1, p2) : tri_hit_f(o, d, tfar, p0, p1, p2);
}

/* ------------------------------------------------------------------------- */
/* ray / box: conservative slab test                                          */
/* ------------------------------------------------------------------------- */

typedef struct { float o[3], d[3], rd[3], tfar, tstart; } ray_t;   /* box tests run over [tstart, tfar - tstart] */

static inline void ray_init(ray_t *r, const float *o, const float *d, float tfar) {
    for (int k = 0; k < 3; k++) {
        r->o[k] = o[k]; r->d[k] = d[k];
        r->rd[k] = (fabsf(d[k]) > 1e-30f) ? 1.0f / d[k] : copysignf(1e30f, d[k]);
    }
    r->tfar = tfar; r->tstart = 0.0f;
}

static inline int box_hit(const ray_t *r, const float *lo, const float *hi) {
    float tmin = r->tstart, tmax = r->tfar - r->tstart;
    for (int k = 0; k < 3; k++) {
        const float t0 = (lo[k] - r->o[k]) * r->rd[k];
        const float t1 = (hi[k] - r->o[k]) * r->rd[k];
        tmin = fmaxf(tmin, fminf(t0, t1));
        tmax = fminf(tmax, fmaxf(t0, t1));
    }
    return tmin <= tmax * 1.0000005f;
}

/* ------------------------------------------------------------------------- */
/* BVH over the grid: recursive split of the quad index rectangle             */
/* (deliberately a different tree from the GPU's Morton LBVH)                 */
/* ------------------------------------------------------------------------- */

static int build_grid(orc_scene *s, int i0, int i1, int j0, int j1) {
    const int me = s->n_gn++;
    gnode *n = &s->gn[me];
    const int ni = i1 - i0, nj = j1 - j0;
    if (ni <= 2 && nj <= 2) {
        n->left = -1; n->right = -1; n->i0 = i0; n->j0 = j0; n->ni = ni; n->nj = nj;
        for (int k = 0; k < 3; k++) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
        for (int i = i0; i <= i1; i++)
            for (int j = j0; j <= j1; j++) {
                const float *p = vtx(s, i, j);
                for (int k = 0; k < 3; k++) {
                    n->lo[k] = fminf(n->lo[k], p[k]);
                    n->hi[k] = fmaxf(n->hi[k], p[k]);
                }
            }
        for (int k = 0; k < 3; k++) { n->lo[k] -= s->pad; n->hi[k] += s->pad; }
        return me;
    }
    int l, r;
    if (ni >= nj) {
        const int m = i0 + ni / 2;
        l = build_grid(s, i0, m, j0, j1);
        r = build_grid(s, m, i1, j0, j1);
    } else {
        const int m = j0 + nj / 2;
        l = build_grid(s, i0, i1, j0, m);
        r = build_grid(s, i0, i1, m, j1);
    }
    n = &s->gn[me];
    n->left = l; n->right = r; n->i0 = n->j0 = n->ni = n->nj = 0;
    for (int k = 0; k < 3; k++) {
        n->lo[k] = fminf(s->gn[l].lo[k], s->gn[r].lo[k]);
        n->hi[k] = fmaxf(s->gn[l].hi[k], s->gn[r].hi[k]);
    }
    return me;
}

/* BVH over the TIN: median split of the centroid along the widest axis       */
static const orc_scene *g_sort_scene; static int g_sort_axis;
static float tin_centroid(const orc_scene *s, int t, int ax) {
    return (s->vs[3 * (size_t)s->ts[3 * t] + ax] + s->vs[3 * (size_t)s->ts[3 * t + 1] + ax]
            + s->vs[3 * (size_t)s->ts[3 * t + 2] + ax]) / 3.0f;
}
static int cmp_tin(const void *a, const void *b) {
    const float ca = tin_centroid(g_sort_scene, *(const int *)a, g_sort_axis);
    const float cb = tin_centroid(g_sort_scene, *(const int *)b, g_sort_axis);
    return (ca > cb) - (ca < cb);
}
static int build_tin(orc_scene *s, int first, int count) {
    const int me = s->n_tn++;
    tnode *n = &s->tn[me];
    for (int k = 0; k < 3; k++) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
    for (int q = first; q < first + count; q++) {
        const int t = s->tri_order[q];
        for (int v = 0; v < 3; v++) {
            const float *p = s->vs + 3 * (size_t)s->ts[3 * t + v];
            for (int k = 0; k < 3; k++) {
                n->lo[k] = fminf(n->lo[k], p[k] - s->pad);
                n->hi[k] = fmaxf(n->hi[k], p[k] + s->pad);
            }
        }
    }
    if (count <= 4) { n->left = -1; n->right = -1; n->first = first; n->count = count; return me; }
    int ax = 0; float ext = -1.0f;
    for (int k = 0; k < 3; k++) if (n->hi[k] - n->lo[k] > ext) { ext = n->hi[k] - n->lo[k]; ax = k; }
    g_sort_scene = s; g_sort_axis = ax;
    qsort(s->tri_order + first, (size_t)count, sizeof(int), cmp_tin);
    const int h = count / 2;
    const int l = build_tin(s, first, h);
    const int r = build_tin(s, first + h, count - h);
    n = &s->tn[me];
    n->left = l; n->right = r; n->first = 0; n->count = 0;
    return me;
}

orc_scene *orc_scene_create(const float *vert_grid, int d0, int d1,
                            const float *vert_simp, int nvs,
                            const int32_t *tri_simp, int nts) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->d0 = d0; s->d1 = d1;
    const size_t nv = (size_t)d0 * (size_t)d1;
    s->vert = (float *)malloc(nv * 3 * sizeof(float));
    memcpy(s->vert, vert_grid, nv * 3 * sizeof(float));
    /* the TIN is only part of the scene if num_vert_simp >= 3: horizon_comp.cpp:199 */
    if (nvs >= 3 && nts >= 1) {
        s->nvs = nvs; s->nts = nts;
        s->vs = (float *)malloc((size_t)nvs * 3 * sizeof(float));
        memcpy(s->vs, vert_simp, (size_t)nvs * 3 * sizeof(float));
        s->ts = (int *)malloc((size_t)nts * 3 * sizeof(int));
        memcpy(s->ts, tri_simp, (size_t)nts * 3 * sizeof(int));
    }
    /* scene bounds -> conservative padding */
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t v = 0; v < nv; v++)
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(lo[k], s->vert[3 * v + k]); hi[k] = fmaxf(hi[k], s->vert[3 * v + k]);
        }
    for (int v = 0; v < s->nvs; v++)
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(lo[k], s->vs[3 * v + k]); hi[k] = fmaxf(hi[k], s->vs[3 * v + k]);
        }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    /* the padding has to survive being added to a coordinate (boxes are kept in the caller's
       frame here): 1e-6 of the diagonal alone is below half an ulp at Swiss-grid offsets and
       left the boxes unpadded, so the tree culled triangles the float test accepts (found by
       tests/test_gpu_fuzz.py: BVH != brute force on a 4 x 45 strip at x = 2.6e6).  Same rule
       as the GPU build (hz_scene.hip): + 4 ulp of the largest coordinate. */
    float maxabs = 0.0f;
    for (int k = 0; k < 3; k++) maxabs = fmaxf(maxabs, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
    s->pad = (float)(1.0e-6 * sqrt((double)ex * ex + (double)ey * ey + (double)ez * ez)
                     + 4.0 * (double)FLT_EPSILON * (double)maxabs) + FLT_MIN;
    /* grid BVH */
    if (d0 >= 2 && d1 >= 2) {
        const size_t nq = (size_t)(d0 - 1) * (size_t)(d1 - 1);
        s->gn = (gnode *)malloc((2 * nq + 8) * sizeof(gnode));
        s->n_gn = 0;
        build_grid(s, 0, d0 - 1, 0, d1 - 1);
        s->gn = (gnode *)realloc(s->gn, (size_t)s->n_gn * sizeof(gnode));
    }
    if (s->nts > 0) {
        s->tn = (tnode *)malloc((size_t)(2 * s->nts + 8) * sizeof(tnode));
        s->tri_order = (int *)malloc((size_t)s->nts * sizeof(int));
        for (int t = 0; t < s->nts; t++) s->tri_order[t] = t;
        s->n_tn = 0;
        build_tin(s, 0, s->nts);
    }
    return s;
}

void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    free(s->vert); free(s->vs); free(s->ts); free(s->gn); free(s->tn); free(s->tri_order);
    free(s);
}

/* mode 0: BVH + float test; 1: brute force + float test; 2: brute force + double test */
static int occluded_impl(const orc_scene *s, const float *o, const float *d, float tfar,
                         int mode, orc_counters *cnt);
static int occluded(const orc_scene *s, const float *o, const float *d, float tfar,
                    int mode, orc_counters *cnt) {
    tl_tri_mode = g_tri_mode;
    const int hit = occluded_impl(s, o, d, tfar, mode, cnt);
    if (g_tri_compare >= 0) {                          /* sensitivity run: the same ray with another triangle test */
        tl_tri_mode = g_tri_compare;
        const int other = occluded_impl(s, o, d, tfar, mode, NULL);
        tl_tri_mode = g_tri_mode;
        __atomic_fetch_add(&g_cmp_rays, 1, __ATOMIC_RELAXED);
        if (other != hit) __atomic_fetch_add(&g_cmp_flips, 1, __ATOMIC_RELAXED);
    }
    return hit;
}
static int occluded_impl(const orc_scene *s, const float *o, const float *d, float tfar,
                         int mode, orc_counters *cnt) {
    if (mode == 0) {
        ray_t r; ray_init(&r, o, d, tfar); r.tstart = -g_box_start_pads * s->pad;
        int stack[128]; int sp = 0;
        if (s->n_gn > 0) {
            stack[sp++] = 0;
            while (sp > 0) {
                const gnode *n = &s->gn[stack[--sp]];
                if (cnt) cnt->nodes++;
                if (!box_hit(&r, n->lo, n->hi)) continue;
                if (n->left < 0) {
                    for (int i = n->i0; i < n->i0 + n->ni; i++)
                        for (int j = n->j0; j < n->j0 + n->nj; j++)
                            if (quad_hit(s, i, j, o, d, tfar, 0, cnt)) return 1;
                } else { stack[sp++] = n->right; stack[sp++] = n->left; }
            }
        }
        if (s->n_tn > 0) {
            sp = 0; stack[sp++] = 0;
            while (sp > 0) {
                const tnode *n = &s->tn[stack[--sp]];
                if (cnt) cnt->nodes++;
                if (!box_hit(&r, n->lo, n->hi)) continue;
                if (n->left < 0) {
                    for (int q = n->first; q < n->first + n->count; q++)
                        if (tin_hit(s, s->tri_order[q], o, d, tfar, 0, cnt)) return 1;
                } else { stack[sp++] = n->right; stack[sp++] = n->left; }
            }
        }
        return 0;
    }
    const int dbl = (mode == 2);
    for (int i = 0; i < s->d0 - 1; i++)
        for (int j = 0; j < s->d1 - 1; j++)
            if (quad_hit(s, i, j, o, d, tfar, dbl, cnt)) return 1;
    for (int t = 0; t < s->nts; t++)
        if (tin_hit(s, t, o, d, tfar, dbl, cnt)) return 1;
    return 0;
}

/* closest hit: the minimum t over ALL triangles accepted with the caller's tfar (the result does
 * not depend on the visiting order); mode 0 = BVH, 1 = brute force.  Returns 1 when hit.        */
static int closest(const orc_scene *s, const float *o, const float *d, float tfar, int mode, float *dist) {
    float best = INFINITY; int any = 0; float t;
    tl_tri_mode = g_tri_mode;
#define TRY_QUAD(i, j) do { \
        const float *a = vtx(s, (i), (j)), *b = vtx(s, (i), (j) + 1); \
        const float *c = vtx(s, (i) + 1, (j)), *e = vtx(s, (i) + 1, (j) + 1); \
        if (tri_hit_t_f(o, d, tfar, a, b, c, &t)) { any = 1; if (t < best) best = t; } \
        if (tri_hit_t_f(o, d, tfar, b, e, c, &t)) { any = 1; if (t < best) best = t; } } while (0)
#define TRY_TIN(tt) do { \
        const float *p0 = s->vs + 3 * (size_t)s->ts[3 * (tt) + 0], *p1 = s->vs + 3 * (size_t)s->ts[3 * (tt) + 1]; \
        const float *p2 = s->vs + 3 * (size_t)s->ts[3 * (tt) + 2]; \
        if (tri_hit_t_f(o, d, tfar, p0, p1, p2, &t)) { any = 1; if (t < best) best = t; } } while (0)
    if (mode == 0) {
        ray_t r; ray_init(&r, o, d, tfar); r.tstart = -g_box_start_pads * s->pad;
        int stack[128]; int sp = 0;
        if (s->n_gn > 0) {
            stack[sp++] = 0;
            while (sp > 0) {
                const gnode *n = &s->gn[stack[--sp]];
                r.tfar = (best < tfar) ? best * 1.0001f : tfar;      /* prune, conservatively */
                if (!box_hit(&r, n->lo, n->hi)) continue;
                if (n->left < 0) {
                    for (int i = n->i0; i < n->i0 + n->ni; i++)
                        for (int j = n->j0; j < n->j0 + n->nj; j++) TRY_QUAD(i, j);
                } else { stack[sp++] = n->right; stack[sp++] = n->left; }
            }
        }
        if (s->n_tn > 0) {
            sp = 0; stack[sp++] = 0;
            while (sp > 0) {
                const tnode *n = &s->tn[stack[--sp]];
                r.tfar = (best < tfar) ? best * 1.0001f : tfar;
                if (!box_hit(&r, n->lo, n->hi)) continue;
                if (n->left < 0) {
                    for (int q = n->first; q < n->first + n->count; q++) TRY_TIN(s->tri_order[q]);
                } else { stack[sp++] = n->right; stack[sp++] = n->left; }
            }
        }
    } else {
        for (int i = 0; i < s->d0 - 1; i++)
            for (int j = 0; j < s->d1 - 1; j++) TRY_QUAD(i, j);
        for (int tt = 0; tt < s->nts; tt++) TRY_TIN(tt);
    }
#undef TRY_QUAD
#undef TRY_TIN
    if (any) *dist = best;
    return any;
}

void orc_closest_batch(const orc_scene *s, int64_t n, const float *org, const float *dir,
                       const float *tfar, int mode, uint8_t *hit, float *dist) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < n; q++) {
        float t = NAN;
        hit[q] = (uint8_t)closest(s, org + 3 * q, dir + 3 * q, tfar[q], mode, &t);
        dist[q] = t;
    }
}

void orc_occluded_batch(const orc_scene *s, int64_t n, const float *org,
                        const float *dir, const float *tfar, int mode,
                        uint8_t *out) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < n; q++)
        out[q] = (uint8_t)occluded(s, org + 3 * q, dir + 3 * q, tfar[q], mode, NULL);
}

/* ------------------------------------------------------------------------- */
/* trig tables (horizon_comp.cpp:711-731) -- built once per call              */
/* ------------------------------------------------------------------------- */

typedef struct {
    int azim_num, elev_num;
    float *azim_sin, *azim_cos, *elev_ang, *elev_sin, *elev_cos;
    float hori_acc, low, up, dist;  /* radians / metres after conversion      */
} tables_t;

static void tables_build(tables_t *t, int azim_num, float hori_acc_deg,
                         float low_deg, float dist_km) {
    float up_deg = 89.98;                          /* horizon_comp.cpp:648    */
    t->hori_acc = deg2rad_f(hori_acc_deg);         /* :667-669                */
    t->low = deg2rad_f(low_deg);
    t->up = deg2rad_f(up_deg);
    t->dist = (float)((double)dist_km * 1000.0);   /* :670                    */
    t->azim_num = azim_num;
    t->azim_sin = (float *)malloc(sizeof(float) * (size_t)azim_num);
    t->azim_cos = (float *)malloc(sizeof(float) * (size_t)azim_num);
    for (int i = 0; i < azim_num; i++) {           /* :714-718                */
        float ang = (float)(((2 * M_PI) / azim_num) * i);
        t->azim_sin[i] = sinf(ang);
        t->azim_cos[i] = cosf(ang);
    }
    const double step = (double)t->hori_acc / 5.0; /* (hori_acc / 5.0)        */
    t->elev_num = (int)ceil((double)(t->up - t->low) / step) + 1;  /* :721-722 */
    t->elev_ang = (float *)malloc(sizeof(float) * (size_t)t->elev_num);
    t->elev_sin = (float *)malloc(sizeof(float) * (size_t)t->elev_num);
    t->elev_cos = (float *)malloc(sizeof(float) * (size_t)t->elev_num);
    for (int i = 0; i < t->elev_num; i++) {        /* :726-731                */
        float ang = (float)((double)t->up - step * (double)i);
        t->elev_ang[t->elev_num - i - 1] = ang;
        t->elev_sin[t->elev_num - i - 1] = sinf(ang);
        t->elev_cos[t->elev_num - i - 1] = cosf(ang);
    }
}
static void tables_free(tables_t *t) {
    free(t->azim_sin); free(t->azim_cos); free(t->elev_ang); free(t->elev_sin); free(t->elev_cos);
}

/* exported so tests can compare the product's host-built tables bit for bit */
int orc_tables(int azim_num, float hori_acc_deg, float low_deg, float dist_km,
               float *azim_sin, float *azim_cos, int elev_cap, float *elev_ang,
               float *elev_sin, float *elev_cos, float *scalars4) {
    tables_t t; tables_build(&t, azim_num, hori_acc_deg, low_deg, dist_km);
    const int n = t.elev_num;
    if (azim_sin) memcpy(azim_sin, t.azim_sin, sizeof(float) * (size_t)azim_num);
    if (azim_cos) memcpy(azim_cos, t.azim_cos, sizeof(float) * (size_t)azim_num);
    if (elev_ang && n <= elev_cap) {
        memcpy(elev_ang, t.elev_ang, sizeof(float) * (size_t)n);
        memcpy(elev_sin, t.elev_sin, sizeof(float) * (size_t)n);
        memcpy(elev_cos, t.elev_cos, sizeof(float) * (size_t)n);
    }
    if (scalars4) { scalars4[0] = t.hori_acc; scalars4[1] = t.low; scalars4[2] = t.up; scalars4[3] = t.dist; }
    tables_free(&t);
    return n;
}

/* ------------------------------------------------------------------------- */
/* per-cell search algorithms (horizon_comp.cpp:302-498)                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    const orc_scene *s; const tables_t *t; int mode;
    float o[3]; float rot[3][3];
    uint64_t rays, guards; orc_counters *cnt;
    int want_dist;            /* closest-hit queries (the *_hori_dist variants, :519-612)  */
    float dist, dist_hit;     /* persist across azimuths exactly as there (:526-527)       */
} cell_t;

/* direction for (elevation index, azimuth index) + occlusion query          */
static inline int shoot(cell_t *c, int ie, int k) {
    const tables_t *t = c->t;
    /* :318-320 / :357-359: local (east, north, up) components              */
    const float ray[3] = {t->elev_cos[ie] * t->azim_sin[k],
                          t->elev_cos[ie] * t->azim_cos[k],
                          t->elev_sin[ie]};
    float rr[3];                                   /* mat_vec_mult :55-62     */
    for (int a = 0; a < 3; a++)
        rr[a] = (c->rot[a][0] * ray[0] + c->rot[a][1] * ray[1]) + c->rot[a][2] * ray[2];
    c->rays++;
    if (c->want_dist) {                            /* castRay_intersect1, :268-292 */
        const int hit = closest(c->s, c->o, rr, t->dist, c->mode, &c->dist);
        if (hit) c->dist_hit = c->dist;            /* :545-547 / :589-591 */
        return hit;
    }
    return occluded(c->s, c->o, rr, t->dist, c->mode, c->cnt);
}

static inline int ind_of(const tables_t *t, float elev_samp) {
    /* (int)roundf((elev_samp - low) / (hori_acc / 5.0)) : :351-352           */
    return (int)roundf((float)((double)(elev_samp - t->low) / ((double)t->hori_acc / 5.0)));
}

static void ray_discrete_sampling(cell_t *c, float *hori, float *dist_out) {   /* :302-333, :519-555 */
    const tables_t *t = c->t;
    for (int k = 0; k < t->azim_num; k++) {
        int ind_elev = 0, ind_elev_prev = 0, hit = 1;
        while (hit) {
            ind_elev_prev = ind_elev;
            ind_elev = (ind_elev + 10 < t->elev_num - 1) ? ind_elev + 10 : t->elev_num - 1;
            hit = shoot(c, ind_elev, k);
            if (hit && ind_elev == t->elev_num - 1) {
                c->guards++; break;                /* the reference never ends */
            }
        }
        hori[k] = (float)((double)(t->elev_ang[ind_elev_prev] + t->elev_ang[ind_elev]) / 2.0);
        if (dist_out) dist_out[k] = c->dist_hit;
    }
}

static int binary_azim(cell_t *c, int k, float *out) {            /* :346-375  */
    const tables_t *t = c->t;
    float lim_up = t->up, lim_low = t->low;
    float elev_samp = (float)((double)(lim_up + lim_low) / 2.0);
    int ind_elev = ind_of(t, elev_samp);
    while (fmaxf(lim_up - t->elev_ang[ind_elev], t->elev_ang[ind_elev] - lim_low) > t->hori_acc) {
        const int hit = shoot(c, ind_elev, k);
        if (hit) lim_low = t->elev_ang[ind_elev]; else lim_up = t->elev_ang[ind_elev];
        elev_samp = (float)((double)(lim_up + lim_low) / 2.0);
        ind_elev = ind_of(t, elev_samp);
    }
    *out = elev_samp;
    return ind_elev;
}

static void ray_binary_search(cell_t *c, float *hori, float *dist_out) {   /* :339-381, :561-612 */
    for (int k = 0; k < c->t->azim_num; k++) {
        binary_azim(c, k, &hori[k]);
        if (dist_out) dist_out[k] = c->dist_hit;
    }
}

static void ray_guess_const(cell_t *c, float *hori) {             /* :387-498  */
    const tables_t *t = c->t;
    int ind_elev_prev_azim = binary_azim(c, 0, &hori[0]);         /* :398-429  */
    for (int k = 1; k < t->azim_num; k++) {
        /* move upwards :439-458 */
        int ind_elev = (ind_elev_prev_azim - 5 > 0) ? ind_elev_prev_azim - 5 : 0;
        int ind_elev_prev = 0, hit = 1, count = 0;
        while (hit) {
            ind_elev_prev = ind_elev;
            ind_elev = (ind_elev + 10 < t->elev_num - 1) ? ind_elev + 10 : t->elev_num - 1;
            hit = shoot(c, ind_elev, k);
            count += 1;
            if (hit && ind_elev == t->elev_num - 1) { c->guards++; break; }
        }
        if (count > 1) {                                           /* :460-467  */
            const float es = (float)((double)(t->elev_ang[ind_elev_prev] + t->elev_ang[ind_elev]) / 2.0);
            ind_elev = ind_of(t, es);
            hori[k] = t->elev_ang[ind_elev];
            ind_elev_prev_azim = ind_elev;
            continue;
        }
        /* move downwards :472-488 */
        ind_elev = (ind_elev_prev_azim + 5 < t->elev_num - 1) ? ind_elev_prev_azim + 5 : t->elev_num - 1;
        hit = 0;
        while (!hit) {
            ind_elev_prev = ind_elev;
            ind_elev = (ind_elev - 10 > 0) ? ind_elev - 10 : 0;
            hit = shoot(c, ind_elev, k);
            if (!hit && ind_elev == 0) { c->guards++; break; }
        }
        const float es = (float)((double)(t->elev_ang[ind_elev_prev] + t->elev_ang[ind_elev]) / 2.0);
        ind_elev = ind_of(t, es);                                  /* :490-494  */
        hori[k] = t->elev_ang[ind_elev];
        ind_elev_prev_azim = ind_elev;
    }
}

/* ------------------------------------------------------------------------- */
/* gridded driver (horizon_comp.cpp:629-822)                                  */
/* stats[0..5] = rays, guard events, nodes, tris, build ns, ray-loop ns       */
/* rows [row_begin, row_end) of the inner domain are computed; others untouched */
/* ------------------------------------------------------------------------- */

int orc_horizon_gridded(const float *vert_grid, int dem_dim_0, int dem_dim_1,
                        const float *vec_norm, const float *vec_north,
                        int offset_0, int offset_1, float *hori_buffer,
                        int dim_in_0, int dim_in_1, int azim_num,
                        float dist_search, float hori_acc,
                        const char *ray_algorithm, const char *geom_type,
                        const float *vert_simp, int num_vert_simp,
                        const int32_t *tri_ind_simp, int num_tri_simp,
                        float elev_ang_low_lim, const uint8_t *mask,
                        float hori_fill, float ray_org_elev,
                        int mode, int row_begin, int row_end, int count_work,
                        uint64_t *stats) {
    (void)geom_type;  /* all three strings describe the same surface (:139-183) */
    int alg;
    if (strcmp(ray_algorithm, "discrete_sampling") == 0) alg = 0;
    else if (strcmp(ray_algorithm, "binary_search") == 0) alg = 1;
    else if (strcmp(ray_algorithm, "guess_constant") == 0) alg = 2;
    else return 1;
    const double t_start = now_s();
    orc_scene *s = orc_scene_create(vert_grid, dem_dim_0, dem_dim_1, vert_simp,
                                    num_vert_simp, tri_ind_simp, num_tri_simp);
    tables_t t; tables_build(&t, azim_num, hori_acc, elev_ang_low_lim, dist_search);
    const double t_built = now_s();
    if (row_begin < 0) row_begin = 0;
    if (row_end > dim_in_0 || row_end < 0) row_end = dim_in_0;
    uint64_t rays = 0, guards = 0, nodes = 0, tris = 0;
    /* cells are independent (the reference hands rows to TBB, :739-744); chunks of 16
       cells keep all cores busy for narrow slabs too */
#pragma omp parallel for collapse(2) schedule(dynamic, 16) reduction(+ : rays, guards, nodes, tris)
    for (int i = row_begin; i < row_end; i++) {
        for (int j = 0; j < dim_in_1; j++) {
            const size_t ind_arr = (size_t)i * (size_t)dim_in_1 + (size_t)j;
            float *hori = hori_buffer + ind_arr * (size_t)azim_num;
            if (mask[ind_arr] == 1) {                              /* :750      */
                const float *nm = vec_norm + 3 * ind_arr, *nr = vec_north + 3 * ind_arr;
                const float norm_x = nm[0], norm_y = nm[1], norm_z = nm[2];
                const float north_x = nr[0], north_y = nr[1], north_z = nr[2];
                const float *p = vert_grid + 3 * ((size_t)(i + offset_0) * (size_t)dem_dim_1
                                                  + (size_t)(j + offset_1));
                cell_t c; c.s = s; c.t = &t; c.mode = mode; c.rays = 0; c.guards = 0;
                c.want_dist = 0; c.dist = 0.0f; c.dist_hit = 0.0f;
                orc_counters cn = {0, 0}; c.cnt = count_work ? &cn : NULL;
                c.o[0] = p[0] + norm_x * ray_org_elev;            /* :763-770  */
                c.o[1] = p[1] + norm_y * ray_org_elev;
                c.o[2] = p[2] + norm_z * ray_org_elev;
                /* east = north x norm (:773-776), rot_inv columns (east, north, norm) */
                const float east_x = north_y * norm_z - north_z * norm_y;
                const float east_y = north_z * norm_x - north_x * norm_z;
                const float east_z = north_x * norm_y - north_y * norm_x;
                c.rot[0][0] = east_x; c.rot[0][1] = north_x; c.rot[0][2] = norm_x;
                c.rot[1][0] = east_y; c.rot[1][1] = north_y; c.rot[1][2] = norm_y;
                c.rot[2][0] = east_z; c.rot[2][1] = north_z; c.rot[2][2] = norm_z;
                if (alg == 0) ray_discrete_sampling(&c, hori, NULL);
                else if (alg == 1) ray_binary_search(&c, hori, NULL);
                else ray_guess_const(&c, hori);
                rays += c.rays; guards += c.guards; nodes += cn.nodes; tris += cn.tris;
            } else {
                for (int k = 0; k < azim_num; k++) hori[k] = hori_fill;  /* :789-794 */
            }
        }
    }
    if (stats) {
        stats[0] = rays; stats[1] = guards; stats[2] = nodes; stats[3] = tris;
        stats[4] = (uint64_t)((t_built - t_start) * 1e9);      /* scene build [ns]  */
        stats[5] = (uint64_t)((now_s() - t_built) * 1e9);      /* ray loop [ns]     */
    }
    tables_free(&t);
    orc_scene_destroy(s);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* locations driver (horizon_comp.cpp:828-1094)                               */
/* stats[0] = search rays (as the reference counts), stats[1] = guard events, */
/* stats[2] = locations that found the surface along +/- normal               */
/* ------------------------------------------------------------------------- */
int orc_horizon_locations(const float *vert_grid, int dem_dim_0, int dem_dim_1,
                          const float *coords, const float *vec_norm, const float *vec_north,
                          float *hori_buffer, float *hori_dist_buffer, int num_loc,
                          int azim_num, float dist_search, float hori_acc,
                          const char *ray_algorithm, const char *geom_type,
                          float elev_ang_low_lim, const float *ray_org_elev,
                          int hori_dist_out, int mode, uint64_t *stats) {
    (void)geom_type;
    int alg;
    if (strcmp(ray_algorithm, "discrete_sampling") == 0) alg = 0;
    else if (strcmp(ray_algorithm, "binary_search") == 0) alg = 1;
    else if (strcmp(ray_algorithm, "guess_constant") == 0) alg = 2;
    else return 1;
    if (hori_dist_out && alg == 2) return 2;       /* horizon.pyx: not implemented */
    /* no simplified outer mesh for locations (:848-852) */
    orc_scene *s = orc_scene_create(vert_grid, dem_dim_0, dem_dim_1, NULL, 0, NULL, 0);
    tables_t t; tables_build(&t, azim_num, hori_acc, elev_ang_low_lim, dist_search);
    uint64_t rays = 0, guards = 0, found = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays, guards, found)
    for (int i = 0; i < num_loc; i++) {
        const float norm_x = vec_norm[3 * i], norm_y = vec_norm[3 * i + 1], norm_z = vec_norm[3 * i + 2];
        const float north_x = vec_north[3 * i], north_y = vec_north[3 * i + 1], north_z = vec_north[3 * i + 2];
        const float ini[3] = {coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]};
        float dist = 0.0f;                                         /* :947-957 */
        const float up[3] = {norm_x, norm_y, norm_z}, down[3] = {-norm_x, -norm_y, -norm_z};
        int hit = closest(s, ini, up, 100000.0f, mode, &dist);
        if (!hit) {
            hit = closest(s, ini, down, 100000.0f, mode, &dist);
            dist = (float)((double)dist * -1.0);
        }
        if (!hit) continue;
        found++;
        cell_t c; c.s = s; c.t = &t; c.mode = mode; c.rays = 0; c.guards = 0; c.cnt = NULL;
        c.want_dist = hori_dist_out ? 1 : 0; c.dist = 0.0f; c.dist_hit = 0.0f;
        c.o[0] = ini[0] + norm_x * (dist + ray_org_elev[i]);       /* :961-963 */
        c.o[1] = ini[1] + norm_y * (dist + ray_org_elev[i]);
        c.o[2] = ini[2] + norm_z * (dist + ray_org_elev[i]);
        const float east_x = north_y * norm_z - north_z * norm_y;
        const float east_y = north_z * norm_x - north_x * norm_z;
        const float east_z = north_x * norm_y - north_y * norm_x;
        c.rot[0][0] = east_x; c.rot[0][1] = north_x; c.rot[0][2] = norm_x;
        c.rot[1][0] = east_y; c.rot[1][1] = north_y; c.rot[1][2] = norm_y;
        c.rot[2][0] = east_z; c.rot[2][1] = north_z; c.rot[2][2] = norm_z;
        float *hori = hori_buffer + (size_t)i * (size_t)azim_num;
        float *dout = hori_dist_out ? hori_dist_buffer + (size_t)i * (size_t)azim_num : NULL;
        if (alg == 0) ray_discrete_sampling(&c, hori, dout);
        else if (alg == 1) ray_binary_search(&c, hori, dout);
        else ray_guess_const(&c, hori);
        rays += c.rays; guards += c.guards;
    }
    if (stats) { stats[0] = rays; stats[1] = guards; stats[2] = found; }
    tables_free(&t);
    orc_scene_destroy(s);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* shadow / sw_dir_cor (shadow_comp.cpp:96-159, 318-605)                      */
/* ------------------------------------------------------------------------- */

static inline void vec_unit(float *x, float *y, float *z) {       /* :96-106   */
    const float mag = sqrtf((*x * *x + *y * *y) + *z * *z);
    *x = *x / mag; *y = *y / mag; *z = *z / mag;
}

static inline void vec_rot(float kx, float ky, float kz, float theta,
                           float *vx, float *vy, float *vz) {     /* :109-132  */
    const float ct = r_cosf(theta), st = r_sinf(theta);
    const float part = (float)((double)((kx * *vx + ky * *vy) + kz * *vz) * (1.0 - (double)ct));
    const float rx = (*vx * ct + (ky * *vz - kz * *vy) * st) + kx * part;
    const float ry = (*vy * ct + (kz * *vx - kx * *vz) * st) + ky * part;
    const float rz = (*vz * ct + (kx * *vy - ky * *vx) * st) + kz * part;
    *vx = rx; *vy = ry; *vz = rz;
}

static inline float atmos_refrac(float elev_ang_true, float temp, float pressure) { /* :135-159 */
    const float lower = -1.0f, upper = 90.0f;
    elev_ang_true = fmaxf(lower, fminf(elev_ang_true, upper));
    float refrac_cor = (float)(1.02 / (double)r_tanf(deg2rad_f(
        (float)((double)elev_ang_true + 10.3 / ((double)elev_ang_true + 5.11)))));
    refrac_cor = (float)((double)refrac_cor + 0.0019279);
    refrac_cor = (float)((double)refrac_cor * (((double)pressure / 101.0) * (283.0 / (273.0 + (double)temp))));
    return (float)((double)refrac_cor * (1.0 / 60.0));
}

typedef struct {
    orc_scene *s;
    int d0, d1, off0, off1, in0, in1;
    float *tilt, *norm, *enl, *elev; uint8_t *mask;
    float fill, ang_max; int refrac;
    float temperature_ref, pressure_ref, lapse_rate, expo;
} orc_terrain;

orc_terrain *orc_terrain_create(const float *vert_grid, int d0, int d1, int off0, int off1,
                                const float *vec_tilt, const float *vec_norm,
                                int in0, int in1, const float *surf_enl_fac,
                                const float *elevation, const uint8_t *mask,
                                float sw_dir_cor_fill, float ang_max, int refrac_cor) {
    orc_terrain *t = (orc_terrain *)calloc(1, sizeof(orc_terrain));
    t->s = orc_scene_create(vert_grid, d0, d1, NULL, 0, NULL, 0);   /* no TIN: shadow_comp.cpp:198-298 */
    t->d0 = d0; t->d1 = d1; t->off0 = off0; t->off1 = off1; t->in0 = in0; t->in1 = in1;
    const size_t nc = (size_t)in0 * (size_t)in1;
    t->tilt = (float *)malloc(nc * 12); memcpy(t->tilt, vec_tilt, nc * 12);
    t->norm = (float *)malloc(nc * 12); memcpy(t->norm, vec_norm, nc * 12);
    t->enl = (float *)malloc(nc * 4); memcpy(t->enl, surf_enl_fac, nc * 4);
    t->elev = (float *)malloc(nc * 4); memcpy(t->elev, elevation, nc * 4);
    t->mask = (uint8_t *)malloc(nc); memcpy(t->mask, mask, nc);
    t->fill = sw_dir_cor_fill; t->ang_max = ang_max; t->refrac = refrac_cor;
    t->temperature_ref = 283.15;                                   /* :349-354  */
    t->pressure_ref = 101.0;
    t->lapse_rate = 0.0065;
    const float g = 9.81, R_d = 287.0;
    t->expo = g / (R_d * t->lapse_rate);
    return t;
}

void orc_terrain_destroy(orc_terrain *t) {
    if (!t) return;
    orc_scene_destroy(t->s);
    free(t->tilt); free(t->norm); free(t->enl); free(t->elev); free(t->mask); free(t);
}

/* which = 0: shadow (u8 out), 1: sw_dir_cor (f32 out); stats[0] = rays cast  */
static void terrain_run(const orc_terrain *t, const float *sun_position, int which,
                        uint8_t *out_u8, float *out_f32, int mode, uint64_t *stats) {
    const float ray_org_elev = 0.05;                               /* :388, :497 */
    const float dot_prod_min = cosf(deg2rad_f(t->ang_max));        /* :498      */
    uint64_t rays = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays)
    for (int i = 0; i < t->in0; i++) {
        for (int j = 0; j < t->in1; j++) {
            const size_t ind_arr = (size_t)i * (size_t)t->in1 + (size_t)j;
            if (t->mask[ind_arr] != 1) {
                if (which == 0) out_u8[ind_arr] = 3; else out_f32[ind_arr] = t->fill;
                continue;
            }
            const float *tl = t->tilt + 3 * ind_arr, *nm = t->norm + 3 * ind_arr;
            const float tilt_x = tl[0], tilt_y = tl[1], tilt_z = tl[2];
            const float norm_x = nm[0], norm_y = nm[1], norm_z = nm[2];
            const float *p = t->s->vert + 3 * ((size_t)(i + t->off0) * (size_t)t->d1 + (size_t)(j + t->off1));
            float o[3];
            o[0] = p[0] + norm_x * ray_org_elev;
            o[1] = p[1] + norm_y * ray_org_elev;
            o[2] = p[2] + norm_z * ray_org_elev;
            float sun_x = sun_position[0] - o[0];                  /* :422-425  */
            float sun_y = sun_position[1] - o[1];
            float sun_z = sun_position[2] - o[2];
            vec_unit(&sun_x, &sun_y, &sun_z);
            float dot_prod_ns = (norm_x * sun_x + norm_y * sun_y) + norm_z * sun_z;
            if (t->refrac == 1) {                                  /* :430-446  */
                const float elev_ang_true = (float)(90.0 - (double)rad2deg_f(r_acosf(dot_prod_ns)));
                const float temperature = t->temperature_ref - (t->lapse_rate * t->elev[ind_arr]);
                const float pressure = t->pressure_ref * r_powf(temperature / t->temperature_ref, t->expo);
                const float refrac_cor = atmos_refrac(elev_ang_true, K2degC_f(temperature), pressure);
                float k_x = sun_y * norm_z - sun_z * norm_y;
                float k_y = sun_z * norm_x - sun_x * norm_z;
                float k_z = sun_x * norm_y - sun_y * norm_x;
                vec_unit(&k_x, &k_y, &k_z);
                vec_rot(k_x, k_y, k_z, deg2rad_f(refrac_cor), &sun_x, &sun_y, &sun_z);
                dot_prod_ns = (norm_x * sun_x + norm_y * sun_y) + norm_z * sun_z;
            }
            const float dot_prod_ts = (tilt_x * sun_x + tilt_y * sun_y) + tilt_z * sun_z;
            const float sd[3] = {sun_x, sun_y, sun_z};
            if (which == 0) {
                if (dot_prod_ts > 0.0f) {                          /* :451-470  */
                    rays++;
                    out_u8[ind_arr] = occluded(t->s, o, sd, INFINITY, mode, NULL) ? 2 : 0;
                } else out_u8[ind_arr] = 1;
            } else {
                if (dot_prod_ts > dot_prod_min) {                  /* :561-590  */
                    rays++;
                    if (occluded(t->s, o, sd, INFINITY, mode, NULL)) out_f32[ind_arr] = 0.0f;
                    else {
                        if (dot_prod_ns < dot_prod_min) dot_prod_ns = dot_prod_min;
                        out_f32[ind_arr] = (dot_prod_ts / dot_prod_ns) * t->enl[ind_arr];
                    }
                } else out_f32[ind_arr] = 0.0f;
            }
        }
    }
    if (stats) stats[0] = rays;
}

void orc_terrain_shadow(const orc_terrain *t, const float *sun_position, uint8_t *buf,
                        int mode, uint64_t *stats) {
    terrain_run(t, sun_position, 0, buf, NULL, mode, stats);
}
void orc_terrain_sw_dir_cor(const orc_terrain *t, const float *sun_position, float *buf,
                            int mode, uint64_t *stats) {
    terrain_run(t, sun_position, 1, NULL, buf, mode, stats);
}

/* ------------------------------------------------------------------------- */
/* sky view factor (topo_param.pyx:412-460)                                   */
/* float32 state, libm double trig as Cython's libc.math gives                */
/* ------------------------------------------------------------------------- */

void orc_sky_view_factor(const float *azim, const float *hori, const float *vec_tilt,
                         int len_0, int len_1, int len_2, float *svf) {
    float *azim_sin = (float *)malloc(sizeof(float) * (size_t)len_2);
    float *azim_cos = (float *)malloc(sizeof(float) * (size_t)len_2);
    for (int k = 0; k < len_2; k++) {
        azim_sin[k] = (float)sin((double)azim[k]);
        azim_cos[k] = (float)cos((double)azim[k]);
    }
    const float azim_spac = azim[1] - azim[0];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < len_0; i++)
        for (int j = 0; j < len_1; j++) {
            const size_t c = (size_t)i * (size_t)len_1 + (size_t)j;
            const float tx = vec_tilt[3 * c], ty = vec_tilt[3 * c + 1], tz = vec_tilt[3 * c + 2];
            float agg = 0.0f;
            for (int k = 0; k < len_2; k++) {
                const float h = hori[c * (size_t)len_2 + (size_t)k];
                const float hori_plane = (float)atan((double)(-azim_sin[k] * tx / tz - azim_cos[k] * ty / tz));
                const float hori_elev = (h >= hori_plane) ? h : hori_plane;
                const double ce = cos((double)hori_elev);
                agg = (float)((double)agg + ((double)(tx * azim_sin[k] + ty * azim_cos[k])
                        * ((M_PI / 2.0) - (double)hori_elev - (sin(2.0 * (double)hori_elev) / 2.0))
                        + (double)tz * (ce * ce)));
            }
            svf[c] = (float)(((double)azim_spac / (2.0 * M_PI)) * (double)agg);
        }
    free(azim_sin); free(azim_cos);
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
