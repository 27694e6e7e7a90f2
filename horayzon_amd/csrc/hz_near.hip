// hz_near.hip -- near-field certificates for the horizon kernel (gfx950).
//
// Every ray of a cell starts a few centimetres above a grid vertex, i.e. INSIDE the padded boxes of the quads
// around that vertex and of all their ancestors: it walks the root-to-leaf path(s) around its own origin and tests
// the four adjacent quads before it sees any other terrain -- about 2/3 of the work of a ray, identical for the
// ~780 rays of a cell ("origin neighbourhood").  Most rays do not need it: a ray that looks
// above everything NEAR the cell can only be blocked by terrain further away.
//
// This pre-pass computes, once per cell and launch, a certificate for exactly that:
//   window W  = the (2w)^2 quads around the cell's vertex (w = HZ_NEAR_W cells in each direction),
//   near_idx[cell][k] = the smallest elevation-table index such that a ray of azimuth k with an index >= it
//                       passes strictly above every triangle of W (with margins),
//   near_r[cell]      = a distance r such that every triangle NOT in W lies further than r from the ray origin.
// k_horizon then starts such a ray at parameter r instead of 0 (its box tests run from the shifted origin), so
// the traversal never descends into the origin's neighbourhood.  Any triangle that is still reached gets the
// unchanged exact test with the unchanged origin: hit decisions cannot change -- only boxes are culled, and only
// boxes that contain nothing the ray can hit (`opts.verify_near` re-traces every shortened
// ray from parameter 0 in the counting instantiation and counts disagreements: there must be none).
//
// Geometry (exact arithmetic; the float margins are listed with the code):
//  * Local frame of the cell (east, north, norm), origin o.  A ray of azimuth phi_k lies in the vertical half-plane
//    H_k = { r (sin phi_k, cos phi_k, 0) + z (0, 0, 1), r >= 0 }.  It can hit a triangle T only where T meets H_k,
//    and T /\ H_k is a segment along which z / r (the tangent of the elevation seen from o) is monotone, so its
//    maximum sits at an end point, and the end points are crossings of H_k with EDGES of T.  Hence
//        tanE[k] = max over window edges crossing H_k of z / r at the crossing point
//    bounds the elevation of everything in W along azimuth k.  (The six spokes from the cell's own vertex are not
//    evaluated: the origin lies above the planes of the six triangles around that vertex -- checked per cell -- so
//    z / r = slope + gamma / r with gamma < 0 increases with r on each of them and their maximum is taken on the inner
//    ring's outer edges.  DESIGN.md section 4.3 lists every statement of this argument with the check that backs it.)
//  * That bound needs the window's surface to be a graph over the cell's LOCAL horizontal plane (one curve z(r) per
//    azimuth, starting directly below o): checked per cell (orientation of every window triangle in the local (east,
//    north) projection); a cell whose frame is tilted against terrain steeper than the tilt allows gets no certificate.
//  * The mesh is a height field over the world (x, y) plane, so every triangle outside W projects outside W's
//    boundary polygon, and its 3-D distance from o is at least the horizontal distance from o to that polygon:
//    near_r = that distance (shrunk).  Quads that break the height-field property and outer-domain TIN triangles do
//    not obey this: the scene build marks their (x, y) footprints in a coarse bitmap (HZ_BLOB_BAD_MAP) and a cell whose
//    window box touches a marked bitmap cell gets no certificate (round 5; rounds 3-4 switched the whole scene off).
#include "hz_internal.h"

namespace hz {

#ifndef HZ_NEAR_W
#define HZ_NEAR_W 2   // measured on the 3601^2 tile: w = 2 and w = 3 give the same total (kernel 196 + 4.8 ms vs 192 + 8.4 ms)
#endif

struct NearParams {
    const float *verts;
    const float *vec_norm, *vec_north;
    const uint8_t *mask;
    const float *azim_sin, *azim_cos;
    int d0, d1, offset_0, offset_1, dim_in_1;
    int row_begin, n_cells;            // cells [row_begin * dim_in_1, +n_cells) of the inner domain
    int azim_num, elev_num;
    float ray_org_elev, low, step, up, pad;
    unsigned short *near_idx;          // [n_cells][azim_num]
    float *near_r;                     // [n_cells]
    unsigned *reasons;                 // null, or 20 counters: [0] cells, [1] with a certificate, [2 + b] refused for reason bit b, [16] tasks, [17] bins
    const uint32_t *bad_bits;          // HZ_BLOB_BAD_MAP: the scene's bitmap of bad (x, y) cells (hz_common.h), or null
    int bad_nb;
    float bad_x0, bad_y0, bad_sx, bad_sy;
};

// why a cell gets no certificate (bits of the per-wave flag word; HZ_NEAR_REASONS=1 prints the histogram of a call)
enum { HZ_NR_FRAME = 1, HZ_NR_VERTEX_ON_AXIS = 2, HZ_NR_EDGE_OVER_AXIS = 4, HZ_NR_AZ_TOLERANCE = 8, HZ_NR_INPLANE_EDGE = 16,
       HZ_NR_CROSSING_NEAR_AXIS = 32, HZ_NR_INTERVAL = 64, HZ_NR_PRECISION = 128, HZ_NR_EDGE_ON = 256, HZ_NR_ORIENTATION = 512,
       HZ_NR_ORIGIN_BELOW = 1024, HZ_NR_AXIS_IN_TRIANGLE = 2048, HZ_NR_WINDOW = 4096, HZ_NR_BAD_MESH = 8192 };

// order-preserving float <-> int so that LDS atomicMax works on floats
__device__ __forceinline__ int f2o(float f) { const int i = __float_as_int(f); return i >= 0 ? i : (i ^ 0x7fffffff); }
__device__ __forceinline__ float o2f(int i) { return __int_as_float(i >= 0 ? i : (i ^ 0x7fffffff)); }

#define HZ_NEAR_MAXT 256      // tasks of the edge phase per cell (<= 255 edges, task numbers in bytes)
template <int W>
__host__ __device__ constexpr int near_per_wave_words(int A) {
    constexpr int NV = 2 * W + 1, NVERT = NV * NV, NEDGE = 2 * NV * (NV - 1) + (NV - 1) * (NV - 1);
    return NVERT * 5 + NVERT * 3 + A + 4 + 2 * NEDGE + NEDGE + 1 + 8 * NEDGE + HZ_NEAR_MAXT / 4;
}

template <int W>
__global__ __launch_bounds__(256) void k_near_cert(NearParams p) {
    constexpr int NV = 2 * W + 1, NVERT = NV * NV, CENTRE = W * NV + W;
    constexpr int NHOR = NV * (NV - 1), NDIAG = (NV - 1) * (NV - 1), NEDGE = 2 * NHOR + NDIAG;
    constexpr int NSEG = 8 * W;        // boundary segments of the window polygon
    static_assert(NVERT <= 64 && NSEG <= 64, "one lane per window vertex / boundary segment");
    constexpr int NEC = 8;             // per-edge constants kept in LDS: a_e a_n a_z b_e b_n b_z r_a r_b
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_near[];
    const int A = p.azim_num;
    // LDS: [ (sin phi_k, cos phi_k) pairs : 2 A floats, shared ] then per wave
    //      [ q: NVERT x 5 | vp: NVERT x 3 | E: A | flags: 4 | first azimuth bin, bins per edge: 2 NEDGE | task prefix: NEDGE + 1 |
    //        edge constants: NEC x NEDGE | task -> edge map: HZ_NEAR_MAXT bytes ]
    const int per_wave = near_per_wave_words<W>(A);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *tab = reinterpret_cast<float *>(smem_near);
    float *q = tab + 2 * A + (size_t)wave * per_wave;               // [NVERT][5]: e, n, z, world dx, dy
    float *vp = q + NVERT * 5;                                      // [NVERT][3]: horizontal distance r, azimuth, bound x of the vertex seen in a plane
    int *E = reinterpret_cast<int *>(vp + NVERT * 3);               // [A]
    int *flags = E + A;                                             // [0]: certificate unusable
    int *ek = flags + 4;                                            // [NEDGE][2]
    int *pre = ek + 2 * NEDGE;                                      // [NEDGE + 1]
    float *ec = reinterpret_cast<float *>(pre + NEDGE + 1);         // [NEDGE][NEC]
    unsigned char *tmap = reinterpret_cast<unsigned char *>(ec + NEC * NEDGE);   // [HZ_NEAR_MAXT]
    for (int k = threadIdx.x; k < A; k += 256) { tab[2 * k] = p.azim_sin[k]; tab[2 * k + 1] = p.azim_cos[k]; }   // (sin, cos) pairs: one 8 B read
    const int cl = blockIdx.x * 4 + wave;                           // cell of this wave (launch local)
    const bool have = cl < p.n_cells;
    const int i = have ? p.row_begin + cl / p.dim_in_1 : 0, j = have ? cl % p.dim_in_1 : 0;
    const int gi = i + p.offset_0, gj = j + p.offset_1;
    const size_t cell = (size_t)i * p.dim_in_1 + j;
    bool valid = have && p.mask[cell] == 1 && gi - W >= 0 && gi + W <= p.d0 - 1 && gj - W >= 0 && gj + W <= p.d1 - 1;
    // ---- window vertices in the local frame (the ray set-up of k_horizon, same operations) ------------------
    if (valid && lane < NVERT) {
        const float nx = p.vec_norm[3 * cell], ny = p.vec_norm[3 * cell + 1], nz = p.vec_norm[3 * cell + 2];
        const float tx = p.vec_north[3 * cell], ty = p.vec_north[3 * cell + 1], tz = p.vec_north[3 * cell + 2];
        const float ex = ty * nz - tz * ny, ey = tz * nx - tx * nz, ez = tx * ny - ty * nx;
        const float *v = p.verts + 3 * ((size_t)gi * p.d1 + (size_t)gj);
        const float ox = v[0] + nx * p.ray_org_elev, oy = v[1] + ny * p.ray_org_elev, oz = v[2] + nz * p.ray_org_elev;
        const int a = lane / NV - W, b = lane % NV - W;
        const float *w = p.verts + 3 * ((size_t)(gi + a) * p.d1 + (size_t)(gj + b));
        const float rx = w[0] - ox, ry = w[1] - oy, rz = w[2] - oz;
        const float qe = (rx * ex + ry * ey) + rz * ez, qn = (rx * tx + ry * ty) + rz * tz, qz = (rx * nx + ry * ny) + rz * nz;
        q[5 * lane + 0] = qe;
        q[5 * lane + 1] = qn;
        q[5 * lane + 2] = qz;
        q[5 * lane + 3] = rx; q[5 * lane + 4] = ry;
        // Polar form of the vertex, once per vertex (round 5; rounds 2-4 formed it per edge end, i.e. ~4 times): horizontal
        // distance, azimuth clockwise from north, and the bound of z / r for a vertex that lies IN a half-plane (|d| <= tol):
        // it is then seen at r in [r sqrt(1 - (tol / r)^2), r], the larger of z / r over that range (tol / r <= 0.126
        // because the azimuth tolerance of its edges is <= 0.25).  Vertices within 1 mm of the axis: refused by their edges.
        const float vr = __builtin_sqrtf(qe * qe + qn * qn);
        float vx = 0.0f;
        if (vr > 1.0e-3f) {
            const float vt = (1.0e-3f * vr + 0.01f) / vr;
            vx = qz > 0.0f ? qz / (vr * __builtin_sqrtf(1.0f - vt * vt)) : qz / vr;
        }
        vp[3 * lane + 0] = vr; vp[3 * lane + 1] = atan2f(qe, qn); vp[3 * lane + 2] = vx;
    }
    // HZ_BLOB_BAD_MAP (hz_common.h): the distance bound near_r (row R of DESIGN.md section 4.3) holds for this cell if no
    // bad quad and no TIN triangle projects into the window's (x, y) bounding box.  Those have marked the scene's coarse
    // bitmap; the window's box spans a few of its cells (one lane each; more than 8 x 8: refused).
    bool bad_mesh = false;
    if (p.bad_bits != nullptr && valid) {           // (`valid` is wave uniform)
        float wx = 0.0f, wy = 0.0f;
        if (lane < NVERT) {
            const int a = lane / NV - W, b = lane % NV - W;
            const float *w = p.verts + 3 * ((size_t)(gi + a) * p.d1 + (size_t)(gj + b));
            wx = w[0]; wy = w[1];
        }
        float xl = lane < NVERT ? wx : __builtin_inff(), xh = lane < NVERT ? wx : -__builtin_inff();
        float yl = lane < NVERT ? wy : __builtin_inff(), yh = lane < NVERT ? wy : -__builtin_inff();
        for (int off = 32; off > 0; off >>= 1) {
            xl = __builtin_fminf(xl, __shfl_xor(xl, off)); xh = __builtin_fmaxf(xh, __shfl_xor(xh, off));
            yl = __builtin_fminf(yl, __shfl_xor(yl, off)); yh = __builtin_fmaxf(yh, __shfl_xor(yh, off));
        }
        const int nb = p.bad_nb;
        const int i0 = min(max((int)__builtin_floorf((xl - p.bad_x0) * p.bad_sx), 0), nb - 1);
        const int i1 = min(max((int)__builtin_floorf((xh - p.bad_x0) * p.bad_sx), 0), nb - 1);
        const int j0 = min(max((int)__builtin_floorf((yl - p.bad_y0) * p.bad_sy), 0), nb - 1);
        const int j1 = min(max((int)__builtin_floorf((yh - p.bad_y0) * p.bad_sy), 0), nb - 1);
        if (i1 - i0 >= 8 || j1 - j0 >= 8) bad_mesh = true;
        else {
            const int i = i0 + (lane & 7), j = j0 + (lane >> 3);
            const bool hit = i <= i1 && j <= j1 && ((p.bad_bits[((size_t)j * nb + i) >> 5] >> (((size_t)j * nb + i) & 31)) & 1u) != 0u;
            bad_mesh = __ballot(hit) != 0ull;
        }
    }
    for (int k = lane; k < A; k += 64) E[k] = f2o(-__builtin_inff());
    if (lane == 0) {
        // The whole construction works in the cell's (east, north, norm) coordinates and assumes that the ray of table
        // entry (k, i) has azimuth phi_k and elevation elev_ang[i] IN THOSE coordinates: true only for an orthonormal
        // frame.  The reference accepts any pair of vectors (horizon_comp.cpp:751-779); a frame that is off by more than
        // 1e-4 (unit length, right angle) gets no certificate (budget: DESIGN.md section 4.3, term F).
        int bad = 0;
        if (valid) {
            const float nx = p.vec_norm[3 * cell], ny = p.vec_norm[3 * cell + 1], nz = p.vec_norm[3 * cell + 2];
            const float tx = p.vec_north[3 * cell], ty = p.vec_north[3 * cell + 1], tz = p.vec_north[3 * cell + 2];
            const float nn = (nx * nx + ny * ny) + nz * nz, tt = (tx * tx + ty * ty) + tz * tz, nt = (nx * tx + ny * ty) + nz * tz;
            if (!(__builtin_fabsf(nn - 1.0f) <= 1.0e-4f && __builtin_fabsf(tt - 1.0f) <= 1.0e-4f && __builtin_fabsf(nt) <= 1.0e-4f)) bad = 1;
            // Round 6 (sweep seed 64003, configurations 734 and 1377): how far the ray of table azimuth k can lie BESIDE the
            // half-plane H_k this construction cuts the window with.  In the coordinates used here (dot products with east, north,
            // norm) the direction a e + b t + c n has the components (a |e|^2, b |t|^2 + c n.t, c |n|^2 + b n.t): its azimuth is off
            // by <= | |n|^2 - 1 | + | |t|^2 - 1 | (|e|^2 / |t|^2 = |n|^2 (1 - (n.t)^2)) and it leaves H_k sideways by |n.t| per unit of
            // HEIGHT.  At a crossing at horizontal distance r and height z the plane of the ray therefore passes within
            //     d_lat = |z| k_nt + r k_len
            // of the crossing computed for H_k.  On ordinary terrain that is nothing; next to a 300 m spike on a 1 m grid (lateral
            // slope 300) 9 mm sideways are 2.7 m of height -- the certificate of the cell beside the spike let rays start behind a
            // face they hit (one horizon value per configuration one search step low; found by the full-length re-trace).
            // The two factors travel in flags[1], flags[2]; phase 1 widens the bins by d_lat / r, phase 2 the crossing interval.
            flags[1] = __float_as_int(__builtin_fabsf(nt) + 1.0e-7f);
            flags[2] = __float_as_int(__builtin_fabsf(nn - 1.0f) + __builtin_fabsf(tt - 1.0f) + 1.0e-6f);
        } else { flags[1] = 0; flags[2] = 0; }
        flags[0] = (bad ? HZ_NR_FRAME : 0) | (bad_mesh ? HZ_NR_BAD_MESH : 0);
    }
    __syncthreads();
    const float dphi = 6.283185307179586f / (float)A;
    auto edge_ends = [&](int e, int &ia, int &ib) {                  // end points (window vertex numbers) of edge e
        if (e < NHOR) { const int r = e / (NV - 1), c = e % (NV - 1); ia = r * NV + c; ib = ia + 1; }
        else if (e < 2 * NHOR) { const int f = e - NHOR, r = f / NV, c = f % NV; ia = r * NV + c; ib = ia + NV; }
        else { const int f = e - 2 * NHOR, r = f / (NV - 1), c = f % (NV - 1); ia = r * NV + c + 1; ib = (r + 1) * NV + c; }   // (i, j+1) - (i+1, j)
    };
    // ---- edges, phase 1: per edge the azimuth bins it spans and the constants phase 2 needs (once per edge, in LDS) ------
    static_assert(NEDGE <= 255, "edge numbers travel in bytes");
    int bins_total = 0;
    for (int e0 = 0; e0 < NEDGE; e0 += 64) {
        const int e = e0 + lane;
        int k_lo = 0, bins = 0;
        if (valid && e < NEDGE) {
            int ia, ib;
            edge_ends(e, ia, ib);
            if (ia != CENTRE && ib != CENTRE) {                      // spokes: see the header
                const float ae = q[5 * ia], an = q[5 * ia + 1], az = q[5 * ia + 2];
                const float be = q[5 * ib], bn = q[5 * ib + 1], bz = q[5 * ib + 2];
                const float ra = vp[3 * ia], rb = vp[3 * ib];
                const float rmin = __builtin_fminf(ra, rb);
                if (!(rmin > 1.0e-3f)) atomicOr(&flags[0], HZ_NR_VERTEX_ON_AXIS);                 // a vertex (almost) above / below the origin
                else {
                    const float pa = vp[3 * ia + 1], pb = vp[3 * ib + 1];   // azimuth clockwise from north
                    float dl = pb - pa;
                    if (dl > 3.14159265f) dl -= 6.2831853f;
                    if (dl < -3.14159265f) dl += 6.2831853f;
                    if (__builtin_fabsf(dl) > 2.9f) atomicOr(&flags[0], HZ_NR_EDGE_OVER_AXIS);    // the edge passes (almost) over the origin
                    else {
                        const float lo = dl >= 0.0f ? pa : pb, span = __builtin_fabsf(dl);
                        // end points within the tolerance count as in the plane; + how far the ray's own plane can be off (flags[1..2])
                        const float m_az = 2.0e-3f + 0.02f / rmin + (__builtin_fmaxf(__builtin_fabsf(az), __builtin_fabsf(bz)) * __int_as_float(flags[1])) / rmin + __int_as_float(flags[2]);
                        // (grids with centimetre spacing: the tolerance would span a large part of the circle and the
                        //  bins would leave the (-A, 2 A) range phase 2 wraps once -- no certificate for such a cell)
                        if (m_az > 0.25f) atomicOr(&flags[0], HZ_NR_AZ_TOLERANCE);
                        else {
                            k_lo = (int)__builtin_floorf((lo - m_az) / dphi);
                            bins = (int)__builtin_ceilf((lo + span + m_az) / dphi) - k_lo + 1;
                            float *c = ec + NEC * e;
                            c[0] = ae; c[1] = an; c[2] = az; c[3] = be; c[4] = bn; c[5] = bz; c[6] = ra; c[7] = rb;
                        }
                    }
                }
            }
            ek[2 * e] = k_lo; ek[2 * e + 1] = bins;
        }
        bins_total += bins;
    }
    for (int off = 32; off > 0; off >>= 1) bins_total += __shfl_xor(bins_total, off);
    // Tasks of CH consecutive bins of one edge, one lane per task.  CH is chosen per cell so that the tasks fill two
    // rounds of the wave (<= 128 + one ragged task per edge): with a fixed CH = 16 (rounds 2-3) most cells had 70 ... 150
    // tasks, i.e. a nearly empty second or third round of 16 iterations each.
    const int CH = max(4, (bins_total + 99) / 100);
    int carry = 0;
    for (int e0 = 0; e0 < NEDGE; e0 += 64) {
        const int e = e0 + lane;
        const int tasks = (valid && e < NEDGE) ? (ek[2 * e + 1] + CH - 1) / CH : 0;
        int inc = tasks;                                             // inclusive scan over the wave
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
        const int start = carry + inc - tasks;
        if (e < NEDGE) pre[e] = start;
        for (int t = 0; t < tasks; t++) if (start + t < HZ_NEAR_MAXT) tmap[start + t] = (unsigned char)e;   // task -> edge
        carry += __shfl(inc, 63);
    }
    if (carry > HZ_NEAR_MAXT && lane == 0) atomicOr(&flags[0], HZ_NR_AZ_TOLERANCE);   // (cannot happen for A <= 2048: refusal, not an error)
    if (p.reasons != nullptr && valid && lane == 0) { atomicAdd(&p.reasons[17], (unsigned)bins_total); atomicAdd(&p.reasons[16], (unsigned)carry); }
    // (everything below the azimuth table is private to the wave: its LDS instructions execute in program order, so a
    //  fence against compiler reordering is enough and the four cells of a workgroup need not wait for each other)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- edges, phase 2: one task = one edge x <= CH azimuths; where does the edge cross those vertical planes? ------
    if (valid) {
        const int n_task = min(carry, HZ_NEAR_MAXT);
        for (int c = lane; c < n_task; c += 64) {
            const int e = (int)tmap[c];
            const float *cc = ec + NEC * e;
            const float ae = cc[0], an = cc[1], az = cc[2], be = cc[3], bn = cc[4], bz = cc[5], ra = cc[6], rb = cc[7];
            const float rmin = __builtin_fminf(ra, rb);
            const float tol_a = 1.0e-3f * ra + 0.01f, tol_b = 1.0e-3f * rb + 0.01f;
            const int first = ek[2 * e] + (c - pre[e]) * CH, last = min(first + CH, ek[2 * e] + ek[2 * e + 1]);
            const float r_big = __builtin_fmaxf(ra, rb);
            // absolute error bounds of the interpolated (r, z) of a crossing: the window coordinates q carry ~3 roundings of
            // magnitude <= 2^-24 |q| each, the interpolation two more (term C)
            const float er = 6.0e-7f * (ra + rb), ez = 6.0e-7f * ((__builtin_fabsf(az) + __builtin_fabsf(bz)) + (ra + rb));
            const float dz = bz - az;
            const float ninf = -__builtin_inff();
            // lateral uncertainty of the cutting plane at this edge: rounding of da / db (two rounded products and a difference:
            // <= 4 * 2^-24 r each) + the ray's own offset from H_k (frame terms of this cell: flags[1..2], see above)
            const float dl = 4.0e-7f * r_big + __builtin_fmaxf(__builtin_fabsf(az), __builtin_fabsf(bz)) * __int_as_float(flags[1]) + r_big * __int_as_float(flags[2]);
            int k = first;                                           // first lies in (-A, 2 A): wrapped once, then stepped
            if (k < 0) k += A;
            if (k >= A) k -= A;
            for (int kk = first; kk < last; kk++, k = (k + 1 == A) ? 0 : k + 1) {
#pragma clang fp contract(fast)      // bounds, not the bit-exact contract: FMAs only remove roundings the error terms allow for
                const float2 sc = reinterpret_cast<const float2 *>(tab)[k];
                const float sp = sc.x, cp = sc.y;
                const float da = ae * cp - an * sp, db = be * cp - bn * sp;     // signed distances from the plane
                const float fa = ae * sp + an * cp, fb = be * sp + bn * cp;     // along the azimuth (r of the end points)
                // (end points that lie IN the plane -- within the tolerance -- bound themselves: the vertex phase below)
                float cand = ninf;
                if (__builtin_fminf(da, db) <= dl && __builtin_fmaxf(da, db) >= -dl) {
                    // The edge crosses the plane -- or ends within dl of it: the plane of the ray may cross it there -- at parameter
                    // t = da / (da - db), clamped to the edge (an edge that only comes NEAR is evaluated at its near end).  The plane
                    // is known to within dl sideways, so |t - t_exact| <= dt = dl / |da - db| (a factor 1.6 in hand for the rounding
                    // part).  With x(t) = z(t) / r(t), z and r linear in t:  x(tau) - x(t) = (tau - t) (dz r(t) - z(t) df) /
                    // (r(tau) r(t))  EXACTLY, hence |x(tau) - x(t)| <= dt (|dz| + |x| |df|) / (0.95 r) for |tau - t| <= dt as
                    // long as dt |df| <= 0.05 r: the candidate is x + that bound (v_rcp_f32: 1 ulp, inside the margins).
                    const float rden = __builtin_amdgcn_rcpf(da - db);
                    const float dt = dl * __builtin_fabsf(rden) + 1.0e-6f;
                    if (!(dt <= 0.1f)) {
                        // both end points within rounding of the plane: the edge lies IN it; its end-point values bound it
                        if (!(__builtin_fabsf(da) <= tol_a && __builtin_fabsf(db) <= tol_b)) atomicOr(&flags[0], HZ_NR_INPLANE_EDGE);
                    } else {
                        const float t = __builtin_fminf(__builtin_fmaxf(da * rden, 0.0f), 1.0f), df = fb - fa;
                        const float r = fa + t * df, z = az + t * dz;
                        if (r > 0.0f) {
                            const float rr = __builtin_amdgcn_rcpf(r);
                            const float x = z * rr, adf = __builtin_fabsf(df), ax = __builtin_fabsf(x);
                            const float slope = __builtin_fabsf(dz) + ax * adf;
                            // refusals: a crossing close to the axis; an interval that is not small against r; an
                            // elevation error (ez + |x| er) / r / (1 + x^2) of this candidate above 5e-4 rad
                            if (!(r >= 0.25f * rmin)) atomicOr(&flags[0], HZ_NR_CROSSING_NEAR_AXIS);
                            if (!(dt * adf <= 0.05f * r)) atomicOr(&flags[0], HZ_NR_INTERVAL);
                            if (!(ez + ax * er <= 5.0e-4f * r * (1.0f + x * x))) atomicOr(&flags[0], HZ_NR_PRECISION);
                            cand = __builtin_fmaxf(cand, x + 1.06f * dt * slope * rr);
                        } else if (!(r < -0.05f * rmin)) atomicOr(&flags[0], HZ_NR_CROSSING_NEAR_AXIS);      // (within rounding of the axis)
                    }
                }
                if (cand > ninf) atomicMax(&E[k], f2o(cand));
            }
        }
        // ---- vertices: a window vertex that lies IN the half-plane of azimuth k (|d| <= 1e-3 r + 1 cm: the sign tests of the
        // crossings above cannot be trusted for it) on the ray's side of the axis bounds itself.  One lane per vertex, the few
        // bins within the azimuth tolerance of the vertex (2e-3 + 0.02 / r >= 2 asin(tol / r): every bin in which the test can
        // hold).  Rounds 2-4 ran this test for both end points of every edge in every bin of the edge: the same (vertex, bin)
        // pairs pass -- the bins of an edge cover the tolerance of its end points -- at ~10 of the edge loop's 52 VALU instructions.
        if (lane < NVERT && lane != CENTRE) {
            const float ve = q[5 * lane], vn = q[5 * lane + 1];
            const float vr = vp[3 * lane], pv = vp[3 * lane + 1], vx = vp[3 * lane + 2];
            const float m_v = 2.0e-3f + 0.02f / vr;
            if (vr > 1.0e-3f && !(m_v > 0.25f)) {       // (else: its edges have refused the cell)
                const float tol = 1.0e-3f * vr + 0.01f, half_r = 0.5f * vr;
                const int k0 = (int)__builtin_floorf((pv - m_v) / dphi), k1 = (int)__builtin_ceilf((pv + m_v) / dphi);
                int k = k0;                                          // in (-A, 2 A): wrapped once, then stepped
                if (k < 0) k += A;
                if (k >= A) k -= A;
                for (int kk = k0; kk <= k1; kk++, k = (k + 1 == A) ? 0 : k + 1) {
#pragma clang fp contract(fast)
                    const float2 sc = reinterpret_cast<const float2 *>(tab)[k];
                    const float d = ve * sc.y - vn * sc.x, f = ve * sc.x + vn * sc.y;
                    if (__builtin_fabsf(d) <= tol && f > half_r) atomicMax(&E[k], f2o(vx));
                }
            }
        }
    }
    // a window triangle (other than the six at the cell's own vertex) that contains the local vertical axis would be
    // seen at every azimuth, up to the zenith: its edges alone do not bound it -> no certificate for this cell
    // The same triangles must form a GRAPH over the cell's local horizontal plane: every window triangle projects onto
    // the local (east, north) plane with the same orientation and is not seen edge-on (|n . norm| > 1e-3 |n|).  The
    // tangent bound above rests on it -- along an azimuth the surface is then one curve z(r) that starts directly below
    // the origin.  With a frame tilted against very steep terrain (found by the random sweep of round 3: 1 m x 90 m
    // cells, 119 m steps, vec_norm 0.4 degrees off the vertical) an ADJACENT triangle can face away from vec_norm; the
    // origin then lies behind it, rays leaving upwards hit it at once, and its spokes -- which the edge phase skips
    // because they "cross the half-plane at the vertex below the origin" -- bound nothing.  No certificate then.
    if (valid) {
        bool pos = false, neg = false;
        for (int t = lane; t < 2 * NDIAG; t += 64) {
            const int qd = t >> 1, r = qd / (NV - 1), c = qd % (NV - 1);
            const int va = r * NV + c, vb = va + 1, vc = va + NV, vd = vc + 1;     // quad corners a, b / c, d
            const int i0 = (t & 1) ? vb : va, i1 = (t & 1) ? vd : vb, i2 = vc;     // (a, b, c) and (b, d, c)
            const float x0 = q[5 * i0], y0 = q[5 * i0 + 1], x1 = q[5 * i1], y1 = q[5 * i1 + 1], x2 = q[5 * i2], y2 = q[5 * i2 + 1];
            const float ux = x1 - x0, uy = y1 - y0, uz = q[5 * i1 + 2] - q[5 * i0 + 2];
            const float vx = x2 - x0, vy = y2 - y0, vz = q[5 * i2 + 2] - q[5 * i0 + 2];
            const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;   // nz: local projected area
            if (!(nz * nz > 1.0e-6f * ((nx * nx + ny * ny) + nz * nz))) atomicOr(&flags[0], HZ_NR_EDGE_ON);
            else if (nz > 0.0f) pos = true;
            else neg = true;
            if (i0 == CENTRE || i1 == CENTRE || i2 == CENTRE) {
                // The six triangles at the cell's own vertex: the ray origin must lie strictly ABOVE each of their planes
                // (local z of the plane at the axis: gamma < 0).  Then z / r = slope + gamma / r increases with r on
                // each of them, the surface along an azimuth is one continuous curve, and its maximum over the inner
                // ring is taken on the ring's outer edges -- which is why the spokes may be skipped.  Normally
                // gamma = -ray_org_elev; with coordinates of 1e6 m the origin v + norm * ray_org_elev is rounded to a
                // 0.25 m grid and can land ON or BELOW a steep adjacent triangle: no certificate then.
                const float z00 = q[5 * i0 + 2];
                const float lift = (nx * x0 + ny * y0) / nz;                     // nz != 0: checked above
                const float gamma = z00 + lift;
                const float tol_g = 1.0e-5f * (__builtin_fabsf(z00) + __builtin_fabsf(lift)) + 1.0e-6f;
                if (!(gamma < -tol_g)) atomicOr(&flags[0], HZ_NR_ORIGIN_BELOW);
                continue;
            }
            const float c0 = x0 * y1 - x1 * y0, c1 = x1 * y2 - x2 * y1, c2 = x2 * y0 - x0 * y2;   // origin vs the three edges
            const float tol = 1.0e-3f * (__builtin_fabsf(c0) + __builtin_fabsf(c1) + __builtin_fabsf(c2));
            if ((c0 >= -tol && c1 >= -tol && c2 >= -tol) || (c0 <= tol && c1 <= tol && c2 <= tol)) atomicOr(&flags[0], HZ_NR_AXIS_IN_TRIANGLE);
        }
        if (__ballot(pos) != 0ull && __ballot(neg) != 0ull && lane == 0) atomicOr(&flags[0], HZ_NR_ORIENTATION);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- distance to everything outside the window: the boundary polygon in the world (x, y) plane -----------------
    float rin = __builtin_inff();
    if (valid && lane < NSEG) {
        // boundary vertex ring, clockwise from the top-left corner
        int s0, s1;
        const int side = lane / (2 * W), t = lane % (2 * W);
        if (side == 0) { s0 = t; s1 = t + 1; }                                         // top row, left -> right
        else if (side == 1) { s0 = t * NV + (NV - 1); s1 = (t + 1) * NV + (NV - 1); }  // right column, down
        else if (side == 2) { s0 = (NV - 1) * NV + (NV - 1 - t); s1 = s0 - 1; }        // bottom row, right -> left
        else { s0 = (NV - 1 - t) * NV; s1 = s0 - NV; }                                 // left column, up
        const float ax = q[5 * s0 + 3], ay = q[5 * s0 + 4], bx = q[5 * s1 + 3], by = q[5 * s1 + 4];
        const float ux = bx - ax, uy = by - ay;
        const float uu = ux * ux + uy * uy;
        float t01 = uu > 0.0f ? -(ax * ux + ay * uy) / uu : 0.0f;
        t01 = __builtin_fminf(__builtin_fmaxf(t01, 0.0f), 1.0f);
        const float cx = ax + t01 * ux, cy = ay + t01 * uy;
        rin = __builtin_sqrtf(cx * cx + cy * cy);
    }
    for (int off = 32; off > 0; off >>= 1) rin = __builtin_fminf(rin, __shfl_xor(rin, off));
    // (a certificate whose radius is not positive would start the cell's box tests at 0 instead of -tau (hz_common.h): none)
    const float near_rad = 0.97f * rin - 4.0f * p.pad;
    const bool ok = valid && flags[0] == 0 && rin < 1.0e30f && near_rad > 0.0f;
    if (p.reasons != nullptr && have && lane == 0) {      // debug histogram (HZ_NEAR_REASONS=1): why cells got no certificate
        const int f = valid ? flags[0] : HZ_NR_WINDOW;
        atomicAdd(&p.reasons[0], 1u);
        if (ok) atomicAdd(&p.reasons[1], 1u);
        for (int b = 0; b < 14; b++) if (f & (1 << b)) atomicAdd(&p.reasons[2 + b], 1u);
    }
    // ---- table index per azimuth ---------------------------------------------------------------------------------
    if (have) {
        // elevation margin: two table steps, at least 2 mrad (a ray that clears the window by less is not shortened)
        const float marg = __builtin_fmaxf(2.0f * p.step, 2.0e-3f);
        for (int k = lane; k < A; k += 64) {
            unsigned short out = 65535;
            if (ok) {
                const float t = o2f(E[k]);
                if (t == -__builtin_inff()) out = 0;
                else if (t == t && t < 1.0e30f) {
                    // elev_ang[i] = up - step (elev_num - 1 - i)  (horizon_comp.cpp:723-730): first index at or above el, + 1
                    const float el = atanf(t) + marg;
                    const float fi = __builtin_ceilf((el - p.up) / p.step) + (float)(p.elev_num - 1) + 1.0f;
                    out = (unsigned short)__builtin_fminf(__builtin_fmaxf(fi, 0.0f), 65535.0f);
                }
            }
            p.near_idx[(size_t)cl * A + k] = out;
        }
        if (lane == 0) p.near_r[cl] = ok ? near_rad : 0.0f;
    }
}

int near_launch(const Scene *sc, const NearArgs &a, hipStream_t st) {
    NearParams p;
    p.verts = sc->verts();
    p.vec_norm = a.vec_norm; p.vec_north = a.vec_north; p.mask = a.mask;
    p.azim_sin = a.azim_sin; p.azim_cos = a.azim_cos;
    p.d0 = sc->hdr.d0; p.d1 = sc->hdr.d1; p.offset_0 = a.offset_0; p.offset_1 = a.offset_1; p.dim_in_1 = a.dim_in_1;
    p.row_begin = a.row_begin; p.n_cells = (a.row_end - a.row_begin) * a.dim_in_1;
    p.azim_num = a.azim_num; p.elev_num = a.elev_num;
    p.ray_org_elev = a.ray_org_elev; p.low = a.low; p.step = (float)((double)a.hori_acc / 5.0); p.up = a.up;
    p.pad = sc->hdr.pad;
    p.near_idx = a.near_idx; p.near_r = a.near_r; p.reasons = a.reasons;
    // per-cell guard for scenes that are not a height field everywhere (a.ignore_bad_map: the tests' override)
    p.bad_bits = nullptr; p.bad_nb = 0; p.bad_x0 = p.bad_y0 = p.bad_sx = p.bad_sy = 0.0f;
    if ((sc->hdr.flags & HZ_BLOB_BAD_MAP) && !a.ignore_bad_map) {
        p.bad_bits = reinterpret_cast<const uint32_t *>((const char *)sc->blob + sc->hdr.off_bad);
        p.bad_nb = sc->hdr.bad_nb; p.bad_x0 = sc->hdr.bad_x0; p.bad_y0 = sc->hdr.bad_y0; p.bad_sx = sc->hdr.bad_sx; p.bad_sy = sc->hdr.bad_sy;
    }
    if (p.n_cells <= 0) return HZ_OK;
    const size_t lds = ((size_t)2 * a.azim_num + (size_t)4 * near_per_wave_words<HZ_NEAR_W>(a.azim_num)) * sizeof(float);
    HZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_near_cert<HZ_NEAR_W>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int grid = (p.n_cells + 3) / 4;
    hipLaunchKernelGGL(k_near_cert<HZ_NEAR_W>, dim3(grid), dim3(256), lds, st, p);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

// largest azimuth count the certificates support (LDS of the pre-pass)
int near_max_azim() { return 2048; }

}  // namespace hz
