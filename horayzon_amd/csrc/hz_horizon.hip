// hz_horizon.hip -- terrain horizon kernels for gfx950 (wave64).
//
// Replaces the TBB row loop + per-cell search functions + Embree rtcOccluded1 of
// the reference (horizon_comp.cpp:739-800, :302-498, :241-262).
//
// Work decomposition
//   one lane      = one inner-domain grid cell; it walks ALL azimuth sectors
//                   sequentially (guess_constant carries the elevation index from
//                   one azimuth to the next, horizon_comp.cpp:431-496)
//   one wavefront = an 8 x 8 block of cells -> neighbouring lanes shoot nearly
//                   parallel rays and fetch the same BVH nodes (coalesced by the TA)
//   one workgroup = 4 wavefronts; per-lane traversal stacks and the output
//                   staging (4 azimuths -> one 16 B store) live in LDS
//   blocks        -> the launch has as many workgroups as are resident at once and every WAVE pulls
//                   8 x 8 blocks from the queue of the XCD it runs on (persistent waves, round 5); the order
//                   of a queue is that of the XCD-aware tile map: each of the 8 XCDs walks compact patches
//                   of 16 x 16 tiles, so that one XCD's L2 sees one region of the BVH.  (Launches with fewer
//                   tiles than that, redo launches and the counting monitor: workgroup b = tile b, wave = quadrant.)
//   leftover cells: a block ends when at most 36 of its cells are unfinished (hz_opts.left_min); their records are sorted by
//                   (azimuths left, position) and a follow-up launch (LEFT) finishes them, 64 per wave.
//
// Per lane a small state machine (Search) produces the next elevation sample as
// soon as the previous occlusion query finishes; lanes never wait for an azimuth
// barrier.  When fewer than `regroup` lanes of a wave are still traversing, the
// wave leaves the traversal loop (ballot + popcount) so idle lanes can fetch
// their next ray: this is the ray compaction step.  A ray expected to be blocked first
// walks the subtree above the leaf that blocked the cell's previous ray (hit cache).
//
// The float/double promotion pattern of the reference's index arithmetic is
// reproduced exactly (hard part 2); tables are built on the
// host with the reference's expressions (hz_api.hip) and only read here.
#include "hz_search.h"
#include <vector>

namespace hz {

#ifndef HZ_TPB
#define HZ_TPB 256    // threads per workgroup: 4 waves = a 16 x 16 tile (64 / 128: probes with 1 / 2 waves per workgroup; the tile stays the unit of the XCD mapping)
#endif
#define HZ_WPB (HZ_TPB / 64)
#ifndef HZ_QLEN
#define HZ_QLEN 2       // leaves a lane sets aside before it needs a leaf step (3 and 4 measured slower)
#endif

struct HorizonParams {
    SceneView sv;
    Tables tb;
    const float *vec_norm, *vec_north;
    const uint8_t *mask;
    float *hori;
    int offset_0, offset_1, dim_in_1;
    int row_begin, row_end;
    TileMap tm;                    // tile grid of the slab -> workgroups (XCD aware)
    float dist, hori_fill, ray_org_elev;
    float dist_box, neg_tau;       // dist + 2 tau and -tau (hz_common.h: where the box tests start and end), formed on the host: uniform
                                   // float arithmetic in the kernel would sit in vector registers, and there is none to spare
    int top_nodes, regroup, stack_bytes, leaf_bias, stage_bytes, hit_cache, stack_cap;
    int pre_bytes;                 // LDS in front of the stack: the output staging buffer, or (no staging) two padding rows --
                                   // the fast stack reads the rows below its sentinel together with the top (hz_trace)
    const unsigned short *near_idx;   // near-field certificates of this launch's rows (hz_near.hip) or null
    const float *near_r;
    float *scratch_row;            // COUNT only: null, or the one row every lane stores into instead of its row of `hori` (certificate monitor)
    int verify_near;               // 0: off; else re-trace the shortened rays selected by verify_mask over their full length
    unsigned verify_mask;          //   (power of two - 1: one of every verify_mask + 1 shortened rays; 0: all of them)
    unsigned long long *counters;
    const int *tile_list;          // null, or the n_list blocks (workgroup number * 4 + wave, of the full launch) to repeat
    int n_list;
    int *redo_list;                // !LEVELSTACK: waves whose stack overflowed append their block here (count: counters[8])
    // Leftover cells (rounds 5 - 6): a block ENDS when at most left_min of its cells are unfinished; those lanes append their cell's state
    // (HZ_LEFT_WORDS words per cell) to the OUT region of left_rec (out_ctl[0] = slots allocated there), and a later launch (left_mode =
    // level >= 1, the LEFT instantiation) finishes them, 64 records per wave -- and may hand over again, to the next level's region
    // (left_min > 0 there too).  Between the two launches the records are SORTED (left_sort below: by azimuths still to do, then by
    // position): a LEFT wave takes 64 consecutive entries of the sorted permutation left_perm -- cells with the same number of
    // azimuths left, next to each other on the DEM -- so that its rays are coherent and its lanes finish together.  in_ctl[1] = valid
    // records (written by the sort's key kernel), in_ctl[8 + x] = groups handed to XCD x.  A region out of room stops the hand-over.
    int left_min, left_mode;
    unsigned left_in_base;                   // LEFT: first record of the region to finish
    unsigned left_out_base, left_out_cap;    // left_min > 0: first record and capacity of the region this launch hands over to
    unsigned *left_rec, *left_in_ctl, *left_out_ctl;
    const unsigned *left_perm;               // LEFT: sorted record numbers (relative to left_in_base)
    int persist;                   // 1: persistent waves -- the launch has as many workgroups as are resident at once and every WAVE pulls 8 x 8
    unsigned *queue;               //    blocks from the queue of its XCD (queue[x] = blocks of XCD x handed out so far) until all are empty
};

// LDS: [ per-lane stacks int[depth][256] | output staging float[4][256] | top-of-tree nodelet Node[top_nodes] ]
// LEVELSTACK: the traversal's stack discipline (hz_common.h).  false = one LDS entry per pending sibling: fewest VALU
// instructions, `stack_cap` entries, overflow flagged in counters[8]; true = one entry per tree level, cannot overflow.
// (the counting instantiation carries ~20 more live values: at 5 workgroups per CU it spilled 10 - 22 VGPRs, so it is built
//  for 4.  Its tallies -- rays, node visits, triangle tests, wave iterations -- are functions of the lanes' states only, not
//  of the schedule, so they are the production launch's numbers: the ray counts of the two instantiations are compared by the tests.)
// One leftover record (written once per hand-over of a cell, at the end of a block).  Word 0 is the launch-local cell number
// (row - row_begin) * dim_in_1 + column (32 bits: the host switches the hand-over off for launches of 2^32 - 1 cells or more);
// ~0u marks a slot that was reserved and never filled.
__device__ __forceinline__ void hz_left_write(unsigned *w, unsigned cert, unsigned k, unsigned flags,
                                                        unsigned ind, unsigned prev, unsigned pazim, unsigned count, float lim_up, float lim_low,
                                                        float elev_samp, float ev, unsigned cache, float st0, float st1, float st2) {
    w[0] = cert;
    w[1] = k; w[2] = flags; w[3] = ind; w[4] = prev; w[5] = pazim; w[6] = count;
    w[7] = __float_as_uint(lim_up); w[8] = __float_as_uint(lim_low); w[9] = __float_as_uint(elev_samp); w[10] = __float_as_uint(ev);
    w[11] = cache; w[12] = __float_as_uint(st0); w[13] = __float_as_uint(st1); w[14] = __float_as_uint(st2); w[15] = 0u;
}

// LEFT: the leftover instantiation (p.left_mode): a wave takes 64 records of cells that production blocks left unfinished instead of
// an 8 x 8 block.  A template parameter and not a run-time switch: the restore code in front of the main loop cost the production
// kernel 5 % (two rematerialised instructions in the node step, scratch reloads in the refill) although it never ran there.
template <int ALG, bool COUNT, bool STAGE, bool NODELET, bool LEVELSTACK, bool LEFT = false>
#ifndef HZ_WG_PER_CU
#define HZ_WG_PER_CU 5     // resident workgroups per CU the register allocation is held to (6: 80 VGPRs, measured slower)
#endif
__global__ __launch_bounds__(HZ_TPB, (COUNT ? 4 : HZ_WG_PER_CU) * (4 / HZ_WPB)) void k_horizon(HorizonParams p_arg) {
    // The parameters are read through the kernel-argument segment (constant address space: scalar loads), and -- see the block loop
    // below -- through a pointer the compiler cannot see through at the top of every pass: with the plain by-value argument every
    // parameter load and everything derived from parameters only was hoisted out of the block loop and then lived through the
    // traversal loop (66 spilled SGPRs, 25 v_readlane per node / leaf step, 7 VGPRs in scratch; one block per launch: 15 / 0 / 0).
    typedef const __attribute__((address_space(4))) HorizonParams *hz_kparams;
    hz_kparams pk = (hz_kparams)__builtin_amdgcn_kernarg_segment_ptr();
    // (the host pass of hipcc only parses this body; it has no constant address space to copy from)
#if defined(__HIP_DEVICE_COMPILE__)
#define HZ_LOAD_PARAMS(dst) dst = *pk
#else
#define HZ_LOAD_PARAMS(dst) dst = p_arg
#endif
    HorizonParams p;
    HZ_LOAD_PARAMS(p);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *stack = reinterpret_cast<int *>(smem + p.pre_bytes);
    const float4 *top = reinterpret_cast<const float4 *>(smem + p.pre_bytes + p.stack_bytes);
    const int tid = threadIdx.x;
    const int ntop = p.top_nodes;
    if (NODELET && ntop > 0) {   // stage the breadth-first top of the tree in LDS (coalesced 16 B per lane)
        float4 *dst = reinterpret_cast<float4 *>(smem + p.pre_bytes + p.stack_bytes);
        const float4 *src = reinterpret_cast<const float4 *>(p.sv.nodes);
        for (int i = tid; i < ntop * 2; i += HZ_TPB) dst[i] = src[i];
        __syncthreads();
    }

    // wave -> 8 x 8 block of cells: quadrant `wave` of the tile of workgroup blockIdx.x, or -- in a launch that repeats
    // blocks whose fast stack overflowed -- the block the list names (entry = workgroup number * 4 + quadrant)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar register: it lives through the block loop)
    // Persistent waves (round 5, p.persist): a workgroup's slot -- its LDS -- is only free again when the SLOWEST of its four waves
    // has finished its block, so with one tile per workgroup 6 - 9 % of the wave slots stood empty in the middle of a launch
    // (profiles/r05/wg_trace_wpb4_slab447.json: wave lifetimes 29 ... 38 ms around a mean of 33).  Now the launch has only as many
    // workgroups as are resident at once, and every wave, on its own, pulls the next 8 x 8 block from the queue of the XCD it
    // runs on (the order inside an XCD is the order of the tile map: the waves resident on an XCD still work on one compact
    // piece of the DEM), then from the other XCDs' queues.  A block is computed exactly as before: results cannot change.
    const int xcc_own = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);      // HW_REG_XCC_ID, bits 3:0
    int xcc_off = 0;                 // queues xcc_own .. xcc_own + xcc_off - 1 (mod 8) were found empty
    int left_on = 1;                 // 0: the region of the handed-over cells is full, this wave's blocks run to their end
  for (;;) {                         // one 8 x 8 block per pass (exactly one pass without p.persist)
    asm volatile("" : "+s"(pk));
    HZ_LOAD_PARAMS(p);
    // (the same for what is derived from the thread number -- lane, lane >> 3, lane & 7, LDS addresses: formed again in every pass
    //  from a copy the compiler cannot see through, or they are hoisted out of the block loop and live through the traversal)
    int tid_b = tid;
    asm volatile("" : "+v"(tid_b));
    const int lane = tid_b & 63;
    // (what the traversal loop and the refill read stays in registers: the compiler would otherwise re-load these from the
    //  argument segment INSIDE those loops -- an s_load and its wait in front of every vote)
#define HZ_KEEP(x) asm volatile("" : "+s"(x))
    HZ_KEEP(p.leaf_bias); HZ_KEEP(p.regroup); HZ_KEEP(p.neg_tau); HZ_KEEP(p.sv.cx); HZ_KEEP(p.sv.cy); HZ_KEEP(p.sv.cz);
    HZ_KEEP(p.near_idx); HZ_KEEP(p.near_r);
#undef HZ_KEEP
    int ti = 0, tj = 0;
    int blk = (int)blockIdx.x * HZ_WPB + wave;
    unsigned rec0 = 0u, rec_n = 0u;                  // LEFT: first entry of the sorted permutation of this wave's group, entries in it
    if (LEFT) {
        // the next group of 64 sorted records.  The sorted order is cut into runs of 32 groups (2048 records: neighbours on the DEM
        // with the same number of azimuths left) that go round robin to the XCDs; a wave takes the next group of its XCD's runs,
        // then of the other XCDs'.
        blk = -1;
        const unsigned groups = (p.left_in_ctl[1] + 63u) / 64u;
        if (p.tile_list) {           // the groups whose fast stack overflowed, one per wave (left_mode launch with a list)
            const int b = (int)blockIdx.x * HZ_WPB + wave;
            const int g = b < p.n_list ? p.tile_list[b] : -1;
            if (g >= 0 && (unsigned)g < groups) { blk = g; rec0 = (unsigned)g * 64u; rec_n = min(64u, p.left_in_ctl[1] - (unsigned)g * 64u); }
        } else
        while (xcc_off < 8) {
            const unsigned x = (unsigned)((xcc_own + xcc_off) & 7);
            unsigned q = 0u;
            if (lane == 0) q = atomicAdd(&p.left_in_ctl[8 + x], 1u);
            q = (unsigned)__builtin_amdgcn_readfirstlane((int)q);
            const unsigned g = (((q >> 5) * 8u + x) << 5) | (q & 31u);
            if (g < groups) { blk = (int)g; rec0 = g * 64u; rec_n = min(64u, p.left_in_ctl[1] - g * 64u); break; }
            xcc_off++;                                           // (g grows with q: this XCD's runs are used up)
        }
        if (blk < 0) break;
    } else if (p.persist) {
        blk = -1;
        while (xcc_off < 8) {
            const int x = (xcc_own + xcc_off) & 7;
            int q = 0;
            if (lane == 0) q = (int)atomicAdd(&p.queue[x], 1u);
            q = __builtin_amdgcn_readfirstlane(q);
            if (q < p.tm.per_xcd * 4) { blk = (((q >> 2) * 8 + x) << 2) | (q & 3); break; }
            xcc_off++;
        }
        if (blk < 0) break;
    } else if (p.tile_list) blk = (blk < p.n_list) ? p.tile_list[blk] : -1;
    else if (HZ_WPB != 4) {
        // fewer than 4 waves per workgroup: the 4 / HZ_WPB workgroups of one tile follow each other ON THE SAME XCD
        // (workgroup b runs on XCD b % 8), so the tile -> XCD mapping is that of the 4-wave kernel
        const int b = (int)blockIdx.x, x = b & 7, t = b >> 3, per = 4 / HZ_WPB;
        blk = (((t / per) * 8 + x) * 4) + (t % per) * HZ_WPB + wave;
    }
    blk = __builtin_amdgcn_readfirstlane(blk);       // (wave uniform on every path: keep it out of the vector registers)
    bool has_tile = blk >= 0 && (LEFT || hz_tile_of_block(p.tm, blk >> 2, &ti, &tj));
    int i = p.row_begin + ti * 16 + ((blk >> 1) & 1) * 8 + (lane >> 3);
    int j = tj * 16 + (blk & 1) * 8 + (lane & 7);
    const unsigned *rec = nullptr;                   // left_mode: this lane's record
    if (LEFT) {
        has_tile = (unsigned)lane < rec_n;
        if (has_tile) {
            rec = p.left_rec + ((size_t)p.left_in_base + (size_t)p.left_perm[rec0 + (unsigned)lane]) * HZ_LEFT_WORDS;
            const unsigned c0 = rec[0];
            has_tile = c0 != 0xffffffffu;
            const unsigned di = c0 / (unsigned)p.dim_in_1;
            i = p.row_begin + (int)di; j = (int)(c0 - di * (unsigned)p.dim_in_1);
        }
    }
    const bool in_dom = has_tile && (i < p.row_end) && (j < p.dim_in_1);

    const Tables &t = p.tb;
    const unsigned long long t_start = COUNT ? (unsigned long long)wall_clock64() : 0ull;
    const size_t cell = in_dom ? ((size_t)i * p.dim_in_1 + j) : 0;
    bool done = !in_dom;
    // launch-local cell number: the only per-cell address state kept across the traversal (the output pointer
    // and the certificate row are rebuilt from it at every refill: 1 VGPR instead of 2 + 2)
    const unsigned cert = in_dom ? (unsigned)(i - p.row_begin) * (unsigned)p.dim_in_1 + (unsigned)j : 0u;
    // (the certificate monitor -- a counting launch NEXT TO the production launch, hz_api.hip -- stores into one scratch row: the
    //  rows of `hori` belong to the production launch)
    const bool to_scratch = COUNT && p.scratch_row != nullptr;
    float *const hori0 = to_scratch ? p.scratch_row : p.hori + (size_t)p.row_begin * p.dim_in_1 * (size_t)t.azim_num;   // first cell of this launch
    const unsigned cell_stride = to_scratch ? 0u : 1u;
    Sink out;
    out.hori = COUNT ? hori0 + (size_t)(cert * cell_stride) * (size_t)t.azim_num : hori0 + (size_t)cert * (size_t)t.azim_num;
    out.dist = nullptr; out.dist_hit = 0.0f;
    out.stage = reinterpret_cast<float *>(smem) + tid_b;   // only touched when STAGE
    out.stride = HZ_TPB;
    float ox = 0, oy = 0, oz = 0;
    float r01 = 0, r02 = 0, r11 = 0, r12 = 0, r21 = 0, r22 = 0;
    const bool masked = in_dom && p.mask[cell] != 1;
    {   // masked cells get hori_fill for every azimuth (horizon_comp.cpp:789-794).  The wave fills them together, one
        // cell after the other with consecutive lanes on consecutive azimuths: 256 B per store instruction instead of
        // 64 scattered 4 B stores at stride 4 A (8 x write amplification on ocean-masked domains).
        unsigned long long m = __ballot(masked);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            float *row = COUNT ? hori0 + (size_t)((unsigned)__shfl((int)cert, src) * cell_stride) * (size_t)t.azim_num
                               : hori0 + (size_t)__shfl((int)cert, src) * (size_t)t.azim_num;
            for (int k = lane; k < t.azim_num; k += 64) row[k] = p.hori_fill;
        }
    }
    if (in_dom) {
        if (masked) {
            done = true;
        } else {                                              // :751-779
            const float norm_x = p.vec_norm[3 * cell], norm_y = p.vec_norm[3 * cell + 1], norm_z = p.vec_norm[3 * cell + 2];
            const float north_x = p.vec_north[3 * cell], north_y = p.vec_north[3 * cell + 1], north_z = p.vec_north[3 * cell + 2];
            const float *v = p.sv.verts + 3 * ((size_t)(i + p.offset_0) * p.sv.d1 + (size_t)(j + p.offset_1));
            ox = v[0] + norm_x * p.ray_org_elev;
            oy = v[1] + norm_y * p.ray_org_elev;
            oz = v[2] + norm_z * p.ray_org_elev;
            // (east = north x norm is formed again at every refill: 9 instructions there against 3 registers that would
            //  live through the traversal -- the kernel sits exactly at the 96 VGPRs of 5 workgroups per CU)
            r01 = north_x; r02 = norm_x;
            r11 = north_y; r12 = norm_y;
            r21 = north_z; r22 = norm_z;
        }
    }
    // (the origin in the scene-centred frame is formed from (ox, oy, oz) where it is needed -- three subtractions per ray
    //  instead of three more registers that live through the traversal; the empty asm stops the compiler from hoisting them)
#define HZ_OC(ocx, ocy, ocz) float ocx, ocy, ocz; { float ax_ = ox, ay_ = oy, az_ = oz; \
        asm volatile("" : "+v"(ax_), "+v"(ay_), "+v"(az_)); \
        ocx = ax_ - p.sv.cx; ocy = ay_ - p.sv.cy; ocz = az_ - p.sv.cz; }
    const float tfar = p.dist;

    Search s;
    s.k = 0; s.phase = PH_NEWAZ; s.ind = 0; s.prev = 0; s.pazim = 0; s.count = 0;
    s.lim_up = 0; s.lim_low = 0; s.elev_samp = 0; s.ev = 0;
    unsigned rays = 0, guards = 0, w_adv = 0;
    TravCounters tc; tc.nodes = 0; tc.tris = 0; tc.w_nodes = 0; tc.w_leaves = 0;   // COUNT only
    const unsigned cells_cnt = (in_dom && !done && !LEFT) ? 1u : 0u;
    bool ray_active = false, last_hit = false;
    float dx = 0, dy = 0, dz = 1;
    RayBox rb = hz_raybox(0, 0, 0, 0, 0, 1);
    TravState ts; hz_trav_reset(ts);
    int cache = 0;           // hit cache: subtree above the leaf that blocked this cell's last blocked ray
    bool second = false;     // the cache walk found nothing: the root traversal is still due
    bool overflow = false;   // !LEVELSTACK: a ray needed more stack entries than this launch has (see hz_trace)
    // near-field certificate of this cell (hz_near.hip): rays of azimuth k with a table index >= near_idx[k] clear
    // everything within near_r of the origin and start their box tests at parameter near_r
    // verify_near (COUNT): a shortened ray that is selected (want_v) is traced a second time over its full length
    // (verifying) and the two decisions are compared.  Lane state is kept in flags only (lane masks in scalar registers)
    // and the tallies are wave-uniform popcounts.
    unsigned shortened = 0;                        // COUNT only
    unsigned w_verified = 0, w_violations = 0;     // wave-uniform
    bool verifying = false, first_result = false, want_v = false;
    float tn = 0.0f;
    bool had_guard = false;          // left_mode: the cell was counted as a guard cell by the launch that started it
    if (LEFT && !done) {
        // a cell another wave left unfinished: its search state, its hit cache, the output values it had staged -- and the ray that
        // was in flight, issued again here from (s.ind, s.k) over its full length from the root (the decision cannot differ: no
        // certificate, no cache walk, section 4 of DESIGN.md); it was counted when it was issued first
        s.k = (int)rec[1]; s.phase = (int)(rec[2] & 0xffu); s.ind = (int)rec[3]; s.prev = (int)rec[4]; s.pazim = (int)rec[5]; s.count = (int)rec[6];
        s.lim_up = __uint_as_float(rec[7]); s.lim_low = __uint_as_float(rec[8]); s.elev_samp = __uint_as_float(rec[9]); s.ev = __uint_as_float(rec[10]);
        cache = (int)rec[11];
        last_hit = (rec[2] & 0x100u) != 0u;
        had_guard = (rec[2] & 0x400u) != 0u;
        if (STAGE) { out.stage[0] = __uint_as_float(rec[12]); out.stage[out.stride] = __uint_as_float(rec[13]); out.stage[2 * out.stride] = __uint_as_float(rec[14]); }
        if (rec[2] & 0x200u) {
            const float ec = t.elev_cos[s.ind], es = t.elev_sin[s.ind];
            const float rx = ec * t.azim_sin[s.k], ry = ec * t.azim_cos[s.k], rz = es;
            const float r00 = r11 * r22 - r21 * r12, r10 = r21 * r02 - r01 * r22, r20 = r01 * r12 - r11 * r02;      // east = north x norm
            dx = (r00 * rx + r01 * ry) + r02 * rz;
            dy = (r10 * rx + r11 * ry) + r12 * rz;
            dz = (r20 * rx + r21 * ry) + r22 * rz;
            HZ_OC(ocx, ocy, ocz)
            rb = hz_raybox(ocx + p.neg_tau * dx, ocy + p.neg_tau * dy, ocz + p.neg_tau * dz, dx, dy, dz);
            hz_trav_reset(ts);
            ray_active = true;
        }
    }

    // (a block that starts with few cells -- the ragged rim of the domain -- is not worth a hand-over)
    int left_min = (COUNT || !left_on || __popcll(__ballot(!done)) <= p.left_min + 8) ? 0 : p.left_min;
    while (__ballot(!done) != 0ull) {
        if (!COUNT && left_min > 0 && __popcll(__ballot(!done)) <= left_min) {
            // cells still unfinished (<= left_min of them): they go to the next leftover launch.  A wave whose fast stack overflowed
            // hands nothing over: its block is computed again.
            const unsigned long long um = __ballot(!done);
            if (um != 0ull && !(!LEVELSTACK && __ballot(overflow) != 0ull)) {
                const unsigned n = (unsigned)__popcll(um);
                unsigned *const out_rec = p.left_rec + (size_t)p.left_out_base * HZ_LEFT_WORDS;
                unsigned base = 0u;
                if (lane == 0) base = atomicAdd(&p.left_out_ctl[0], n);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (base + n > p.left_out_cap) {
                    // the region is full: this block runs to its end, as do this wave's later ones; the slots the counter moved
                    // over below the capacity stay unfilled (~0u in word 0: the sort's key kernel skips them)
                    left_on = 0; left_min = 0;
                    if (base + (unsigned)lane < p.left_out_cap) out_rec[(size_t)(base + (unsigned)lane) * HZ_LEFT_WORDS] = 0xffffffffu;
                } else {
                    if (lane == 0) atomicAdd(&p.counters[LEFT ? 29 : 28], (unsigned long long)n);      // (statistics: hz_stats.left_cells / left_again)
                    if (!done) {
                        const unsigned rank = (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
                        const unsigned flags = (unsigned)s.phase | (last_hit ? 0x100u : 0u) | (ray_active ? 0x200u : 0u) | ((guards != 0u || had_guard) ? 0x400u : 0u);
                        hz_left_write(out_rec + (size_t)(base + rank) * HZ_LEFT_WORDS, cert, (unsigned)s.k, flags,
                                      (unsigned)s.ind, (unsigned)s.prev, (unsigned)s.pazim, (unsigned)s.count, s.lim_up, s.lim_low, s.elev_samp, s.ev, (unsigned)cache,
                                      STAGE ? out.stage[0] : 0.0f, STAGE ? out.stage[out.stride] : 0.0f, STAGE ? out.stage[2 * out.stride] : 0.0f);
                        done = true;
                    }
                }
            }
        }

        if (__ballot(!done) == 0ull) break;
        // ---- refill: lanes without a ray take the next sample of their search -----------------
        if (!done && !ray_active) {
            if (COUNT) HZ_WAVE_TICK(w_adv, lane);
            // (the per-cell addresses are rebuilt from `cert` HERE, at every refill: the empty asm keeps the compiler from
            //  hoisting the 64-bit products out of the loop -- it then spilled them, and every refill paid three scratch
            //  reloads with a full memory wait each: vector-memory instructions are what this kernel is short of)
            unsigned cert_r = cert;
            asm volatile("" : "+v"(cert_r));
            out.hori = COUNT ? hori0 + (size_t)(cert_r * cell_stride) * (size_t)t.azim_num : hori0 + (size_t)cert_r * (size_t)t.azim_num;
            if (advance<ALG, STAGE>(s, last_hit, t, out, guards)) {
                // local direction (east, north, up) and rotation: horizon_comp.cpp:357-361, :55-62
                // every load of the new ray is issued before the first one is used: the table entries, and the certificate of
                // (cell, azimuth) with the cell's radius -- read unconditionally: a load that waits for the comparison of
                // another load is one more memory round trip per refill (three in a row cost 3.5 %)
                const float ec = t.elev_cos[s.ind], es = t.elev_sin[s.ind];
                const float asn = t.azim_sin[s.k], acs = t.azim_cos[s.k];
                int near_i = 0x7fffffff;
                float near_rad = 0.0f;
                if (p.near_idx != nullptr) {
                    near_i = (int)p.near_idx[(size_t)cert_r * (size_t)t.azim_num + s.k];
                    near_rad = p.near_r[cert_r];
                }
                const float rx = ec * asn, ry = ec * acs, rz = es;
                float nx_ = r01, ny_ = r11, nz_ = r21;        // (the empty asm keeps the products inside the loop)
                asm volatile("" : "+v"(nx_), "+v"(ny_), "+v"(nz_));
                const float r00 = ny_ * r22 - nz_ * r12;      // east = north x norm (horizon_comp.cpp:763-766)
                const float r10 = nz_ * r02 - nx_ * r22;
                const float r20 = nx_ * r12 - ny_ * r02;
                dx = (r00 * rx + r01 * ry) + r02 * rz;
                dy = (r10 * rx + r11 * ry) + r12 * rz;
                dz = (r20 * rx + r21 * ry) + r22 * rz;
                    tn = (s.ind >= near_i) ? near_rad : p.neg_tau;    // no certificate: the box tests start at -tau (hz_common.h)
                if (COUNT && tn > 0.0f) shortened++;
                if (COUNT) want_v = p.verify_near && tn > 0.0f && (((rays + cert) & p.verify_mask) == 0u);
                HZ_OC(ocx, ocy, ocz)
                rb = hz_raybox(ocx + tn * dx, ocy + tn * dy, ocz + tn * dz, dx, dy, dz);
                hz_trav_reset(ts);
                // a ray below the previous azimuth's horizon is expected to be blocked near the same ridge
                second = p.hit_cache && (cache != 0) && (s.ind <= s.pazim) && (s.k > 0);
                if (second) {
                    ts.node = cache;
                    // Fast stack (round 5): the ROOT waits in entry 1, below the cached subtree.  A cache walk that finds nothing
                    // pops it and carries on with the full traversal inside hz_trace -- the same node visits and triangle tests
                    // as leaving the loop with "miss" and coming back with a reset state (which is what `second` still does for
                    // the level stack, whose entries cannot name the root), but the lane does not idle until its wave leaves.
                    // Depth: the walk below the cached node needs <= 3 * anc_levels entries above this one, far below the
                    // root traversal's own maximum, so no launch overflows that did not before.
                    if (!LEVELSTACK) { ts.sp = 1; stack[HZ_TPB + tid] = 0; second = false; }
                }
                ray_active = true;
                rays++;
            } else {
                done = true;
            }
        }
        // ---- traversal (hz_common.h: speculative while-while, one postponed leaf per lane) ------
        bool start_v = false, viol = false;
        if (ray_active) {
            int r;
            r = hz_trace<HZ_TPB, COUNT, HZ_QLEN, NODELET, LEVELSTACK>(p.sv.nodes, p.sv.prims, top, ntop, stack, tid, ox, oy, oz,
                                                 dx, dy, dz, tfar, p.dist_box, rb, ts, p.regroup, p.leaf_bias, tc, p.stack_cap, overflow);
            if (r == 0 && second) {                      // nothing in the cached subtree: full traversal
                second = false; hz_trav_reset(ts);
            } else if (COUNT && want_v && r != 2) {
                // the shortened ray is done: trace it again over its full length and compare the decisions
                want_v = false; verifying = true; first_result = (r == 1); start_v = true;
                HZ_OC(ocx, ocy, ocz)
                rb = hz_raybox(ocx + p.neg_tau * dx, ocy + p.neg_tau * dy, ocz + p.neg_tau * dz, dx, dy, dz);
                hz_trav_reset(ts);
            } else if (r != 2) {
                if (COUNT && verifying) { viol = ((r == 1) != first_result); verifying = false; }
                ray_active = false; last_hit = (r == 1);
                if (r == 1) cache = p.sv.anc[HZ_LEAF_ID(ts.lq0)];   // ts.lq0 is the leaf that blocked the ray
            }
        }
        if (COUNT) {
            w_verified += (unsigned)__popcll(__ballot(start_v));
            w_violations += (unsigned)__popcll(__ballot(viol));
        }
    }
    // one atomic per wave and counter
    unsigned long long r = rays, g = guards, nc = tc.nodes, tcn = tc.tris, cc = cells_cnt;
    unsigned long long wn = tc.w_nodes, wl = tc.w_leaves, wa = w_adv;
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_xor(r, off); g += __shfl_xor(g, off); cc += __shfl_xor(cc, off);
        if (COUNT) {
            nc += __shfl_xor(nc, off); tcn += __shfl_xor(tcn, off);
            wn += __shfl_xor(wn, off); wl += __shfl_xor(wl, off); wa += __shfl_xor(wa, off);
        }
    }
    // !LEVELSTACK: a wave in which a ray ran out of stack entries does not count; its block is computed again by the
    // one-entry-per-level kernel (horizon_run), which overwrites everything this wave wrote
    const bool wave_overflowed = !LEVELSTACK && __ballot(overflow) != 0ull;
    if (wave_overflowed) {
        if (lane == 0) {
            // (LEFT: the GROUP is computed again -- its records are only read here -- and has a count and a list of its own)
            const unsigned long long slot = atomicAdd(&p.counters[LEFT ? 30 : 8], 1ull);
            if (slot < (unsigned long long)HZ_REDO_CAP) p.redo_list[(LEFT ? HZ_REDO_CAP : 0) + (int)slot] = blk;
        }
    } else {
    const int guard_cells = __popcll(__ballot(guards != 0u && !(LEFT && had_guard)));   // cells where the reference's search would not terminate (a leftover cell: counted once)
    if (lane == 0) {
        if (r) atomicAdd(&p.counters[0], r);
        if (g) { atomicAdd(&p.counters[1], g); atomicAdd(&p.counters[11], (unsigned long long)guard_cells); }
        if (cc) atomicAdd(&p.counters[4], cc);
        if (COUNT) {
            atomicAdd(&p.counters[2], nc); atomicAdd(&p.counters[3], tcn);
            atomicAdd(&p.counters[5], wn); atomicAdd(&p.counters[6], wl); atomicAdd(&p.counters[7], wa);
        }
    }
    if (COUNT) {
        // when did the last wave of each XCD finish?  (counters[12 + xcc] = latest end, counters[20] = ~earliest start, on
        // the 100 MHz real-time counter; HZ_XCD_TRACE=1 prints the spans: the 8 XCDs own fixed regions of the tile)
        if (lane == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID, bits 3:0
            atomicMax(&p.counters[12 + xcc], (unsigned long long)wall_clock64());
            atomicMax(&p.counters[20], ~t_start);
        }
    }
    if (COUNT) {
        unsigned long long sh = shortened;
        for (int off = 32; off > 0; off >>= 1) sh += __shfl_xor(sh, off);
        if (lane == 0) {
            if (sh) atomicAdd(&p.counters[9], sh);
            if (w_violations) atomicAdd(&p.counters[10], (unsigned long long)w_violations);
            if (w_verified) atomicAdd(&p.counters[21], (unsigned long long)w_verified);
        }
    }
    }   // (!wave_overflowed)
    if (!p.persist) break;
  }     // next block of this wave
}

// Workgroups of `func` (HZ_TPB threads, `lds` bytes of dynamic LDS) the current device keeps resident at once; 0: unknown.  Asked once
// per (kernel, LDS size, device): the occupancy query and hipGetDeviceProperties are host-side work in front of every launch otherwise.
static long long resident_workgroups(const void *func, size_t lds) {
    struct Key { const void *f; size_t lds; int dev; long long n; };
    static std::mutex mu;
    static std::vector<Key> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lock(mu);
    for (const Key &k : cache) if (k.f == func && k.lds == lds && k.dev == dev) return k.n;
    int per_cu = 0;
    hipDeviceProp_t prop;
    long long n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, func, HZ_TPB, lds) == hipSuccess && per_cu > 0 &&
        hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        n = (long long)per_cu * prop.multiProcessorCount;
    else (void)hipGetLastError();
    cache.push_back(Key{func, lds, dev, n});
    return n;
}

template <int ALG, bool COUNT, bool STAGE, bool NODELET, bool LEVELSTACK>
static int launch_one(const HorizonParams &p_in, int grid, size_t lds, int persist_grid, hipStream_t st) {
    const void *func = reinterpret_cast<const void *>(k_horizon<ALG, COUNT, STAGE, NODELET, LEVELSTACK>);
    HZ_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HorizonParams p = p_in;
    if (p.persist) {
        // persistent waves: as many workgroups as this instantiation keeps resident (registers and this launch's LDS), each
        // wave pulls blocks until the queues are empty.  A launch with fewer tiles than that stays one tile per workgroup.
        // (persist_grid > 0: test hook -- that many workgroups, so that small grids run several blocks per wave too)
        const long long resident = persist_grid > 0 ? (long long)persist_grid : resident_workgroups(func, lds);
        if (resident <= 0 || (long long)grid <= resident) p.persist = 0;
        else {
            grid = (int)resident;
            HZ_HIP(hipMemsetAsync(p.queue, 0, 8 * sizeof(unsigned), st));
        }
    }
    hipLaunchKernelGGL((k_horizon<ALG, COUNT, STAGE, NODELET, LEVELSTACK>), dim3(grid), dim3(HZ_TPB), lds, st, p);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

// The LEFT instantiation: always persistent -- how many records there are is only known on the device (left_in_ctl), so the launch has
// the resident number of workgroups and every wave pulls groups of 64 sorted records until none is left.
// (with a list -- the groups to repeat after a stack overflow -- one group per wave of a plain launch)
template <int ALG, bool STAGE, bool LS>
static int launch_left(const HorizonParams &p, size_t lds, int persist_grid, hipStream_t st) {
    const void *func = reinterpret_cast<const void *>(k_horizon<ALG, false, STAGE, false, LS, true>);
    HZ_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long long grid = p.tile_list ? (long long)((p.n_list + HZ_WPB - 1) / HZ_WPB)
                                 : (persist_grid > 0 ? (long long)persist_grid : resident_workgroups(func, lds));
    if (grid <= 0) grid = 1024;
    hipLaunchKernelGGL((k_horizon<ALG, false, STAGE, false, LS, true>), dim3((unsigned)grid), dim3(HZ_TPB), lds, st, p);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

// Sort keys of the records of one region (24 bits): the azimuths a cell still has to do, most first -- a LEFT wave's 64 cells
// then finish together, and the longest groups start first --, then a Morton code of the cell's 16 x 16 tile, so that cells with
// the same count are neighbours: at the same azimuth their rays are nearly parallel (coalesced node fetches, one traversal
// order).  Slots that were never filled sort last; ctl[1] <- number of valid records.
__device__ __forceinline__ unsigned hz_spread16(unsigned v) {
    v &= 0xffffu; v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__global__ __launch_bounds__(256) void k_left_keys(const unsigned *__restrict__ rec, unsigned cap, unsigned *__restrict__ ctl, unsigned dim_in_1,
                                                   unsigned azim_num, int morton_shift, int class_shift, unsigned *__restrict__ keys,
                                                   unsigned *__restrict__ vals) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned cnt = min(ctl[0], cap);
    bool valid = false;
    unsigned key = 0x00ffffffu;
    if (idx < cnt) {
        const unsigned c = rec[(size_t)idx * HZ_LEFT_WORDS];
        if (c != 0xffffffffu) {
            valid = true;
            const unsigned k = rec[(size_t)idx * HZ_LEFT_WORDS + 1];
            const unsigned r = min(azim_num > k ? azim_num - k : 0u, 511u);
            const unsigned di = c / dim_in_1, j = c - di * dim_in_1;
            const unsigned m = ((hz_spread16(di >> 4) << 1) | hz_spread16(j >> 4)) >> morton_shift;
            key = (((511u - r) >> class_shift) << 15) | (m & 0x7fffu);
        }
    }
    if (idx < cap) { keys[idx] = key; vals[idx] = idx; }
    const unsigned long long vm = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && vm != 0ull) atomicAdd(&ctl[1], (unsigned)__popcll(vm));
}

// Keys + sort of region `r` of the leftover records (a.left_base[r], a.left_cap[r] slots; how many are filled is known on the device
// only, so all slots are sorted).  Leaves the sorted record numbers in a.left_sort + 3 * cap_max (3 passes: a -> b -> a -> b).
int left_sort(const HorizonArgs &a, int r, hipStream_t st) {
    const unsigned cap = a.left_cap[r];
    if (cap == 0 || a.left_rec == nullptr || a.left_sort == nullptr) return HZ_OK;
    unsigned *ctl = reinterpret_cast<unsigned *>(a.counters + HZ_CNT_LEFT) + 16 * r;
    const int rows = a.row_end - a.row_begin;
    int bits = 0;
    while ((1 << bits) < std::max((rows + 15) / 16, (a.dim_in_1 + 15) / 16)) bits++;
    const int shift = std::max(0, 2 * bits - 15);
    uint32_t *ka = a.left_sort, *va = ka + a.left_cap_max, *kb = va + a.left_cap_max, *vb = kb + a.left_cap_max, *tmp = vb + a.left_cap_max;
    hipLaunchKernelGGL(k_left_keys, dim3((cap + 255u) / 256u), dim3(256), 0, st, a.left_rec + (size_t)a.left_base[r] * HZ_LEFT_WORDS, cap, ctl,
                       (unsigned)a.dim_in_1, (unsigned)a.azim_num, shift, a.left_key_shift, ka, va);
    HZ_HIP(hipGetLastError());
    return radix_sort_pairs_u32(ka, va, kb, vb, cap, tmp, st, 3);
}

template <int ALG>
static int launch_alg(const HorizonParams &p, int grid, size_t lds, bool count, bool level_stack, int pg, hipStream_t st) {
    const bool stage = p.stage_bytes != 0;
    if (p.left_mode) {
        if (level_stack) return stage ? launch_left<ALG, true, true>(p, lds, pg, st) : launch_left<ALG, false, true>(p, lds, pg, st);
        return stage ? launch_left<ALG, true, false>(p, lds, pg, st) : launch_left<ALG, false, false>(p, lds, pg, st);
    }
    if (ALG == ALG_GUESS && !count && p.top_nodes > 0) {    // opt-in LDS nodelet variant (opts.top_nodes > 0)
        if (!level_stack) return stage ? launch_one<ALG_GUESS, false, true, true, false>(p, grid, lds, pg, st)
                                       : launch_one<ALG_GUESS, false, false, true, false>(p, grid, lds, pg, st);
        return stage ? launch_one<ALG_GUESS, false, true, true, true>(p, grid, lds, pg, st)
                     : launch_one<ALG_GUESS, false, false, true, true>(p, grid, lds, pg, st);
    }
    if (level_stack) {
        if (count) return stage ? launch_one<ALG, true, true, false, true>(p, grid, lds, pg, st) : launch_one<ALG, true, false, false, true>(p, grid, lds, pg, st);
        return stage ? launch_one<ALG, false, true, false, true>(p, grid, lds, pg, st) : launch_one<ALG, false, false, false, true>(p, grid, lds, pg, st);
    }
    if (count) return stage ? launch_one<ALG, true, true, false, false>(p, grid, lds, pg, st) : launch_one<ALG, true, false, false, false>(p, grid, lds, pg, st);
    return stage ? launch_one<ALG, false, true, false, false>(p, grid, lds, pg, st) : launch_one<ALG, false, false, false, false>(p, grid, lds, pg, st);
}

// 8 x 8 blocks (workgroup number * 4 + wave) a full launch over the rows of `a` numbers: the domain of HorizonArgs::tile_list
int horizon_num_blocks(const HorizonArgs &a) {
    const int rows = a.row_end - a.row_begin;
    if (rows <= 0 || a.dim_in_1 <= 0) return 0;
    const TileMap tm = make_tile_map((rows + 15) / 16, (a.dim_in_1 + 15) / 16);
    return tm.per_xcd * 8 * 4;
}

int horizon_launch(const Scene *sc, const HorizonArgs &a, hipStream_t st, int *used_level_stack) {
    HorizonParams p;
    p.sv = scene_view(sc);
    p.tb.azim_sin = a.azim_sin; p.tb.azim_cos = a.azim_cos;
    p.tb.elev_ang = a.elev_ang; p.tb.elev_sin = a.elev_sin; p.tb.elev_cos = a.elev_cos; p.tb.mid_idx = a.mid_idx;
    p.tb.azim_num = a.azim_num; p.tb.elev_num = a.elev_num;
    p.tb.hori_acc = a.hori_acc; p.tb.low = a.low; p.tb.up = a.up;
    p.tb.step = (double)a.hori_acc / 5.0;
    p.vec_norm = a.vec_norm; p.vec_north = a.vec_north;
    p.mask = a.mask; p.hori = a.hori;
    p.offset_0 = a.offset_0; p.offset_1 = a.offset_1; p.dim_in_1 = a.dim_in_1;
    p.row_begin = a.row_begin; p.row_end = a.row_end;
    const int rows = a.row_end - a.row_begin;
    if (rows <= 0 || a.dim_in_1 <= 0) return HZ_OK;
    const int tiles_i = (rows + 15) / 16;
    p.tm = make_tile_map(tiles_i, (a.dim_in_1 + 15) / 16);
    p.dist = a.dist; p.hori_fill = a.hori_fill; p.ray_org_elev = a.ray_org_elev;
    p.dist_box = a.dist + 2.0f * p.sv.tau; p.neg_tau = -p.sv.tau;
    // stack.  The fast discipline keeps every pending sibling as its own LDS entry (fewest VALU instructions in a
    // VALU-issue-bound kernel) and gets the entries that still allow 5 workgroups per CU next to the 4 KB output
    // staging: 27 (measured: <= 31 KiB of LDS per workgroup -> 5 resident, 32 KiB -> 4); rays of the 3601^2 tile use
    // <= 20.  Its worst case (3 per level) does not fit, so an overflowing wave raises counters[8] and horizon_run
    // repeats that launch with the one-entry-per-level discipline (`height` entries, cannot overflow, +10 % VALU) and
    // keeps it for the scene.  Shallow trees whose worst case fits never need the second kernel.
    const int stage = ((a.azim_num & 3) == 0 && (reinterpret_cast<size_t>(a.hori) & 15) == 0) ? 4 * HZ_TPB * 4 : 0;
    // (two rows: a lane that has popped its sentinel but still has queued leaves reads the two rows below the stack)
    const int pre = stage ? stage : 2 * HZ_TPB * 4;
    const int height = std::max(sc->hdr.height, 1);
    // (a.level_stack < 0: test hook, the fast discipline with that many entries)
    // (with the opt-in LDS nodelet the fast stack gives up the entries the nodelet's bytes need, so that 5 workgroups stay resident)
    const int want_top = (a.top_nodes > 0 && a.alg == ALG_GUESS && !a.count_work) ? std::min(a.top_nodes, sc->hdr.n_top) : 0;
    // (bytes of LDS per workgroup the fast stack may use together with staging and nodelet: 31 KiB keeps 5 workgroups per CU resident)
    const int lds_budget = 31 * 1024 / (4 / HZ_WPB);
    // (entry 0 of the fast stack is the sentinel, and a node step wants three free entries above the top: at least 4)
    const int fast_cap = a.level_stack < 0 ? std::max(-a.level_stack, 4)
                                           : std::max((lds_budget - pre - want_top * (int)sizeof(Node)) / (HZ_TPB * 4), 4);
    const bool level_stack = a.level_stack > 0;
    const int depth = level_stack ? height : std::min(fast_cap, 3 * height + 1);
    if (used_level_stack) *used_level_stack = (level_stack || depth >= 3 * height + 1) ? 1 : 0;   // 1: cannot overflow
    p.stack_cap = depth;
    p.stack_bytes = depth * HZ_TPB * 4;
    // output staging (4 azimuths per lane) when the 16 B stores are aligned: azim_num % 4 == 0
    p.stage_bytes = stage;
    p.pre_bytes = pre;
    // LDS nodelet (opt-in, guess_constant only): opts.top_nodes > 0 stages that many top-of-tree nodes;
    // the default reads every node through L1 (measured 2 % faster)
    int top = (a.top_nodes > 0 && a.alg == ALG_GUESS && !a.count_work) ? a.top_nodes : 0;
    top = std::min(top, std::max(0, (80 * 1024 - (p.stack_bytes + p.pre_bytes)) / (int)sizeof(Node)));
    top = std::min(top, sc->hdr.n_top);
    p.top_nodes = top;
    // defaults from the sweep on the 3601^2 tile: refill when fewer than 36 lanes
    // are traversing (40 until round 5: re-swept after the loop rewrite, profiles/r05/ab_tri_fma_and_regroup_sweep.log: 32 - 36
    // beat 40 by 1.2 %, 28 and 40 tie, 16 costs 7 %); leaf step when 24 n_leaf > 16 n_node.  opts.regroup = threshold | bias << 8.
    // (<= 0: the default -- a zeroed hz_opts must not switch the ray compaction off: 2.9 s instead of 2.15 s per tile)
    p.regroup = (a.regroup <= 0) ? 36 : std::min(a.regroup & 0xff, 64);
    if (a.left_mode && a.left_regroup > 0) p.regroup = std::min(a.left_regroup, 64);
    p.leaf_bias = (a.regroup >= 256) ? (a.regroup >> 8) : 24;      // (20 until round 4: re-swept after the node step lost 18 instructions, profiles/r04/regroup_sweep.log)
    p.hit_cache = (a.hit_cache != 0) ? 1 : 0;
    p.near_idx = a.near_idx; p.near_r = a.near_r;
    p.verify_near = (a.verify_near > 0 && a.near_idx != nullptr) ? 1 : 0;
    p.scratch_row = a.count_work ? a.scratch_row : nullptr;
    {   // one of every N shortened rays, N rounded up to a power of two (1: all)
        unsigned n = a.verify_near > 1 ? (unsigned)a.verify_near : 1u, m = 1u;
        while (m < n && m < (1u << 30)) m <<= 1;
        p.verify_mask = m - 1u;
    }
    p.counters = a.counters;
    p.tile_list = a.tile_list; p.n_list = a.n_list;
    p.redo_list = reinterpret_cast<int *>(a.counters + HZ_CNT_N);
    // leftover cells: level l's records (l = 1 ..) live in region l - 1 of a.left_rec, its control words (16 per level) behind the counters
    p.left_mode = std::max(a.left_mode, 0);
    p.left_rec = a.left_rec;
    unsigned *const left_ctl = reinterpret_cast<unsigned *>(a.counters + HZ_CNT_LEFT);
    const bool hands_over = a.left_rec != nullptr && !a.tile_list && !a.count_work && a.left_min > 0 && p.left_mode < HZ_LEFT_LEVELS &&
                            a.left_cap[p.left_mode] >= 64u;
    p.left_min = hands_over ? std::min(a.left_min, 56) : 0;
    p.left_in_base = p.left_mode ? a.left_base[p.left_mode - 1] : 0u;
    p.left_perm = (p.left_mode && a.left_sort) ? a.left_sort + 3 * a.left_cap_max : nullptr;      // (left_sort: sorted values end in vals_b)
    p.left_in_ctl = left_ctl + 16 * (p.left_mode ? p.left_mode - 1 : 0);
    p.left_out_base = hands_over ? a.left_base[p.left_mode] : 0u;
    p.left_out_cap = hands_over ? a.left_cap[p.left_mode] : 0u;
    p.left_out_ctl = left_ctl + 16 * std::min(p.left_mode, HZ_LEFT_LEVELS - 1);
    // persistent waves (k_horizon): the default for full launches; a.no_persist restores one tile per workgroup (same-box A/Bs)
    p.persist = (a.tile_list == nullptr && ((!a.no_persist && HZ_WPB == 4) || p.left_mode)) ? 1 : 0;
    p.queue = reinterpret_cast<unsigned *>(a.counters + 24);
    const size_t lds = (size_t)p.pre_bytes + (size_t)p.stack_bytes + (size_t)top * sizeof(Node);
    const int grid = a.tile_list ? (a.n_list + HZ_WPB - 1) / HZ_WPB : p.tm.per_xcd * 8 * (4 / HZ_WPB);
    if (grid <= 0) return HZ_OK;
    const bool count = a.count_work != 0;
    int rc_launch;
    switch (a.alg) {
        case ALG_DISCRETE: rc_launch = launch_alg<ALG_DISCRETE>(p, grid, lds, count, level_stack, a.persist_grid, st); break;
        case ALG_BINARY: rc_launch = launch_alg<ALG_BINARY>(p, grid, lds, count, level_stack, a.persist_grid, st); break;
        default: rc_launch = launch_alg<ALG_GUESS>(p, grid, lds, count, level_stack, a.persist_grid, st); break;
    }
    return rc_launch;
}

// ---------------------------------------------------------------------------------------
// reductions over the azimuth axis of a horizon array (topo_param.pyx):
//   KIND 0  sky view factor        _sky_view_factor_cy        :412-460
//   KIND 1  visible sky fraction   _visible_sky_fraction_cy   :499-543
//   KIND 2  topographic openness   _topographic_openness_cy   :577-603
//
// k_topo: one lane per cell for the arithmetic, one wave per 64 consecutive cells for the memory: the wave loads a
// [64 cells] x [32 azimuths] block of `hori` with 128 B contiguous per half wave into LDS (row stride 33 floats: the
// transposed reads are conflict free) and every lane then walks ITS cell's 32 values in azimuth order.  The float32
// accumulator with a float64 add per azimuth is the reference's (topo_param.pyx:446-458: the order of the additions is
// kept).  (Round 2 read `hori` with one lane per cell straight from HBM: 4 B loads at a stride of 4 A bytes, 1.7 - 3.4 x
// over-fetch and 31.5 ms per 3601^2 tile for an 18 GB read.)
//
// Trigonometry: the Cython code calls libm's float64 atan / cos / sin three times per (cell, azimuth) and adds the float64 term
// to a FLOAT32 accumulator (topo_param.pyx:446); all arguments lie in [-pi/2, pi/2].  Here ONE sine / cosine pair of the horizon
// angle gives cos^2, sin(2 h) / 2 = sin cos and -- through tan(h) >= x  <=>  sin >= x cos -- the comparison with the tilted plane's
// own horizon atan(x) without an arctangent; the arctangent is only evaluated where the plane limits.
// Round 6: the pair and the term are evaluated in FLOAT32 (fdlibm-style kernel polynomials on the halved argument + the
// double-angle formulas, |error| < 2e-7; every multiply and add rounds separately, -ffp-contract=off).  With the float64 pair of
// rounds 3 - 5 the kernel was bound by its float64 instructions (7.8e9 wave instructions at 4 cycles = 12.7 ms for 18.35 GB:
// 1.6 TB/s), not by memory.  What decides the result's accuracy is the float32 accumulator both sides share (half an ulp of a sum
// of ~100: 4e-6 per add before the division by azim_num), not the 1e-7 of a term: against the reference-made fixtures the float32
// terms differ by <= 2.4e-7 (tests/test_gpu_prep.py; the bar is 1e-5, the reference itself is built with -ffast-math), and
// k_topo_wide keeps libm's float64 routines per azimuth as the Cython code has them.  Each lane prefetches its share of the next
// block into registers before it reduces the current one.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void hz_sincos_halfpi_f(float x, float &s, float &c) {
    // sin / cos for |x| <= pi/2 (+ a little): polynomials in z = y^2 on y = x / 2 (|y| <= pi/4: truncation < 2e-9), then
    // sin x = 2 sin y cos y, cos x = 1 - 2 sin^2 y
    const float y = 0.5f * x, z = y * y;
    float ps = 2.7557314e-06f;                                  //  1/9!
    ps = ps * z + -1.9841270e-04f;                              // -1/7!
    ps = ps * z + 8.3333338e-03f;                               //  1/5!
    ps = ps * z + -1.6666667e-01f;                              // -1/3!
    float pc = -2.7557314e-07f;                                 // -1/10!
    pc = pc * z + 2.4801587e-05f;                               //  1/8!
    pc = pc * z + -1.3888889e-03f;                              // -1/6!
    pc = pc * z + 4.1666668e-02f;                               //  1/4!
    const float sy = (y * z) * ps + y;
    const float cy = (z * z) * pc + (1.0f - 0.5f * z);
    s = (2.0f * sy) * cy;
    c = 1.0f - (2.0f * sy) * sy;
}

#define HZ_TOPO_CH 32      // azimuths per LDS block
#ifndef HZ_TOPO_WG
#define HZ_TOPO_WG 3      // workgroups per CU the register allocation is held to (3: 155 VGPRs; 4: see DESIGN.md section 5)
#endif
template <int KIND>
__global__ __launch_bounds__(256, HZ_TOPO_WG) void k_topo(const float *__restrict__ azim, const float *__restrict__ hori,
                                             const float *__restrict__ vec_tilt, size_t ncell, int A,
                                             float *__restrict__ out) {
    extern __shared__ float topo_lds[];                 // [2 A] sin / cos of the azimuths, then 4 x [64][33] blocks
    float *az_tab = topo_lds;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *tile = topo_lds + 2 * A + wave * (64 * (HZ_TOPO_CH + 1));
    if (KIND != 2) {                                    // float32 of the float64 value, topo_param.pyx:425-426
        for (int k = threadIdx.x; k < A; k += blockDim.x) {
            az_tab[k] = (float)sin((double)azim[k]);
            az_tab[A + k] = (float)cos((double)azim[k]);
        }
    }
    const size_t cell0 = ((size_t)blockIdx.x * 4 + wave) * 64;
    const size_t c = cell0 + lane;
    const bool have = c < ncell;
    float tx = 0.0f, ty = 0.0f, tz = 1.0f;
    if (KIND != 2 && have) { tx = vec_tilt[3 * c]; ty = vec_tilt[3 * c + 1]; tz = vec_tilt[3 * c + 2]; }
    // tangent of the tilted plane's own horizon: x = -sin(az) tx / tz - cos(az) ty / tz (:441-443); the two quotients
    // are formed once per cell (the Cython expression divides per azimuth: same value to a float rounding)
    const float qx = -tx / tz, qy = -ty / tz;
    const double half_pi = 3.14159265358979323846 / 2.0;
    const float half_pi_f = 1.57079632679489661923f;
    float agg = 0.0f;
    const int half = lane >> 5, col = lane & 31;
    // this lane's share of a block: rows half, half + 2, ... of column `col` (128 B contiguous per half wave and row)
    const size_t n_rows = cell0 < ncell ? min((size_t)64, ncell - cell0) : 0;
    const float *src = hori + (cell0 + half) * (size_t)A + col;
    float pre[32];
    auto fetch = [&](int k0) {
        const int n = min(HZ_TOPO_CH, A - k0);
#pragma unroll
        for (int r = 0; r < 32; r++) {
            pre[r] = 0.0f;
            if (col < n && (size_t)(2 * r + half) < n_rows) pre[r] = src[(size_t)(2 * r) * A + k0];
        }
    };
    fetch(0);
    __syncthreads();                                    // az_tab is written
    // A wave's tile is private to it: LDS instructions of one wave execute in program order, so the transposed reads
    // below see the stores above without a workgroup barrier (the waves of a workgroup drift apart and overlap each
    // other's load and arithmetic phases); the fences only keep the compiler from reordering across them.
    for (int k0 = 0; k0 < A; k0 += HZ_TOPO_CH) {
        const int n = min(HZ_TOPO_CH, A - k0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 32; r++) tile[(2 * r + half) * (HZ_TOPO_CH + 1) + col] = pre[r];
        if (k0 + HZ_TOPO_CH < A) fetch(k0 + HZ_TOPO_CH);   // the next block's loads fly while this one is reduced
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!have) continue;
        const float *row = tile + lane * (HZ_TOPO_CH + 1);
#ifndef HZ_TOPO_UNROLL
#define HZ_TOPO_UNROLL 2      // two terms in flight per lane (the float32 accumulation stays sequential): 15.7 -> 14.8 ms; 4: 16.5 (VGPRs)
#endif
#pragma unroll HZ_TOPO_UNROLL
        for (int kk = 0; kk < n; kk++) {
            const float hv = row[kk];
            if (KIND == 2) {
                agg = (float)(((double)agg + half_pi) - (double)hv);                         // :599
                continue;
            }
            const float as = az_tab[k0 + kk], ac = az_tab[A + k0 + kk];
            const float xf = as * qx + ac * qy;
            float sn, cs;
            hz_sincos_halfpi_f(hv, sn, cs);
            float he = hv;
            // hv < atan(x)  <=>  sin < x cos: the tilted plane hides the horizon (:444 takes the larger angle)
            if (!(sn >= xf * cs)) {
                const float hp = atanf(xf);
                // sine and cosine of the plane's horizon come from its tangent: cos = 1 / sqrt(1 + x^2), sin = x cos
                if (!(hv >= hp)) {
                    he = hp;
                    const float rc = __builtin_amdgcn_rsqf(1.0f + xf * xf);
                    cs = rc; sn = xf * rc;
                }
            }
            if (KIND == 0) {
                agg = agg + ((tx * as + ty * ac) * ((half_pi_f - he) - sn * cs) + tz * (cs * cs));      // :446-452
            } else {
                agg = agg + (1.0f - sn);                           // 1 - cos(pi/2 - he), :540
            }
        }
    }
    if (!have) return;
    if (KIND == 2) { out[c] = agg / (float)A; return; }                                     // :601
    const float azim_spac = azim[1] - azim[0];
    out[c] = (float)(((double)azim_spac / (2.0 * 3.14159265358979323846)) * (double)agg);
}

// fallback for azimuth counts whose sine / cosine table does not fit in LDS: one lane per cell, libm calls as the
// Cython code has them (the round-2 kernel)
template <int KIND>
__global__ __launch_bounds__(256) void k_topo_wide(const float *__restrict__ azim, const float *__restrict__ hori,
                                                  const float *__restrict__ vec_tilt, size_t ncell, int A,
                                                  float *__restrict__ out) {
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    float tx = 0.0f, ty = 0.0f, tz = 1.0f;
    if (KIND != 2) { tx = vec_tilt[3 * c]; ty = vec_tilt[3 * c + 1]; tz = vec_tilt[3 * c + 2]; }
    const float *h = hori + c * (size_t)A;
    float agg = 0.0f;
    for (int k = 0; k < A; k++) {
        const float hv = h[k];
        if (KIND == 2) {
            agg = (float)(((double)agg + (3.14159265358979323846 / 2.0)) - (double)hv);
            continue;
        }
        const float as = (float)sin((double)azim[k]);
        const float ac = (float)cos((double)azim[k]);
        const float hori_plane = (float)atan((double)(-as * tx / tz - ac * ty / tz));
        const float he = (hv >= hori_plane) ? hv : hori_plane;
        if (KIND == 0) {
            const double ce = cos((double)he);
            agg = (float)((double)agg + ((double)(tx * as + ty * ac)
                          * ((3.14159265358979323846 / 2.0) - (double)he - (sin(2.0 * (double)he) / 2.0))
                          + (double)tz * (ce * ce)));
        } else {
            agg = (float)((double)agg + (1.0 - cos((3.14159265358979323846 / 2.0) - (double)he)));
        }
    }
    if (KIND == 2) { out[c] = agg / (float)A; return; }
    const float azim_spac = azim[1] - azim[0];
    out[c] = (float)(((double)azim_spac / (2.0 * 3.14159265358979323846)) * (double)agg);
}

int topo_launch(int kind, const float *azim, const float *hori, const float *vec_tilt, int len_0, int len_1,
                int len_2, float *out, hipStream_t st) {
    const size_t ncell = (size_t)len_0 * len_1;
    if (ncell == 0) return HZ_OK;
    const size_t lds = (2 * (size_t)len_2 + 4 * 64 * (HZ_TOPO_CH + 1)) * sizeof(float);
    // (hz_debug_set("topo_wide", 1) forces the fallback kernel -- tests compare the two paths on the same input, so that they
    //  cannot drift apart: k_topo takes the float32 arctangent only where the tilted plane limits, k_topo_wide calls libm's
    //  float64 routines per azimuth as the Cython code does; both are held to 1e-5 against the reference-made fixtures)
    if (lds <= 60 * 1024 && g_topo_wide.load(std::memory_order_relaxed) == 0) {
        const dim3 grid((unsigned)((ncell + 255) / 256)), block(256);
        if (kind == 0) hipLaunchKernelGGL(k_topo<0>, grid, block, lds, st, azim, hori, vec_tilt, ncell, len_2, out);
        else if (kind == 1) hipLaunchKernelGGL(k_topo<1>, grid, block, lds, st, azim, hori, vec_tilt, ncell, len_2, out);
        else hipLaunchKernelGGL(k_topo<2>, grid, block, lds, st, azim, hori, vec_tilt, ncell, len_2, out);
    } else {
        const dim3 grid((unsigned)((ncell + 255) / 256)), block(256);
        if (kind == 0) hipLaunchKernelGGL(k_topo_wide<0>, grid, block, 0, st, azim, hori, vec_tilt, ncell, len_2, out);
        else if (kind == 1) hipLaunchKernelGGL(k_topo_wide<1>, grid, block, 0, st, azim, hori, vec_tilt, ncell, len_2, out);
        else hipLaunchKernelGGL(k_topo_wide<2>, grid, block, 0, st, azim, hori, vec_tilt, ncell, len_2, out);
    }
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

int svf_launch(const float *azim, const float *hori, const float *vec_tilt, int len_0, int len_1,
               int len_2, float *svf, hipStream_t st) {
    return topo_launch(0, azim, hori, vec_tilt, len_0, len_1, len_2, svf, st);
}

}  // namespace hz
