// hz_search.h -- the reference's per-cell horizon search algorithms as a resumable per-lane
// state machine (ray_discrete_sampling / ray_binary_search / ray_guess_const,
// horizon_comp.cpp:302-498, and the *_hori_dist variants :519-612).  Shared by the gridded
// and the locations kernels.  The float/double promotion pattern of the reference's index
// arithmetic is reproduced exactly; tables are built on the host (hz_api.hip) and only read here.
#pragma once
#include "hz_internal.h"

namespace hz {

enum { ALG_DISCRETE = 0, ALG_BINARY = 1, ALG_GUESS = 2 };
enum { PH_NEWAZ = 0, PH_BIN = 1, PH_UP = 2, PH_DOWN = 3, PH_EMIT = 4 };

struct Tables {
    const float *azim_sin, *azim_cos, *elev_ang, *elev_sin, *elev_cos;
    const int *mid_idx;   // pairs: mid_idx[2 i] = ind_of(half_sum(elev_ang[i], elev_ang[i + 10])), [2 i + 1] = bits of elev_ang[that], built on the host
    int azim_num, elev_num;
    float hori_acc, low, up;
    double step;   // (double)hori_acc / 5.0
};

struct Search {
    int k, phase, ind, prev, pazim, count;
    float lim_up, lim_low, elev_samp;
    float ev;   // value to emit for azimuth k (phase PH_EMIT): one emit site keeps the code small
};

// (int)roundf((elev_samp - low) / (hori_acc / 5.0)), horizon_comp.cpp:351-352
__device__ __forceinline__ int ind_of(const Tables &t, float elev_samp) {
    return (int)__builtin_roundf((float)((double)(elev_samp - t.low) / t.step));
}
// (a + b) / 2.0 -> float, horizon_comp.cpp:350, :330, :462, :490
__device__ __forceinline__ float half_sum(float a, float b) {
    return (float)((double)(a + b) / 2.0);
}

// index of the table entry nearest to the midpoint of entries `a` and `b` (horizon_comp.cpp:462-464,
// :490-492).  The search steps by 10 entries, so the midpoint index comes from a host-built table
// (same float/double expressions, hz_api.hip); clamped steps at the table ends take the general path.
// *ev = elev_ang[that index], the value the search emits: the table holds (index, value bits) pairs -- one 8 B load.
__device__ __forceinline__ int mid_index(const Tables &t, int a, int b, float *ev) {
    const int lo = min(a, b);
    if (max(a, b) - lo == 10) {
        const int2 v = reinterpret_cast<const int2 *>(t.mid_idx)[lo];
        *ev = __int_as_float(v.y);
        return v.x;
    }
    const int ind = ind_of(t, half_sum(t.elev_ang[a], t.elev_ang[b]));
    *ev = t.elev_ang[ind];
    return ind;
}

// per-cell output sink
struct Sink {
    float *hori;         // &hori_buffer[cell * azim_num]
    float *dist;         // &hori_dist_buffer[cell * azim_num] or null
    float dist_hit;      // distance of the last hit ray; persists across azimuths (horizon_comp.cpp:526-527)
    float *stage;        // LDS staging of 4 consecutive azimuths of this lane (stage[j * stride]) or null
    int stride;
};

// A lane produces its azimuths one at a time; a 4 B store per lane at stride 4*A bytes costs a
// 32 B memory sector each (8x write amplification measured).  With STAGE, four consecutive
// values of a lane are collected in LDS and leave as one 16 B store (requires azim_num % 4 == 0
// and a 16 B aligned buffer; checked by the launcher).
template <bool STAGE>
__device__ __forceinline__ void emit(Sink &s, const Tables &, int k, float h) {
    if (STAGE) {
        if ((k & 3) == 3) {
            const float4 v = make_float4(s.stage[0], s.stage[s.stride], s.stage[2 * s.stride], h);
            *reinterpret_cast<float4 *>(s.hori + (k - 3)) = v;
        } else {
            s.stage[(k & 3) * s.stride] = h;
        }
    } else {
        s.hori[k] = h;
    }
    if (s.dist) s.dist[k] = s.dist_hit;   // :552, :608
}

// Consume the result of the previous ray (if any) and produce the next sample.
// Returns true with s.ind / s.k identifying the next ray, false when the cell is finished.
template <int ALG, bool STAGE>
__device__ __forceinline__ bool advance(Search &s, bool hit, const Tables &t, Sink &out,
                                        unsigned &guards) {
    const int top = t.elev_num - 1;
    for (;;) {
        if (s.phase == PH_EMIT) {                              // the only place that writes output
            emit<STAGE>(out, t, s.k, s.ev);
            s.k++; s.phase = PH_NEWAZ;
        }
        if (s.phase == PH_NEWAZ) {
            if (s.k >= t.azim_num) return false;
            const bool binary = (ALG == ALG_BINARY) || (ALG == ALG_GUESS && s.k == 0);
            if (binary) {                                   // horizon_comp.cpp:348-354 / :398-404
                s.lim_up = t.up; s.lim_low = t.low;
                s.elev_samp = half_sum(s.lim_up, s.lim_low);
                s.ind = ind_of(t, s.elev_samp);
                s.phase = PH_BIN;
                const float e = t.elev_ang[s.ind];
                if (__builtin_fmaxf(s.lim_up - e, e - s.lim_low) > t.hori_acc) return true;
                s.ev = s.elev_samp; s.pazim = s.ind; s.phase = PH_EMIT;
                continue;
            }
            // move upwards: horizon_comp.cpp:311-317 (discrete, from index 0) / :439-446
            s.ind = (ALG == ALG_DISCRETE) ? 0 : max(s.pazim - 5, 0);
            s.prev = s.ind;
            s.ind = min(s.ind + 10, top);
            s.count = 1;
            s.phase = PH_UP;
            return true;
        }
        if (s.phase == PH_BIN) {                             // :367-374 / :417-424
            const float e0 = t.elev_ang[s.ind];
            if (hit) s.lim_low = e0; else s.lim_up = e0;
            s.elev_samp = half_sum(s.lim_up, s.lim_low);
            s.ind = ind_of(t, s.elev_samp);
            const float e = t.elev_ang[s.ind];
            if (__builtin_fmaxf(s.lim_up - e, e - s.lim_low) > t.hori_acc) return true;
            s.ev = s.elev_samp; s.pazim = s.ind; s.phase = PH_EMIT;   // :376 / :428
            continue;
        }
        if (s.phase == PH_UP) {
            const bool guard = hit && (s.ind == top);        // the reference never leaves this loop
            if (hit && !guard) {
                s.prev = s.ind;
                s.ind = min(s.ind + 10, top);
                s.count++;
                return true;
            }
            if (guard) guards++;
            if (ALG == ALG_DISCRETE) {                       // :330
                s.ev = half_sum(t.elev_ang[s.prev], t.elev_ang[s.ind]); s.phase = PH_EMIT;
                s.pazim = s.prev;                            // last blocked sample (hit-cache hint only)
                continue;
            }
            if (s.count > 1) {                               // :460-467
                s.ind = mid_index(t, s.prev, s.ind, &s.ev);
                s.pazim = s.ind; s.phase = PH_EMIT;
                continue;
            }
            // move downwards: :472-477
            s.ind = min(s.pazim + 5, top);
            s.prev = s.ind;
            s.ind = max(s.ind - 10, 0);
            s.phase = PH_DOWN;
            return true;
        }
        // PH_DOWN: :474-488
        {
            const bool guard = (!hit) && (s.ind == 0);
            if (!hit && !guard) {
                s.prev = s.ind;
                s.ind = max(s.ind - 10, 0);
                return true;
            }
            if (guard) guards++;
            s.ind = mid_index(t, s.prev, s.ind, &s.ev);                       // :490-494
            s.pazim = s.ind; s.phase = PH_EMIT;
            continue;
        }
    }
}

}  // namespace hz
