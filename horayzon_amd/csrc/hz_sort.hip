// hz_sort.hip -- hand-written device primitives for the LBVH build (gfx950, wave64):
//   * stable LSD radix sort of (uint32 key, uint32 value) pairs, 8 bits per pass
//   * exclusive prefix sum of uint32
// One pass of the sort = k_hist (per-tile digit histogram, digit-major) -> exclusive scan of the
// histogram -> k_scatter (stable ranks inside the tile from wave-level digit matching + ordered
// per-wave counters in LDS).  A tile is 4 rounds x 256 keys; all streams are HBM bound.
#include "hz_internal.h"

namespace hz {

#define SORT_TPB 256
#define SORT_ROUNDS 4
#define SORT_TILE (SORT_TPB * SORT_ROUNDS)
#define SCAN_TPB 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_TPB * SCAN_ITEMS)

// ---------------------------------------------------------------------------------------
// exclusive scan: tile sums -> (recursive) scan of the sums -> tile scan with carried offset
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread workgroup; returns the exclusive
// prefix and the workgroup total (wsum: LDS, 4 entries)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wsum, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_TPB) void k_scan_sums(const uint32_t *__restrict__ in, size_t n,
                                                       uint32_t *__restrict__ sums) {
    __shared__ uint32_t wsum[4];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += in[base + k];
    uint32_t total;
    (void)block_exclusive_scan(s, wsum, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// in == out is allowed (the scan of the tile sums runs in place)
__global__ __launch_bounds__(SCAN_TPB) void k_scan_tiles(const uint32_t *in, size_t n, const uint32_t *offsets,
                                                        uint32_t *out) {
    __shared__ uint32_t wsum[4];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (base + k < n) ? in[base + k] : 0u; s += v[k]; }
    uint32_t total;
    uint32_t run = block_exclusive_scan(s, wsum, &total) + (offsets ? offsets[blockIdx.x] : 0u);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// temp must hold scan_temp_elems(n) uint32
size_t scan_temp_elems(size_t n) {
    size_t total = 0;
    while (n > SCAN_TILE) { n = (n + SCAN_TILE - 1) / SCAN_TILE; total += n; }
    return total + 1;
}

int exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *temp, hipStream_t st) {
    if (n == 0) return HZ_OK;
    const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tiles == 1) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_TPB), 0, st, in, n, (const uint32_t *)nullptr, out);
        HZ_HIP(hipGetLastError());
        return HZ_OK;
    }
    uint32_t *sums = temp;                       // tile sums, scanned in place
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)tiles), dim3(SCAN_TPB), 0, st, in, n, sums);
    int rc = exclusive_scan_u32(sums, sums, tiles, temp + tiles, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(SCAN_TPB), 0, st, in, n, (const uint32_t *)sums, out);
    HZ_HIP(hipGetLastError());
    return HZ_OK;
}

// ---------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_TPB) void k_hist(const uint32_t *__restrict__ keys, size_t n, int shift,
                                                  uint32_t n_tiles, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const size_t i = base + (size_t)r * SORT_TPB + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];      // digit-major
}

__global__ __launch_bounds__(SORT_TPB) void k_scatter(const uint32_t *__restrict__ keys_in,
                                                     const uint32_t *__restrict__ vals_in, size_t n, int shift,
                                                     uint32_t n_tiles, const uint32_t *__restrict__ offs,
                                                     uint32_t *__restrict__ keys_out,
                                                     uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t running[256];          // keys of each digit already placed by earlier rounds
    __shared__ uint32_t wcnt[4][256];          // per wave: keys of each digit in this round
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    running[threadIdx.x] = offs[(size_t)threadIdx.x * n_tiles + blockIdx.x];
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_ROUNDS; r++) {
#pragma unroll
        for (int w = 0; w < 4; w++) wcnt[w][threadIdx.x] = 0;
        __syncthreads();
        const size_t i = base + (size_t)r * SORT_TPB + threadIdx.x;
        const bool valid = i < n;
        uint32_t key = 0, val = 0, d = 0;
        if (valid) { key = keys_in[i]; val = vals_in[i]; d = (key >> shift) & 255u; }
        // lanes of this wave holding the same digit (8 ballots), rank among them = lower lanes
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned long long lower = same & ((1ull << lane) - 1ull);
        const uint32_t rank_in_wave = (uint32_t)__popcll(lower);
        if (valid && lower == 0ull) wcnt[wave][d] = (uint32_t)__popcll(same);   // first lane of the digit
        __syncthreads();
        if (valid) {
            uint32_t pos = running[d] + rank_in_wave;
            for (int w = 0; w < wave; w++) pos += wcnt[w][d];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        running[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        __syncthreads();
    }
}

// temp (uint32 elements) needed by radix_sort_pairs_u32
size_t sort_temp_elems(size_t n) {
    const size_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    return 2 * 256 * tiles + scan_temp_elems(256 * tiles);
}

// Sorts n pairs by the low 8 * passes bits of the key (stable).  keys_a/vals_a hold the input and, after an even number of
// passes (4: a -> b -> a -> b -> a), the sorted output; after an odd number the output is in keys_b/vals_b.
int radix_sort_pairs_u32(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, size_t n,
                         uint32_t *temp, hipStream_t st, int passes) {
    if (n == 0) return HZ_OK;
    const size_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    if (tiles > 0x7fffffffull / 256) return set_error(HZ_ERR_ARG, "too many primitives for the radix sort");
    uint32_t *hist = temp, *offs = temp + 256 * tiles, *scan_tmp = temp + 2 * 256 * tiles;
    uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    for (int pass = 0; pass < passes; pass++) {
        const int shift = 8 * pass;
        hipLaunchKernelGGL(k_hist, dim3((unsigned)tiles), dim3(SORT_TPB), 0, st, ki, n, shift, (uint32_t)tiles, hist);
        int rc = exclusive_scan_u32(hist, offs, 256 * tiles, scan_tmp, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_scatter, dim3((unsigned)tiles), dim3(SORT_TPB), 0, st, ki, vi, n, shift, (uint32_t)tiles,
                           offs, ko, vo);
        HZ_HIP(hipGetLastError());
        uint32_t *t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    return HZ_OK;
}

}  // namespace hz
