// hz_bench.hip -- machine calibration kernels behind hz_debug_valu_peak / hz_debug_copy_peak.
//
// The traversal kernels are bound by VALU issue, not by HBM.  To price them
// against something measured on the box -- not against a spec sheet -- bench.py runs these two
// kernels in its untimed section:
//   k_valu_peak : every wave runs long chains of independent v_fma_f32 (or v_pk_fma_f32); reports
//                 wave-level VALU instructions per second and SIMD.  This is the denominator of
//                 roofline.bound = "valu_issue".
//   k_copy_peak : float4 grid-stride copy; reports read + write GB/s (what an HBM-bound kernel reaches).
#include "hz_internal.h"

namespace hz {

typedef float v2f __attribute__((ext_vector_type(2)));

// 16 independent chains per lane, 4 x unrolled: 64 VALU instructions per loop trip (the loop counter
// lives in SGPRs: s_add / s_cmp / s_cbranch issue on the scalar port)
template <bool PACKED>
__global__ __launch_bounds__(256) void k_valu_peak(float *__restrict__ out, int trips, float m, float c) {
    const float seed = (float)threadIdx.x * 1.0e-3f;
    if (PACKED) {
        v2f x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = (v2f){seed + (float)k, seed - (float)k};
        const v2f mm = {m, m}, cc = {c, c};
        for (int t = 0; t < trips; t++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = __builtin_elementwise_fma(x[k], mm, cc);
            }
        }
        v2f s = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 16; k++) s += x[k];
        if (s.x + s.y == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s.x;   // keeps the chains alive
    } else {
        float x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = seed + (float)k;
        for (int t = 0; t < trips; t++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = __builtin_fmaf(x[k], m, c);
            }
        }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; k++) s += x[k];
        if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void k_copy_peak(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int bench_valu_peak(int packed, int waves_per_simd, double *winst_per_s_per_simd, double *clock_ghz, int *simds) {
    hipDeviceProp_t prop;
    int dev = 0;
    HZ_HIP(hipGetDevice(&dev));
    HZ_HIP(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const int w = std::min(std::max(waves_per_simd, 1), 8);
    // one workgroup = 4 waves = one wave per SIMD of a CU.  cus * w * 8 workgroups of 22 VGPRs are all resident at
    // once for w = 1 (8 waves per SIMD) and run in w rounds otherwise, so `w` only lengthens the run: 3.5 ms per
    // round.  Keep the run short: a burst of a few ms is issued at the full engine clock (4.06 cycles per instruction
    // at the nominal 2.4 GHz), a 28 ms run of nothing but FMAs is power-limited (4.2 cycles) -- the traversal
    // kernels, with their mixed instructions, are not (they reach 4.07 for seconds).
    const int grid = cus * w * 8;
    const int trips = 4096;
    float *out = nullptr;
    HZ_HIP(hipMalloc((void **)&out, (size_t)grid * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    HZ_HIP(hipEventCreate(&e0)); HZ_HIP(hipEventCreate(&e1));
    float best = 1.0e30f;
    for (int rep = 0; rep < 4; rep++) {     // first repetition warms the clocks
        HZ_HIP(hipEventRecord(e0, nullptr));
        if (packed) hipLaunchKernelGGL(k_valu_peak<true>, dim3(grid), dim3(256), 0, nullptr, out, trips, 0.999f, 0.001f);
        else hipLaunchKernelGGL(k_valu_peak<false>, dim3(grid), dim3(256), 0, nullptr, out, trips, 0.999f, 0.001f);
        HZ_HIP(hipEventRecord(e1, nullptr));
        HZ_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        HZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(out);
    const double winst = (double)grid * 4.0 * (double)trips * 64.0;      // wave-level VALU instructions
    const int n_simd = cus * 4;
    if (winst_per_s_per_simd) *winst_per_s_per_simd = winst / ((double)best * 1.0e-3) / (double)n_simd;
    if (clock_ghz) *clock_ghz = (double)prop.clockRate * 1.0e-6;
    if (simds) *simds = n_simd;
    return HZ_OK;
}

int bench_copy_peak(size_t bytes, double *gbs) {
    const size_t n = std::max<size_t>(bytes / 16, 1);
    float4 *src = nullptr, *dst = nullptr;
    HZ_HIP(hipMalloc((void **)&src, n * 16));
    HZ_HIP(hipMalloc((void **)&dst, n * 16));
    HZ_HIP(hipMemset(src, 1, n * 16));
    hipEvent_t e0, e1;
    HZ_HIP(hipEventCreate(&e0)); HZ_HIP(hipEventCreate(&e1));
    float best = 1.0e30f;
    for (int rep = 0; rep < 6; rep++) {
        HZ_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(k_copy_peak, dim3(256 * 16), dim3(256), 0, nullptr, src, dst, n);
        HZ_HIP(hipEventRecord(e1, nullptr));
        HZ_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        HZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(dst);
    if (gbs) *gbs = 2.0 * (double)n * 16.0 / ((double)best * 1.0e-3) / 1.0e9;
    return HZ_OK;
}

}  // namespace hz
