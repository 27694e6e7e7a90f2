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

// Issue rate of single VALU instructions the traversal kernels use (cycles per wave64 instruction on one SIMD, with 8
// resident waves per SIMD so that latency is hidden): 16 independent registers, 64 instructions per loop trip.
//   0 v_fma_f32   1 v_cvt_f32_ubyte0   2 v_cvt_f32_u32 (SDWA WORD_1)   3 v_perm_b32   4 v_max3_f32   5 v_cndmask_b32
//   6 v_lshl_add_u64   7 v_mad_u64_u32   8 v_lshlrev_b64   9 v_cmp_le_f32   10 v_lshl_add_u32   11 v_mul_f32
//   12 v_rcp_f32   13 v_fma_f64   14 v_add_co_u32 + v_addc_co_u32 pair (counted as 2)   15 v_min_f32
//   19 v_add_f32  20 v_sub_f32  21 v_max_f32  22 v_and_b32  23 v_add_u32  24 v_mov_b32  25 v_cndmask_b32 (no hazard)
//   26 v_fmac_f32  27 v_fma_mix_f32 (f16 source)  28 v_min3_f32 (three sources)  29 v_pk_fma_f32  30 v_cvt_f32_f16
//   16 v_fma_f32 (VOP3) alternating with v_mul_f32 (VOP2)   17 v_fma_f32 with three different VGPR sources   18 v_fma_f32 v, v, 1.0, 0.5
#define HZ_RATE_16(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)
template <int OP>
__global__ __launch_bounds__(256) void k_inst_rate(unsigned *__restrict__ out, int trips, unsigned seed) {
    unsigned x[16];
    unsigned long long y[16];
    double d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { x[k] = seed + threadIdx.x * 17u + (unsigned)k; y[k] = x[k]; d[k] = 1.0 + 1e-3 * (double)k; }
    const unsigned sel = 0x02000103u, one = 0x3f800000u;
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (OP == 0) {
#define I0(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I0)
            } else if (OP == 1) {
#define I1(k) asm volatile("v_cvt_f32_ubyte0_e32 %0, %0" : "+v"(x[k]));
                HZ_RATE_16(I1)
            } else if (OP == 2) {
#define I2(k) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x[k]));
                HZ_RATE_16(I2)
            } else if (OP == 3) {
#define I3(k) asm volatile("v_perm_b32 %0, 0, %0, %1" : "+v"(x[k]) : "v"(sel));
                HZ_RATE_16(I3)
            } else if (OP == 4) {
#define I4(k) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I4)
            } else if (OP == 5) {
#define I5(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(one) : "vcc");
                HZ_RATE_16(I5)
            } else if (OP == 6) {
#define I6(k) asm volatile("v_lshl_add_u64 %0, %0, 6, %0" : "+v"(y[k]));
                HZ_RATE_16(I6)
            } else if (OP == 7) {
#define I7(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, 48, %0" : "+v"(y[k]) : "v"(x[k]) : "vcc");
                HZ_RATE_16(I7)
            } else if (OP == 8) {
#define I8(k) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(y[k]));
                HZ_RATE_16(I8)
            } else if (OP == 9) {
#define I9(k) asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" : : "v"(x[k]), "v"(one) : "vcc");
                HZ_RATE_16(I9)
            } else if (OP == 10) {
#define I10(k) asm volatile("v_lshl_add_u32 %0, %0, 10, %1" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I10)
            } else if (OP == 11) {
#define I11(k) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I11)
            } else if (OP == 12) {
#define I12(k) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(x[k]));
                HZ_RATE_16(I12)
            } else if (OP == 13) {
#define I13(k) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[k]));
                HZ_RATE_16(I13)
            } else if (OP == 14) {
#define I14(k) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(x[k]), "+v"(x[(k + 8) & 15]) : : "vcc");
                I14(0) I14(1) I14(2) I14(3) I14(4) I14(5) I14(6) I14(7)
            } else if (OP == 15) {
#define I15(k) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I15)
            } else if (OP == 16) {   // VOP3 and VOP2 alternating
#define I16(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_mul_f32_e32 %3, %1, %3" : "+v"(x[k]) : "v"(one), "v"(sel), "v"(x[(k + 8) & 15]));
                I16(0) I16(1) I16(2) I16(3) I16(4) I16(5) I16(6) I16(7)
            } else if (OP == 17) {   // VOP3 with three different source registers
#define I17(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(one), "v"(sel));
                HZ_RATE_16(I17)
            } else if (OP == 18) {   // VOP3 whose two constant operands are inline constants
#define I18(k) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(x[k]));
                HZ_RATE_16(I18)
            } else if (OP == 19) {
#define I19(k) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I19)
            } else if (OP == 20) {
#define I20(k) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I20)
            } else if (OP == 21) {
#define I21(k) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I21)
            } else if (OP == 22) {
#define I22(k) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(x[k]) : "v"(sel));
                HZ_RATE_16(I22)
            } else if (OP == 23) {
#define I23(k) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x[k]) : "v"(sel));
                HZ_RATE_16(I23)
            } else if (OP == 24) {
#define I24(k) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x[k]) : "v"(x[(k + 1) & 15]));
                HZ_RATE_16(I24)
            } else if (OP == 25) {   // vcc is never written in this kernel: no hazard wait states
#define I25(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I25)
            } else if (OP == 26) {
#define I26(k) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[k]) : "v"(one), "v"(sel));
                HZ_RATE_16(I26)
            } else if (OP == 27) {   // f16 operand read straight from the high half of a register
#define I27(k) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[k]) : "v"(one), "v"(sel));
                HZ_RATE_16(I27)
            } else if (OP == 28) {
#define I28(k) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(one), "v"(sel));
                HZ_RATE_16(I28)
            } else if (OP == 29) {
#define I29(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[k]) : "v"(y[(k + 1) & 15]));
                HZ_RATE_16(I29)
            } else if (OP == 30) {
#define I30(k) asm volatile("v_cvt_f32_f16_e32 %0, %0" : "+v"(x[k]));
                HZ_RATE_16(I30)
            } else if (OP == 31) {   // one v_cndmask followed by three fast-class instructions (counted as 4)
#define I31(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_fma_f32 %3, %3, %1, %2\n\tv_add_f32_e32 %4, %1, %4\n\tv_mul_f32_e32 %5, %2, %5" \
                            : "+v"(x[k]) : "v"(one), "v"(sel), "v"(x[k + 4]), "v"(x[k + 8]), "v"(x[k + 12]));
                I31(0) I31(1) I31(2) I31(3)
            } else if (OP == 32) {   // select with the mask in an SGPR pair other than vcc
#define I32(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x[k]) : "v"(one));
                HZ_RATE_16(I32)
            } else if (OP == 33) {   // v_fma_f32 whose three sources sit in the same VGPR bank (register number mod 4)
                asm volatile("v_fma_f32 v100, v100, v104, v108\n\tv_fma_f32 v112, v112, v104, v108\n\tv_fma_f32 v116, v116, v104, v108\n\tv_fma_f32 v120, v120, v104, v108\n\t"
                             "v_fma_f32 v124, v124, v104, v108\n\tv_fma_f32 v128, v128, v104, v108\n\tv_fma_f32 v132, v132, v104, v108\n\tv_fma_f32 v136, v136, v104, v108\n\t"
                             "v_fma_f32 v100, v100, v104, v108\n\tv_fma_f32 v112, v112, v104, v108\n\tv_fma_f32 v116, v116, v104, v108\n\tv_fma_f32 v120, v120, v104, v108\n\t"
                             "v_fma_f32 v124, v124, v104, v108\n\tv_fma_f32 v128, v128, v104, v108\n\tv_fma_f32 v132, v132, v104, v108\n\tv_fma_f32 v136, v136, v104, v108"
                             : : : "v100", "v104", "v108", "v112", "v116", "v120", "v124", "v128", "v132", "v136");
            } else if (OP == 34) {   // the same with the sources in three different banks
                asm volatile("v_fma_f32 v100, v100, v105, v110\n\tv_fma_f32 v112, v112, v105, v110\n\tv_fma_f32 v116, v116, v105, v110\n\tv_fma_f32 v120, v120, v105, v110\n\t"
                             "v_fma_f32 v124, v124, v105, v110\n\tv_fma_f32 v128, v128, v105, v110\n\tv_fma_f32 v132, v132, v105, v110\n\tv_fma_f32 v136, v136, v105, v110\n\t"
                             "v_fma_f32 v100, v100, v105, v110\n\tv_fma_f32 v112, v112, v105, v110\n\tv_fma_f32 v116, v116, v105, v110\n\tv_fma_f32 v120, v120, v105, v110\n\t"
                             "v_fma_f32 v124, v124, v105, v110\n\tv_fma_f32 v128, v128, v105, v110\n\tv_fma_f32 v132, v132, v105, v110\n\tv_fma_f32 v136, v136, v105, v110"
                             : : : "v100", "v105", "v110", "v112", "v116", "v120", "v124", "v128", "v132", "v136");
            } else {
                // 35 .. 42: do VGPR bank collisions (register number mod 4) also slow the slow-class instructions of the
                // node step down?  odd selector: every source in bank 0 (v100 / v104 / v108); even: three banks (v100 / v105 / v110)
#define HZ_B8(INS, A, B, TAIL) INS " v100, v100, " A ", " B TAIL "\n\t" INS " v112, v112, " A ", " B TAIL "\n\t" INS " v116, v116, " A ", " B TAIL "\n\t" \
                INS " v120, v120, " A ", " B TAIL "\n\t" INS " v124, v124, " A ", " B TAIL "\n\t" INS " v128, v128, " A ", " B TAIL "\n\t" \
                INS " v132, v132, " A ", " B TAIL "\n\t" INS " v136, v136, " A ", " B TAIL
#define HZ_B16(INS, A, B, TAIL) asm volatile(HZ_B8(INS, A, B, TAIL) "\n\t" HZ_B8(INS, A, B, TAIL) : : : "v100", "v104", "v105", "v108", "v110", \
                "v112", "v116", "v120", "v124", "v128", "v132", "v136")
                if (OP == 35) HZ_B16("v_fma_mix_f32", "v104", "v108", " op_sel:[1,0,0] op_sel_hi:[1,0,0]");
                else if (OP == 36) HZ_B16("v_fma_mix_f32", "v105", "v110", " op_sel:[1,0,0] op_sel_hi:[1,0,0]");
                else if (OP == 37) HZ_B16("v_perm_b32", "v104", "v108", "");
                else if (OP == 38) HZ_B16("v_perm_b32", "v105", "v110", "");
                else if (OP == 39) HZ_B16("v_max3_f32", "v104", "v108", "");
                else if (OP == 40) HZ_B16("v_max3_f32", "v105", "v110", "");
                else if (OP == 41) asm volatile("v_max_f32_e32 v100, v100, v104\n\tv_max_f32_e32 v112, v112, v104\n\tv_max_f32_e32 v116, v116, v104\n\tv_max_f32_e32 v120, v120, v104\n\t"
                                                "v_max_f32_e32 v124, v124, v104\n\tv_max_f32_e32 v128, v128, v104\n\tv_max_f32_e32 v132, v132, v104\n\tv_max_f32_e32 v136, v136, v104\n\t"
                                                "v_max_f32_e32 v100, v100, v104\n\tv_max_f32_e32 v112, v112, v104\n\tv_max_f32_e32 v116, v116, v104\n\tv_max_f32_e32 v120, v120, v104\n\t"
                                                "v_max_f32_e32 v124, v124, v104\n\tv_max_f32_e32 v128, v128, v104\n\tv_max_f32_e32 v132, v132, v104\n\tv_max_f32_e32 v136, v136, v104"
                                                : : : "v100", "v104", "v112", "v116", "v120", "v124", "v128", "v132", "v136");
                else asm volatile("v_max_f32_e32 v100, v100, v105\n\tv_max_f32_e32 v112, v112, v105\n\tv_max_f32_e32 v116, v116, v105\n\tv_max_f32_e32 v120, v120, v105\n\t"
                                  "v_max_f32_e32 v124, v124, v105\n\tv_max_f32_e32 v128, v128, v105\n\tv_max_f32_e32 v132, v132, v105\n\tv_max_f32_e32 v136, v136, v105\n\t"
                                  "v_max_f32_e32 v100, v100, v105\n\tv_max_f32_e32 v112, v112, v105\n\tv_max_f32_e32 v116, v116, v105\n\tv_max_f32_e32 v120, v120, v105\n\t"
                                  "v_max_f32_e32 v124, v124, v105\n\tv_max_f32_e32 v128, v128, v105\n\tv_max_f32_e32 v132, v132, v105\n\tv_max_f32_e32 v136, v136, v105"
                                  : : : "v100", "v105", "v112", "v116", "v120", "v124", "v128", "v132", "v136");
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) acc ^= x[k] ^ (unsigned)y[k] ^ (unsigned)(y[k] >> 32) ^ (unsigned)__double_as_longlong(d[k]);
    if (acc == 0x12345u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
static float inst_rate_ms(int grid, unsigned *out, int trips) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1.0e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(k_inst_rate<OP>, dim3(grid), dim3(256), 0, nullptr, out, trips, 12345u);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0) best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return best;
}

int bench_inst_rate(int op, double *cycles_per_inst) {
    hipDeviceProp_t prop;
    int dev = 0;
    HZ_HIP(hipGetDevice(&dev));
    HZ_HIP(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const int grid = cus * 8;            // 8 waves per SIMD, one round
    const int trips = 1024;
    unsigned *out = nullptr;
    HZ_HIP(hipMalloc((void **)&out, (size_t)grid * 256 * sizeof(unsigned)));
    float ms = 0.0f;
    switch (op) {
        case 0: ms = inst_rate_ms<0>(grid, out, trips); break;
        case 1: ms = inst_rate_ms<1>(grid, out, trips); break;
        case 2: ms = inst_rate_ms<2>(grid, out, trips); break;
        case 3: ms = inst_rate_ms<3>(grid, out, trips); break;
        case 4: ms = inst_rate_ms<4>(grid, out, trips); break;
        case 5: ms = inst_rate_ms<5>(grid, out, trips); break;
        case 6: ms = inst_rate_ms<6>(grid, out, trips); break;
        case 7: ms = inst_rate_ms<7>(grid, out, trips); break;
        case 8: ms = inst_rate_ms<8>(grid, out, trips); break;
        case 9: ms = inst_rate_ms<9>(grid, out, trips); break;
        case 10: ms = inst_rate_ms<10>(grid, out, trips); break;
        case 11: ms = inst_rate_ms<11>(grid, out, trips); break;
        case 12: ms = inst_rate_ms<12>(grid, out, trips); break;
        case 13: ms = inst_rate_ms<13>(grid, out, trips); break;
        case 14: ms = inst_rate_ms<14>(grid, out, trips); break;
        case 15: ms = inst_rate_ms<15>(grid, out, trips); break;
        case 16: ms = inst_rate_ms<16>(grid, out, trips); break;
        case 17: ms = inst_rate_ms<17>(grid, out, trips); break;
        case 18: ms = inst_rate_ms<18>(grid, out, trips); break;
        case 19: ms = inst_rate_ms<19>(grid, out, trips); break;
        case 20: ms = inst_rate_ms<20>(grid, out, trips); break;
        case 21: ms = inst_rate_ms<21>(grid, out, trips); break;
        case 22: ms = inst_rate_ms<22>(grid, out, trips); break;
        case 23: ms = inst_rate_ms<23>(grid, out, trips); break;
        case 24: ms = inst_rate_ms<24>(grid, out, trips); break;
        case 25: ms = inst_rate_ms<25>(grid, out, trips); break;
        case 26: ms = inst_rate_ms<26>(grid, out, trips); break;
        case 27: ms = inst_rate_ms<27>(grid, out, trips); break;
        case 28: ms = inst_rate_ms<28>(grid, out, trips); break;
        case 29: ms = inst_rate_ms<29>(grid, out, trips); break;
        case 30: ms = inst_rate_ms<30>(grid, out, trips); break;
        case 31: ms = inst_rate_ms<31>(grid, out, trips); break;
        case 32: ms = inst_rate_ms<32>(grid, out, trips); break;
        case 33: ms = inst_rate_ms<33>(grid, out, trips); break;
        case 34: ms = inst_rate_ms<34>(grid, out, trips); break;
        case 35: ms = inst_rate_ms<35>(grid, out, trips); break;
        case 36: ms = inst_rate_ms<36>(grid, out, trips); break;
        case 37: ms = inst_rate_ms<37>(grid, out, trips); break;
        case 38: ms = inst_rate_ms<38>(grid, out, trips); break;
        case 39: ms = inst_rate_ms<39>(grid, out, trips); break;
        case 40: ms = inst_rate_ms<40>(grid, out, trips); break;
        case 41: ms = inst_rate_ms<41>(grid, out, trips); break;
        case 42: ms = inst_rate_ms<42>(grid, out, trips); break;
        default: (void)hipFree(out); return set_error(HZ_ERR_ARG, "unknown instruction selector %d", op);
    }
    (void)hipFree(out);
    HZ_HIP(hipGetLastError());
    // 8 waves per SIMD, each trips x 64 instructions, at the nominal engine clock
    const double inst_per_simd = 8.0 * (double)trips * 64.0;
    if (cycles_per_inst) *cycles_per_inst = (double)ms * 1.0e-3 * (double)prop.clockRate * 1.0e3 / inst_per_simd;
    return HZ_OK;
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
