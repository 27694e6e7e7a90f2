"""Issue rate of the VALU instructions the traversal kernels are made of (cycles per wave64 instruction per SIMD)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from horayzon_amd import _lib
NAMES = ["v_fma_f32", "v_cvt_f32_ubyte0", "v_cvt_f32_u32_sdwa_word1", "v_perm_b32", "v_max3_f32", "v_cndmask_b32",
         "v_lshl_add_u64", "v_mad_u64_u32", "v_lshlrev_b64", "v_cmp_le_f32", "v_lshl_add_u32", "v_mul_f32", "v_rcp_f32",
         "v_fma_f64", "v_add_co_u32+v_addc_co_u32 (per instruction)", "v_min_f32", "v_fma_f32 (VOP3) / v_mul_f32 (VOP2) alternating",
         "v_fma_f32 three different VGPR sources", "v_fma_f32 v, v, 1.0, 0.5 (one VGPR source)",
         "v_add_f32", "v_sub_f32", "v_max_f32", "v_and_b32", "v_add_u32", "v_mov_b32", "v_cndmask_b32 (vcc never written)",
         "v_fmac_f32", "v_fma_mix_f32 (f16 hi-half source)", "v_min3_f32 three different sources", "v_pk_fma_f32 (2 FMAs)",
         "v_cvt_f32_f16", "v_cndmask + 3 fast-class instructions (per instruction)", "v_cndmask_b32_e64 (mask in s[10:11])",
         "v_fma_f32 all sources in one VGPR bank (3 waves/SIMD: 144 VGPRs)", "v_fma_f32 sources in three banks (3 waves/SIMD)",
         "v_fma_mix_f32 sources in one bank", "v_fma_mix_f32 sources in three banks", "v_perm_b32 sources in one bank",
         "v_perm_b32 sources in three banks", "v_max3_f32 sources in one bank", "v_max3_f32 sources in three banks",
         "v_max_f32 sources in one bank", "v_max_f32 sources in two banks"]
import sys as _s
REPS = 3
L = _lib.lib()
out = {}
for rep in range(REPS):          # the first pass also warms the clocks; keep the fastest
    for op, name in enumerate(NAMES):
        r = C.c_double(0)
        _lib.check(L.hz_debug_inst_rate(0, op, C.byref(r)))
        out[name] = min(out.get(name, 1e9), round(r.value, 3))
print(json.dumps(out, indent=1))
