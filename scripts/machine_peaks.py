#!/usr/bin/env python
"""Measured machine ceilings of the box (MI355X): wave-level VALU issue rate per SIMD for v_fma_f32 and
v_pk_fma_f32 (8 waves per SIMD, runs of 1 ... 8 rounds of 3.5 ms: the longer a pure-FMA run, the more the power limit
shows), and the float4 copy bandwidth.  Prints one JSON object."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from horayzon_amd import _lib


def main():
    L = _lib.lib()
    out = {"device": _lib.device_info(0), "valu": [], "copy_gbs": {}}
    for packed in (0, 1):
        for w in (1, 2, 4, 5, 8):
            r, clk, simds = C.c_double(0), C.c_double(0), C.c_int(0)
            _lib.check(L.hz_debug_valu_peak(0, packed, w, C.byref(r), C.byref(clk), C.byref(simds)))
            out["valu"].append({"packed": packed, "rounds_of_3p5_ms": w, "winst_per_s_per_simd": r.value,
                                "clock_ghz": clk.value, "simds": simds.value,
                                "cycles_per_wave_inst": clk.value * 1e9 / r.value})
    for mb in (64, 1024, 4096):
        g = C.c_double(0)
        _lib.check(L.hz_debug_copy_peak(0, mb << 20, C.byref(g)))
        out["copy_gbs"]["%d MiB" % mb] = g.value
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
