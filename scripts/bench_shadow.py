"""Config 4 probe: shadow mask / sw_dir_cor for the 3601^2 synthetic tile over 144 diurnal sun
positions on one MI355X (outputs stay in HBM; kernel time from HIP events in hz_stats)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import horayzon_amd as hz
from horayzon_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=3601)
ap.add_argument("--suns", type=int, default=144)
args = ap.parse_args()

n, off = args.n, 16
g = synth.fractal_tile(n=n, offset=off)
in0 = in1 = n - 2 * off
vec_tilt, enl = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], off)
vec_norm, _ = synth.planar_frames(in0, in1)
elev = np.ascontiguousarray(g["z"][off:off + in0, off:off + in1])
mask = np.ones((in0, in1), np.uint8)
suns, alt, az = synth.sun_positions(num=args.suns)
L = _lib.lib()
sc = hz.Scene.create(g["vert_grid"], n, n)
res = {"tile": n, "suns": args.suns, "cells": in0 * in1, "bvh_build_s": sc.stats["t_bvh_s"],
       "sun_alt_deg_minmax": [float(np.rad2deg(alt.min())), float(np.rad2deg(alt.max()))]}
for refrac in (0, 1):
    th = C.c_void_p()
    _lib.check(L.hz_terrain_create(0, C.byref(th)))
    _lib.check(L.hz_terrain_initialise_scene(th, sc._h, off, off, _lib.ptr(vec_tilt), _lib.ptr(vec_norm), in0, in1,
                                             _lib.ptr(enl), _lib.ptr(elev), _lib.ptr(mask), float("nan"), 89.0, refrac))
    for which, name, dt in ((0, "shadow", torch.uint8), (1, "sw_dir_cor", torch.float32)):
        out = torch.empty((args.suns, in0, in1), dtype=dt, device="cuda:0")
        fn = L.hz_terrain_shadow_batch if which == 0 else L.hz_terrain_sw_dir_cor_batch
        for rep in range(2):
            st = _lib.hz_stats()
            t = time.time()
            _lib.check(fn(th, _lib.ptr(suns), args.suns, out.data_ptr(), C.byref(st)))
            wall = time.time() - t
        key = "%s_refrac%d" % (name, refrac)
        res[key] = {"kernel_s": st.t_kernel_s, "wall_s": wall, "rays": int(st.num_rays),
                    "cells_per_s": args.suns * in0 * in1 / st.t_kernel_s,
                    "mray_per_s": st.num_rays / st.t_kernel_s / 1e6,
                    "ms_per_sun": 1e3 * st.t_kernel_s / args.suns}
        if which == 0:
            o = out.cpu().numpy()
            res[key]["frac_codes_0123"] = [float((o == c).mean()) for c in range(4)]
        del out
    L.hz_terrain_destroy(th)
print(json.dumps(res))
