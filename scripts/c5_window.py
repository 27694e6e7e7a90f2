"""Config 5, second part: the horizon array itself for a 3601^2 inner window of the 14401^2 mosaic
(NumPy in, 18.7 GB NumPy out, one GPU)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import horayzon_amd as hz
from horayzon_amd import synth
n, w = 14401, 3601
g = synth.fractal_tile(n=n, offset=16)
off = (n - w) // 2
vec_norm, vec_north = synth.planar_frames(w, w)
t = time.time()
hori, azim = hz.horizon.horizon_gridded(g["vert_grid"], n, n, vec_norm, vec_north, off, off, 50.0, azim_num=360)
wall = time.time() - t
st = hz.horizon.last_stats
print(json.dumps({"tile": n, "window": w, "wall_s": wall, "t_bvh_s": st["t_bvh_s"], "t_h2d_s": st["t_h2d_s"],
                  "t_kernel_s": st["t_kernel_s"], "t_d2h_s": st["t_d2h_s"], "gb_out": hori.nbytes / 1e9,
                  "rays": st["num_rays"], "cells_per_s_kernel": w * w / st["t_kernel_s"],
                  "mray_per_s_kernel": st["num_rays"] / st["t_kernel_s"] / 1e6, "cells_per_s_wall": w * w / wall,
                  "nan": int(np.isnan(hori[::61, ::67]).sum())}))
