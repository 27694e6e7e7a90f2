"""End-to-end drop-in call on config 3: NumPy in, NumPy out (18.3 GB horizon array)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import horayzon_amd as hz
from horayzon_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3601
devices = [int(d) for d in sys.argv[2].split(",")] if len(sys.argv) > 2 else None   # e.g. 0,1,2,3
g = synth.fractal_tile(n=n, offset=16)
kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
for rep in range(2):
    t = time.time()
    hori, azim = hz.horizon.horizon_gridded(**kw, dist_search=50.0, azim_num=360, devices=devices)
    wall = time.time() - t
    st = hz.horizon.last_stats
    print(json.dumps({"wall_s": wall, "t_bvh_s": st["t_bvh_s"], "t_h2d_s": st["t_h2d_s"], "t_kernel_s": st["t_kernel_s"],
                      "t_d2h_s": st["t_d2h_s"], "t_total_s": st["t_total_s"], "gb": hori.nbytes / 1e9,
                      "cells_per_s_end_to_end": hori.shape[0] * hori.shape[1] / wall,
                      "nan": int(np.isnan(hori[::97, ::89]).sum())}), flush=True)
    del hori
