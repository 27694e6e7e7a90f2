"""Config 5 on ONE GPU in SVF-fused mode: 14401^2 synthetic mosaic, 360 azimuths, 50 km; the horizon
(298 GB) is never materialised, only the sky view factor (4 B / cell) comes back."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import horayzon_amd as hz
from horayzon_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14401
off = 16
t = time.time(); g = synth.fractal_tile(n=n, offset=off); t_synth = time.time() - t
kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
vec_tilt, _ = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], off)
t = time.time()
_, azim, svf = hz.horizon.horizon_gridded(**kw, dist_search=50.0, azim_num=360, svf_vec_tilt=vec_tilt, svf_only=True)
wall = time.time() - t
st = hz.horizon.last_stats
cells = svf.size
print(json.dumps({"tile": n, "cells": cells, "synth_s": t_synth, "wall_s": wall, "t_bvh_s": st["t_bvh_s"],
                  "t_h2d_s": st["t_h2d_s"], "t_kernel_s": st["t_kernel_s"], "t_svf_s": st["t_svf_s"],
                  "t_d2h_s": st["t_d2h_s"], "scene_bytes": st["scene_bytes"], "bvh_height": st["bvh_height"],
                  "rays": st["num_rays"], "cells_per_s_kernel": cells / st["t_kernel_s"],
                  "mray_per_s_kernel": st["num_rays"] / st["t_kernel_s"] / 1e6, "cells_per_s_wall": cells / wall,
                  "svf_min_max_nan": [float(np.nanmin(svf)), float(np.nanmax(svf)), int(np.isnan(svf).sum())]}))
