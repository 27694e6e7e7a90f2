"""Config 5 scale check on ONE GPU: 4 x 4 mosaic = 14401 x 14401 synthetic tile (207 M vertices,
415 M triangles).  Builds the scene, computes a slab of rows with 360 azimuths (horizon + SVF,
outputs resident in HBM) and optionally compares one row with the CPU oracle."""
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
ap.add_argument("--n", type=int, default=14401)
ap.add_argument("--rows", type=int, default=32)
ap.add_argument("--oracle-rows", type=int, default=1)
args = ap.parse_args()
n, off, A = args.n, 16, 360
t = time.time()
g = synth.fractal_tile(n=n, offset=off)
res = {"n": n, "synth_s": time.time() - t}
in0 = in1 = n - 2 * off
t = time.time()
sc = hz.Scene.create(g["vert_grid"], n, n)
res["scene_create_wall_s"] = time.time() - t
res["scene"] = sc.stats
dev = "cuda:0"
d_norm = torch.zeros((in0, in1, 3), dtype=torch.float32, device=dev); d_norm[..., 2] = 1.0
d_north = torch.zeros((in0, in1, 3), dtype=torch.float32, device=dev); d_north[..., 1] = 1.0
d_mask = torch.ones((in0, in1), dtype=torch.uint8, device=dev)
rb = in0 // 2
rows = args.rows
d_hori = torch.empty((rows, in1, A), dtype=torch.float32, device=dev)
opts = _lib.hz_opts(); opts.device = 0; opts.top_nodes = -1; opts.regroup = -1
opts.row_begin, opts.row_end = rb, rb + rows
L = _lib.lib()
for rep in range(2):
    st = _lib.hz_stats()
    _lib.check(L.hz_horizon_gridded_scene(sc._h, d_norm.data_ptr(), d_north.data_ptr(), off, off,
                                          d_hori.data_ptr() - 4 * rb * in1 * A, in0, in1, A, 50.0, 0.25,
                                          b"guess_constant", -15.0, d_mask.data_ptr(), 0.0, 0.01, C.byref(opts),
                                          C.byref(st)))
res["slab"] = {"rows": rows, "cells": int(st.num_cells), "rays": int(st.num_rays), "kernel_s": st.t_kernel_s,
               "cells_per_s": st.num_cells / st.t_kernel_s, "mray_per_s": st.num_rays / st.t_kernel_s / 1e6,
               "guards": int(st.guard_events), "rays_per_cell_az": st.num_rays / (st.num_cells * A)}
res["hbm_allocated_gb"] = torch.cuda.memory_allocated() / 1e9
if args.oracle_rows > 0:
    from oracle import oracle as orc
    kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
    t = time.time()
    ref, _, so = orc.horizon_gridded(**kw, dist_search=50.0, azim_num=A, rows=(rb, rb + args.oracle_rows),
                                     slab_only=True, return_stats=True)
    got = d_hori[:args.oracle_rows].cpu().numpy()
    res["oracle"] = {"rows": args.oracle_rows, "wall_s": time.time() - t, "build_s": so["t_build_s"],
                     "rays_s": so["t_rays_s"], "bit_identical": bool(np.array_equal(got, ref)),
                     "max_abs_diff": float(np.abs(got - ref).max())}
print(json.dumps(res))
