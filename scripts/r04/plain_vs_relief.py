"""Why does the half-plain DEM defeat the cost probe?  Counters and times of the plain slab and of the relief slab."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import horayzon_amd as hz
from horayzon_amd import synth
n, off, A = 2049, 16, 72
g = synth.fractal_tile(n=n, offset=off, plain_fraction=0.5)
kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
sc = hz.Scene.create(g["vert_grid"], n, n)
for name, rows in (("plain", (100, 900)), ("relief", (1100, 1900)), ("boundary", (960, 1060))):
    for count in (False, True):
        for rep in range(2):
            hz.horizon.horizon_gridded(**kw, dist_search=50.0, azim_num=A, scene=sc, rows=rows, count_work=count)
        st = hz.horizon.last_stats
        cells = st["num_cells"]
        print(json.dumps(dict(slab=name, count=count, kernel_ms=1e3 * st["t_kernel_s"], us_per_row=1e6 * st["t_kernel_s"] / (rows[1] - rows[0]),
                              rays_per_cell_az=st["num_rays"] / cells / A, nodes_per_ray=st["nodes_visited"] / max(st["num_rays"], 1),
                              tris_per_ray=st["tris_tested"] / max(st["num_rays"], 1), wave_node_iters_per_cell=st["wave_node_iters"] / cells,
                              wave_leaf_iters_per_cell=st["wave_leaf_iters"] / cells, wave_refills_per_cell=st["wave_refills"] / cells,
                              guard=st["guard_events"], near_ms=1e3 * st["t_near_s"], shortened=st["rays_shortened"] / max(st["num_rays"], 1))), flush=True)
