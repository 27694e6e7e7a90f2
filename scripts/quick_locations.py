"""Quick probe of horizon_locations on the C3 scene: N random locations, 360 azimuths, binary_search (bench.py's extras line)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import horayzon_amd as hz
from horayzon_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=3601)
ap.add_argument("--locations", type=int, default=1000000)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
n, m = args.n, args.locations
g = synth.fractal_tile(n=n, offset=16)
sc = hz.Scene.create(g["vert_grid"], n, n)
rng = np.random.default_rng(5)
ci = rng.integers(40, n - 40, m); cj = rng.integers(40, n - 40, m)
coords = np.stack([g["x"][cj] + rng.uniform(-8.0, 8.0, m), g["y"][ci] + rng.uniform(-8.0, 8.0, m),
                   g["z"][ci, cj] + rng.uniform(-30.0, 60.0, m)], axis=1).astype(np.float32)
vn = np.zeros((m, 3), np.float32); vn[:, 2] = 1.0
vo = np.zeros((m, 3), np.float32); vo[:, 1] = 1.0
for rep in range(args.reps):
    r = hz.horizon.horizon_locations(g["vert_grid"], n, n, coords, vn, vo, 50.0, azim_num=360, scene=sc)
    st = hz.horizon.last_stats
    print("rep %d kernel %.4fs locations/s %.0f Mray/s %.1f rays %d" % (rep, st["t_kernel_s"], m / st["t_kernel_s"], st["num_rays"] / st["t_kernel_s"] / 1e6, st["num_rays"]), flush=True)
