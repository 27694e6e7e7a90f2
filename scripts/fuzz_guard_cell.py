import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import horayzon_amd as hip
from oracle import oracle as orc
from tests import cases
seed, target, ci, cj = [int(x) for x in sys.argv[1:5]]
rng = np.random.default_rng(seed)
for it in range(target + 1):
    kw, par, extra, tilt = cases.fuzz_case(rng)
    if it % 4 == 1: rng.integers(4, 15)
in0, in1 = kw["vec_norm"].shape[:2]
par = dict(par); par.pop("mask", None)
m = np.zeros((in0, in1), np.uint8); m[ci, cj] = 1
for name, kwx in (("default", {}), ("no near", dict(_near_skip=False)), ("level stack", dict(_level_stack=True)), ("no hit cache", dict(_hit_cache=False)),
                  ("no near, no cache", dict(_near_skip=False, _hit_cache=False))):
    hip.horizon.horizon_gridded(**kw, **par, mask=m, rows=(ci, ci + 1), **kwx)
    s = hip.horizon.last_stats
    print("gpu", name, "guards", s["guard_events"], "rays", s["num_rays"])
for mode in (0, 1, 2):
    _, _, so = orc.horizon_gridded(**kw, **par, mask=m, rows=(ci, ci + 1), return_stats=True, mode=mode)
    print("oracle mode", mode, "guards", so["guards"], "rays", so["rays"])
# the ray at the top table index, every azimuth: who says hit?
tb = hip.horizon.horizon_tables(par["azim_num"], par["hori_acc"], par["elev_ang_low_lim"])
o0, o1 = kw["offset_0"], kw["offset_1"]
V = kw["vert_grid"][:3 * kw["dem_dim_0"] * kw["dem_dim_1"]].reshape(kw["dem_dim_0"], kw["dem_dim_1"], 3)
n = kw["vec_norm"][ci, cj]; no = kw["vec_north"][ci, cj]
org = (V[ci + o0, cj + o1] + n * np.float32(par["ray_org_elev"])).astype(np.float32)
east = np.cross(no, n).astype(np.float32)
top = tb["elev_num"] - 1
dirs = []
for k in range(par["azim_num"]):
    r = np.array([tb["elev_cos"][top] * tb["azim_sin"][k], tb["elev_cos"][top] * tb["azim_cos"][k], tb["elev_sin"][top]], np.float32)
    dirs.append((east * r[0] + no * r[1]) + n * r[2])
dirs = np.ascontiguousarray(np.array(dirs, np.float32))
sc = orc.Scene(kw["vert_grid"], kw["dem_dim_0"], kw["dem_dim_1"])
tf = np.float32(par["dist_search"] * 1000.0)
for mode in (0, 1, 2):
    print("top-index rays, oracle mode", mode, sc.occluded(org, dirs, tf, mode=mode).astype(int).tolist())
print("origin", org, "norm", n, "elev_num", tb["elev_num"], "top angle deg", np.rad2deg(tb["elev_ang"][top]))
hip.horizon.horizon_gridded(**kw, **par, mask=m, rows=(ci, ci + 1), count_work=True, _verify_near=True)
s = hip.horizon.last_stats
print("verify: rays", s["num_rays"], "shortened", s["rays_shortened"], "violations", s["near_violations"], "guards", s["guard_events"])
print("window vertices (rel. to the cell's vertex):")
W = 2
for a in range(-W, W + 1):
    print("  ", [tuple(np.round(V[ci + o0 + a, cj + o1 + b] - V[ci + o0, cj + o1], 2)) for b in range(-W, W + 1)])
