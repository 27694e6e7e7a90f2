import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import horayzon_amd as hip
from oracle import oracle as orc
from tests import cases
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(target + 1):
    kw, par, extra, tilt = cases.fuzz_case(rng)
    if it % 4 == 1: rng.integers(4, 15)
in0, in1 = kw["vec_norm"].shape[:2]
r0, r1 = extra.get("rows", (0, in0))
par = dict(par); par.pop("mask", None)
bad = 0
for i in range(r0, r1):
    for j in range(in1):
        m = np.zeros((in0, in1), np.uint8); m[i, j] = 1
        hg = hip.horizon.horizon_gridded(**kw, **par, mask=m, rows=(i, i + 1))[0]
        sg = hip.horizon.last_stats
        hc, _, so = orc.horizon_gridded(**kw, **par, mask=m, rows=(i, i + 1), return_stats=True)
        if sg["guard_events"] != so["guards"] or sg["num_rays"] != so["rays"]:
            bad += 1
            if bad <= 4:
                print("cell", i, j, "gpu guards/rays", sg["guard_events"], sg["num_rays"], "cpu", so["guards"], so["rays"], "equal", np.array_equal(hg[i, j], hc[i, j]))
                print("  hori idx-ish", np.round(np.rad2deg(hc[i, j]), 2))
print("cells with different guard counts:", bad, "of", (r1 - r0) * in1, "elev table: hori_acc", par["hori_acc"], "low", par["elev_ang_low_lim"])
