"""Replay configuration K of the gridded random sweep (tests/test_gpu_fuzz.py) for a seed and print what differs.
usage: python scripts/fuzz_replay.py <seed> <k>"""
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
    stack = {"_level_stack": -int(rng.integers(4, 15))} if it % 4 == 1 else {}
verify = (target % 3 == 0)
out = hip.horizon.horizon_gridded(**kw, **par, **extra, **stack, count_work=verify, _verify_near=verify)
st = hip.horizon.last_stats
ro = {"rows": extra["rows"]} if "rows" in extra else {}
h_cpu, a_cpu, so = orc.horizon_gridded(**kw, **par, **ro, return_stats=True)
print({k: v for k, v in par.items() if np.isscalar(v)}, {k: v for k, v in extra.items() if k != "svf_vec_tilt"}, stack)
print("hori equal", np.array_equal(out[0], h_cpu, equal_nan=True), "rays", st["num_rays"], so["rays"], "guards", st["guard_events"], so["guards"],
      "violations", st["near_violations"], "height_field", st["height_field"], "near_used", st["near_used"])
d = np.abs(out[0] - h_cpu); print("hori max diff", np.nanmax(d), "n diff", int(np.nansum(d > 0)))
print("stats", {k: st[k] for k in ("num_cells", "guard_events", "guard_cells", "stack_fallbacks", "stack_redo_blocks", "rays_shortened", "bvh_height")})
for name, kwx in (("no near", dict(_near_skip=False)), ("level stack", dict(_level_stack=True)), ("no hit cache", dict(_hit_cache=False)), ("count", dict(count_work=True))):
    o2 = hip.horizon.horizon_gridded(**kw, **par, **extra, **kwx)
    s2 = hip.horizon.last_stats
    print(name, "guards", s2["guard_events"], "rays", s2["num_rays"], "fallbacks", s2["stack_fallbacks"], s2["stack_redo_blocks"], "equal", np.array_equal(o2[0], h_cpu, equal_nan=True))
if tilt is not None:
    in0 = kw["vec_norm"].shape[0]
    r0, r1 = extra.get("rows", (0, in0))
    svf_cpu = orc.sky_view_factor(a_cpu, h_cpu[r0:r1], tilt[r0:r1])
    e = np.abs(out[2][r0:r1] - svf_cpu)
    i = np.unravel_index(np.nanargmax(e), e.shape)
    print("svf max err", e.max(), "at", i, "gpu", out[2][r0:r1][i], "cpu", svf_cpu[i], "tilt", tilt[r0:r1][i], "hori", h_cpu[r0:r1][i])
