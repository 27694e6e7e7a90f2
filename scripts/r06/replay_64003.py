"""Replay of the adversarial sweep seed 64003 + 7 (GPU call 8: failed under HZ_TEST_SCHEDULE="persist_grid=3,left_min=0x1020"):
find the first failing configuration and run it under several schedules against the oracle."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import horayzon_amd as hz
from oracle import oracle as orc
from tests import cases
rng = np.random.default_rng(64003 + 7)
sched = dict(persist_grid=3, left_min=0x1020)
n_max = int(sys.argv[1]) if len(sys.argv) > 1 else 2600
for it in range(n_max):
    kw, par, desc = cases.adversarial_near_case(rng)
    ho, ao, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    res = {}
    variants = (("count_verify", dict(count_work=True, _verify_near=1)),)
    if it % 4 == 0:
        variants += (
                        ("two_levels_monitor", dict(_verify_near=4, _persist_grid=3, _left_min=0x1020)),
                        ("two_levels", dict(_persist_grid=3, _left_min=0x1020)),
                        ("one_level", dict(_persist_grid=3, _left_min=0x20)),
                        ("no_hand_over", dict(_persist_grid=3, _left_min=-1)),
                        ("default", dict()))
    for name, extra in variants:
        h, a = hz.horizon.horizon_gridded(**kw, **par, **extra)
        st = hz.horizon.last_stats
        res[name] = dict(equal=bool(np.array_equal(h, ho, equal_nan=True)), ndiff=int((h != ho).sum() - (np.isnan(h) & np.isnan(ho)).sum()),
                         rays=int(st["num_rays"]) - int(so["rays"]), guards=int(st["guard_events"]) - int(so["guards"]),
                         viol=int(st["near_violations"]), left=int(st["left_cells"]), again=int(st["left_again"]))
    bad = {k: v for k, v in res.items() if not v["equal"] or v["rays"] or v["guards"] or v["viol"]}
    if bad:
        print("FAIL it", it, json.dumps(desc), json.dumps({k: (v if k in bad else "ok") for k, v in res.items()}), flush=True)
        if "--all" not in sys.argv:
            break
print("done", it)
