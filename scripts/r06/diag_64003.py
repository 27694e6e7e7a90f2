"""Adversarial sweep seed 64003 + 7, configurations 734 and 1377: a shortened ray whose full-length re-trace disagrees (found by GPU
call 9).  Which cells / azimuths, and does the production path (which uses the shortened decision) differ from the oracle?"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import horayzon_amd as hz
from oracle import oracle as orc
from tests import cases
rng = np.random.default_rng(64003 + 7)
want = (734, 1377)
for it in range(max(want) + 1):
    kw, par, desc = cases.adversarial_near_case(rng)
    if it not in want:
        continue
    print("=== it", it, json.dumps(desc), {k: v for k, v in par.items() if np.isscalar(v) or isinstance(v, str)}, flush=True)
    ho, ao, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    for name, extra in (("count_verify", dict(count_work=True, _verify_near=1)), ("production", dict()), ("production_no_left", dict(_left_min=-1, _persist_grid=-1)),
                        ("no_certificates", dict(_near_skip=False)), ("count_no_verify", dict(count_work=True))):
        h, a = hz.horizon.horizon_gridded(**kw, **par, **extra)
        st = hz.horizon.last_stats
        d = np.argwhere(h != ho)
        print(name, "ndiff", len(d), "rays", int(st["num_rays"]) - int(so["rays"]), "guards", int(st["guard_events"]) - int(so["guards"]),
              "viol", int(st["near_violations"]), "shortened", int(st["rays_shortened"]), "first diffs", d[:6].tolist(),
              [(float(h[tuple(x)]), float(ho[tuple(x)])) for x in d[:3]], flush=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "r06_10", "adv_64003_%d.npz" % it), **{k: np.asarray(v) for k, v in kw.items()},
             par=json.dumps({k: (v if not isinstance(v, np.generic) else v.item()) for k, v in par.items() if np.isscalar(v) or isinstance(v, str)}), ho=ho)
