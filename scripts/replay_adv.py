"""Replay ONE configuration of scripts/fuzz_near_adversarial.py (seed, index) and say what differs from the oracle.
usage: python scripts/replay_adv.py <seed> <index>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import horayzon_amd as hz          # noqa: E402
from tests import cases            # noqa: E402
from oracle import oracle as orc   # noqa: E402

seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(idx + 1):
    kw, par, desc = cases.adversarial_near_case(rng)
ho, ao, so = orc.horizon_gridded(**kw, **par, return_stats=True)
for label, extra in (("certificates + count", dict(count_work=True, _verify_near=1)), ("no certificates", dict(count_work=True, _near_skip=False)),
                     ("production", dict())):
    h, a = hz.horizon.horizon_gridded(**kw, **par, **extra)
    st = hz.horizon.last_stats
    d = ~((h == ho) | (np.isnan(h) & np.isnan(ho)))
    print(label, json.dumps(dict(lib=os.environ.get("HORAYZON_HIP_LIB", "product"), horizon_equal=bool(not d.any()), differing=int(d.sum()),
          rays=int(st["num_rays"]), rays_oracle=int(so["rays"]), guards=int(st["guard_events"]), guards_oracle=int(so["guards"]),
          redo=int(st["stack_redo_blocks"]), fallbacks=int(st["stack_fallbacks"]), shortened=int(st["rays_shortened"]))))
    if d.any():
        w = np.argwhere(d)[:5]
        for (i, j, k) in w:
            print("   cell", int(i), int(j), "azimuth", int(k), "gpu", float(h[i, j, k]), "oracle", float(ho[i, j, k]))
