"""CPU sweep: does the oracle's TREE ever change a decision against BRUTE FORCE?  
Runs the random generator of tests/test_gpu_fuzz.py (cases.random_config) and the adversarial generator of the near-field
sweeps (cases.adversarial_near_case: cliffs, spikes, terraces, tilted frames, coordinates of 2.6e6 -- the one that found
the grazing-at-the-origin counter-example) and compares horizon, ray count and guard count of the two acceleration paths.
One JSON line per PROBLEM, a summary line at the end.
usage: python scripts/sweep_bvh_brute.py --n-random 3000 --n-adversarial 3000 --seed 51001 [--box-start 0] --out profiles/r05/x.jsonl"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cases            # noqa: E402
from oracle import oracle as orc   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-random", type=int, default=100)
ap.add_argument("--n-adversarial", type=int, default=100)
ap.add_argument("--seed", type=int, default=51001)
ap.add_argument("--box-start", type=float, default=None, help="pads (default: the contract, oracle.BOX_START_PADS)")
ap.add_argument("--out", default="")
args = ap.parse_args()
if args.box_start is not None:
    orc.set_box_start(args.box_start)
out = open(args.out, "w") if args.out else sys.stdout
tot = dict(seed=args.seed, box_start_pads=orc.BOX_START_PADS if args.box_start is None else args.box_start,
           random=0, adversarial=0, rays=0, guards=0, problems=0)
t0 = time.time()


def one(kind, it, kw, par, desc):
    a, _, sa = orc.horizon_gridded(**kw, **par, return_stats=True, mode=orc.MODE_BVH)
    b, _, sb = orc.horizon_gridded(**kw, **par, return_stats=True, mode=orc.MODE_BRUTE)
    ok = bool(np.array_equal(a, b, equal_nan=True) and sa["rays"] == sb["rays"] and sa["guards"] == sb["guards"])
    tot[kind] += 1; tot["rays"] += int(sb["rays"]); tot["guards"] += int(sb["guards"])
    if not ok:
        tot["problems"] += 1
        rec = dict(kind=kind, i=it, desc=desc, differing=int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum()),
                   rays=[int(sa["rays"]), int(sb["rays"])], guards=[int(sa["guards"]), int(sb["guards"])])
        out.write(json.dumps(rec) + "\n"); out.flush()
        print("PROBLEM", json.dumps(rec), file=sys.stderr, flush=True)


rng = np.random.default_rng(args.seed)
for it in range(args.n_random):
    kw, par = cases.random_config(rng, max_n=34)
    one("random", it, kw, par, {k: v for k, v in par.items() if np.isscalar(v)})
rng = np.random.default_rng(args.seed + 1)
for it in range(args.n_adversarial):
    kw, par, desc = cases.adversarial_near_case(rng)
    one("adversarial", it, kw, par, desc)
tot["seconds"] = round(time.time() - t0, 1)
out.write(json.dumps({"summary": tot}) + "\n")
out.flush()
print(json.dumps({"summary": tot}), file=sys.stderr)
sys.exit(1 if tot["problems"] else 0)
