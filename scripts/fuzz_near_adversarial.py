"""Adversarial sweep of the near-field certificates: tests/cases.py: adversarial_near_case.
Every configuration runs the counting instantiation with opts.verify_near = 1 (every shortened ray is traced a second
time over its full length); every `--oracle-every`-th one is also compared with the CPU oracle (horizon, ray and guard
counts).  One JSON line per configuration, a summary line at the end.
usage: python scripts/fuzz_near_adversarial.py --n 5000 --seed 41001 --out gpurun_out/r04_fuzz_near_41001.jsonl"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import horayzon_amd as hz          # noqa: E402
from tests import cases            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200)
ap.add_argument("--seed", type=int, default=41001)
ap.add_argument("--oracle-every", type=int, default=4)
ap.add_argument("--out", default="")
args = ap.parse_args()
orc = None
if args.oracle_every > 0:
    from oracle import oracle as orc
rng = np.random.default_rng(args.seed)
out = open(args.out, "w") if args.out else sys.stdout
tot = dict(configs=0, rays=0, shortened=0, retraced=0, violations=0, with_certificates=0, oracle_compared=0, oracle_mismatch=0)
t0 = time.time()
for it in range(args.n):
    kw, par, desc = cases.adversarial_near_case(rng)
    h, a = hz.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=1)
    st = hz.horizon.last_stats
    rec = dict(i=it, **desc, cells=int(st["num_cells"]), rays=int(st["num_rays"]), shortened=int(st["rays_shortened"]),
               retraced=int(st["near_verified"]), violations=int(st["near_violations"]), near_used=int(st["near_used"]),
               guards=int(st["guard_events"]))
    if orc is not None and it % args.oracle_every == 0:
        ho, ao, so = orc.horizon_gridded(**kw, **par, return_stats=True)
        ok = bool(np.array_equal(h, ho, equal_nan=True) and st["num_rays"] == so["rays"] and st["guard_events"] == so["guards"])
        rec["oracle_equal"] = ok
        tot["oracle_compared"] += 1
        tot["oracle_mismatch"] += int(not ok)
    tot["configs"] += 1
    tot["rays"] += rec["rays"]; tot["shortened"] += rec["shortened"]; tot["retraced"] += rec["retraced"]
    tot["violations"] += rec["violations"]; tot["with_certificates"] += int(rec["shortened"] > 0)
    out.write(json.dumps(rec) + "\n")
    if rec["violations"] or rec.get("oracle_equal") is False:
        print("PROBLEM", json.dumps(rec), file=sys.stderr, flush=True)
tot["seconds"] = time.time() - t0
tot["seed"] = args.seed
out.write(json.dumps({"summary": tot}) + "\n")
out.flush()
print(json.dumps({"summary": tot}), file=sys.stderr)
sys.exit(1 if (tot["violations"] or tot["oracle_mismatch"]) else 0)
