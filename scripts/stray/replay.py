#!/usr/bin/env python
"""Replay of the random-sweep configurations that showed the "stray element" of round 1
(HZ_FUZZ_SEED=9002, configurations 915..924; the last one is a 4 x 19 DEM + outer TIN with
rows=(1, 3)): GPU call, CPU oracle call, comparison -- exactly the flow of
tests/test_gpu_fuzz.py::test_random_configurations.  Exit code 3 when an array is dirty.

    python scripts/stray/replay.py [--seed 9002 --first 915 --last 924 --loops 1]
    LD_PRELOAD=scripts/stray/libhzq.so python scripts/stray/replay.py     # heap tripwire
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=9002)
    ap.add_argument("--first", type=int, default=915)
    ap.add_argument("--last", type=int, default=924)
    ap.add_argument("--loops", type=int, default=1)
    ap.add_argument("--no-gpu", action="store_true", help="oracle only (checks the tooling on a CPU box)")
    args = ap.parse_args()
    from tests import cases
    from oracle import oracle as orc
    orc.build()
    hip = None
    if not args.no_gpu:
        import horayzon_amd as hip
    hzq = None
    try:
        hzq = C.CDLL(None).hzq_check
    except AttributeError:
        pass
    bad = 0
    for loop in range(args.loops):
        rng = np.random.default_rng(args.seed)
        for it in range(args.last + 1):
            kw, par, extra, tilt = cases.fuzz_case(rng)
            if it < args.first:
                continue
            in0, in1 = kw["vec_norm"].shape[:2]
            ro = {"rows": extra["rows"]} if "rows" in extra else {}
            if hip is not None:
                out = hip.horizon.horizon_gridded(**kw, **par, **extra)
                h_gpu = out[0]
            else:
                h_gpu = None
            h_cpu, a_cpu, so = orc.horizon_gridded(**kw, **par, **ro, return_stats=True)
            r0, r1 = extra.get("rows", (0, in0))
            for name, h in (("gpu", h_gpu), ("cpu", h_cpu)):
                if h is None:
                    continue
                outside = np.ones(h.shape, bool); outside[r0:r1] = False
                dirty = outside & ~np.isnan(h)
                if dirty.any():
                    bad += 1
                    idx = np.flatnonzero(dirty)
                    print("DIRTY loop %d config %d: %s array (data at %#x, %d bytes) holds non-NaN outside rows [%d, %d): "
                          "flat indices %s values %s" % (loop, it, name, h.ctypes.data, h.nbytes, r0, r1, idx[:8].tolist(),
                                                         h.ravel()[idx[:8]].tolist()), flush=True)
            if h_gpu is not None and not np.array_equal(h_gpu[r0:r1], h_cpu[r0:r1]):
                bad += 1
                print("MISMATCH inside the slab, loop %d config %d" % (loop, it), flush=True)
            if hzq is not None and hzq() > 0:
                print("hzq reported damage by loop %d config %d" % (loop, it), flush=True)
    print("replay done: %d problems" % bad, flush=True)
    sys.exit(3 if bad else 0)


if __name__ == "__main__":
    main()
