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
    ap.add_argument("--watch", choices=("pre", "post"), default=None,
                    help="hardware write watchpoint (scripts/stray/libhzwatch.so) on the stray element's address in the "
                         "last configuration: 'pre' arms it BEFORE the GPU call of loop k+1 on the address the oracle's "
                         "array had in loop k (needs --loops >= 2; whoever owns that heap address during the GPU call "
                         "shows up with a backtrace); 'post' arms it right after the GPU call on the address the next "
                         "912-byte NumPy block will get")
    ap.add_argument("--watch-offset", type=int, default=888)
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
    watch = None
    if args.watch:
        watch = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhzwatch.so"))
        watch.hzwatch_arm.argtypes = [C.c_void_p]
    learned = None
    bad = 0
    for loop in range(args.loops):
        rng = np.random.default_rng(args.seed)
        for it in range(args.last + 1):
            kw, par, extra, tilt = cases.fuzz_case(rng)
            if it < args.first:
                continue
            in0, in1 = kw["vec_norm"].shape[:2]
            ro = {"rows": extra["rows"]} if "rows" in extra else {}
            armed = False
            if watch is not None and it == args.last and args.watch == "pre" and learned is not None:
                print("arming (pre) on %#x = learned array address %#x + %d" % (learned + args.watch_offset, learned, args.watch_offset), flush=True)
                watch.hzwatch_arm(C.c_void_p(learned + args.watch_offset)); armed = True
            if hip is not None:
                out = hip.horizon.horizon_gridded(**kw, **par, **extra)
                h_gpu = out[0]
            else:
                h_gpu = None
            if watch is not None and it == args.last and args.watch == "post":
                probe = np.full((in0, in1, par["azim_num"]), np.nan, np.float32)
                addr = probe.ctypes.data
                del probe
                watch.hzwatch_arm(C.c_void_p(addr + args.watch_offset)); armed = True
            h_cpu, a_cpu, so = orc.horizon_gridded(**kw, **par, **ro, return_stats=True)
            if it == args.last:
                print("loop %d: oracle array at %#x, gpu array at %s" % (loop, h_cpu.ctypes.data, hex(h_gpu.ctypes.data) if h_gpu is not None else None), flush=True)
                learned = h_cpu.ctypes.data
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
            if armed:
                watch.hzwatch_disarm()
            if hzq is not None and hzq() > 0:
                print("hzq reported damage by loop %d config %d" % (loop, it), flush=True)
    print("replay done: %d problems" % bad, flush=True)
    sys.exit(3 if bad else 0)


if __name__ == "__main__":
    main()
