#!/usr/bin/env python
"""Pin harness: run the INSTALLED reference (ChristianSteger/HORAYZON with Intel Embree 4 + oneTBB, e.g. from the
conda environment of its README) on the parity configurations of this repository and write its outputs to
tests/golden/embree_*.npz.  `pytest -m gpu tests/test_gpu_embree_pin.py` (HIP kernels) and
`pytest tests/test_oracle.py -k embree` (CPU oracle) then compare against those files and report the mismatch
fraction against the north-star bar (1e-4 rad horizon, 1e-5 SVF).  It also records the reference's own printed
"Ray tracing time" / ray count per case and for a band of the 3601^2 benchmark tile, the Embree / TBB libraries it ran
with and the core count, in tests/golden/embree_timing.json -- bench.py then reports that as `cpu_baseline.embree`.

The build environment of this repository has neither Embree nor a network, so the files cannot be produced here:
until a maintainer runs this script once, parity stays "unpinned" for the ray-casting decisions.

    conda activate horayzon            # environment with the reference installed
    python scripts/make_embree_fixtures.py [--out tests/golden]

The inputs are regenerated from seeds by tests/cases.py / horayzon_amd/synth.py (NumPy only; no GPU needed); only
parameters and the reference's OUTPUTS are stored.
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def import_reference():
    """The installed reference package -- not this repository's alias package of the same name."""
    saved = list(sys.path)
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    for name in [m for m in sys.modules if m == "horayzon" or m.startswith("horayzon.")]:
        del sys.modules[name]
    try:
        ref = importlib.import_module("horayzon")
        hz = importlib.import_module("horayzon.horizon")
        sh = importlib.import_module("horayzon.shadow")
        tp = importlib.import_module("horayzon.topo_param")
    finally:
        sys.path = saved
    f = getattr(hz, "__file__", "") or ""
    if os.path.abspath(f).startswith(ROOT) or not f.endswith((".so", ".pyd")):
        raise SystemExit("the imported 'horayzon' (%s) is not the compiled reference package: install "
                         "ChristianSteger/HORAYZON (Embree 4, TBB) and run this script from its environment" % f)
    return ref, hz, sh, tp


def import_stub():
    """--dry-run: stand-ins with the reference's call signatures and its stdout report (horizon_comp.cpp:225-227,
    802-818), backed by this repository's CPU oracle.  The files a dry run writes say so ("version": "DRY-RUN ...") and
    are only good for checking this script (case list, shapes, dtypes, file layout, timing schema) -- never as fixtures."""
    import types
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc

    def horizon_gridded(vert_grid, dem_dim_0, dem_dim_1, vec_norm, vec_north, offset_0, offset_1, dist_search=50.0, **kw):
        hori, azim, st = orc.horizon_gridded(vert_grid, dem_dim_0, dem_dim_1, vec_norm, vec_north, offset_0, offset_1,
                                             dist_search=dist_search, return_stats=True, **kw)
        mask = kw.get("mask")
        cells = int((np.asarray(mask) == 1).sum()) if mask is not None else int(vec_norm.shape[0] * vec_norm.shape[1])
        # the reference reports from C++ (printf on file descriptor 1): write there, not through sys.stdout
        os.write(1, ("BVH build time: %g s\nNumber of grid cells for which horizon is computed: %d \nRay tracing time: %g s\n"
                     "Number of rays shot: %d\nTotal run time: %g s\n"
                     % (st["t_build_s"], cells, st["t_rays_s"], st["rays"], st["t_build_s"] + st["t_rays_s"])).encode())
        return hori, azim

    ref = types.SimpleNamespace(__version__="DRY-RUN (CPU oracle of this repository, NOT the Embree reference)")
    hz = types.SimpleNamespace(horizon_gridded=horizon_gridded)
    sh = types.SimpleNamespace(Terrain=orc.Terrain)
    tp = types.SimpleNamespace(sky_view_factor=orc.sky_view_factor)
    return ref, hz, sh, tp


def validate(out_dir, n_bench_rows):
    """Shape / dtype / layout check of everything main() wrote, against the case list the pin tests replay
    (tests/embree_pin.py).  Raises AssertionError with the offending key."""
    import json
    hz_npz = np.load(os.path.join(out_dir, "embree_horizon.npz"))
    names = []
    for name, kw, par in pin_cases():
        in0, in1 = kw["vec_norm"].shape[:2]
        A = par.get("azim_num", 360)
        h, a = hz_npz["hori__" + name], hz_npz["azim__" + name]
        assert h.dtype == np.float32 and h.shape == (in0, in1, A), ("hori__" + name, h.dtype, h.shape)
        assert a.dtype == np.float32 and a.shape == (A,), ("azim__" + name, a.dtype, a.shape)
        assert np.isfinite(h).all(), "hori__" + name
        names.append(name)
    first = names[0]
    assert hz_npz["svf__" + first].shape == hz_npz["hori__" + first].shape[:2] and hz_npz["svf__" + first].dtype == np.float32
    assert "version" in hz_npz.files
    sh_npz = np.load(os.path.join(out_dir, "embree_shadow.npz"))
    g, (vec_tilt, vec_norm, enl, elev, mask), suns = shadow_case()
    assert sh_npz["suns"].shape == suns.shape and sh_npz["suns"].dtype == np.float32
    for refrac in (0, 1):
        for geom in ("triangle", "grid"):
            k = "%s_refrac%d" % (geom, refrac)
            assert sh_npz["shadow__" + k].shape == (suns.shape[0],) + mask.shape and sh_npz["shadow__" + k].dtype == np.uint8, k
            assert sh_npz["sw_dir_cor__" + k].shape == (suns.shape[0],) + mask.shape and sh_npz["sw_dir_cor__" + k].dtype == np.float32, k
            assert sh_npz["shadow__" + k].max() <= 3, k
    tj = json.load(open(os.path.join(out_dir, "embree_timing.json")))
    assert set(tj) >= {"reference_version", "cases", "environment"} and set(tj["cases"]) == set(names)
    for name, rep in tj["cases"].items():
        assert {"ray_tracing_s", "rays"} <= set(rep), (name, rep)       # what bench.py / DESIGN.md quote
    assert tj["environment"]["logical_cores"] >= 1
    if n_bench_rows > 0:
        c3 = tj["c3_tile"]
        assert {"cells_per_s", "mray_per_s", "rows", "ray_tracing_s", "rays", "tile"} <= set(c3), c3   # bench.py: cpu_baseline.embree
        assert c3["rows"] == n_bench_rows and c3["cells_per_s"] > 0
    return len(names)


def pin_cases():
    """(name, grid kwargs, parameters): the configurations tests/test_gpu_embree_pin.py replays.  Searches that the
    reference cannot finish (guard events) are avoided by the choice of elev_ang_low_lim."""
    sys.path.insert(0, ROOT)
    from tests import cases
    from horayzon_amd import synth
    out = []
    c2 = cases.grid_kwargs(cases.c2_hill())
    for alg in cases.ALGS:
        for geom in ("triangle", "quad", "grid"):
            out.append(("c2_%s_%s" % (alg, geom), c2, dict(dist_search=10.0, azim_num=36, ray_algorithm=alg, geom_type=geom)))
    g = cases.rough_terrain(96, 110, seed=5, offset=6, relief=1200.0)
    out.append(("rough_grid", cases.grid_kwargs(g), dict(dist_search=4.0, azim_num=72, elev_ang_low_lim=-60.0)))
    g = cases.rough_terrain(80, 70, seed=8, offset=5, relief=700.0, tilt_frames=True, origin=(2.6e6, 1.2e6))
    out.append(("tilted_large_coords", cases.grid_kwargs(g), dict(dist_search=3.0, azim_num=45, hori_acc=0.1,
                                                                elev_ang_low_lim=-89.98, ray_algorithm="binary_search")))

    def dem(z, dx=30.0, dy=30.0, offset=4):
        n0, n1 = z.shape
        x = (np.arange(n1) * dx).astype(np.float32)
        y = ((n0 - 1 - np.arange(n0)) * dy).astype(np.float32)
        xx, yy = np.meshgrid(x, y)
        vn, vo = synth.planar_frames(n0 - 2 * offset, n1 - 2 * offset)
        return dict(vert_grid=synth.pack_vertices(xx, yy, z.astype(np.float32)), dem_dim_0=n0, dem_dim_1=n1,
                    vec_norm=vn, vec_north=vo, offset_0=offset, offset_1=offset)
    # the degenerate shapes of tests/test_gpu_parity.py::test_degenerate_terrain_shapes
    out.append(("flat", dem(np.full((40, 44), 250.0)), dict(dist_search=2.0, azim_num=16, elev_ang_low_lim=-89.98)))
    yy, xx = np.mgrid[0:48, 0:52]
    out.append(("terraces", dem(100.0 * ((xx // 6) % 4) + 50.0 * ((yy // 5) % 3)), dict(dist_search=2.0, azim_num=24, elev_ang_low_lim=-89.98)))
    spike = np.zeros((33, 35)); spike[16, 17] = 500.0
    out.append(("spike", dem(spike), dict(dist_search=2.0, azim_num=32, elev_ang_low_lim=-89.98)))
    strip = 200.0 * np.random.default_rng(5).random((3, 400))
    out.append(("strip", dem(strip, offset=0), dict(dist_search=20.0, azim_num=12, elev_ang_low_lim=-89.98)))
    return out


def shadow_case():
    sys.path.insert(0, ROOT)
    from tests import cases
    from horayzon_amd import synth
    g = cases.c2_hill(height=1500.0)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    mask[5:9, 5:20] = 0
    suns, _, _ = synth.sun_positions(num=24)
    suns = suns + np.array([5000.0, 5000.0, 0.0], np.float32)
    return g, (vec_tilt, vec_norm, enl, elev, mask), suns


class CaptureStdout:
    """The reference reports from C++ (printf, horizon_comp.cpp:225-227, :802-810): capture file descriptor 1."""

    def __enter__(self):
        import tempfile
        sys.stdout.flush()
        self._saved = os.dup(1)
        self._tmp = tempfile.TemporaryFile(mode="w+b")
        os.dup2(self._tmp.fileno(), 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        self._tmp.seek(0)
        self.text = self._tmp.read().decode("utf-8", "replace")
        self._tmp.close()
        sys.stdout.write(self.text)
        return False


def parse_report(text):
    """"BVH build time", "Ray tracing time", "Number of rays shot", cell count from the reference's stdout."""
    import re
    out = {}
    for key, pat in (("bvh_build_s", r"BVH build time:\s*([0-9.eE+-]+)"), ("ray_tracing_s", r"Ray tracing time:\s*([0-9.eE+-]+)"),
                     ("rays", r"Number of rays shot:\s*([0-9]+)"), ("cells", r"horizon is computed:\s*([0-9]+)"),
                     ("total_run_s", r"Total run time:\s*([0-9.eE+-]+)")):
        m = re.search(pat, text)
        if m:
            out[key] = float(m.group(1)) if key not in ("rays", "cells") else int(m.group(1))
    return out


def environment():
    """What the timings were taken on: CPU model, logical cores, the Embree / TBB libraries mapped into this process."""
    import platform
    env = {"host": platform.node(), "logical_cores": os.cpu_count(), "python": platform.python_version()}
    try:
        with open("/proc/cpuinfo") as f:
            env["cpu_model"] = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), None)
    except OSError:
        pass
    try:
        with open("/proc/self/maps") as f:
            libs = sorted({l.split()[-1] for l in f if ("embree" in l or "libtbb" in l) and "/" in l})
        env["embree_tbb_libraries"] = [os.path.realpath(x) for x in libs]
    except OSError:
        pass
    return env


def bench_tile(hz, n=3601, rows=64):
    """The headline configuration (BASELINE.json config 3: 3601^2 synthetic tile, 360 azimuths, guess_constant,
    dist_search 50 km) with the reference itself, restricted by `mask` to `rows` rows in the middle of the tile (the
    reference has no row-slab argument; rim cells are avoided because its search does not terminate there,
    horizon_comp.cpp:474-488).  Returns the reference's own "Ray tracing time" and ray count for those cells."""
    sys.path.insert(0, ROOT)
    from horayzon_amd import synth
    off = 16
    g = synth.fractal_tile(n=n, offset=off)
    in0 = n - 2 * off
    mask = np.zeros((in0, in0), np.uint8)
    mask[in0 // 2:in0 // 2 + rows] = 1
    with CaptureStdout() as cap:
        hz.horizon_gridded(g["vert_grid"], n, n, g["vec_norm"], g["vec_north"], off, off, 50.0, azim_num=360, mask=mask)   # noqa
    rep = parse_report(cap.text)
    rep.update(tile=n, rows=rows, cells_expected=rows * in0, azim_num=360, dist_search_km=50.0, ray_algorithm="guess_constant")
    if rep.get("ray_tracing_s") and rep.get("rays"):
        rep["mray_per_s"] = rep["rays"] / rep["ray_tracing_s"] / 1e6
        rep["cells_per_s"] = rep.get("cells", rows * in0) / rep["ray_tracing_s"]
    return rep


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--bench-rows", type=int, default=64, help="rows of the 3601^2 tile timed with the reference (0: skip)")
    ap.add_argument("--bench-tile", type=int, default=3601, help="size of the timed tile (the dry run uses a small one)")
    ap.add_argument("--dry-run", action="store_true",
                    help="run the whole case list against a stand-in for the reference (this repository's CPU oracle) to check "
                         "the script itself: shapes, dtypes, file layout, timing schema.  Writes to --out, which must NOT be "
                         "tests/golden")
    args = ap.parse_args(argv)
    if args.dry_run and os.path.abspath(args.out) == os.path.join(ROOT, "tests", "golden"):
        raise SystemExit("--dry-run must not write into tests/golden (its outputs are not fixtures): pass --out <tmp dir>")
    ref, hz, sh, tp = import_stub() if args.dry_run else import_reference()
    os.makedirs(args.out, exist_ok=True)
    store = {}
    timing = {"reference_version": getattr(ref, "__version__", "unknown"), "cases": {}}
    for name, kw, par in pin_cases():
        print("reference horizon_gridded:", name, flush=True)
        with CaptureStdout() as cap:
            hori, azim = hz.horizon_gridded(kw["vert_grid"], kw["dem_dim_0"], kw["dem_dim_1"], kw["vec_norm"], kw["vec_north"],
                                            kw["offset_0"], kw["offset_1"], **par)
        timing["cases"][name] = parse_report(cap.text)       # the reference's printed "Ray tracing time" / ray count
        store["hori__" + name] = hori
        store["azim__" + name] = azim
    # SVF of the first case through the reference's own topo_param (pins the fused SVF too)
    name, kw, par = pin_cases()[0]
    tilt = np.zeros(kw["vec_norm"].shape, np.float32); tilt[..., 2] = 1.0
    store["svf__" + name] = tp.sky_view_factor(store["azim__" + name], store["hori__" + name], tilt)
    np.savez_compressed(os.path.join(args.out, "embree_horizon.npz"), version=getattr(ref, "__version__", "unknown"), **store)
    g, (vec_tilt, vec_norm, enl, elev, mask), suns = shadow_case()
    sstore = {"suns": suns}
    for refrac in (False, True):
        for geom in ("triangle", "grid"):
            t = sh.Terrain()
            t.initialise(g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask, geom_type=geom,
                         refrac_cor=refrac, sw_dir_cor_fill=-9.0)
            shm = np.empty((suns.shape[0],) + mask.shape, np.uint8)
            swc = np.empty((suns.shape[0],) + mask.shape, np.float32)
            for s in range(suns.shape[0]):
                t.shadow(suns[s], shm[s]); t.sw_dir_cor(suns[s], swc[s])
            sstore["shadow__%s_refrac%d" % (geom, int(refrac))] = shm
            sstore["sw_dir_cor__%s_refrac%d" % (geom, int(refrac))] = swc
    np.savez_compressed(os.path.join(args.out, "embree_shadow.npz"), **sstore)
    # the CPU baseline north_star asks for: the reference's own TBB / Embree path on this machine's cores.  bench.py
    # reports it as cpu_baseline.embree when tests/golden/embree_timing.json exists (labelled with the host it was
    # recorded on -- the GPU box has no Embree).
    timing["environment"] = environment()
    if args.bench_rows > 0:
        print("reference horizon_gridded: 3601^2 tile, %d rows (timing)" % args.bench_rows, flush=True)
        timing["c3_tile"] = bench_tile(hz, n=args.bench_tile, rows=args.bench_rows)
    import json
    with open(os.path.join(args.out, "embree_timing.json"), "w") as f:
        json.dump(timing, f, indent=1)
    n = validate(args.out, args.bench_rows)
    print("wrote", os.path.join(args.out, "embree_horizon.npz"), ", embree_shadow.npz and embree_timing.json (%d horizon cases, layout "
          "validated)%s" % (n, " -- DRY RUN, not fixtures" if args.dry_run else ""))


if __name__ == "__main__":
    main()
