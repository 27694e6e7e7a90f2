"""Shared by the GPU and the oracle pin tests: replay the configurations of scripts/make_embree_fixtures.py with
a given implementation and compare with the reference outputs in tests/golden/embree_*.npz (when a maintainer has
produced them -- they need the installed reference with Embree, which the build environment lacks)."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HORIZON = os.path.join(HERE, "golden", "embree_horizon.npz")
SHADOW = os.path.join(HERE, "golden", "embree_shadow.npz")
MISSING = ("tests/golden/embree_*.npz not present: the ray-casting decisions stay UNPINNED against the Embree reference -- "
           "run scripts/make_embree_fixtures.py in an environment with the reference installed (DESIGN.md section 3)")


def _harness():
    spec = importlib.util.spec_from_file_location("make_embree_fixtures", os.path.join(os.path.dirname(HERE), "scripts",
                                                                                    "make_embree_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def compare_horizon(horizon_gridded, bar=1.0e-4):
    """Returns a report dict; raises AssertionError when more than 1e-3 of the values miss the bar or any value is off
    by more than two search brackets."""
    ref = np.load(HORIZON)
    rep = {}
    for name, kw, par in _harness().pin_cases():
        h = horizon_gridded(**kw, **par)[0]
        r = ref["hori__" + name]
        d = np.abs(h.astype(np.float64) - r.astype(np.float64))
        frac = float((d > bar).mean())
        rep[name] = dict(values=int(h.size), mismatch_fraction=frac, max_abs=float(d.max()))
        acc = np.deg2rad(par.get("hori_acc", 0.25))
        assert frac <= 1.0e-3 and d.max() <= 2.5 * acc, (name, rep[name])
    return rep


def compare_shadow(make_terrain, tol_cells=1.0e-4):
    ref = np.load(SHADOW)
    g, (vec_tilt, vec_norm, enl, elev, mask), suns = _harness().shadow_case()
    rep = {}
    for refrac in (False, True):
        t = make_terrain()
        t.initialise(g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask, refrac_cor=refrac,
                     sw_dir_cor_fill=-9.0)
        for geom in ("triangle", "grid"):
            key = "%s_refrac%d" % (geom, int(refrac))
            bad = tot = 0
            worst = 0.0
            for s in range(suns.shape[0]):
                a = np.empty(mask.shape, np.uint8); f = np.empty(mask.shape, np.float32)
                t.shadow(suns[s], a); t.sw_dir_cor(suns[s], f)
                bad += int((a != ref["shadow__" + key][s]).sum()); tot += a.size
                both = (f != 0) & (ref["sw_dir_cor__" + key][s] != 0) & (mask == 1)
                if both.any():
                    worst = max(worst, float(np.abs(f[both] / ref["sw_dir_cor__" + key][s][both] - 1.0).max()))
            rep[key] = dict(cells=tot, differing_codes=bad, sw_dir_cor_max_rel=worst)
            assert bad / tot <= tol_cells and worst <= 1.0e-4, (key, rep[key])
    return rep
