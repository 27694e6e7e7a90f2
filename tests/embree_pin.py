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


def compare_horizon(horizon_gridded, bar=1.0e-4, path=None, check=True):
    """Returns a report per case: values, the FRACTION of values that miss the north-star bar (1e-4 rad), the largest
    difference in radians and in search brackets, and the five worst cells (row, column, azimuth index, got, reference).
    With `check` it then raises AssertionError when more than 1e-3 of a case's values miss the bar or any value is off by
    more than two search brackets -- after the whole report has been printed, so a failing pin still says where."""
    import json
    ref = np.load(path or HORIZON)
    rep, failed = {}, []
    for name, kw, par in _harness().pin_cases():
        h = horizon_gridded(**kw, **par)[0]
        r = ref["hori__" + name]
        d = np.abs(h.astype(np.float64) - r.astype(np.float64))
        frac = float((d > bar).mean())
        acc = np.deg2rad(par.get("hori_acc", 0.25))
        worst = []
        for flat in np.argsort(d, axis=None)[::-1][:5]:
            i, j, k = np.unravel_index(int(flat), d.shape)
            if d[i, j, k] > 0:
                worst.append([int(i), int(j), int(k), float(h[i, j, k]), float(r[i, j, k])])
        rep[name] = dict(values=int(h.size), mismatch_fraction=frac, mismatches=int((d > bar).sum()), max_abs=float(d.max()),
                         max_abs_in_brackets=float(d.max() / (2.0 * acc)), worst_cells_row_col_azim_got_ref=worst)
        if not (frac <= 1.0e-3 and d.max() <= 2.5 * acc):
            failed.append(name)
    print(json.dumps({"embree_pin_horizon": rep}))
    if check:
        assert not failed, (failed, {k: rep[k] for k in failed})
    return rep


def compare_shadow(make_terrain, tol_cells=1.0e-4, path=None, check=True):
    """Shadow codes and sw_dir_cor against the reference's: per case the number and fraction of differing codes, the
    confusion counts (got -> reference) of the differing cells, the largest relative sw_dir_cor difference; asserts after
    the report is printed."""
    import json
    ref = np.load(path or SHADOW)
    g, (vec_tilt, vec_norm, enl, elev, mask), suns = _harness().shadow_case()
    rep, failed = {}, []
    for refrac in (False, True):
        t = make_terrain()
        t.initialise(g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask, refrac_cor=refrac,
                     sw_dir_cor_fill=-9.0)
        for geom in ("triangle", "grid"):
            key = "%s_refrac%d" % (geom, int(refrac))
            bad = tot = 0
            worst = 0.0
            confusion = {}
            for s in range(suns.shape[0]):
                a = np.empty(mask.shape, np.uint8); f = np.empty(mask.shape, np.float32)
                t.shadow(suns[s], a); t.sw_dir_cor(suns[s], f)
                r = ref["shadow__" + key][s]
                diff = a != r
                bad += int(diff.sum()); tot += a.size
                for x, y in zip(a[diff].tolist(), r[diff].tolist()):
                    confusion["%d->%d" % (x, y)] = confusion.get("%d->%d" % (x, y), 0) + 1
                both = (f != 0) & (ref["sw_dir_cor__" + key][s] != 0) & (mask == 1)
                if both.any():
                    worst = max(worst, float(np.abs(f[both] / ref["sw_dir_cor__" + key][s][both] - 1.0).max()))
            rep[key] = dict(cells=tot, differing_codes=bad, differing_fraction=bad / tot, confusion_got_to_ref=confusion,
                            sw_dir_cor_max_rel=worst)
            if not (bad / tot <= tol_cells and worst <= 1.0e-4):
                failed.append(key)
    print(json.dumps({"embree_pin_shadow": rep}))
    if check:
        assert not failed, (failed, {k: rep[k] for k in failed})
    return rep
