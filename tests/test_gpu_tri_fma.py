"""The build-time switch -DHZ_TRI_FMA (hz_common.h): the triangle test's cross and dot products with fused multiply-adds, the
way Embree's vector code evaluates them on an FMA machine.  NOT the contract -- the switch exists so that adopting that
evaluation (should the Embree pin of README.md ask for it) is a flag and a re-validation.  This test keeps the two sides of
the switch together: a library built with the flag against the oracle's triangle mode "plain_fma", bit for bit (horizon,
ray and guard counts, closest-hit distances), in a subprocess because the library is chosen at import."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "horayzon_amd", "libhorayzon_hip_fma.so")

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
try:
    import torch  # noqa: F401  (tests/conftest.py: load order of the two HIP runtimes)
except Exception:
    pass
import horayzon_amd as hz
from horayzon_amd import _lib
from oracle import oracle as orc
from tests import cases
assert _lib.LIB_PATH.endswith("libhorayzon_hip_fma.so"), _lib.LIB_PATH
orc.build()
out = {"cases": 0, "differs_from_plain": 0}
rng = np.random.default_rng(5150)
todo = [(cases.grid_kwargs(cases.c2_hill()), dict(dist_search=10.0, azim_num=36)),
        (cases.grid_kwargs(cases.rough_terrain(93, 117, seed=7, offset=6, tilt_frames=True)),
         dict(dist_search=2.0, azim_num=24, elev_ang_low_lim=-60.0, ray_algorithm="binary_search"))]
for _ in range(12):
    todo.append(cases.random_config(rng, max_n=34))
for it in range(6):
    kw, par, _ = cases.adversarial_near_case(rng)
    todo.append((kw, par))
for kw, par in todo:
    h, a = hz.horizon.horizon_gridded(**kw, **par)
    st = dict(hz.horizon.last_stats)
    orc.set_tri_mode("plain_fma")
    r, ar, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    orc.set_tri_mode("plain")
    p, _, sp = orc.horizon_gridded(**kw, **par, return_stats=True)
    assert np.array_equal(h, r, equal_nan=True), "horizon differs from the oracle's plain_fma mode"
    assert st["num_rays"] == so["rays"] and st["guard_events"] == so["guards"], (st["num_rays"], so["rays"], st["guard_events"], so["guards"])
    out["cases"] += 1
    out["differs_from_plain"] += int(not (np.array_equal(r, p, equal_nan=True) and so["rays"] == sp["rays"]))
# closest hit: horizon_locations with distances (the *_hori_dist variants)
g = cases.rough_terrain(70, 80, seed=57, offset=0, relief=700.0)
n = 300
ci = rng.integers(3, 67, n); cj = rng.integers(3, 77, n)
coords = np.stack([g["x"][cj] + rng.uniform(-12, 12, n), g["y"][ci] + rng.uniform(-12, 12, n),
                   g["z"][ci, cj] + rng.uniform(-150, 300, n)], axis=1).astype(np.float32)
vn = np.zeros((n, 3), np.float32); vn[:, 2] = 1.0
vo = np.zeros((n, 3), np.float32); vo[:, 1] = 1.0
par = dict(azim_num=16, ray_algorithm="binary_search", hori_dist_out=True, elev_ang_low_lim=-60.0)
a = hz.horizon.horizon_locations(g["vert_grid"], 70, 80, coords, vn, vo, 2.5, **par)
orc.set_tri_mode("plain_fma")
b = orc.horizon_locations(g["vert_grid"], 70, 80, coords, vn, vo, 2.5, **par)
orc.set_tri_mode("plain")
assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)
out["closest_hit"] = "ok"
print("RESULT " + json.dumps(out))
"""


def _ensure_lib():
    if os.path.exists(LIB):
        return True
    try:
        subprocess.run(["bash", os.path.join(ROOT, "scripts", "build_variant.sh"), "fma", "-DHZ_TRI_FMA"], cwd=ROOT,
                       check=True, capture_output=True, timeout=900)
    except Exception:
        return False
    return os.path.exists(LIB)


def test_fma_build_equals_the_oracles_plain_fma_mode(hip):
    if not _ensure_lib():
        pytest.skip("libhorayzon_hip_fma.so is not built and hipcc could not build it here")
    env = dict(os.environ, HORAYZON_HIP_LIB=LIB)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["cases"] == 20
    print(res)
