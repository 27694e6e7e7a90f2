"""Why do cells get no certificate?  HZ_NEAR_REASONS histograms (stderr of the library) for adversarial configurations
grouped by cell aspect, and for a 256-row slab of the c3 tile."""
import os, sys
os.environ["HZ_NEAR_REASONS"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import horayzon_amd as hz
from horayzon_amd import synth
from tests import cases
rng = np.random.default_rng(777)
seen = {}
while min([seen.get(a, 0) for a in cases.ADV_ASPECTS]) < 6:
    kw, par, desc = cases.adversarial_near_case(rng)
    a = (desc["dx"], desc["dy"])
    if seen.get(a, 0) >= 6 or desc["skew"] > 1e-4:
        continue
    seen[a] = seen.get(a, 0) + 1
    print("CASE", a, "origin", desc["origin"][0], "tilt", desc["tilt_deg"], "relief", round(desc["relief"], 1), desc["feats"], "elev", desc["ray_org_elev"], "off", desc["off"], file=sys.stderr, flush=True)
    hz.horizon.horizon_gridded(**kw, **par)
g = synth.fractal_tile(n=3601, offset=16)
print("CASE c3 tile rows 1600..1856", file=sys.stderr, flush=True)
hz.horizon.horizon_gridded(**{k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")},
                           dist_search=50.0, azim_num=360, rows=(1600, 1856))
