"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import numpy as np

from horayzon_amd import synth

GRID_KEYS = ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")


def grid_kwargs(g):
    return {k: g[k] for k in GRID_KEYS}


def c2_hill(height=1000.0):
    """BASELINE config 1/2: 200 x 200 Gaussian hill (1000 m: no search of the
    reference runs into its non-terminating loops; 1500 m does)."""
    return synth.gaussian_hill(n=200, dx=50.0, height=height, sigma=1500.0, offset=10)


def rough_terrain(n0, n1, seed, dx=30.0, dy=30.0, relief=800.0, offset=4, tilt_frames=False,
                  origin=(0.0, 0.0)):
    """Small fractal terrain; optionally with per-cell rotated (norm, north) frames that
    mimic a curved-earth ENU set-up (non axis-aligned rays)."""
    rng = np.random.default_rng(seed)
    z = synth.fractal_elevation(n0, n1, seed=seed, z_min=100.0, z_max=100.0 + relief)
    x = (origin[0] + np.arange(n1) * dx).astype(np.float32)
    y = (origin[1] + (n0 - 1 - np.arange(n0)) * dy).astype(np.float32)
    xx, yy = np.meshgrid(x, y)
    in0, in1 = n0 - 2 * offset, n1 - 2 * offset
    vec_norm, vec_north = synth.planar_frames(in0, in1)
    if tilt_frames:
        # small smooth rotations of the local frame (as on a curved DEM)
        ax = (0.02 * (np.arange(in0)[:, None] - in0 / 2) / in0 + 0 * np.arange(in1)[None, :]).astype(np.float64)
        ay = (0.02 * (np.arange(in1)[None, :] - in1 / 2) / in1 + 0 * np.arange(in0)[:, None]).astype(np.float64)
        nrm = np.stack([np.sin(ay), -np.sin(ax) * np.cos(ay), np.cos(ax) * np.cos(ay)], axis=2)
        north0 = np.array([0.0, 1.0, 0.0])
        north = north0[None, None, :] - (nrm * north0).sum(axis=2, keepdims=True) * nrm
        north /= np.linalg.norm(north, axis=2, keepdims=True)
        # rotate north slightly about the normal
        ang = 0.01 * rng.standard_normal((in0, in1, 1))
        east = np.cross(north, nrm)
        north = np.cos(ang) * north + np.sin(ang) * east
        vec_norm = np.ascontiguousarray(nrm, np.float32)
        vec_north = np.ascontiguousarray(north, np.float32)
    return dict(vert_grid=synth.pack_vertices(xx, yy, z), dem_dim_0=n0, dem_dim_1=n1,
                vec_norm=vec_norm, vec_north=vec_north, offset_0=offset, offset_1=offset,
                x=x, y=y, z=z)


def outer_tin(g, margin=3000.0, zval=900.0):
    """A coarse ring of triangles around the DEM (simplified outer domain,
    horizon_comp.cpp:199-218): 8 vertices, 8 triangles."""
    x0, x1 = float(g["x"].min()), float(g["x"].max())
    y0, y1 = float(g["y"].min()), float(g["y"].max())
    v = np.array([[x0, y0, zval * 0.2], [x1, y0, zval * 0.3], [x1, y1, zval * 0.25], [x0, y1, zval * 0.2],
                  [x0 - margin, y0 - margin, zval], [x1 + margin, y0 - margin, zval * 1.1],
                  [x1 + margin, y1 + margin, zval * 0.9], [x0 - margin, y1 + margin, zval]], np.float32)
    t = np.array([[0, 4, 5], [0, 5, 1], [1, 5, 6], [1, 6, 2], [2, 6, 7], [2, 7, 3], [3, 7, 4], [3, 4, 0]], np.int32)
    return v.ravel().copy(), v.shape[0], t.ravel().copy(), t.shape[0]


def terrain_inputs(g):
    """Terrain.initialise inputs for a planar case dict."""
    o = g["offset_0"]
    vec_tilt, surf_enl_fac = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], o)
    in0, in1 = vec_tilt.shape[:2]
    vec_norm, _ = synth.planar_frames(in0, in1)
    elevation = np.ascontiguousarray(g["z"][o:o + in0, o:o + in1], np.float32)
    mask = np.ones((in0, in1), np.uint8)
    return vec_tilt, vec_norm, surf_enl_fac, elevation, mask


ALGS = ("guess_constant", "binary_search", "discrete_sampling")


def random_config(rng, max_n=90):
    """One random horizon_gridded configuration (grid kwargs, parameters): sizes, spacings, relief,
    coordinate offsets, frames, algorithm, table resolution, masks and an outer TIN all vary."""
    n0, n1 = int(rng.integers(2, max_n)), int(rng.integers(2, max_n))
    off = int(rng.integers(0, max(1, min(n0, n1) // 3)))
    if n0 - 2 * off < 1 or n1 - 2 * off < 1:
        off = 0
    dx, dy = float(rng.choice([1.0, 10.0, 30.0, 90.0, 500.0])), float(rng.choice([1.0, 10.0, 30.0, 90.0, 500.0]))
    relief = float(rng.choice([0.0, 5.0, 300.0, 3000.0])) * (dx / 30.0) ** 0.5
    origin = (float(rng.choice([0.0, 2.6e6, -4.0e5])), float(rng.choice([0.0, 1.2e6])))
    g = rough_terrain(n0, n1, seed=int(rng.integers(1 << 30)), dx=dx, dy=dy, relief=relief, offset=off,
                            tilt_frames=bool(rng.integers(2)), origin=origin)
    kw = grid_kwargs(g)
    span = max(n0 * dy, n1 * dx) / 1000.0
    par = dict(dist_search=float(rng.choice([0.3, 1.0, 3.0])) * span,
               azim_num=int(rng.choice([1, 3, 8, 12, 30, 45])),
               hori_acc=float(rng.choice([0.1, 0.25, 1.0, 3.0])),
               ray_algorithm=str(rng.choice(ALGS)),
               geom_type=str(rng.choice(["triangle", "quad", "grid"])),
               elev_ang_low_lim=float(rng.choice([-15.0, -45.0, -89.98])),
               ray_org_elev=float(rng.choice([0.005, 0.01, 0.5, 20.0])),
               hori_fill=float(rng.choice([0.0, -1.0])))
    in0, in1 = kw["vec_norm"].shape[:2]
    if rng.integers(3) == 0:
        par["mask"] = (rng.random((in0, in1)) < 0.6).astype(np.uint8)
    if rng.integers(4) == 0 and n0 > 3 and n1 > 3:
        vs, nvs, ts, nts = outer_tin(g, margin=3.0 * span * 100.0, zval=100.0 + relief)
        par.update(vert_simp=vs, num_vert_simp=nvs, tri_ind_simp=ts, num_tri_simp=nts)
    return kw, par


def fuzz_case(rng):
    """One configuration of the gridded random sweep (tests/test_gpu_fuzz.py): random_config plus the
    extras (row slab, chunked host output, tiny LDS stacks, fused SVF).  Consumes `rng` exactly as the
    sweep does, so `scripts/stray/replay.py` can regenerate configuration k of a seed."""
    kw, par = random_config(rng)
    in0, in1 = kw["vec_norm"].shape[:2]
    extra = {}
    if rng.integers(3) == 0 and in0 > 2:                 # a row slab (the multi-GPU sharding unit)
        r0 = int(rng.integers(0, in0 - 1))
        extra["rows"] = (r0, int(rng.integers(r0 + 1, in0 + 1)))
    if rng.integers(3) == 0:                             # streamed host output in small chunks
        extra["_chunk_rows"] = int(rng.integers(1, 9))
    if rng.integers(3) == 0:                             # (round 1 drew a tiny LDS stack size here; the draw keeps the
        rng.integers(3, 13)                              #  configuration sequence of a seed unchanged)
    tilt = None
    if rng.integers(3) == 0 and par["azim_num"] >= 2:   # fused sky view factor
        a, b = rng.uniform(-0.4, 0.4, (in0, in1)), rng.uniform(-0.4, 0.4, (in0, in1))
        tilt = np.stack([np.sin(b), -np.sin(a) * np.cos(b), np.cos(a) * np.cos(b)], axis=2).astype(np.float32)
        extra["svf_vec_tilt"] = tilt
    return kw, par, extra, tilt


ADV_ASPECTS = ((1.0, 90.0), (90.0, 1.0), (0.5, 45.0), (45.0, 0.5), (2.0, 180.0), (30.0, 30.0), (10.0, 10.0), (1.0, 1.0),
               (10.0, 900.0), (900.0, 10.0), (5.0, 20.0), (20.0, 5.0))


def adversarial_near_case(rng, max_n=40):
    """One configuration aimed at the near-field certificates (hz_near.hip): cells of aspect 1:90 and
    90:1, per-cell frames tilted by up to 5 degrees in random directions, cliffs / single spikes / pits / terraces next to
    the cells, ray origins 0.005 ... 20 m above the surface, coordinates of 2.6e6 (where the ray origin is rounded to a
    0.25 m grid).  No outer TIN and a plain height field, so the certificates are active.  Returns (grid kwargs,
    parameters, description)."""
    n0, n1 = int(rng.integers(9, max_n)), int(rng.integers(9, max_n))
    off = int(rng.integers(0, 4))
    dx, dy = ADV_ASPECTS[int(rng.integers(len(ADV_ASPECTS)))]
    scale = float(min(dx, dy))
    relief = float(rng.choice([0.0, 3.0, 30.0, 300.0])) * scale ** 0.5
    origin = (float(rng.choice([0.0, 2.6e6])), float(rng.choice([0.0, 1.2e6])))
    z = synth.fractal_elevation(n0, n1, seed=int(rng.integers(1 << 30)), z_min=100.0, z_max=100.0 + relief).astype(np.float64)
    feats = []
    ii, jj = np.meshgrid(np.arange(n0), np.arange(n1), indexing="ij")
    if rng.integers(2):                       # a cliff along a row, a column or a diagonal
        h = float(rng.choice([5.0, 40.0, 400.0])) * float(rng.choice([scale ** 0.5, 1.0]))
        kind = int(rng.integers(3))
        c = int(rng.integers(2, max(3, min(n0, n1) - 2)))
        side = (ii > c) if kind == 0 else ((jj > c) if kind == 1 else (ii + jj > 2 * c))
        z = z + h * side
        feats.append("cliff%d:%g" % (kind, h))
    for _ in range(int(rng.integers(0, 5))):  # single spikes and pits
        h = float(rng.choice([-200.0, -20.0, 3.0, 30.0, 300.0]))
        z[int(rng.integers(n0)), int(rng.integers(n1))] += h
        feats.append("spike:%g" % h)
    if rng.integers(4) == 0:                  # terraces: flat steps with vertical-ish risers
        t = float(rng.choice([1.0, 10.0, 100.0]))
        z = np.round(z / t) * t
        feats.append("terrace:%g" % t)
    z = z.astype(np.float32)
    x = (origin[0] + np.arange(n1) * dx).astype(np.float32)
    y = (origin[1] + (n0 - 1 - np.arange(n0)) * dy).astype(np.float32)
    xx, yy = np.meshgrid(x, y)
    in0, in1 = n0 - 2 * off, n1 - 2 * off
    tmax = np.deg2rad(float(rng.choice([0.0, 0.4, 2.0, 5.0])))
    mag = tmax * rng.random((in0, in1))
    dirn = rng.uniform(0.0, 2.0 * np.pi, (in0, in1))
    nrm = np.stack([np.sin(mag) * np.cos(dirn), np.sin(mag) * np.sin(dirn), np.cos(mag)], axis=2)
    north0 = np.array([0.0, 1.0, 0.0])
    north = north0[None, None, :] - (nrm * north0).sum(axis=2, keepdims=True) * nrm
    north /= np.linalg.norm(north, axis=2, keepdims=True)
    ang = float(rng.choice([0.0, 0.05, 1.0])) * rng.standard_normal((in0, in1, 1))
    east = np.cross(north, nrm)
    north = np.cos(ang) * north + np.sin(ang) * east
    skew = float(rng.choice([0.0, 0.0, 0.0, 3.0e-5, 2.0e-3]))      # frames that are not quite orthonormal (the library
    if skew:                                                     #  must refuse certificates beyond 1e-4, not err)
        north = north + skew * nrm
        nrm = nrm * (1.0 + 0.5 * skew)
    kw = dict(vert_grid=synth.pack_vertices(xx, yy, z), dem_dim_0=n0, dem_dim_1=n1,
              vec_norm=np.ascontiguousarray(nrm, np.float32), vec_north=np.ascontiguousarray(north, np.float32),
              offset_0=off, offset_1=off)
    span = max(n0 * dy, n1 * dx) / 1000.0
    par = dict(dist_search=float(rng.choice([0.3, 1.0, 3.0])) * span,
               azim_num=int(rng.choice([8, 36, 90, 360])),
               hori_acc=float(rng.choice([0.1, 0.25, 1.0])),
               ray_algorithm=str(rng.choice(["guess_constant", "guess_constant", "binary_search", "discrete_sampling"])),
               elev_ang_low_lim=float(rng.choice([-15.0, -45.0, -89.98])),
               ray_org_elev=float(rng.choice([0.005, 0.01, 0.1, 2.0, 20.0])))
    if par["ray_algorithm"] == "discrete_sampling":         # ~45 ... 900 rays per azimuth: keep the CPU oracle's share small
        par["azim_num"] = min(par["azim_num"], 36)
        par["hori_acc"] = max(par["hori_acc"], 0.25)
    desc = dict(dem=[n0, n1], off=off, dx=dx, dy=dy, relief=relief, origin=list(origin), tilt_deg=float(np.rad2deg(tmax)),
                skew=skew, feats=feats, **{k: v for k, v in par.items()})
    return kw, par, desc
