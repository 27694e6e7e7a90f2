"""CPU tests of the oracle (oracle/hz_oracle.c): golden vectors generated from the
reference's importable modules, analytic known answers, and agreement of the oracle's
three intersection paths.  These pin the checker the GPU parity tests rely on."""
import os

import numpy as np
import pytest

from horayzon_amd import synth
from tests import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------------------------------
# golden vectors from the reference (topo_param.sky_view_factor, tests/golden/make_fixtures.py)
# ---------------------------------------------------------------------------------------
def test_svf_golden_reference(orc):
    d = np.load(os.path.join(GOLD, "svf_reference.npz"))
    for n in "abc":
        svf = orc.sky_view_factor(d["azim_" + n], d["hori_" + n], d["tilt_" + n])
        assert np.abs(svf - d["svf_" + n]).max() <= 1.0e-5, n     # north-star SVF tolerance
    azim = d["azim_a"]
    flat = np.zeros((2, 2, 36), np.float32)
    up = np.zeros((2, 2, 3), np.float32); up[..., 2] = 1.0
    assert np.abs(orc.sky_view_factor(azim, flat, up) - d["svf_flat"]).max() <= 1e-6
    assert np.abs(d["svf_flat"] - 1.0).max() <= 1e-6              # closed form
    h30 = flat + np.float32(np.deg2rad(30.0))
    assert np.abs(orc.sky_view_factor(azim, h30, up) - d["svf_30deg"]).max() <= 1e-6
    assert np.abs(d["svf_30deg"] - 0.75).max() <= 1e-6            # cos^2(30 deg)


def test_svf_argument_checks(orc):
    azim = np.zeros(4, np.float32); hori = np.zeros((2, 2, 4), np.float32); tilt = np.zeros((2, 2, 3), np.float32)
    with pytest.raises(ValueError):
        orc.sky_view_factor(azim[:3], hori, tilt)
    with pytest.raises(ValueError):
        orc.sky_view_factor(azim.astype(np.float64), hori, tilt)


# ---------------------------------------------------------------------------------------
# analytic known answers
# ---------------------------------------------------------------------------------------
def _flat(n=41, dx=25.0, off=8):
    x = (np.arange(n) * dx).astype(np.float32)
    xx, yy = np.meshgrid(x, x)
    z = np.zeros_like(xx)
    vn, vo = synth.planar_frames(n - 2 * off, n - 2 * off)
    return dict(vert_grid=synth.pack_vertices(xx, yy, z), dem_dim_0=n, dem_dim_1=n, vec_norm=vn,
                vec_north=vo, offset_0=off, offset_1=off), x, z


@pytest.mark.parametrize("alg", ("binary_search", "guess_constant", "discrete_sampling"))
def test_flat_plane(orc, alg):
    """Flat plane: every ray below 0 deg hits within the DEM, every ray above misses ->
    the horizon sits in the bracket just below 0."""
    kw, _, _ = _flat()
    acc = np.deg2rad(0.25)
    h, azim, st = orc.horizon_gridded(**kw, dist_search=5.0, azim_num=24, ray_algorithm=alg,
                                      elev_ang_low_lim=-60.0, return_stats=True)
    assert st["guards"] == 0
    assert (h <= acc + 1e-6).all() and (h >= -2.0 * acc - 1e-6).all()   # bracket midpoint around 0
    assert np.allclose(azim, np.float32(2 * np.pi / 24) * np.arange(24), atol=1e-6)


def test_wall_known_angle(orc):
    """A ridge of height H at distance D north of a cell: horizon(azimuth 0) = atan((H - e) / D)."""
    n, dx, off = 61, 20.0, 10
    x = (np.arange(n) * dx).astype(np.float32)
    y = ((n - 1 - np.arange(n)) * dx).astype(np.float32)       # row 0 is the northern edge
    xx, yy = np.meshgrid(x, y)
    z = np.zeros_like(xx)
    H = 300.0
    z[5, :] = H                                                 # east-west ridge
    vn, vo = synth.planar_frames(n - 2 * off, n - 2 * off)
    kw = dict(vert_grid=synth.pack_vertices(xx, yy, z), dem_dim_0=n, dem_dim_1=n, vec_norm=vn,
              vec_north=vo, offset_0=off, offset_1=off)
    acc_deg = 0.25
    h, _ = orc.horizon_gridded(**kw, dist_search=5.0, azim_num=4, ray_algorithm="binary_search",
                               hori_acc=acc_deg, elev_ang_low_lim=-30.0)
    for i_in in (0, 10, 25):
        D = (i_in + off - 5) * dx
        expect = np.arctan((H - 0.01) / D)
        assert abs(h[i_in, 20, 0] - expect) <= np.deg2rad(acc_deg) + 1e-6
    # looking south / east / west only the flat plane is visible
    assert (np.abs(h[:, 10:30, 2]) <= np.deg2rad(acc_deg) + 1e-6).all()


def test_gaussian_hill_symmetry_and_algorithms(orc):
    """180-degree rotational symmetry of the mesh (the quad diagonal keeps its orientation
    under a half turn) and agreement of the three search algorithms within their accuracy."""
    g = cases.c2_hill()
    kw = cases.grid_kwargs(g)
    acc = np.deg2rad(0.25)
    res = {}
    for alg in ("guess_constant", "binary_search"):
        res[alg], _, st = orc.horizon_gridded(**kw, dist_search=10.0, azim_num=36, ray_algorithm=alg,
                                              return_stats=True)
        assert st["guards"] == 0
    hb = res["binary_search"]
    rot = np.roll(hb[::-1, ::-1, :], 18, axis=2)                # (i, j, k) -> (N-1-i, N-1-j, k + 18)
    assert np.abs(hb - rot).max() <= 2.0 * acc
    assert (np.abs(hb - rot) > 1e-6).mean() < 0.02
    assert np.abs(res["guess_constant"] - hb).max() <= 2.5 * acc
    # looking from the plain towards the hill: horizon close to the apparent summit angle
    i, j = 90, 0                                                # inner cell west of the summit row
    d = np.hypot(g["x"][j + 10] - g["x"].mean(), g["y"][i + 10] - g["y"].mean())
    assert hb[i, j, 9] > 0.5 * np.arctan(1000.0 / d)            # azimuth 90 deg = east


# ---------------------------------------------------------------------------------------
# the three intersection paths agree
# ---------------------------------------------------------------------------------------
def test_bvh_equals_brute_force(orc):
    g = cases.rough_terrain(40, 44, seed=5, offset=0, relief=600.0)
    sc = orc.Scene(g["vert_grid"], 40, 44)
    rng = np.random.default_rng(1)
    nray = 60000
    ci = rng.integers(0, 40, nray); cj = rng.integers(0, 44, nray)
    org = np.stack([g["x"][cj], g["y"][ci], g["z"][ci, cj] + np.float32(0.01)], axis=1).astype(np.float32)
    az = rng.uniform(0, 2 * np.pi, nray); el = np.deg2rad(rng.uniform(-30.0, 45.0, nray))
    # include exactly axis-aligned and diagonal rays (they run along mesh edges)
    az[:8000] = np.deg2rad(rng.choice([0.0, 45.0, 90.0, 135.0, 180.0, 225.0, 270.0, 315.0], 8000))
    d = np.stack([np.cos(el) * np.sin(az), np.cos(el) * np.cos(az), np.sin(el)], axis=1).astype(np.float32)
    tfar = np.where(rng.random(nray) < 0.3, np.inf, rng.uniform(50.0, 3000.0, nray)).astype(np.float32)
    h_bvh = sc.occluded(org, d, tfar, orc.MODE_BVH)
    h_brute = sc.occluded(org, d, tfar, orc.MODE_BRUTE)
    h_f64 = sc.occluded(org, d, tfar, orc.MODE_BRUTE_F64)
    assert np.array_equal(h_bvh, h_brute)          # the tree never changes a decision
    assert 0.2 < h_brute.mean() < 0.95
    assert (h_brute != h_f64).mean() < 2e-3         # float32 vs exact geometry: only grazing rays


def test_degenerate_terrain_bvh_equals_brute_force(orc):
    """Flat, terraced (vertical walls), single-spike and strip DEMs: boxes without extent and many
    coplanar faces must not change a decision either (same shapes as the GPU parity test)."""
    from horayzon_amd import synth

    def dem(z, offset):
        n0, n1 = z.shape
        x = (np.arange(n1) * 30.0).astype(np.float32)
        y = ((n0 - 1 - np.arange(n0)) * 30.0).astype(np.float32)
        xx, yy = np.meshgrid(x, y)
        vn, vo = synth.planar_frames(n0 - 2 * offset, n1 - 2 * offset)
        return dict(vert_grid=synth.pack_vertices(xx, yy, z.astype(np.float32)), dem_dim_0=n0, dem_dim_1=n1,
                    vec_norm=vn, vec_north=vo, offset_0=offset, offset_1=offset)
    yy, xx = np.mgrid[0:26, 0:30]
    spike = np.zeros((21, 23)); spike[10, 11] = 500.0
    shapes = [(np.full((20, 22), 250.0), 2), (100.0 * ((xx // 6) % 4) + 50.0 * ((yy // 5) % 3), 2), (spike, 2),
              (200.0 * np.random.default_rng(5).random((3, 120)), 0)]
    for z, off in shapes:
        kw = dem(z, off)
        a, _, sa = orc.horizon_gridded(**kw, dist_search=2.0, azim_num=16, elev_ang_low_lim=-80.0, return_stats=True)
        b, _, sb = orc.horizon_gridded(**kw, dist_search=2.0, azim_num=16, elev_ang_low_lim=-80.0,
                                       mode=orc.MODE_BRUTE, return_stats=True)
        assert np.array_equal(a, b) and sa["rays"] == sb["rays"]
    # flat plane: every horizon sits in the bracket just below 0
    a, _ = orc.horizon_gridded(**dem(np.full((20, 22), 250.0), 2), dist_search=2.0, azim_num=16)
    assert (a <= 0.0).all() and (a > np.deg2rad(-0.3)).all()


def test_random_configurations_bvh_equals_brute_force(orc):
    """The generator of tests/test_gpu_fuzz.py on the CPU side: the oracle's tree never changes a
    decision, whatever the spacing, relief or coordinate offset (HZ_FUZZ_N / HZ_FUZZ_SEED widen it)."""
    import os
    n = int(os.environ.get("HZ_FUZZ_N", "12"))
    rng = np.random.default_rng(int(os.environ.get("HZ_FUZZ_SEED", "777")))
    for it in range(n):
        kw, par = cases.random_config(rng, max_n=34)
        a, _, sa = orc.horizon_gridded(**kw, **par, return_stats=True)
        b, _, sb = orc.horizon_gridded(**kw, **par, mode=orc.MODE_BRUTE, return_stats=True)
        desc = "config %d: dem %dx%d %s" % (it, kw["dem_dim_0"], kw["dem_dim_1"],
                                           {k: v for k, v in par.items() if np.isscalar(v)})
        assert np.array_equal(a, b), desc
        assert sa["rays"] == sb["rays"] and sa["guards"] == sb["guards"], desc


def test_known_counter_example_grazing_ray_at_its_origin(orc):
    """DESIGN.md section 4 item 3, found by scripts/fuzz_near_adversarial.py --seed 48001 (configuration 2536): the float
    triangle test accepts, at t = 0, a ray whose origin lies 5 mm OUTSIDE the triangle's box -- it grazes the plane of a
    cliff triangle and T cancels to exactly 0.  Brute force reports the hit; a tree whose box tests start at exactly 0 culls
    the box (round 4: guard count 88 against 89).  Since round 5 the box tests of the oracle's tree (and of the HIP
    kernels, tests/test_gpu_fuzz.py::test_grazing_ray_counter_example_on_the_gpu) run over [-tau, tfar + tau]: tree = brute force."""
    rng = np.random.default_rng(48001)
    for _ in range(2537):
        kw, par, desc = cases.adversarial_near_case(rng)
    assert desc["dem"] == [17, 22] and par["ray_algorithm"] == "discrete_sampling"
    sc = orc.Scene(kw["vert_grid"], 17, 22)
    o = np.array([[140.00006, 50.000305, 533.005]], np.float32)
    d = np.array([[0.01114424, 0.06120159, 0.9980782]], np.float32)
    assert bool(sc.occluded(o, d, 132.0, mode=orc.MODE_BRUTE)[0]) and bool(sc.occluded(o, d, 132.0, mode=orc.MODE_BVH)[0])
    h0, _, s0 = orc.horizon_gridded(**kw, **par, return_stats=True, mode=orc.MODE_BVH)
    h1, _, s1 = orc.horizon_gridded(**kw, **par, return_stats=True, mode=orc.MODE_BRUTE)
    assert np.array_equal(h0, h1, equal_nan=True) and s0["rays"] == s1["rays"]
    assert (s0["guards"], s1["guards"]) == (89, 89)
    # the hole is real: with box tests that start at exactly 0 (the round-4 tree) the triangle's box is culled;
    # 4 pads are the least that find it, the contract has 16
    try:
        orc.set_box_start(0.0)
        assert not bool(sc.occluded(o, d, 132.0, mode=orc.MODE_BVH)[0])
        _, _, s2 = orc.horizon_gridded(**kw, **par, return_stats=True, mode=orc.MODE_BVH)
        assert s2["guards"] == 88
        orc.set_box_start(2.0)
        assert not bool(sc.occluded(o, d, 132.0, mode=orc.MODE_BVH)[0])
        orc.set_box_start(4.0)
        assert bool(sc.occluded(o, d, 132.0, mode=orc.MODE_BVH)[0])
    finally:
        orc.set_box_start()


def _contract_old_vs_shipped(orc, kw, **par):
    """horizon + ray / guard counts under the shipped contract and under the published Embree test (den != 0, box tests from 0)."""
    new, _, sn = orc.horizon_gridded(**kw, **par, return_stats=True)
    try:
        orc.set_den_noise(0.0); orc.set_box_start(0.0)
        old, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    finally:
        orc.set_den_noise(); orc.set_box_start()
    return new, sn, old, so


def test_round5_contract_clauses_move_no_baseline_result(orc):
    """Round 5 gave the numerical contract two clauses, on the product's and the oracle's side at once:
    the triangle test's parallel check is |den| > 2^-20 sum|n_i d_i| instead of Embree's published den != 0, and the tree's box
    tests run over [-tau, tfar + tau] instead of [0, tfar].  Both exist to make tree == brute force on adversarial inputs (the
    counter-example above); neither may change what the published test decides on the BASELINE inputs.  Here: config 2 with the
    three algorithms, the large-coordinate and the tilted-frame cases, and rows of config 3 -- every horizon value, ray count and
    guard count equal with the two clauses switched off (orc.set_den_noise(0), orc.set_box_start(0))."""
    total = 0
    g = cases.c2_hill()
    for alg in ("guess_constant", "binary_search", "discrete_sampling"):
        new, sn, old, so = _contract_old_vs_shipped(orc, cases.grid_kwargs(g), dist_search=10.0, azim_num=36, ray_algorithm=alg)
        assert np.array_equal(new, old) and sn["rays"] == so["rays"] and sn["guards"] == so["guards"], alg
        total += new.size
    g = cases.c2_hill(height=1500.0)            # (the guard-event variant of config 2)
    new, sn, old, so = _contract_old_vs_shipped(orc, cases.grid_kwargs(g), dist_search=10.0, azim_num=36)
    assert np.array_equal(new, old) and sn["rays"] == so["rays"] and sn["guards"] == so["guards"] > 0
    g = cases.rough_terrain(80, 90, seed=11, dx=25.0, dy=25.0, offset=5, origin=(668000.0, 172000.0))   # (tests/test_gpu_parity.py::test_large_coordinates)
    for alg in ("guess_constant", "binary_search", "discrete_sampling"):
        new, sn, old, so = _contract_old_vs_shipped(orc, cases.grid_kwargs(g), dist_search=1.5, azim_num=30, elev_ang_low_lim=-70.0,
                                                   ray_algorithm=alg)
        assert np.array_equal(new, old) and sn["rays"] == so["rays"] and sn["guards"] == so["guards"], alg
        total += new.size
    g = cases.rough_terrain(93, 117, seed=7, offset=6, tilt_frames=True)
    for alg in ("guess_constant", "binary_search", "discrete_sampling"):
        new, sn, old, so = _contract_old_vs_shipped(orc, cases.grid_kwargs(g), dist_search=2.0, azim_num=24, elev_ang_low_lim=-60.0,
                                                   ray_algorithm=alg)
        assert np.array_equal(new, old) and sn["rays"] == so["rays"] and sn["guards"] == so["guards"], alg
        total += new.size
    assert total > 1.5e6


def test_round5_contract_clauses_move_no_config3_row(orc):
    """The same on BASELINE config 3 (3601^2 synthetic tile, 360 azimuths, 50 km): inner-domain rows 1777-1778 (two of the five rows
    tests/test_gpu_fullsize.py compares with the GPU; the host here has 8 cores)."""
    g = synth.fractal_tile(n=3601, offset=16)
    new, sn, old, so = _contract_old_vs_shipped(orc, cases.grid_kwargs(g), dist_search=50.0, azim_num=360, rows=(1777, 1779), slab_only=True)
    assert new.shape == (2, 3569, 360) and not np.isnan(new).any()
    assert np.array_equal(new, old) and sn["rays"] == so["rays"] and sn["guards"] == so["guards"]


def test_horizon_bvh_equals_brute_force_and_tin(orc):
    g = cases.rough_terrain(34, 38, seed=15, offset=3, relief=500.0)
    kw = cases.grid_kwargs(g)
    vs, nvs, ts, nts = cases.outer_tin(g, margin=1500.0, zval=500.0)
    for extra in ({}, dict(vert_simp=vs, num_vert_simp=nvs, tri_ind_simp=ts, num_tri_simp=nts)):
        a, _, sa = orc.horizon_gridded(**kw, dist_search=4.0, azim_num=16, elev_ang_low_lim=-50.0,
                                       return_stats=True, **extra)
        b, _, sb = orc.horizon_gridded(**kw, dist_search=4.0, azim_num=16, elev_ang_low_lim=-50.0,
                                       mode=orc.MODE_BRUTE, return_stats=True, **extra)
        assert np.array_equal(a, b) and sa["rays"] == sb["rays"]
    # num_vert_simp < 3 -> the TIN is ignored (horizon_comp.cpp:199)
    c, _ = orc.horizon_gridded(**kw, dist_search=4.0, azim_num=16, elev_ang_low_lim=-50.0,
                               vert_simp=vs, num_vert_simp=2, tri_ind_simp=ts, num_tri_simp=nts)
    d, _ = orc.horizon_gridded(**kw, dist_search=4.0, azim_num=16, elev_ang_low_lim=-50.0)
    assert np.array_equal(c, d)


def test_closest_hit_and_locations(orc):
    """Closest hit: BVH == brute force (the minimum over all accepted triangles is order
    independent); a vertical ray from above returns the height above ground."""
    g = cases.rough_terrain(40, 44, seed=5, offset=0, relief=600.0)
    sc = orc.Scene(g["vert_grid"], 40, 44)
    rng = np.random.default_rng(2)
    n = 20000
    ci = rng.integers(1, 39, n); cj = rng.integers(1, 43, n)
    org = np.stack([g["x"][cj], g["y"][ci], g["z"][ci, cj] + np.float32(0.01)], axis=1).astype(np.float32)
    az = rng.uniform(0, 2 * np.pi, n); el = np.deg2rad(rng.uniform(-40.0, 30.0, n))
    d = np.stack([np.cos(el) * np.sin(az), np.cos(el) * np.cos(az), np.sin(el)], axis=1).astype(np.float32)
    h0, t0 = sc.closest(org, d, 2500.0, orc.MODE_BVH)
    h1, t1 = sc.closest(org, d, 2500.0, orc.MODE_BRUTE)
    assert np.array_equal(h0, h1) and np.array_equal(t0[h0], t1[h1])
    assert np.array_equal(h0, sc.occluded(org, d, 2500.0))            # same acceptance as any-hit
    assert (t0[h0] >= 0).all() and (t0[h0] <= 2500.0 * 1.0001).all()
    up = np.zeros((50, 3), np.float32); up[:, 2] = -1.0
    o2 = np.stack([g["x"][cj[:50]], g["y"][ci[:50]], g["z"][ci[:50], cj[:50]] + np.float32(123.0)], axis=1).astype(np.float32)
    h2, t2 = sc.closest(o2, up, 1.0e5)
    assert h2.all() and np.abs(t2 - 123.0).max() < 1e-2
    # locations driver: points above / below the surface are snapped, far away ones stay NaN
    vn = np.zeros((50, 3), np.float32); vn[:, 2] = 1.0
    vo = np.zeros((50, 3), np.float32); vo[:, 1] = 1.0
    o2[:25, 2] -= 200.0
    o2[0, 0] += 1.0e6
    h, azim, st = orc.horizon_locations(g["vert_grid"], 40, 44, o2, vn, vo, 1.0, azim_num=8, return_stats=True)
    assert st["found"] == 49 and np.isnan(h[0]).all() and not np.isnan(h[1:]).any()
    hg, _ = orc.horizon_gridded(g["vert_grid"], 40, 44, np.repeat(vn[:1], 1, 0).reshape(1, 1, 3),
                                np.repeat(vo[:1], 1, 0).reshape(1, 1, 3), int(ci[1]), int(cj[1]), 1.0, azim_num=8,
                                ray_algorithm="binary_search", elev_ang_low_lim=-89.98)
    assert np.abs(h[1] - hg[0, 0]).max() <= np.deg2rad(0.25) + 1e-6     # same cell via the gridded driver


def test_curved_dem_fixture(orc):
    """ENU vertices / normals / north vectors produced by the REFERENCE's transform and
    direction modules (tests/golden/make_fixtures.py): non axis-aligned frames."""
    d = np.load(os.path.join(GOLD, "curved_dem_reference.npz"))
    off = int(d["offset"])
    n0, n1 = d["x_enu"].shape
    kw = dict(vert_grid=synth.pack_vertices(d["x_enu"], d["y_enu"], d["z_enu"]), dem_dim_0=n0,
              dem_dim_1=n1, vec_norm=d["vec_norm"], vec_north=d["vec_north"], offset_0=off, offset_1=off)
    assert np.abs((d["vec_norm"] * d["vec_north"]).sum(axis=2)).max() < 1e-5     # orthogonal frames
    h, _, st = orc.horizon_gridded(**kw, dist_search=3.0, azim_num=18, elev_ang_low_lim=-89.98,
                                   return_stats=True)
    b, _ = orc.horizon_gridded(**kw, dist_search=3.0, azim_num=18, elev_ang_low_lim=-89.98,
                               mode=orc.MODE_BRUTE)
    assert np.array_equal(h, b)
    assert st["guards"] == 0 and not np.isnan(h).any()
    assert -1.58 < h.min() and h.max() < 1.3


# ---------------------------------------------------------------------------------------
# shadow
# ---------------------------------------------------------------------------------------
def test_shadow_known_answers(orc):
    g = cases.c2_hill(height=1500.0)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    mask[0, :3] = 0
    t = orc.Terrain()
    t.initialise(g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask, sw_dir_cor_fill=-1.0)
    centre = np.array([4975.0, 4975.0, 0.0], np.float32)
    sh = np.empty(mask.shape, np.uint8)
    # sun at the zenith: nothing is shaded
    t.shadow(centre + np.array([0, 0, 1.5e11], np.float32), sh)
    assert np.all(sh[mask == 1] == 0) and np.all(sh[mask == 0] == 3)
    # sun below the horizontal plane: every cell of the plain is self-shaded
    t.shadow(centre + np.array([1e10, 0, -1e9], np.float32), sh)
    plain = (vec_tilt[..., 2] > 0.99999) & (mask == 1)
    assert plain.sum() > 1000 and np.all(sh[plain] == 1)
    # low sun in the east: terrain shadow west of the hill, none on the sunlit plain east of it
    t.shadow(centre + np.array([1.5e11, 0, 1.5e11 * np.tan(np.deg2rad(8.0))], np.float32), sh)
    assert np.all(sh[85:95, 0:15] == 2)          # plain behind the hill: terrain shadow
    assert np.all(sh[85:95, 30:80] == 1)         # west flank: tilted away from the sun
    assert np.all(sh[85:95, 100:178] == 0)       # east flank and plain: lit
    # sw_dir_cor: flat, lit cells have tilt == norm and enlargement 1 -> exactly 1
    f = np.empty(mask.shape, np.float32)
    t.sw_dir_cor(centre + np.array([1.5e11, 0, 1.5e11 * np.tan(np.deg2rad(40.0))], np.float32), f)
    assert np.all(f[mask == 0] == -1.0)
    assert np.abs(f[2:6, 150:175] - 1.0).max() < 1e-2 and f[90, 120] > 1.5 and f[90, 60] < 0.5
    # refraction lifts the apparent sun: more (or equally many) lit cells than without it
    tr = orc.Terrain()
    tr.initialise(g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask, refrac_cor=True)
    sun = centre + np.array([1.5e11, 0, 1.5e11 * np.tan(np.deg2rad(3.0))], np.float32)
    a = np.empty(mask.shape, np.uint8); b = a.copy()
    t.shadow(sun, a); tr.shadow(sun, b)
    assert (b == 0).sum() >= (a == 0).sum() and (a != b).any()


def test_shadow_consistent_with_horizon(orc):
    """A cell is terrain-shaded exactly when the sun is below its horizon at the sun's azimuth."""
    g = cases.c2_hill(height=1500.0)
    kw = cases.grid_kwargs(g)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    h, azim = orc.horizon_gridded(**kw, dist_search=20.0, azim_num=36, ray_algorithm="binary_search",
                                  hori_acc=0.1, elev_ang_low_lim=-25.0, ray_org_elev=0.05)
    t = orc.Terrain()
    t.initialise(g["vert_grid"], 200, 200, 10, 10, vec_norm.copy(), vec_norm, enl, elev, mask)
    k, sun_el = 9, np.deg2rad(6.0)                              # azimuth 90 deg (east)
    far = 1.0e9
    sun = np.array([4975.0 + far * np.cos(sun_el), 4975.0, far * np.sin(sun_el)], np.float32)
    sh = np.empty(mask.shape, np.uint8)
    t.shadow(sun, sh)
    clear = np.abs(h[:, :, k] - sun_el) > np.deg2rad(0.3)       # skip cells within the accuracy band
    assert np.array_equal(sh[clear] == 2, h[:, :, k][clear] > sun_el)


# ---------------------------------------------------------------------------------------
# the refraction branch's float libm calls (hz_crmath.h, shared by the HIP kernels and the oracle)
# ---------------------------------------------------------------------------------------
def test_crmath_is_the_correctly_rounded_float(orc):
    """Exhaustive over the argument ranges the path uses (a few 1e8 floats): hz_crmath.h equals the float64
    libm result rounded once -- i.e. the correctly rounded float -- for every argument, while the platform's
    float routines (glibc 2.35 here) are NOT a fixed target: acosf / tanf differ from it for 0.2 % / 2 %."""
    exp = np.float32(9.81) / (np.float32(287.0) * np.float32(0.0065))           # shadow_comp.cpp:353-354
    for which, lo, hi, y in (("acos", 0.5, 1.0, 0.0), ("acos", -1.0, -0.5, 0.0), ("acos", 0.30, 0.52, 0.0),
                             ("acos", -0.52, -0.30, 0.0), ("acos", 1.0e-20, 1.0e-3, 0.0),
                             ("tan", 0.02, 1.62, 0.0), ("cos", 1.0e-12, 0.02, 0.0), ("sin", 1.0e-12, 0.02, 0.0),
                             ("pow", 0.6, 1.1, float(exp))):
        if lo < 0:      # floats are swept by bit pattern: negative ranges run from -|hi| "up" to -|lo|
            n, bad_d, bad_f = orc.crmath_sweep(which, hi, lo, y)
        else:
            n, bad_d, bad_f = orc.crmath_sweep(which, lo, hi, y)
        assert n > 100000, (which, lo, hi, n)
        assert bad_d == 0, "%s [%g, %g]: %d of %d differ from the rounded float64 result" % (which, lo, hi, bad_d, n)
    # specials the kernel can meet: |dot| marginally above 1, non-positive temperature ratio
    assert orc.crmath_sweep("acos", 1.0, 1.0)[1] == 0 and orc.crmath_sweep("acos", 1.0000001, 1.0000002)[1] == 0
    assert orc.crmath_sweep("pow", 0.0, 0.0, 5.25)[1] == 0


def test_division_by_a_constant_is_the_ieee_quotient(orc):
    """The HIP kernels form deg2rad / rad2deg's `/ 180.0` and `/ M_PI` (shadow_comp.cpp:43-62) as x rc + fma corrections
    (hz_crmath.h: hz_crm_div_const, four instructions instead of the eleven of a float64 division).  The arguments are floats promoted to
    double, so the claim "this is the IEEE quotient" is checked for EVERY finite float, both signs, both constants."""
    import math
    for c in (180.0, math.pi):
        for lo, hi in ((0x00000000, 0x7f7fffff), (0x80000000, 0xff7fffff)):     # +0 ... FLT_MAX, -0 ... -FLT_MAX
            n, bad = orc.div_const_sweep(c, lo, hi)
            assert n == 0x7f800000 and bad == 0, (c, hex(lo), n, bad)
        # (NaNs stay NaNs; +-inf gives NaN instead of +-inf -- see the header: no finite angle reaches it)
        assert orc.div_const_sweep(c, 0x7fc00000, 0x7fc00000) == (1, 0)
        assert orc.div_const_sweep(c, 0x7f800000, 0x7f800000) == (1, 1)


def test_crmath_equals_the_oracles_own_long_double_evaluation(orc):
    """The shared header against the oracle's OWN evaluation of the contract (long double libm rounded once, set_libm(2)):
    same shadow codes and the same sw_dir_cor bits over a day of sun positions on config 2 -- the CPU-side twin of
    tests/test_gpu_parity.py::test_refraction_against_the_oracles_own_functions."""
    g = cases.c2_hill(height=1500.0)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    from horayzon_amd import synth
    suns, _, _ = synth.sun_positions(num=24)
    suns = suns + np.array([5000.0, 5000.0, 0.0], np.float32)
    t = orc.Terrain()
    t.initialise(g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask, refrac_cor=True)
    try:
        for s in range(suns.shape[0]):
            a = np.empty(mask.shape, np.uint8); b = a.copy()
            fa = np.empty(mask.shape, np.float32); fb = fa.copy()
            orc.set_libm(0); t.shadow(suns[s], a); t.sw_dir_cor(suns[s], fa)
            orc.set_libm(2); t.shadow(suns[s], b); t.sw_dir_cor(suns[s], fb)
            assert np.array_equal(a, b) and np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), s
    finally:
        orc.set_libm(False)


def test_refraction_depends_little_on_the_platform_libm(orc):
    """With the platform's float routines instead of hz_crmath.h a handful of cells per million change: the
    reference's own output moves by that much from one libm to the next."""
    g = cases.c2_hill(height=1500.0)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    from horayzon_amd import synth
    suns, _, _ = synth.sun_positions(num=24)
    suns = suns + np.array([5000.0, 5000.0, 0.0], np.float32)
    t = orc.Terrain()
    t.initialise(g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask, refrac_cor=True)
    diff = tot = 0
    worst = 0.0
    try:
        for s in range(suns.shape[0]):
            a = np.empty(mask.shape, np.uint8); b = a.copy()
            fa = np.empty(mask.shape, np.float32); fb = fa.copy()
            orc.set_libm(False); t.shadow(suns[s], a); t.sw_dir_cor(suns[s], fa)
            orc.set_libm(True); t.shadow(suns[s], b); t.sw_dir_cor(suns[s], fb)
            diff += int((a != b).sum()); tot += a.size
            both = (fa != 0) & (fb != 0)
            if both.any():
                worst = max(worst, float(np.abs(fa[both] / fb[both] - 1.0).max()))
    finally:
        orc.set_libm(False)
    assert diff / tot <= 1.0e-4 and worst <= 2.0e-5, (diff, tot, worst)


def test_embree_quad_vertex_order_changes_nothing(orc):
    """geom_type "quad" / "grid" (the default) reach Embree as quads (v0, v1, v2, v3) = ((i,j), (i,j+1), (i+1,j+1),
    (i+1,j)), which its intersector splits into (v0, v1, v3) and (v2, v3, v1): the triangles of the "triangle"
    topology with the second one's vertices rotated.  The oracle evaluates both orders: on config 2 (all three
    algorithms), on rough terrain with tilted frames and on the degenerate shapes no horizon value and no ray count
    changes (the rotation only reorders roundings of a test whose tolerance is relative)."""
    jobs = []
    kw = cases.grid_kwargs(cases.c2_hill())
    for alg in cases.ALGS:
        jobs.append((kw, dict(dist_search=10.0, azim_num=36, ray_algorithm=alg)))
    g = cases.rough_terrain(90, 84, seed=21, offset=5, relief=1800.0, tilt_frames=True, origin=(2.6e6, 1.2e6))
    jobs.append((cases.grid_kwargs(g), dict(dist_search=3.0, azim_num=60, hori_acc=0.1, elev_ang_low_lim=-60.0)))
    yy, xx = np.mgrid[0:48, 0:52]
    terr = (100.0 * ((xx // 6) % 4) + 50.0 * ((yy // 5) % 3)).astype(np.float32)
    from horayzon_amd import synth
    x = (np.arange(52) * 30.0).astype(np.float32); y = ((47 - np.arange(48)) * 30.0).astype(np.float32)
    vn, vo = synth.planar_frames(40, 44)
    jobs.append((dict(vert_grid=synth.pack_vertices(*np.meshgrid(x, y), terr), dem_dim_0=48, dem_dim_1=52, vec_norm=vn,
                      vec_north=vo, offset_0=4, offset_1=4), dict(dist_search=2.0, azim_num=24, elev_ang_low_lim=-80.0)))
    changed = total = 0
    try:
        for kw, par in jobs:
            orc.set_quad_order(False)
            h0, _, s0 = orc.horizon_gridded(**kw, **par, return_stats=True)
            orc.set_quad_order(True)
            h1, _, s1 = orc.horizon_gridded(**kw, **par, return_stats=True)
            changed += int((h0 != h1).sum()); total += h0.size
            assert s0["rays"] == s1["rays"]
    finally:
        orc.set_quad_order(False)
    assert total > 2.0e6 and changed == 0, (changed, total)


# ---------------------------------------------------------------------------------------
# pin against the Embree reference, when a maintainer has produced the fixtures (scripts/make_embree_fixtures.py)
# ---------------------------------------------------------------------------------------
from tests import embree_pin   # noqa: E402


@pytest.mark.skipif(not os.path.exists(embree_pin.HORIZON), reason=embree_pin.MISSING)
def test_oracle_horizon_against_embree_reference(orc):
    embree_pin.compare_horizon(orc.horizon_gridded)


@pytest.mark.skipif(not os.path.exists(embree_pin.SHADOW), reason=embree_pin.MISSING)
def test_oracle_shadow_against_embree_reference(orc):
    embree_pin.compare_shadow(orc.Terrain)


def test_embree_pin_harness_is_consistent(orc):
    """The harness replays seeded configurations: they must be reproducible (same bytes twice) and the comparison code
    must accept the oracle against itself (run through a temporary fixture made from the oracle's own outputs)."""
    import tempfile
    h = embree_pin._harness()
    a, b = h.pin_cases(), h.pin_cases()
    assert [n for n, _, _ in a] == [n for n, _, _ in b] and len(a) == 15
    for (_, kw0, _), (_, kw1, _) in zip(a, b):
        assert np.array_equal(kw0["vert_grid"], kw1["vert_grid"])
    store = {}
    for name, kw, par in a[:2] + a[9:]:
        hori, azim = orc.horizon_gridded(**kw, **par)
        store["hori__" + name] = hori
    saved = embree_pin.HORIZON
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "embree_horizon.npz")
        # only the cases computed above: restrict the replay accordingly
        np.savez_compressed(path, **store)
        embree_pin.HORIZON = path
        orig = h.pin_cases
        try:
            embree_pin._harness = lambda: type("H", (), {"pin_cases": staticmethod(lambda: a[:2] + a[9:]),
                                                          "shadow_case": staticmethod(h.shadow_case)})
            rep = embree_pin.compare_horizon(orc.horizon_gridded)
        finally:
            embree_pin.HORIZON = saved
            embree_pin._harness = _orig_harness
    assert all(v["mismatch_fraction"] == 0.0 and v["max_abs"] == 0.0 for v in rep.values())


_orig_harness = embree_pin._harness


def test_triangle_test_sensitivity_variants(orc):
    """oracle/hz_oracle.c "SENSITIVITY VARIANTS": how far can Embree's FMA / rcp evaluation of the robust Pluecker test
    sit from the plain-IEEE one that is the contract?  (scripts/embree_sensitivity.py runs the full workloads; this
    keeps the machinery honest: the comparison pass does not disturb the result, the FMA / rcp variant flips at most a
    few rays per million, Moeller-Trumbore -- not watertight along shared edges -- flips orders of magnitude more on
    the grid-aligned hill.)"""
    g = cases.rough_terrain(80, 70, seed=8, offset=5, relief=700.0, tilt_frames=True, origin=(2.6e6, 1.2e6))
    kw = cases.grid_kwargs(g)
    par = dict(dist_search=3.0, azim_num=24, hori_acc=0.25, elev_ang_low_lim=-45.0)
    try:
        base, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
        orc.set_tri_compare("embree_fma_rcp")
        again, _ = orc.horizon_gridded(**kw, **par)
        n, flips = orc.tri_compare_counts()
        orc.set_tri_compare(None)
        assert np.array_equal(again, base) and n == so["rays"]
        assert flips <= 5e-6 * n + 2, (flips, n)
        orc.set_tri_mode("embree_fma_rcp")
        var, _ = orc.horizon_gridded(**kw, **par)
        orc.set_tri_mode("plain")
        assert (var != base).mean() <= 1e-4 and np.abs(var - base).max() <= np.deg2rad(0.5) * 1.01
        # the grid-aligned hill: rays along grid lines and diagonals pass exactly through shared edges
        h = cases.grid_kwargs(cases.c2_hill())
        orc.set_tri_compare("moeller_trumbore")
        orc.horizon_gridded(**h, dist_search=10.0, azim_num=8, rows=(80, 100), slab_only=True)
        n_mt, flips_mt = orc.tri_compare_counts()
        orc.set_tri_compare("embree_fma_rcp")
        orc.horizon_gridded(**h, dist_search=10.0, azim_num=8, rows=(80, 100), slab_only=True)
        n_e, flips_e = orc.tri_compare_counts()
        assert n_mt == n_e and flips_mt > 100 * max(flips_e, 1)
        # "plain_fma" -- the CPU side of the product's build-time switch -DHZ_TRI_FMA (tests/test_gpu_tri_fma.py holds the GPU
        # side to it bit for bit): fused cross / dot products only; it sits as close to the contract as the full Embree variant
        orc.set_tri_compare("plain_fma")
        again, _ = orc.horizon_gridded(**kw, **par)
        n_f, flips_f = orc.tri_compare_counts()
        orc.set_tri_compare(None)
        assert np.array_equal(again, base) and n_f == so["rays"] and flips_f <= 5e-6 * n_f + 2, (flips_f, n_f)
        orc.set_tri_mode("plain_fma")
        var, _, sv = orc.horizon_gridded(**kw, **par, return_stats=True, mode=orc.MODE_BVH)
        brute, _, sb = orc.horizon_gridded(**kw, **par, return_stats=True, mode=orc.MODE_BRUTE)
        orc.set_tri_mode("plain")
        assert np.array_equal(var, brute) and sv["rays"] == sb["rays"]      # the tree is as transparent for this test as for the contract
        assert (var != base).mean() <= 1e-4
    finally:
        orc.set_tri_compare(None)
        orc.set_tri_mode("plain")


def test_embree_fixture_script_parses_the_reference_report():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mef", os.path.join(os.path.dirname(os.path.dirname(__file__)), "scripts",
                                                                      "make_embree_fixtures.py"))
    mef = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mef)
    text = ("BVH build time: 0.731 s\nHorizon detection algorithm: guess horizon from previous azimuth direction\n"
            "Number of grid cells for which horizon is computed: 228416 \nRay tracing time: 12.5 s\n"
            "Number of rays shot: 177000000\nTotal run time: 14.1 s\n")
    rep = mef.parse_report(text)
    assert rep == {"bvh_build_s": 0.731, "ray_tracing_s": 12.5, "rays": 177000000, "cells": 228416, "total_run_s": 14.1}
    env = mef.environment()
    assert env["logical_cores"] >= 1 and "host" in env


def test_embree_fixture_script_dry_run(orc, tmp_path):
    """the one maintainer run of scripts/make_embree_fixtures.py must not be able to fail on a typo.
    --dry-run executes the script's WHOLE case list against a stand-in with the reference's signatures and stdout report
    (the CPU oracle), writes the three files, validates shapes / dtypes / layout / the timing schema bench.py reads, and the
    pin comparison code then replays them (0 mismatches by construction) -- including the worst-cell report."""
    h = embree_pin._harness()
    out = str(tmp_path / "dry")
    h.main(["--dry-run", "--out", out, "--bench-tile", "161", "--bench-rows", "4"])
    import json
    tj = json.load(open(os.path.join(out, "embree_timing.json")))
    assert "DRY-RUN" in tj["reference_version"] and tj["c3_tile"]["tile"] == 161 and tj["c3_tile"]["rays"] > 0
    assert tj["c3_tile"]["cells"] == 4 * (161 - 32) and len(tj["cases"]) == 15
    rep = embree_pin.compare_horizon(orc.horizon_gridded, path=os.path.join(out, "embree_horizon.npz"))
    assert len(rep) == 15 and all(v["mismatches"] == 0 and v["worst_cells_row_col_azim_got_ref"] == [] for v in rep.values())
    rs = embree_pin.compare_shadow(orc.Terrain, path=os.path.join(out, "embree_shadow.npz"))
    assert all(v["differing_codes"] == 0 and v["confusion_got_to_ref"] == {} for v in rs.values())
    # a perturbed fixture is reported with its fraction and its worst cells (and fails the check)
    z = dict(np.load(os.path.join(out, "embree_horizon.npz")))
    z["hori__flat"] = z["hori__flat"].copy(); z["hori__flat"][3, 4, 5] += 0.02
    np.savez_compressed(os.path.join(out, "perturbed.npz"), **z)
    rep = embree_pin.compare_horizon(orc.horizon_gridded, path=os.path.join(out, "perturbed.npz"), check=False)
    assert rep["flat"]["mismatches"] == 1 and rep["flat"]["worst_cells_row_col_azim_got_ref"][0][:3] == [3, 4, 5]
    # and the script refuses to write dry-run output into tests/golden
    with pytest.raises(SystemExit):
        h.main(["--dry-run"])
