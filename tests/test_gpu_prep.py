"""GPU tests of the steps next to the path against golden vectors that the
REFERENCE's own Cython modules produced (tests/golden/prep_reference.npz, make_fixtures.py).
The reference builds those modules with -ffast-math, so the bar is rounding-level agreement."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prep_reference.npz")


def _close(a, b, tol):
    assert a.shape == b.shape and a.dtype == b.dtype
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.abs(a[m].astype(np.float64) - b[m].astype(np.float64)).max() <= tol


def test_visible_sky_fraction_and_openness(hip):
    d = np.load(os.path.join(os.path.dirname(GOLD), "svf_reference.npz"))
    for n in "abc":
        vsf = hip.topo_param.visible_sky_fraction(d["azim_" + n], d["hori_" + n], d["tilt_" + n])
        top = hip.topo_param.topographic_openness(d["azim_" + n], d["hori_" + n])
        svf = hip.topo_param.sky_view_factor(d["azim_" + n], d["hori_" + n], d["tilt_" + n])
        assert np.abs(vsf - d["vsf_" + n]).max() <= 1e-5
        assert np.abs(top - d["top_" + n]).max() <= 1e-5
        assert np.abs(svf - d["svf_" + n]).max() <= 1e-5


def test_slope_planar(hip):
    d = np.load(GOLD)
    tp = hip.topo_param.slope_plane_meth(d["pl_x"], d["pl_y"], d["pl_z"])
    tv = hip.topo_param.slope_vector_meth(d["pl_x"], d["pl_y"], d["pl_z"])
    _close(tp, d["pl_tilt_plane"], 2e-6)
    _close(tv, d["pl_tilt_vector"], 2e-6)
    assert np.isnan(tp[0]).all() and np.isnan(tp[:, -1]).all()
    assert np.abs(np.linalg.norm(tp[1:-1, 1:-1], axis=2) - 1.0).max() < 1e-6


@pytest.mark.parametrize("ellps", ("sphere", "GRS80", "WGS84"))
def test_input_preparation_chain(hip, ellps):
    d = np.load(GOLD)
    k = ellps + "_"
    T, D = hip.transform, hip.direction
    X, Y, Z = T.lonlat2ecef(d[k + "lon"], d[k + "lat"], d[k + "h"], ellps=ellps)
    for got, ref in ((X, d[k + "X"]), (Y, d[k + "Y"]), (Z, d[k + "Z"])):
        assert np.abs(got - ref).max() <= 1e-8 * 6.4e6            # float64, ~1e-15 relative
    org = d[k + "origin"]
    tr = T.TransformerEcef2enu(lon_or=org[0], lat_or=org[1], ellps=ellps)
    assert np.allclose([tr.x_ecef_or, tr.y_ecef_or, tr.z_ecef_or], org[2:], rtol=1e-15, atol=1e-8)
    xe, ye, ze = T.ecef2enu(d[k + "X"], d[k + "Y"], d[k + "Z"], tr)
    for got, ref in ((xe, d[k + "x_enu"]), (ye, d[k + "y_enu"]), (ze, d[k + "z_enu"])):
        _close(got, ref, 4e-3 * 2.0 ** -10)                        # float32 ulp at 1e4..1e5 m
    vn = D.surf_norm(d[k + "lon"], d[k + "lat"])
    _close(vn, d[k + "norm_ecef"], 1e-7)
    vno = D.north_dir(d[k + "X"], d[k + "Y"], d[k + "Z"], d[k + "norm_ecef"], ellps=ellps)
    _close(vno, d[k + "north_ecef"], 1e-7)
    _close(T.ecef2enu_vector(d[k + "norm_ecef"], tr), d[k + "norm_enu"], 1e-7)
    _close(T.ecef2enu_vector(d[k + "north_ecef"], tr), d[k + "north_enu"], 1e-7)
    rot = T.rotation_matrix_glob2loc(d[k + "north_enu"][1:-1, 1:-1], d[k + "norm_enu"][1:-1, 1:-1])
    _close(rot, d[k + "rot"], 1e-7)
    # slope on the curved (ENU) geometry with rotation matrices
    P = hip.topo_param
    _close(P.slope_plane_meth(d[k + "x_enu"], d[k + "y_enu"], d[k + "z_enu"], rot_mat=d[k + "rot"]),
           d[k + "tilt_plane"], 5e-6)
    _close(P.slope_plane_meth(d[k + "x_enu"], d[k + "y_enu"], d[k + "z_enu"], rot_mat=d[k + "rot"], output_rot=True),
           d[k + "tilt_plane_rot"], 5e-6)
    _close(P.slope_vector_meth(d[k + "x_enu"], d[k + "y_enu"], d[k + "z_enu"]), d[k + "tilt_vector"], 2e-6)
    _close(P.slope_vector_meth(d[k + "x_enu"], d[k + "y_enu"], d[k + "z_enu"], rot_mat=d[k + "rot"], output_rot=True),
           d[k + "tilt_vector_rot"], 2e-6)


def test_prepared_input_feeds_the_horizon_path(hip, orc):
    """End to end: lon/lat/elevation -> (device) ENU vertices, normals, north vectors -> horizon;
    identical to the horizon computed from the reference-prepared fixture inputs."""
    from horayzon_amd import synth
    d = np.load(os.path.join(os.path.dirname(GOLD), "curved_dem_reference.npz"))
    lon2, lat2 = np.meshgrid(d["lon"], d["lat"])
    T, D = hip.transform, hip.direction
    X, Y, Z = T.lonlat2ecef(lon2, lat2, d["elevation"], ellps="WGS84")
    tr = T.TransformerEcef2enu(lon_or=d["lon"].mean(), lat_or=d["lat"].mean(), ellps="WGS84")
    xe, ye, ze = T.ecef2enu(X, Y, Z, tr)
    off = int(d["offset"])
    sl = (slice(off, lat2.shape[0] - off), slice(off, lat2.shape[1] - off))
    vn_e = D.surf_norm(lon2[sl], lat2[sl])
    vno_e = D.north_dir(X[sl], Y[sl], Z[sl], vn_e, ellps="WGS84")
    vec_norm = T.ecef2enu_vector(vn_e, tr)
    vec_north = T.ecef2enu_vector(vno_e, tr)
    assert np.abs(xe - d["x_enu"]).max() < 5e-3 and np.abs(vec_norm - d["vec_norm"]).max() < 1e-6
    n0, n1 = xe.shape
    h_mine, _ = hip.horizon.horizon_gridded(synth.pack_vertices(xe, ye, ze), n0, n1, vec_norm, vec_north, off, off,
                                            3.0, azim_num=18, elev_ang_low_lim=-89.98, ray_algorithm="binary_search")
    h_ref, _ = hip.horizon.horizon_gridded(synth.pack_vertices(d["x_enu"], d["y_enu"], d["z_enu"]), n0, n1,
                                           d["vec_norm"], d["vec_north"], off, off, 3.0, azim_num=18,
                                           elev_ang_low_lim=-89.98, ray_algorithm="binary_search")
    # inputs agree to float32 rounding; the horizon may move by at most one search bracket somewhere
    assert (h_mine != h_ref).mean() < 0.02 and np.abs(h_mine - h_ref).max() <= 2.5 * np.deg2rad(0.25)


@pytest.mark.parametrize("n", (1, 63, 1024, 1025, 70001, 3_000_017))
def test_build_primitives_sort_and_scan(hip, n):
    """The hand-written radix sort (stable, by key) and exclusive scan of the LBVH build."""
    from horayzon_amd import _lib
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    keys[::3] = keys[0]                                   # many duplicates: stability matters
    if n > 100:
        keys[5:60] &= np.uint32(0xff)                     # and keys that differ in one digit only
    vals = np.arange(n, dtype=np.uint32)
    k, v = keys.copy(), vals.copy()
    _lib.check(_lib.lib().hz_debug_sort_pairs(k.ctypes.data, v.ctypes.data, n, 0))
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])
    x = rng.integers(0, 5, n, dtype=np.uint32)
    out = np.empty(n, np.uint32)
    _lib.check(_lib.lib().hz_debug_exclusive_scan(x.ctypes.data, out.ctypes.data, n, 0))
    ref = np.concatenate([[0], np.cumsum(x[:-1], dtype=np.uint64)]).astype(np.uint32)
    assert np.array_equal(out, ref)
