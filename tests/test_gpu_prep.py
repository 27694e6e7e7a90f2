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


def test_topo_kernels_agree_with_each_other(hip):
    """The tiled kernel (k_topo: azimuth tables in LDS, float32 arctangent only where the tilted plane limits) and the
    fallback for very large azimuth counts (k_topo_wide: libm's float64 routines per azimuth, as the Cython code) must not
    drift apart: same fixtures, both within 1e-5 of the reference's values and within 2e-6 of each other."""
    d = np.load(os.path.join(os.path.dirname(GOLD), "svf_reference.npz"))
    T = hip.topo_param
    for n in "abc":
        a = (T.sky_view_factor(d["azim_" + n], d["hori_" + n], d["tilt_" + n]),
             T.visible_sky_fraction(d["azim_" + n], d["hori_" + n], d["tilt_" + n]),
             T.topographic_openness(d["azim_" + n], d["hori_" + n]))
        from horayzon_amd import _lib
        _lib.check(_lib.lib().hz_debug_set(b"topo_wide", 1))
        try:
            b = (T.sky_view_factor(d["azim_" + n], d["hori_" + n], d["tilt_" + n]),
                 T.visible_sky_fraction(d["azim_" + n], d["hori_" + n], d["tilt_" + n]),
                 T.topographic_openness(d["azim_" + n], d["hori_" + n]))
        finally:
            _lib.check(_lib.lib().hz_debug_set(b"topo_wide", 0))
        for x, y, key in zip(a, b, ("svf_", "vsf_", "top_")):
            assert np.abs(x - y).max() <= 2.0e-6, key
            assert np.abs(y - d[key + n]).max() <= 1.0e-5, key


def test_slope_planar(hip):
    d = np.load(GOLD)
    tp = hip.topo_param.slope_plane_meth(d["pl_x"], d["pl_y"], d["pl_z"])
    tv = hip.topo_param.slope_vector_meth(d["pl_x"], d["pl_y"], d["pl_z"])
    _close(tp, d["pl_tilt_plane"], 2e-6)
    _close(tv, d["pl_tilt_vector"], 2e-6)
    assert np.isnan(tp[0]).all() and np.isnan(tp[:, -1]).all()
    assert np.abs(np.linalg.norm(tp[1:-1, 1:-1], axis=2) - 1.0).max() < 1e-6


def test_swiss_projection_against_reference_fixture(hip):
    """wgs2swiss / swiss2wgs (transform.pyx:266-432) against values the reference itself produced
    (tests/golden/make_fixtures.py), and the round trip within the accuracy of swisstopo's approximate formulas."""
    d = np.load(os.path.join(os.path.dirname(GOLD), "swiss_reference.npz"))
    T = hip.transform
    e, n, h = T.wgs2swiss(d["lon"], d["lat"], d["h_wgs"])
    assert e.dtype == n.dtype == np.float64 and h.dtype == np.float32 and e.shape == d["lon"].shape
    assert np.abs(e - d["e"]).max() <= 1e-8 and np.abs(n - d["n"]).max() <= 1e-8      # metres, float64
    assert np.abs(h - d["h_ch"]).max() <= 5e-4                                           # float32 at ~4 km
    lon, lat, hw = T.swiss2wgs(d["e2"], d["n2"], d["h_ch2"])
    assert np.abs(lon - d["lon2"]).max() <= 1e-13 and np.abs(lat - d["lat2"]).max() <= 1e-13
    assert np.abs(hw - d["h_wgs2"]).max() <= 5e-4
    lon_b, lat_b, h_b = T.swiss2wgs(e, n, h)                          # approximate formulas: a few metres at the rim
    assert np.abs(lon_b - d["lon"]).max() <= 1e-4 and np.abs(lat_b - d["lat"]).max() <= 1e-4
    assert np.abs(h_b - d["h_wgs"]).max() <= 0.5
    with pytest.raises(ValueError):
        T.wgs2swiss(d["lon"], d["lat"], d["h_wgs"].astype(np.float64))
    with pytest.raises(ValueError):
        T.swiss2wgs(d["e2"][:5], d["n2"], d["h_ch2"])


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


def _device_chain(hip, torch, lon, lat, elev_t, off, azim_num, dist_km, rows=None):
    """lon (n1), lat (n0) float64 [degree], elev_t (n0, n1) float32 torch CUDA tensor -> horizon / SVF / shadow,
    every intermediate a torch tensor in HBM, every step a C-ABI call with device pointers."""
    import ctypes as C
    from horayzon_amd import _lib
    L = _lib.lib()
    dev = elev_t.device
    n0, n1 = elev_t.shape
    lon2 = torch.from_numpy(lon).to(dev)[None, :].expand(n0, n1).contiguous()
    lat2 = torch.from_numpy(lat).to(dev)[:, None].expand(n0, n1).contiguous()
    n = n0 * n1
    f64 = lambda: torch.empty(n, dtype=torch.float64, device=dev)
    f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    X, Y, Z = f64(), f64(), f64()
    _lib.check(L.hz_lonlat2ecef(lon2.data_ptr(), lat2.data_ptr(), elev_t.data_ptr(), n, 2, X.data_ptr(), Y.data_ptr(), Z.data_ptr(), 0))
    lon_or, lat_or = float(lon.mean()), float(lat.mean())
    xe, ye, ze = f32(n0, n1), f32(n0, n1), f32(n0, n1)
    _lib.check(L.hz_ecef2enu(X.data_ptr(), Y.data_ptr(), Z.data_ptr(), n, lon_or, lat_or, 2, xe.data_ptr(), ye.data_ptr(), ze.data_ptr(), 0))
    vert_grid = hip.auxiliary.rearrange_pad_buffer(xe, ye, ze)               # torch in -> torch (HBM) out
    sl = (slice(off, n0 - off), slice(off, n1 - off))
    in0, in1 = n0 - 2 * off, n1 - 2 * off
    lon_i, lat_i = lon2[sl].contiguous(), lat2[sl].contiguous()
    Xi, Yi, Zi = (a.view(n0, n1)[sl].contiguous() for a in (X, Y, Z))
    vn_e, vno_e, vec_norm, vec_north = f32(in0, in1, 3), f32(in0, in1, 3), f32(in0, in1, 3), f32(in0, in1, 3)
    m = in0 * in1
    _lib.check(L.hz_surf_norm(lon_i.data_ptr(), lat_i.data_ptr(), m, vn_e.data_ptr(), 0))
    _lib.check(L.hz_north_dir(Xi.data_ptr(), Yi.data_ptr(), Zi.data_ptr(), vn_e.data_ptr(), m, 2, vno_e.data_ptr(), 0))
    _lib.check(L.hz_ecef2enu_vector(vn_e.data_ptr(), m, lon_or, lat_or, 2, vec_norm.data_ptr(), 0))
    _lib.check(L.hz_ecef2enu_vector(vno_e.data_ptr(), m, lon_or, lat_or, 2, vec_north.data_ptr(), 0))
    tilt_full = f32(n0, n1, 3)
    _lib.check(L.hz_slope_plane_meth(xe.data_ptr(), ye.data_ptr(), ze.data_ptr(), n0, n1, None, 0, tilt_full.data_ptr(), 0))
    vec_tilt = tilt_full[sl].contiguous()
    scene = hip.Scene.create(vert_grid, n0, n1)                                # device vert_grid: no upload
    rb, re = rows if rows else (0, in0)
    hori = f32(re - rb, in1, azim_num)
    svf = torch.full((re - rb, in1), float("nan"), dtype=torch.float32, device=dev)
    mask = torch.ones((in0, in1), dtype=torch.uint8, device=dev)
    opts = _lib.hz_opts(); opts.top_nodes = -1; opts.regroup = -1
    opts.row_begin, opts.row_end, opts.hori_is_slab = rb, re, 1
    opts.svf, opts.vec_tilt = svf.data_ptr(), vec_tilt.data_ptr()
    st = _lib.hz_stats()
    _lib.check(L.hz_horizon_gridded_scene(scene._h, vec_norm.data_ptr(), vec_north.data_ptr(), off, off, hori.data_ptr(),
                                          in0, in1, azim_num, dist_km, 0.25, b"guess_constant", -15.0, mask.data_ptr(),
                                          0.0, 0.01, C.byref(opts), C.byref(st)))
    # shadow for one sun position on the same scene, inputs and output in HBM
    enl = torch.ones((in0, in1), dtype=torch.float32, device=dev)
    elev_i = elev_t[sl].contiguous()
    t = hip.shadow.Terrain()
    _lib.check(L.hz_terrain_initialise_scene(t._h, scene._h, off, off, vec_tilt.data_ptr(), vec_norm.data_ptr(), in0, in1,
                                             enl.data_ptr(), elev_i.data_ptr(), mask.data_ptr(), float("nan"), 89.0, 0))
    t._shape, t._scene = (in0, in1), scene
    sun = np.array([[4.0e10, 2.0e10, 1.2e10]], np.float32)
    sh = torch.full((1, in0, in1), 255, dtype=torch.uint8, device=dev)
    t.shadow_batch(sun, sh)
    torch.cuda.synchronize()
    return dict(xe=xe, ye=ye, ze=ze, vec_norm=vec_norm, vec_north=vec_north, vec_tilt=vec_tilt, hori=hori, svf=svf,
                shadow=sh[0], stats=st, scene=scene, sun=sun[0])


def test_device_resident_chain_equals_host_chain(hip, orc):
    """SURVEY 8f row 4 closed: raw lon / lat / elevation in HBM -> ENU vertices, packed vert_grid
    (hz_pack_vertices), normals, north vectors, slope, scene, horizon + fused SVF, shadow -- no host round trip.
    Bit-identical to the same steps through the NumPy mirrors (same kernels), which the other tests of this file
    pin against the reference-made fixtures."""
    torch = pytest.importorskip("torch")
    d = np.load(os.path.join(os.path.dirname(GOLD), "curved_dem_reference.npz"))
    off = int(d["offset"])
    elev_t = torch.from_numpy(np.ascontiguousarray(d["elevation"], np.float32)).to("cuda:0")
    r = _device_chain(hip, torch, d["lon"], d["lat"], elev_t, off, 18, 3.0)
    # host chain through the mirrors
    lon2, lat2 = np.meshgrid(d["lon"], d["lat"])
    T, D, P = hip.transform, hip.direction, hip.topo_param
    X, Y, Z = T.lonlat2ecef(lon2, lat2, d["elevation"], ellps="WGS84")
    tr = T.TransformerEcef2enu(lon_or=d["lon"].mean(), lat_or=d["lat"].mean(), ellps="WGS84")
    xe, ye, ze = T.ecef2enu(X, Y, Z, tr)
    sl = (slice(off, lat2.shape[0] - off), slice(off, lat2.shape[1] - off))
    vn_e = D.surf_norm(lon2[sl], lat2[sl])
    vec_norm = T.ecef2enu_vector(vn_e, tr)
    vec_north = T.ecef2enu_vector(D.north_dir(X[sl], Y[sl], Z[sl], vn_e, ellps="WGS84"), tr)
    vec_tilt = np.ascontiguousarray(P.slope_plane_meth(xe, ye, ze)[sl])
    vg = hip.auxiliary.rearrange_pad_buffer(xe, ye, ze)
    from horayzon_amd import synth
    assert np.array_equal(vg, synth.pack_vertices(xe, ye, ze)) and len(vg) % 4 == 0 and len(vg) >= 3 * xe.size + 16
    assert np.array_equal(r["xe"].cpu().numpy(), xe) and np.array_equal(r["vec_norm"].cpu().numpy(), vec_norm)
    assert np.array_equal(r["vec_north"].cpu().numpy(), vec_north)
    assert np.array_equal(r["vec_tilt"].cpu().numpy(), vec_tilt, equal_nan=True)
    n0, n1 = xe.shape
    h, azim, svf = hip.horizon.horizon_gridded(vg, n0, n1, vec_norm, vec_north, off, off, 3.0, azim_num=18,
                                               svf_vec_tilt=vec_tilt)
    assert np.array_equal(r["hori"].cpu().numpy(), h) and np.array_equal(r["svf"].cpu().numpy(), svf, equal_nan=True)
    # ... and the oracle agrees on the horizon computed from the device-prepared input
    h_o, _ = orc.horizon_gridded(vg, n0, n1, vec_norm, vec_north, off, off, 3.0, azim_num=18)
    assert np.array_equal(h, h_o)
    sh = np.empty(vec_tilt.shape[:2], np.uint8)
    tc = orc.Terrain()
    tc.initialise(vg, n0, n1, off, off, vec_tilt, vec_norm, np.ones(sh.shape, np.float32),
                  np.ascontiguousarray(d["elevation"][sl], np.float32), np.ones(sh.shape, np.uint8))
    tc.shadow(r["sun"], sh)
    assert np.array_equal(r["shadow"].cpu().numpy(), sh)


def test_device_resident_chain_on_the_c3_tile(hip, orc):
    """The same chain at config-3 size (3601 x 3601 one-arc-second tile on the WGS84 ellipsoid: the CURVED variant of
    config 3), timed: raw tile in HBM -> everything up to the scene, then a 64-row slab of horizon + SVF and one shadow
    mask.  Two rows of the slab are checked against the oracle, bit for bit."""
    torch = pytest.importorskip("torch")
    import json
    import time
    from horayzon_amd import synth
    n, off = 3601, 16
    lon = 8.0 + np.arange(n) / 3600.0
    lat = 47.0 - np.arange(n) / 3600.0
    elev_t = torch.from_numpy(synth.fractal_elevation(n, n)).to("cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = _device_chain(hip, torch, lon, lat, elev_t, off, 360, 50.0, rows=(1760, 1824))
    wall = time.perf_counter() - t0
    st = r["stats"]
    assert st.num_cells == 64 * (n - 2 * off) and st.guard_events == 0
    svf = r["svf"]
    assert bool(torch.isfinite(svf).all().item()) and 0.2 < float(svf.min().item()) and float(svf.max().item()) <= 1.0 + 1e-5
    assert int(r["shadow"].max().item()) <= 2 and float((r["shadow"] == 0).float().mean().item()) > 0.3
    # curved-earth frames: normals tilt away from the ENU z axis towards the tile edges
    vn = r["vec_norm"]
    assert float(vn[..., 2].min().item()) < 0.99999 and abs(float((vn ** 2).sum(-1).mean().item()) - 1.0) < 1e-6
    # the oracle on the device-prepared curved geometry (non axis-aligned frames at full size)
    xe, ye, ze = (r[k].cpu().numpy() for k in ("xe", "ye", "ze"))
    ref, _, so = orc.horizon_gridded(synth.pack_vertices(xe, ye, ze), n, n, r["vec_norm"].cpu().numpy(),
                                     r["vec_north"].cpu().numpy(), off, off, dist_search=50.0, azim_num=360,
                                     rows=(1760, 1762), slab_only=True, return_stats=True)
    assert np.array_equal(r["hori"][:2].cpu().numpy(), ref) and so["guards"] == 0
    print(json.dumps({"chain_wall_s": wall, "bvh_build_s": r["scene"].stats["t_bvh_s"], "slab_kernel_s": st.t_kernel_s,
                      "slab_cells_per_s": st.num_cells / st.t_kernel_s, "rays_per_cell_azimuth": st.num_rays / (st.num_cells * 360.0)}))
