"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical
seeded inputs.  Hit decisions are float32-deterministic on both sides, so horizon
arrays, ray counts and shadow codes must be IDENTICAL (not merely close); the
north-star tolerance (1e-4 rad / 1e-5 SVF) is the outer bound asserted as well."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

ALGS = ("guess_constant", "binary_search", "discrete_sampling")


def _compare(hip, orc, kw, **params):
    h_gpu, a_gpu = hip.horizon.horizon_gridded(**kw, **params)
    st = hip.horizon.last_stats
    h_cpu, a_cpu, so = orc.horizon_gridded(**kw, **params, return_stats=True)
    assert np.array_equal(a_gpu, a_cpu)
    assert not np.isnan(h_gpu).any()
    assert np.abs(h_gpu - h_cpu).max() <= 1.0e-4          # north-star bound [rad]
    assert np.array_equal(h_gpu, h_cpu)                    # and in fact bit-identical
    assert st["num_rays"] == so["rays"]
    assert st["guard_events"] == so["guards"]
    return h_gpu, st


@pytest.mark.parametrize("alg", ALGS)
def test_c2_gaussian_hill(hip, orc, alg):
    """BASELINE config 2: 200 x 200 Gaussian hill, 36 azimuths, 1 x MI355X."""
    g = cases.c2_hill()
    h, st = _compare(hip, orc, cases.grid_kwargs(g), dist_search=10.0, azim_num=36, ray_algorithm=alg)
    assert st["guard_events"] == 0
    assert st["num_cells"] == 180 * 180


def test_c2_guard_events(hip, orc):
    """1500 m hill: searches hit the cases where the reference never terminates; both
    sides stop at the clamped index and count the same events."""
    g = cases.c2_hill(height=1500.0)
    h, st = _compare(hip, orc, cases.grid_kwargs(g), dist_search=10.0, azim_num=36)
    assert st["guard_events"] > 0


@pytest.mark.parametrize("alg", ALGS)
def test_rough_tilted_frames(hip, orc, alg):
    """Ragged size (not a multiple of the 16 x 16 tile), rotated per-cell frames."""
    g = cases.rough_terrain(93, 117, seed=7, offset=6, tilt_frames=True)
    _compare(hip, orc, cases.grid_kwargs(g), dist_search=2.0, azim_num=24, ray_algorithm=alg,
             elev_ang_low_lim=-60.0)


@pytest.mark.parametrize("alg", ALGS)
def test_hit_cache_is_transparent(hip, alg):
    """The hit cache (blocked rays first walk the subtree above the leaf that blocked the
    cell's previous ray) only changes the work done, never a result."""
    g = cases.rough_terrain(150, 170, seed=3, offset=10, tilt_frames=True)
    kw = cases.grid_kwargs(g)
    par = dict(dist_search=3.0, azim_num=72, ray_algorithm=alg, elev_ang_low_lim=-60.0, count_work=True)
    h_on, _ = hip.horizon.horizon_gridded(**kw, **par, _hit_cache=1)
    st_on = dict(hip.horizon.last_stats)
    h_off, _ = hip.horizon.horizon_gridded(**kw, **par, _hit_cache=0)
    st_off = dict(hip.horizon.last_stats)
    assert np.array_equal(h_on, h_off)
    assert st_on["num_rays"] == st_off["num_rays"]
    assert st_on["nodes_visited"] != st_off["nodes_visited"]      # the switch does something


def test_lds_nodelet_variant_is_transparent(hip):
    """opts.top_nodes > 0 selects the kernel variant that serves the top of the tree from LDS."""
    g = cases.rough_terrain(150, 170, seed=3, offset=10, tilt_frames=True)
    kw = cases.grid_kwargs(g)
    par = dict(dist_search=3.0, azim_num=72, elev_ang_low_lim=-60.0)
    h0, _ = hip.horizon.horizon_gridded(**kw, **par)
    rays = hip.horizon.last_stats["num_rays"]
    for top in (1, 21, 300, 100000):
        h1, _ = hip.horizon.horizon_gridded(**kw, **par, _top_nodes=top)
        assert np.array_equal(h0, h1)
        assert hip.horizon.last_stats["num_rays"] == rays


def test_large_coordinates(hip, orc):
    """Swiss-grid like offsets (7e5, 2e5): float32 ulp 0.06 m, the AABB padding must hold."""
    g = cases.rough_terrain(80, 90, seed=11, dx=25.0, dy=25.0, offset=5, origin=(668000.0, 172000.0))
    _compare(hip, orc, cases.grid_kwargs(g), dist_search=1.5, azim_num=30, elev_ang_low_lim=-70.0)


def test_curved_dem_reference_fixture(hip, orc):
    """ENU geometry and per-cell frames produced by the reference's own transform/direction
    modules (tests/golden/curved_dem_reference.npz)."""
    import os
    from horayzon_amd import synth
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "curved_dem_reference.npz"))
    off = int(d["offset"])
    n0, n1 = d["x_enu"].shape
    kw = dict(vert_grid=synth.pack_vertices(d["x_enu"], d["y_enu"], d["z_enu"]), dem_dim_0=n0, dem_dim_1=n1,
              vec_norm=d["vec_norm"], vec_north=d["vec_north"], offset_0=off, offset_1=off)
    for alg in ALGS:
        _compare(hip, orc, kw, dist_search=3.0, azim_num=18, elev_ang_low_lim=-89.98, ray_algorithm=alg)


def test_mask_and_fill(hip, orc):
    g = cases.rough_terrain(70, 70, seed=3, offset=5)
    kw = cases.grid_kwargs(g)
    rng = np.random.default_rng(5)
    mask = (rng.random(kw["vec_norm"].shape[:2]) > 0.4).astype(np.uint8)
    mask[0, 0] = 2      # only == 1 is computed (horizon_comp.cpp:750)
    h, st = _compare(hip, orc, kw, dist_search=1.5, azim_num=16, mask=mask, hori_fill=-1.25,
                     elev_ang_low_lim=-60.0)
    assert np.all(h[mask != 1] == np.float32(-1.25))
    assert st["num_cells"] == int((mask == 1).sum())


def test_outer_tin(hip, orc):
    """Simplified outer domain as a second geometry (horizon_comp.cpp:199-218)."""
    g = cases.rough_terrain(64, 64, seed=9, offset=4)
    vs, nvs, ts, nts = cases.outer_tin(g)
    kw = cases.grid_kwargs(g)
    h1, _ = _compare(hip, orc, kw, dist_search=8.0, azim_num=20, vert_simp=vs, num_vert_simp=nvs,
                     tri_ind_simp=ts, num_tri_simp=nts, elev_ang_low_lim=-30.0, ray_algorithm="binary_search")
    h0, _ = hip.horizon.horizon_gridded(**kw, dist_search=8.0, azim_num=20, elev_ang_low_lim=-30.0,
                                        ray_algorithm="binary_search")
    acc = np.deg2rad(0.25)
    assert (h1 >= h0 - 2 * acc).all() and (h1 > h0 + 2 * acc).any()   # the ring only raises horizons
    _compare(hip, orc, kw, dist_search=8.0, azim_num=20, vert_simp=vs, num_vert_simp=nvs,
             tri_ind_simp=ts, num_tri_simp=nts, elev_ang_low_lim=-30.0)


def test_other_parameters(hip, orc):
    g = cases.rough_terrain(60, 75, seed=21, offset=3)
    kw = cases.grid_kwargs(g)
    _compare(hip, orc, kw, dist_search=1.0, azim_num=7, hori_acc=0.1, elev_ang_low_lim=-89.98,
             ray_org_elev=0.5, ray_algorithm="binary_search", geom_type="triangle")
    _compare(hip, orc, kw, dist_search=0.4, azim_num=45, hori_acc=1.0, elev_ang_low_lim=-40.0,
             geom_type="quad")


def test_odd_parameters(hip, orc):
    """Degenerate / extreme settings: coarse accuracy (few table entries, clamped steps), one azimuth,
    tiny search distance, everything masked, top-of-table guards."""
    g = cases.rough_terrain(50, 44, seed=27, offset=2, relief=900.0)
    kw = cases.grid_kwargs(g)
    for alg in ALGS:
        _compare(hip, orc, kw, dist_search=1.0, azim_num=5, hori_acc=10.0, elev_ang_low_lim=-30.0, ray_algorithm=alg)
        _compare(hip, orc, kw, dist_search=0.02, azim_num=1, hori_acc=0.5, elev_ang_low_lim=-89.98, ray_algorithm=alg)
        _compare(hip, orc, kw, dist_search=3.0, azim_num=9, hori_acc=2.5, elev_ang_low_lim=80.0, ray_algorithm=alg)
    mask = np.zeros(kw["vec_norm"].shape[:2], np.uint8)
    h, st = _compare(hip, orc, kw, dist_search=1.0, azim_num=8, mask=mask, hori_fill=0.5)
    assert st["num_rays"] == 0 and st["num_cells"] == 0 and np.all(h == np.float32(0.5))


def test_terrain_reinitialise_and_near_sun(hip, orc):
    g1 = cases.rough_terrain(48, 52, seed=61, offset=4, relief=600.0)
    g2 = cases.rough_terrain(40, 36, seed=62, offset=3, relief=300.0)
    tg, tc = hip.shadow.Terrain(), orc.Terrain()
    for g, (n0, n1, off) in ((g1, (48, 52, 4)), (g2, (40, 36, 3)), (g1, (48, 52, 4))):
        vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
        tg.initialise(g["vert_grid"], n0, n1, off, off, vec_tilt, vec_norm, enl, elev, mask, ang_max=85.0)
        tc.initialise(g["vert_grid"], n0, n1, off, off, vec_tilt, vec_norm, enl, elev, mask, ang_max=85.0)
        # a "sun" 300 m above the middle of the DEM: per-cell directions differ strongly
        sun = np.array([g["x"].mean(), g["y"].mean(), g["z"].max() + 300.0], np.float32)
        a = np.empty(mask.shape, np.uint8); b = a.copy()
        tg.shadow(sun, a); tc.shadow(sun, b)
        assert np.array_equal(a, b) and tg.last_stats["num_rays"] == tc.rays
        fa = np.empty(mask.shape, np.float32); fb = fa.copy()
        tg.sw_dir_cor(sun, fa); tc.sw_dir_cor(sun, fb)
        assert np.array_equal(fa, fb)


def test_tiny_grids(hip, orc):
    for n0, n1 in ((2, 2), (3, 2), (3, 5)):
        g = cases.rough_terrain(n0, n1, seed=n0 * 10 + n1, offset=0, relief=20.0)
        _compare(hip, orc, cases.grid_kwargs(g), dist_search=1.0, azim_num=8, elev_ang_low_lim=-89.98)


def _dem(z, dx=30.0, dy=30.0, offset=2):
    """grid kwargs for an explicit elevation array (planar frames)."""
    from horayzon_amd import synth
    n0, n1 = z.shape
    x = (np.arange(n1) * dx).astype(np.float32)
    y = ((n0 - 1 - np.arange(n0)) * dy).astype(np.float32)
    xx, yy = np.meshgrid(x, y)
    vec_norm, vec_north = synth.planar_frames(n0 - 2 * offset, n1 - 2 * offset)
    return dict(vert_grid=synth.pack_vertices(xx, yy, z.astype(np.float32)), dem_dim_0=n0, dem_dim_1=n1,
                vec_norm=vec_norm, vec_north=vec_north, offset_0=offset, offset_1=offset)


def test_degenerate_terrain_shapes(hip, orc):
    """Boxes without extent and extreme aspect ratios: a perfectly flat DEM (zero z extent in every
    node), a terraced DEM (vertical-walled steps: many coplanar, axis-parallel faces and equal Morton
    neighbours in z), a single raised cell, and a 3 x 400 strip."""
    flat = np.full((40, 44), 250.0)
    _compare(hip, orc, _dem(flat), dist_search=2.0, azim_num=16, elev_ang_low_lim=-30.0)
    yy, xx = np.mgrid[0:48, 0:52]
    terraces = 100.0 * ((xx // 6) % 4) + 50.0 * ((yy // 5) % 3)
    _compare(hip, orc, _dem(terraces), dist_search=2.0, azim_num=24, elev_ang_low_lim=-80.0)
    spike = np.zeros((33, 35)); spike[16, 17] = 500.0
    _compare(hip, orc, _dem(spike), dist_search=2.0, azim_num=32, elev_ang_low_lim=-30.0)
    rng = np.random.default_rng(5)
    strip = 200.0 * rng.random((3, 400))
    _compare(hip, orc, _dem(strip, offset=0), dist_search=20.0, azim_num=12, elev_ang_low_lim=-89.98)
    strip_t = np.ascontiguousarray(strip.T)
    _compare(hip, orc, _dem(strip_t, offset=0), dist_search=20.0, azim_num=12, elev_ang_low_lim=-89.98)


def test_row_slab(hip, orc):
    """opts.row_begin/row_end: the multi-GPU sharding unit."""
    g = cases.rough_terrain(70, 66, seed=13, offset=3)
    kw = cases.grid_kwargs(g)
    full, _ = hip.horizon.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0)
    part, _ = hip.horizon.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0,
                                          rows=(17, 40))
    assert np.array_equal(part[17:40], full[17:40])
    assert np.isnan(part[:17]).all() and np.isnan(part[40:]).all()


def test_host_output_that_is_already_page_locked(hip):
    """The library page-locks a host result array while it runs (HostPinner).  An array the caller has pinned already
    (torch pinned memory) cannot be registered again: the copies must simply go ahead."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from horayzon_amd import _lib
    g = cases.rough_terrain(70, 66, seed=13, offset=3)
    kw = cases.grid_kwargs(g)
    full, _ = hip.horizon.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0)
    sc = hip.Scene.create(kw["vert_grid"], 70, 66)
    pinned = torch.empty(full.shape, dtype=torch.float32, pin_memory=True)
    pinned.fill_(float("nan"))
    out = pinned.numpy()
    m = np.ones(full.shape[:2], np.uint8)
    for chunk in (0, 9):
        o = _lib.hz_opts(); o.chunk_rows = chunk
        out[:] = np.nan
        _lib.check(_lib.lib().hz_horizon_gridded_scene(sc._h, kw["vec_norm"].ctypes.data, kw["vec_north"].ctypes.data, 3, 3, out.ctypes.data,
                                                       full.shape[0], full.shape[1], 12, 1.0, 0.25, b"guess_constant", -60.0, m.ctypes.data,
                                                       0.0, 0.01, C.byref(o), None))
        assert np.array_equal(out, full)


def test_concurrent_calls_on_one_scene(hip):
    """The C ABI takes `const hz_scene *` and ctypes releases the GIL: two host threads may call on the same scene at
    once.  The scene-owned certificate scratch and the shared stream are protected by the scene's run mutex (the calls
    run one after the other); every call must return what it returns alone."""
    import threading
    g = cases.rough_terrain(90, 84, seed=41, offset=4, relief=900.0)
    kw = cases.grid_kwargs(g)
    sc = hip.Scene.create(kw["vert_grid"], 90, 84)
    pars = [dict(dist_search=2.0, azim_num=36), dict(dist_search=1.0, azim_num=60, hori_acc=0.5),
            dict(dist_search=3.0, azim_num=24, ray_algorithm="binary_search"), dict(dist_search=2.0, azim_num=90, rows=(10, 60))]
    alone = [hip.horizon.horizon_gridded(**kw, **p, scene=sc)[0] for p in pars]
    got, errs = [None] * len(pars), []

    def work(i):
        try:
            for _ in range(3):
                got[i] = hip.horizon.horizon_gridded(**kw, **pars[i], scene=sc)[0]
        except Exception as exc:
            errs.append(exc)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(pars))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for a, b in zip(alone, got):
        assert np.array_equal(a, b, equal_nan=True)


def test_row_slab_c_abi_semantics(hip):
    """C ABI (include/horayzon_hip.h): {0, 0} = whole domain, row_end = -1 = dim_in_0, negative / out-of-range /
    reversed slabs are rejected (not clamped), begin == end is an empty slab that succeeds and writes nothing; with
    opts.inputs_are_slab the per-cell inputs hold only the slab's rows."""
    import ctypes as C
    from horayzon_amd import _lib
    L = _lib.lib()
    g = cases.rough_terrain(40, 37, seed=21, offset=2)
    kw = cases.grid_kwargs(g)
    in0, in1 = kw["vec_norm"].shape[:2]
    A = 8
    sc = hip.Scene.create(kw["vert_grid"], 40, 37)
    mask = np.ones((in0, in1), np.uint8); mask[5:9, 3:20] = 0
    tilt = np.zeros((in0, in1, 3), np.float32); tilt[..., 2] = 1.0

    def call(rb, re, norm=kw["vec_norm"], north=kw["vec_north"], m=mask, t=tilt, slab_in=0, slab_out=0):
        rows = in0 if slab_out == 0 else max(re - rb, 1)
        hori = np.full((rows, in1, A), np.nan, np.float32)
        svf = np.full((rows, in1), np.nan, np.float32)
        o = _lib.hz_opts(); o.row_begin, o.row_end = rb, re
        o.inputs_are_slab, o.hori_is_slab = slab_in, slab_out
        o.svf, o.vec_tilt = svf.ctypes.data, t.ctypes.data
        st = _lib.hz_stats()
        rc = L.hz_horizon_gridded_scene(sc._h, norm.ctypes.data, north.ctypes.data, 2, 2, hori.ctypes.data, in0, in1, A,
                                        2.0, 1.0, b"guess_constant", -30.0, m.ctypes.data, -1.0, 0.01, C.byref(o), C.byref(st))
        return rc, hori, svf, st

    rc, full, svf_full, st = call(0, 0)
    assert rc == 0 and not np.isnan(full).any() and st.num_cells == int(mask.sum())
    rc, h, _, _ = call(0, -1)
    assert rc == 0 and np.array_equal(h, full)
    rc, h, _, _ = call(7, -1)
    assert rc == 0 and np.array_equal(h[7:], full[7:]) and np.isnan(h[:7]).all()
    for rb, re in ((-1, 5), (3, -2), (0, in0 + 1), (9, 4), (-3, -1)):
        rc, h, _, _ = call(rb, re)
        assert rc == 1 and b"row slab" in L.hz_last_error() and np.isnan(h).all(), (rb, re)
    rc, h, s, st = call(6, 6)                                   # empty slab: a rank without rows
    assert rc == 0 and np.isnan(h).all() and np.isnan(s).all() and st.num_cells == 0 and st.num_rays == 0
    # slab-local inputs (and outputs): the caller holds rows [11, 29) only
    rb, re = 11, 29
    cut = lambda a: np.ascontiguousarray(a[rb:re])
    rc, h, s, st = call(rb, re, cut(kw["vec_norm"]), cut(kw["vec_north"]), cut(mask), cut(tilt), slab_in=1, slab_out=1)
    assert rc == 0 and np.array_equal(h, full[rb:re]) and np.array_equal(s, svf_full[rb:re])
    assert st.num_cells == int(mask[rb:re].sum())


@pytest.mark.parametrize("chunk", (1, 7, 16, 1000))
def test_streamed_host_output(hip, chunk):
    """Host `hori`: chunks of rows are double buffered on the device and copied out while the next
    chunk is traced; any chunking (incl. ragged last chunk, one chunk, a row slab) gives the same array."""
    g = cases.rough_terrain(70, 66, seed=13, offset=3)
    kw = cases.grid_kwargs(g)
    par = dict(dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0)
    full, _ = hip.horizon.horizon_gridded(**kw, **par)
    rays = hip.horizon.last_stats["num_rays"]
    got, _ = hip.horizon.horizon_gridded(**kw, **par, _chunk_rows=chunk)
    assert np.array_equal(got, full)
    assert hip.horizon.last_stats["num_rays"] == rays
    part, _ = hip.horizon.horizon_gridded(**kw, **par, rows=(5, 41), _chunk_rows=chunk)
    assert np.array_equal(part[5:41], full[5:41])
    assert np.isnan(part[:5]).all() and np.isnan(part[41:]).all()
    vec_tilt = np.zeros(full.shape[:2] + (3,), np.float32); vec_tilt[..., 2] = 1.0
    # the result array is page-locked chunk by chunk behind the scenes (HostPinner); with that switched off the copies
    # are staged pageable ones: same bytes
    import ctypes as C
    from horayzon_amd import _lib
    sc = hip.Scene.create(kw["vert_grid"], kw["dem_dim_0"], kw["dem_dim_1"])
    plain = np.full(full.shape, np.nan, np.float32)
    o = _lib.hz_opts(); o.chunk_rows = chunk; o.no_host_pin = 1
    m = np.ones(full.shape[:2], np.uint8)
    _lib.check(_lib.lib().hz_horizon_gridded_scene(sc._h, kw["vec_norm"].ctypes.data, kw["vec_north"].ctypes.data, kw["offset_0"], kw["offset_1"],
                                                   plain.ctypes.data, full.shape[0], full.shape[1], 12, 1.0, 0.25, b"guess_constant", -60.0,
                                                   m.ctypes.data, 0.0, 0.01, C.byref(o), None))
    assert np.array_equal(plain, full)
    h2, _, svf = hip.horizon.horizon_gridded(**kw, **par, svf_vec_tilt=vec_tilt, _chunk_rows=chunk)
    h1, _, svf1 = hip.horizon.horizon_gridded(**kw, **par, svf_vec_tilt=vec_tilt)
    assert np.array_equal(h2, full) and np.array_equal(svf, svf1)


@pytest.mark.parametrize("chunk", (0, 5, 16))
def test_svf_only_matches_full_path(hip, chunk):
    """svf_only: the horizon lives only in a bounded device buffer (chunks of rows); same SVF, no hori."""
    g = cases.rough_terrain(75, 88, seed=17, offset=5)
    kw = cases.grid_kwargs(g)
    vec_tilt, *_ = cases.terrain_inputs(g)
    par = dict(dist_search=2.0, azim_num=36, elev_ang_low_lim=-40.0, svf_vec_tilt=vec_tilt)
    h, a, svf = hip.horizon.horizon_gridded(**kw, **par)
    rays = hip.horizon.last_stats["num_rays"]
    none, a2, svf2 = hip.horizon.horizon_gridded(**kw, **par, svf_only=True, _chunk_rows=chunk)
    assert none is None and np.array_equal(a, a2) and np.array_equal(svf, svf2)
    assert hip.horizon.last_stats["num_rays"] == rays
    _, _, svf3 = hip.horizon.horizon_gridded(**kw, **par, svf_only=True, devices=[0, 0], _chunk_rows=chunk)
    assert np.array_equal(svf, svf3)
    with pytest.raises(ValueError, match="svf_only"):
        hip.horizon.horizon_gridded(**kw, dist_search=2.0, azim_num=36, svf_only=True)


def test_traversal_stack_is_one_entry_per_level(hip, orc):
    """Siblings are contiguous in the breadth-first node numbering, so the traversal keeps one stack entry per tree
    level (block + mask of the children still to visit): `height` LDS entries can never overflow -- there is no
    retry machinery.  Deep, irregular trees (a long thin DEM, an outer TIN: mixed node / leaf children get wrapper
    nodes) must give the oracle's result."""
    g = cases.rough_terrain(140, 150, seed=33, offset=8, relief=1200.0)
    kw = cases.grid_kwargs(g)
    par = dict(dist_search=4.0, azim_num=48, elev_ang_low_lim=-70.0)
    ref, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    sc = hip.Scene.create(kw["vert_grid"], 140, 150)
    assert 7 <= sc.stats["bvh_height"] <= 10                  # ~ log4(140 x 150 quads) + 1
    h, _ = hip.horizon.horizon_gridded(**kw, **par, scene=sc)
    assert np.array_equal(h, ref) and hip.horizon.last_stats["num_rays"] == so["rays"]
    # The default launch uses the fast discipline (one LDS entry per pending sibling, fewest instructions) with the
    # entries that fit; a wave that runs out is detected and that launch repeated with the level stack.  Forced here
    # with tiny fast stacks; `_level_stack=True` uses the level stack from the start.
    # A few overflowing tiles are repeated one by one (`stack_redo_blocks`), many repeat the whole launch.
    seen, redo, redo_left = set(), {}, {}
    for cap in (0, -3, -4, -7, -9, -10, -11, -12, -13, -14, -30, 1):
        h2, _ = hip.horizon.horizon_gridded(**kw, **par, _level_stack=cap)
        st = dict(hip.horizon.last_stats)
        assert np.array_equal(h2, ref) and st["num_rays"] == so["rays"] and st["num_cells"] == ref.shape[0] * ref.shape[1], cap
        seen.add((cap, st["stack_fallbacks"]))
        redo[cap] = st["stack_redo_blocks"]
        redo_left[cap] = st["left_redo_groups"]
    assert (0, 0) in seen and (-3, 1) in seen and (-30, 0) in seen and (1, 0) in seen
    assert redo[-3] == 0 and redo[0] == 0                   # all tiles overflow with 3 entries: one full repeat
    assert any(0 < v <= 85 for v in redo.values()), redo    # ... and some cap leaves only a few of the 342 blocks
    # (round 6: the follow-up launch of the handed-over cells runs the fast stack too; a group of 64 that overflows is repeated)
    if not hip.horizon.schedule_overrides:       # (with several hand-over levels the follow-up launches run one entry per level)
        assert any(v > 0 for v in redo_left.values()), redo_left
    # ... and a scene remembers: after one overflow its launches go straight to the level stack
    hip.horizon.horizon_gridded(**kw, **par, scene=sc, _level_stack=-3)
    assert hip.horizon.last_stats["stack_fallbacks"] == 1
    hip.horizon.horizon_gridded(**kw, **par, scene=sc, _level_stack=-3)
    assert hip.horizon.last_stats["stack_fallbacks"] == 0
    # a 3 x 1500 strip: a very unbalanced quadtree over (x, y)
    rng = np.random.default_rng(9)
    strip = cases.rough_terrain(3, 1500, seed=4, offset=0, relief=300.0)
    kws = cases.grid_kwargs(strip)
    hs, _ = hip.horizon.horizon_gridded(**kws, dist_search=8.0, azim_num=12, elev_ang_low_lim=-89.98)
    rs, _ = orc.horizon_gridded(**kws, dist_search=8.0, azim_num=12, elev_ang_low_lim=-89.98)
    assert np.array_equal(hs, rs)
    # grid + TIN triangles interleave in Morton order: nodes with both leaf and internal children
    vs, nvs, ts, nts = cases.outer_tin(g, margin=200.0, zval=1500.0)
    ht, _ = hip.horizon.horizon_gridded(**kw, **par, vert_simp=vs, num_vert_simp=nvs, tri_ind_simp=ts, num_tri_simp=nts)
    rt, _ = orc.horizon_gridded(**kw, **par, vert_simp=vs, num_vert_simp=nvs, tri_ind_simp=ts, num_tri_simp=nts)
    assert np.array_equal(ht, rt)
    # same traversal in the shadow kernels
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    tg, tc = hip.shadow.Terrain(), orc.Terrain()
    for t in (tg, tc):
        t.initialise(g["vert_grid"], 140, 150, 8, 8, vec_tilt, vec_norm, enl, elev, mask)
    sun = np.array([2000.0 + 1.0e7 * np.cos(0.12), 2000.0, 1.0e7 * np.sin(0.12)], np.float32)
    want = np.empty(mask.shape, np.uint8); tc.shadow(sun, want)
    got = np.empty(mask.shape, np.uint8); tg.shadow(sun, got)
    assert np.array_equal(got, want) and (want == 2).any()


def test_devices_threads_match_single_call(hip):
    """devices=[...]: one host thread and one row slab per entry (here the same GPU three times, which
    exercises the slab split, the concurrent calls and the merged statistics)."""
    g = cases.rough_terrain(90, 70, seed=21, offset=4, tilt_frames=True)
    kw = cases.grid_kwargs(g)
    in0, in1 = kw["vec_norm"].shape[:2]
    mask = (np.random.default_rng(3).random((in0, in1)) < 0.8).astype(np.uint8)
    mask[:7] = 0                                      # an empty leading slab is possible
    tilt = np.zeros((in0, in1, 3), np.float32); tilt[..., 2] = 1.0
    par = dict(dist_search=2.0, azim_num=24, elev_ang_low_lim=-60.0, mask=mask, hori_fill=-0.5, svf_vec_tilt=tilt)
    h1, a1, s1 = hip.horizon.horizon_gridded(**kw, **par)
    st1 = dict(hip.horizon.last_stats)
    for devs in ([0], [0, 0, 0], [0] * 7):
        h2, a2, s2 = hip.horizon.horizon_gridded(**kw, **par, devices=devs)
        st2 = dict(hip.horizon.last_stats)
        assert np.array_equal(h1, h2) and np.array_equal(a1, a2) and np.array_equal(s1, s2, equal_nan=True)
        assert st1["num_rays"] == st2["num_rays"] and st1["num_cells"] == st2["num_cells"]
    with pytest.raises(ValueError, match="devices"):
        hip.horizon.horizon_gridded(**kw, **par, devices=[0, 0], rows=(0, 5))
    with pytest.raises(hip.HorayzonHipError):
        hip.horizon.horizon_gridded(**kw, **par, devices=[0, 99])


def test_persistent_scene_and_blob_adopt(hip, orc):
    """A scene blob copied byte for byte (what an RCCL broadcast does) gives identical results."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    g = cases.rough_terrain(64, 80, seed=17, offset=4)
    kw = cases.grid_kwargs(g)
    sc = hip.Scene.create(kw["vert_grid"], kw["dem_dim_0"], kw["dem_dim_1"])
    ref, _ = hip.horizon.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0)
    a, _ = hip.horizon.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0, scene=sc)
    assert np.array_equal(a, ref)
    p, n = sc.blob()
    src = torch.empty(0, dtype=torch.uint8, device="cuda:0")
    clone = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    hiprt = C.CDLL("libamdhip64.so")
    assert hiprt.hipMemcpy(C.c_void_p(clone.data_ptr()), C.c_void_p(p), C.c_size_t(n), 3) == 0
    sc2 = hip.Scene.adopt(clone.data_ptr(), n, 0, keepalive=clone)
    b, _ = hip.horizon.horizon_gridded(**kw, dist_search=1.0, azim_num=12, elev_ang_low_lim=-60.0, scene=sc2)
    assert np.array_equal(b, ref)
    del src


def _locations(seed, n=300, tilt=False):
    g = cases.rough_terrain(70, 80, seed=seed, offset=0, relief=700.0)
    rng = np.random.default_rng(seed + 1)
    ci = rng.integers(3, 67, n); cj = rng.integers(3, 77, n)
    coords = np.stack([g["x"][cj] + rng.uniform(-12, 12, n), g["y"][ci] + rng.uniform(-12, 12, n),
                       g["z"][ci, cj] + rng.uniform(-150, 300, n)], axis=1).astype(np.float32)
    coords[:5, 0] += 1.0e5          # far outside the DEM: the normal never meets the mesh -> NaN rows
    vn = np.zeros((n, 3), np.float32); vn[:, 2] = 1.0
    vo = np.zeros((n, 3), np.float32); vo[:, 1] = 1.0
    if tilt:
        a = rng.uniform(-0.05, 0.05, n); b = rng.uniform(-0.05, 0.05, n)
        nrm = np.stack([np.sin(b), -np.sin(a) * np.cos(b), np.cos(a) * np.cos(b)], axis=1)
        north = np.array([0.0, 1.0, 0.0])[None, :] - nrm[:, 1:2] * nrm
        north /= np.linalg.norm(north, axis=1, keepdims=True)
        vn, vo = nrm.astype(np.float32), north.astype(np.float32)
    return g, coords, vn, vo


@pytest.mark.parametrize("alg", ALGS)
def test_horizon_locations(hip, orc, alg):
    """horizon_locations (horizon_comp.cpp:828-1094): snap onto the mesh along +/- normal, then search."""
    g, coords, vn, vo = _locations(51, tilt=True)
    roe = np.linspace(0.01, 2.0, coords.shape[0]).astype(np.float32)
    h_gpu, a_gpu = hip.horizon.horizon_locations(g["vert_grid"], 70, 80, coords, vn, vo, 2.0, azim_num=24,
                                                 ray_algorithm=alg, ray_org_elev=roe)
    st = hip.horizon.last_stats
    h_cpu, a_cpu, so = orc.horizon_locations(g["vert_grid"], 70, 80, coords, vn, vo, 2.0, azim_num=24,
                                             ray_algorithm=alg, ray_org_elev=roe, return_stats=True)
    assert np.array_equal(a_gpu, a_cpu)
    assert np.isnan(h_gpu[:5]).all() and not np.isnan(h_gpu[5:]).any()
    assert np.array_equal(h_gpu, h_cpu, equal_nan=True)
    assert st["num_rays"] == so["rays"] and st["num_cells"] == so["found"] == coords.shape[0] - 5


@pytest.mark.parametrize("alg", ("binary_search", "discrete_sampling"))
def test_horizon_locations_distance(hip, orc, alg):
    """Distance to the horizon (closest hit, the *_hori_dist variants :519-612)."""
    g, coords, vn, vo = _locations(57)
    out_g = hip.horizon.horizon_locations(g["vert_grid"], 70, 80, coords, vn, vo, 2.5, azim_num=16,
                                          ray_algorithm=alg, hori_dist_out=True, elev_ang_low_lim=-60.0)
    out_c = orc.horizon_locations(g["vert_grid"], 70, 80, coords, vn, vo, 2.5, azim_num=16,
                                  ray_algorithm=alg, hori_dist_out=True, elev_ang_low_lim=-60.0)
    assert np.array_equal(out_g[0], out_c[0], equal_nan=True)          # horizon
    assert np.array_equal(out_g[1], out_c[1], equal_nan=True)          # distance: the same float division
    assert np.nanmax(out_g[1]) <= 2500.0 * 1.0001 and np.nanmin(out_g[1]) >= 0.0
    with pytest.raises(TypeError, match="guess_constant"):
        hip.horizon.horizon_locations(g["vert_grid"], 70, 80, coords, vn, vo, 2.5, ray_algorithm="guess_constant",
                                      hori_dist_out=True)


def test_svf_fused_and_standalone(hip, orc):
    g = cases.rough_terrain(72, 72, seed=23, offset=4)
    kw = cases.grid_kwargs(g)
    vec_tilt, *_ = cases.terrain_inputs(g)
    hori, azim, svf_fused = hip.horizon.horizon_gridded(**kw, dist_search=1.5, azim_num=60,
                                                        elev_ang_low_lim=-60.0, svf_vec_tilt=vec_tilt)
    svf_gpu = hip.topo_param.sky_view_factor(azim, hori, vec_tilt)
    svf_cpu = orc.sky_view_factor(azim, hori, vec_tilt)
    assert np.abs(svf_gpu - svf_cpu).max() <= 1.0e-5       # north-star bound
    assert np.abs(svf_fused - svf_cpu).max() <= 1.0e-5
    assert 0.2 < svf_cpu.min() and svf_cpu.max() <= 1.0 + 1e-5


@pytest.mark.parametrize("refrac", (False, True))
def test_shadow_and_sw_dir_cor(hip, orc, refrac):
    from horayzon_amd import synth
    g = cases.c2_hill(height=1500.0)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    mask[5:9, 5:20] = 0
    tg = hip.shadow.Terrain()
    tc = orc.Terrain()
    args = (g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask)
    tg.initialise(*args, refrac_cor=refrac, sw_dir_cor_fill=-9.0)
    tc.initialise(*args, refrac_cor=refrac, sw_dir_cor_fill=-9.0)
    suns, alt, _ = synth.sun_positions(num=24)
    suns = suns + np.array([5000.0, 5000.0, 0.0], np.float32)
    n_shaded = 0
    for s in range(suns.shape[0]):
        sg = np.full(mask.shape, 255, np.uint8); sc = sg.copy()
        tg.shadow(suns[s], sg); tc.shadow(suns[s], sc)
        rays_g, rays_c = tg.last_stats["num_rays"], tc.rays
        fg = np.full(mask.shape, np.nan, np.float32); fc = fg.copy()
        tg.sw_dir_cor(suns[s], fg); tc.sw_dir_cor(suns[s], fc)
        # bit-identical with and without refraction -- to the correctly-rounded-libm CONTRACT: the refraction branch's float libm calls are the shared,
        # correctly rounded hz_crmath.h routines on both sides (byte and float output: the bar is equality)
        assert np.array_equal(sg, sc)
        assert rays_g == rays_c
        assert np.array_equal(fg, fc)
        assert set(np.unique(sg)).issubset({0, 1, 2, 3})
        assert np.all(sg[mask == 0] == 3) and np.all(fg[mask == 0] == np.float32(-9.0))
        n_shaded += int((sg == 2).sum())
    assert n_shaded > 0


def test_refraction_against_platform_libm(hip, orc):
    """The refraction branch's five float libm calls (shadow_comp.cpp:135-159, :430-446) are glibc's in the reference;
    product and oracle both use the correctly rounded hz_crmath.h, so their equality (test above) holds by
    construction.  This test is the independent one: the oracle switched to the PLATFORM's acosf / tanf / powf / cosf /
    sinf (orc.set_libm(True)).  The contract is then a tolerance, not equality: a last-bit difference in the refraction
    angle may move a sun direction by one ulp, flip a grazing ray and change sw_dir_cor in the last digits."""
    from horayzon_amd import synth
    g = cases.c2_hill(height=1500.0)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    tg, tc = hip.shadow.Terrain(), orc.Terrain()
    args = (g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask)
    tg.initialise(*args, refrac_cor=True, sw_dir_cor_fill=-9.0)
    tc.initialise(*args, refrac_cor=True, sw_dir_cor_fill=-9.0)
    suns, alt, _ = synth.sun_positions(num=24)
    suns = suns + np.array([5000.0, 5000.0, 0.0], np.float32)
    orc.set_libm(True)
    try:
        n, flips, worst = 0, 0, 0.0
        for s in range(suns.shape[0]):
            sg = np.full(mask.shape, 255, np.uint8); sc = sg.copy()
            tg.shadow(suns[s], sg); tc.shadow(suns[s], sc)
            fg = np.full(mask.shape, np.nan, np.float32); fc = fg.copy()
            tg.sw_dir_cor(suns[s], fg); tc.sw_dir_cor(suns[s], fc)
            same = sg == sc
            n += sg.size; flips += int((~same).sum())
            if same.any():
                d = np.abs(fg[same] - fc[same]) / np.maximum(np.abs(fc[same]), 1.0)
                worst = max(worst, float(d.max()))
    finally:
        orc.set_libm(False)
    assert flips <= 1e-3 * n, (flips, n)              # measured: <= 1e-4 of the shadow codes move with the libm
    assert worst <= 1e-4, worst                       # sw_dir_cor where the classification agrees


def test_refraction_against_the_oracles_own_functions(hip, orc):
    """with hz_crmath.h on both sides, GPU = oracle holds by construction.  Here the oracle evaluates the
    same CONTRACT (the correctly rounded float of acos / tan / pow / cos / sin) with its own means -- the x87 long double
    libm, rounded once (orc.set_libm(2)) -- and shares no line with the kernels.  The two agree unless an exact value lies
    within ~1e-14 relative of a float rounding boundary (hz_crmath.h evaluates in float64): 0 differing shadow codes and
    0 differing sw_dir_cor bit patterns are expected on this input, and asserted (24 positions x 32 400 cells x 5 calls)."""
    from horayzon_amd import synth
    g = cases.c2_hill(height=1500.0)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    tg, tc = hip.shadow.Terrain(), orc.Terrain()
    args = (g["vert_grid"], 200, 200, 10, 10, vec_tilt, vec_norm, enl, elev, mask)
    tg.initialise(*args, refrac_cor=True, sw_dir_cor_fill=-9.0)
    tc.initialise(*args, refrac_cor=True, sw_dir_cor_fill=-9.0)
    suns, alt, _ = synth.sun_positions(num=24)
    suns = suns + np.array([5000.0, 5000.0, 0.0], np.float32)
    orc.set_libm(2)
    try:
        for s in range(suns.shape[0]):
            sg = np.full(mask.shape, 255, np.uint8); sc = sg.copy()
            tg.shadow(suns[s], sg); tc.shadow(suns[s], sc)
            fg = np.full(mask.shape, np.nan, np.float32); fc = fg.copy()
            tg.sw_dir_cor(suns[s], fg); tc.sw_dir_cor(suns[s], fc)
            assert np.array_equal(sg, sc), s
            assert np.array_equal(fg.view(np.uint32), fc.view(np.uint32)), s
    finally:
        orc.set_libm(False)


def test_shadow_batch_matches_single(hip):
    from horayzon_amd import synth
    g = cases.rough_terrain(90, 90, seed=31, offset=5, relief=1500.0)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    t = hip.shadow.Terrain()
    t.initialise(g["vert_grid"], 90, 90, 5, 5, vec_tilt, vec_norm, enl, elev, mask)
    suns, _, _ = synth.sun_positions(num=12)
    sb = np.empty((12,) + mask.shape, np.uint8)
    fb = np.empty((12,) + mask.shape, np.float32)
    t.shadow_batch(suns, sb)
    t.sw_dir_cor_batch(suns, fb)
    for s in range(12):
        a = np.empty(mask.shape, np.uint8); t.shadow(suns[s], a)
        b = np.empty(mask.shape, np.float32); t.sw_dir_cor(suns[s], b)
        assert np.array_equal(a, sb[s])
        assert np.array_equal(b, fb[s], equal_nan=True)


def test_no_device_memory_leak(hip):
    """Repeated calls through every handle type leave the free HBM where it was."""
    import gc
    import torch
    from horayzon_amd import synth
    g = cases.rough_terrain(120, 130, seed=9, offset=6, relief=900.0)
    kw = cases.grid_kwargs(g)
    vec_tilt, vec_norm, enl, elev, mask = cases.terrain_inputs(g)
    suns, _, _ = synth.sun_positions(num=4)
    coords = np.stack([g["x"][10:40], g["y"][10:40], g["z"][10:40, 10:40].diagonal() + 5.0], axis=1).astype(np.float32)
    vn = np.zeros((30, 3), np.float32); vn[:, 2] = 1.0
    vo = np.zeros((30, 3), np.float32); vo[:, 1] = 1.0

    def cycle():
        hip.horizon.horizon_gridded(**kw, dist_search=3.0, azim_num=24, svf_vec_tilt=vec_tilt, _chunk_rows=17)
        sc = hip.Scene.create(kw["vert_grid"], 120, 130)
        hip.horizon.horizon_gridded(**kw, dist_search=3.0, azim_num=24, scene=sc, rows=(3, 50))
        hip.horizon.horizon_locations(g["vert_grid"], 120, 130, coords, vn, vo, 3.0, azim_num=12, scene=sc)
        sc.close()
        t = hip.shadow.Terrain()
        t.initialise(g["vert_grid"], 120, 130, 6, 6, vec_tilt, vec_norm, enl, elev, mask)
        out = np.empty((4,) + mask.shape, np.uint8)
        t.shadow_batch(suns, out)
        del t
        gc.collect()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert abs(free0 - free1) <= 32 << 20, (free0, free1)


@pytest.mark.parametrize("case", ("rough", "tilted_large_coords", "steep_fine", "masked_coarse"))
def test_near_field_certificates_are_transparent(hip, orc, case):
    """hz_near.hip: rays that clear the cell's neighbourhood start beyond it.  Same horizon, same ray count as
    with the certificates switched off and as the oracle; every shortened ray re-traced over its full length
    takes the same decision (near_violations == 0); and a useful share of the rays is shortened."""
    if case == "rough":
        g = cases.rough_terrain(96, 110, seed=5, offset=6, relief=1200.0)
        par = dict(dist_search=4.0, azim_num=72, elev_ang_low_lim=-30.0)
    elif case == "tilted_large_coords":
        g = cases.rough_terrain(80, 70, seed=8, offset=5, relief=700.0, tilt_frames=True, origin=(2.6e6, 1.2e6))
        par = dict(dist_search=3.0, azim_num=45, hori_acc=0.1, elev_ang_low_lim=-45.0, ray_algorithm="binary_search")
    elif case == "steep_fine":
        g = cases.rough_terrain(70, 90, seed=12, dx=5.0, dy=8.0, offset=4, relief=900.0)
        par = dict(dist_search=1.0, azim_num=120, hori_acc=0.25, elev_ang_low_lim=-89.98, ray_org_elev=2.0)
    else:
        g = cases.rough_terrain(64, 64, seed=3, dx=90.0, dy=60.0, offset=3, relief=300.0)
        rng = np.random.default_rng(1)
        par = dict(dist_search=8.0, azim_num=7, hori_acc=3.0, elev_ang_low_lim=-15.0, ray_algorithm="discrete_sampling",
                   mask=(rng.random((58, 58)) < 0.7).astype(np.uint8), hori_fill=-1.0)
    kw = cases.grid_kwargs(g)
    h_on, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=True)
    st_on = dict(hip.horizon.last_stats)
    h_off, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True, _near_skip=False)
    st_off = dict(hip.horizon.last_stats)
    h_cpu, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    assert np.array_equal(h_on, h_off) and np.array_equal(h_on, h_cpu)
    assert st_on["num_rays"] == st_off["num_rays"] == so["rays"]
    assert st_on["near_violations"] == 0 and st_off["rays_shortened"] == 0
    # (coarse tables leave less room: the margin is two table steps, 1.2 deg at hori_acc = 3 deg)
    assert st_on["rays_shortened"] > (0.15 if case == "masked_coarse" else 0.3) * st_on["num_rays"], (st_on["rays_shortened"], st_on["num_rays"])
    # (the verifying pass traces the shortened rays twice, so its node count is not the product's)
    h_cnt, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True)
    st_cnt = dict(hip.horizon.last_stats)
    assert np.array_equal(h_cnt, h_on) and st_cnt["nodes_visited"] < st_off["nodes_visited"]


def test_near_field_certificates_with_an_outer_tin_are_per_cell(hip, orc):
    """An outer-domain TIN is not part of the height field the certificates' distance bound relies on: its triangles mark
    the scene's bad-quad bitmap (HZ_BLOB_BAD_MAP) and the cells whose window lies under / next to one run without a
    certificate; the others keep theirs (rounds 1-4: off for the whole scene).  Every shortened ray is re-traced."""
    g = cases.rough_terrain(40, 44, seed=2, offset=3)
    kw = cases.grid_kwargs(g)
    vs, nvs, ts, nts = cases.outer_tin(g)
    h, _ = hip.horizon.horizon_gridded(**kw, dist_search=6.0, azim_num=24, vert_simp=vs, num_vert_simp=nvs,
                                       tri_ind_simp=ts, num_tri_simp=nts, count_work=True, _verify_near=1)
    st = dict(hip.horizon.last_stats)
    assert st["height_field"] == 0 and st["near_used"] == 1 and st["near_violations"] == 0
    assert st["near_verified"] == st["rays_shortened"]
    h_cpu, _ = orc.horizon_gridded(**kw, dist_search=6.0, azim_num=24, vert_simp=vs, num_vert_simp=nvs,
                                   tri_ind_simp=ts, num_tri_simp=nts)
    assert np.array_equal(h, h_cpu)


def _schedule_cases(hip, orc):
    """The parity cases that the schedule tests below run again under other launch schedules."""
    for alg in ALGS:
        test_c2_gaussian_hill(hip, orc, alg)
        test_rough_tilted_frames(hip, orc, alg)
    test_c2_guard_events(hip, orc)
    test_mask_and_fill(hip, orc)
    test_outer_tin(hip, orc)
    test_row_slab(hip, orc)
    test_odd_parameters(hip, orc)
    test_traversal_stack_is_one_entry_per_level(hip, orc)
    for case in ("rough", "tilted_large_coords", "steep_fine", "masked_coarse"):
        test_near_field_certificates_are_transparent(hip, orc, case)
    test_streamed_host_output(hip, 7)


@pytest.fixture
def schedule(hip):
    """Sets hip.horizon.schedule_overrides (defaults of hz_opts.left_min / persist_grid / left_cap_test) for one test."""
    def set_(**kw):
        hip.horizon.schedule_overrides.clear()
        hip.horizon.schedule_overrides.update(kw)
    yield set_
    hip.horizon.schedule_overrides.clear()


@pytest.mark.parametrize("grid", (1, 3))
def test_persistent_waves_on_small_grids(hip, orc, schedule, grid):
    """Round 5: a full launch of k_horizon has as many workgroups as are resident at once and every wave pulls 8 x 8 blocks from
    the queue of its XCD, then from the other XCDs' queues (hz_horizon.hip).  Small grids have fewer tiles than that and run one
    tile per workgroup; opts.persist_grid forces 1 / 3 workgroups, so that the parity tests against the oracle run the block loop
    with many passes per wave -- and the wave-private groups of leftover records, filled over several blocks."""
    schedule(persist_grid=grid)
    _schedule_cases(hip, orc)


@pytest.mark.parametrize("sched", (
    dict(left_min=-1),                                   # no hand-over: every block runs to its end
    dict(left_min=-1, persist_grid=-1),                  # ... and one tile per workgroup (the round-4 schedule)
    dict(left_min=0x10, persist_grid=-1),                # one level, exact allocations (no groups) from non-persistent workgroups
    dict(left_min=0x20, persist_grid=2),                 # one level at 32: half of every block is handed over
    dict(left_min=0x38, persist_grid=-1),                # one level at the cap of 56
    dict(left_min=0x203038, persist_grid=3),             # three levels, large thresholds
    dict(left_min=0x03030303, persist_grid=2),           # four levels, tiny thresholds
    dict(left_min=0x1f1f1f, persist_grid=5),             # three levels at 31
    dict(left_min=0x101010, persist_grid=2, left_cap_test=64),      # regions of ONE group: every level runs out of room
    dict(left_min=0x0810, persist_grid=3, left_cap_test=192),
    dict(left_min=0x1c, persist_grid=4, left_tune=0x0410),            # follow-up launch: classes of 8 azimuths, compaction below 16 lanes
))
def test_leftover_schedules(hip, orc, schedule, sched):
    """Rounds 5 - 6: a block ends when at most t[0] of its cells are unfinished and follow-up launches finish the cells handed
    over, handing over again at t[l] (hz_horizon.hip: leftover cells) -- every other test of this file runs the default schedule
    (16 / 16 / 16).  The same parity cases under other thresholds, level counts, without persistent waves, without the hand-over,
    and with record regions so small that the out-of-room path runs."""
    schedule(**sched)
    _schedule_cases(hip, orc)


def test_leftover_cells_are_reported(hip):
    """hz_stats.left_cells / t_left_s / left_again / scratch_bytes (ABI 5, 6): a grid with full 8 x 8 blocks hands cells over and says so; the counting
    instantiation never does."""
    import horayzon_amd as hz
    from horayzon_amd import synth
    g = synth.gaussian_hill(n=200, dx=50.0, height=1500.0, sigma=1500.0, offset=10)
    kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
    h0, _ = hz.horizon.horizon_gridded(**kw, dist_search=10.0, azim_num=36)
    st = dict(hz.horizon.last_stats)
    h1, _ = hz.horizon.horizon_gridded(**kw, dist_search=10.0, azim_num=36, count_work=True)
    sc = dict(hz.horizon.last_stats)
    assert np.array_equal(h0, h1) and st["num_rays"] == sc["num_rays"]
    assert 0 < st["left_cells"] <= st["num_cells"] and st["t_left_s"] > 0.0 and st["scratch_bytes"] > 0
    assert sc["left_cells"] == 0 and sc["left_again"] == 0
    h2, _ = hz.horizon.horizon_gridded(**kw, dist_search=10.0, azim_num=36, _left_min=-1)
    s2 = dict(hz.horizon.last_stats)
    assert np.array_equal(h0, h2) and s2["num_rays"] == st["num_rays"] and s2["left_cells"] == 0 and s2["t_left_s"] == 0.0
    h3, _ = hz.horizon.horizon_gridded(**kw, dist_search=10.0, azim_num=36, _left_min=0x0c0c0c, _persist_grid=2)
    s3 = dict(hz.horizon.last_stats)
    assert np.array_equal(h0, h3) and s3["num_rays"] == st["num_rays"] and s3["left_cells"] > 0 and s3["left_again"] > 0
