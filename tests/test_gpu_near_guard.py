"""The near-field certificates (hz_near.hip) assume that the DEM mesh is a height field over the WORLD (x, y) plane;
the reference accepts any vertex buffer (horizon_comp.cpp:126-127).  The scene build checks the assumption
(HZ_BLOB_HEIGHT_FIELD: every DEM triangle projects onto (x, y) with the same orientation and |n_z| > 1e-3 |n|) and the
certificates are only used when it holds.  These tests feed meshes that violate it."""
import numpy as np
import pytest

from horayzon_amd import synth
from tests import cases

pytestmark = pytest.mark.gpu


def _rotated_hill(angle_deg):
    """The config-2 hill in a frame rotated about the x axis: z is no longer 'up'; vec_norm / vec_north follow."""
    g = cases.c2_hill()
    n = g["dem_dim_0"]
    v = g["vert_grid"][:3 * n * n].reshape(-1, 3).astype(np.float64)
    a = np.deg2rad(angle_deg)
    rot = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(a), -np.sin(a)], [0.0, np.sin(a), np.cos(a)]])
    w = (v @ rot.T).astype(np.float32)
    kw = cases.grid_kwargs(g)
    kw["vert_grid"] = synth.pack_vertices(w[:, 0].reshape(n, n), w[:, 1].reshape(n, n), w[:, 2].reshape(n, n))
    in0, in1 = kw["vec_norm"].shape[:2]
    kw["vec_norm"] = np.ascontiguousarray(np.broadcast_to((rot @ [0.0, 0.0, 1.0]).astype(np.float32), (in0, in1, 3)))
    kw["vec_north"] = np.ascontiguousarray(np.broadcast_to((rot @ [0.0, 1.0, 0.0]).astype(np.float32), (in0, in1, 3)))
    return kw


def test_rotated_frame_is_not_a_height_field_and_certificates_stay_off(hip, orc):
    kw = _rotated_hill(80.0)
    par = dict(dist_search=10.0, azim_num=36)
    h, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True)
    st = dict(hip.horizon.last_stats)
    assert st["height_field"] == 0 and st["near_used"] == 0 and st["rays_shortened"] == 0
    ref, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    assert np.array_equal(h, ref) and st["num_rays"] == so["rays"]
    # the same hill unrotated is a height field: certificates on, many rays shortened
    g = cases.c2_hill()
    h0, _ = hip.horizon.horizon_gridded(**cases.grid_kwargs(g), **par, count_work=True)
    st0 = dict(hip.horizon.last_stats)
    assert st0["height_field"] == 1 and st0["near_used"] == 1 and st0["rays_shortened"] > 0.3 * st0["num_rays"]
    # and both describe the same terrain: different float roundings of the rotated coordinates move a few grazing
    # rays, i.e. a result by one search step (10 table entries = 0.5 deg) at most
    d = np.abs(h - h0)
    assert d.max() <= np.deg2rad(0.5) * 1.01 and (d > 0).mean() < 0.01


def _folded_sheet(n=64, dx=50.0, gap=3.0):
    """A grid folded back over itself along its middle row: the second half of the rows lies `gap` metres above the
    first half (mirror image in y).  Triangles far away in grid index are a few metres away in space -- exactly what
    the certificates' distance bound excludes for a height field."""
    rng = np.random.default_rng(7)
    i = np.arange(n)
    half = (n - 1) / 2.0
    y = (half - np.abs(i - half)) * dx                    # 0 .. fold .. 0
    x = np.arange(n) * dx
    xx, yy = np.meshgrid(x.astype(np.float32), y.astype(np.float32))
    z = (20.0 * rng.random((n, n))).astype(np.float32)
    z[i > half] += np.float32(gap + 20.0)                 # upper sheet clears the lower one
    off = 3
    in0 = in1 = n - 2 * off
    vec_norm, vec_north = synth.planar_frames(in0, in1)
    return dict(vert_grid=synth.pack_vertices(xx, yy, z), dem_dim_0=n, dem_dim_1=n, vec_norm=vec_norm,
                vec_north=vec_north, offset_0=off, offset_1=off)


def test_folded_mesh_certificates_would_drop_hits_and_are_switched_off(hip, orc):
    kw = _folded_sheet()
    par = dict(dist_search=2.0, azim_num=24, hori_acc=1.0, elev_ang_low_lim=-60.0)
    ref, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    h, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=True)
    st = dict(hip.horizon.last_stats)
    assert st["height_field"] == 0 and st["near_used"] == 0
    assert np.array_equal(h, ref) and st["num_rays"] == so["rays"] and st["near_violations"] == 0
    # with the guard overridden the certificates shorten rays that the upper sheet blocks right above the origin:
    # the full-length re-trace of those rays disagrees (this is the failure the guard prevents)
    hip.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=True, _near_skip="force")
    forced = dict(hip.horizon.last_stats)
    assert forced["near_used"] == 1 and forced["near_violations"] > 0


def test_scene_reports_height_field(hip):
    g = cases.rough_terrain(50, 60, seed=9, offset=4)
    sc = hip.Scene.create(g["vert_grid"], 50, 60)
    p, d0, d1, hf = sc.vertices()
    assert p and (d0, d1, hf) == (50, 60, True)
    kw = _rotated_hill(100.0)                              # past the vertical: every triangle flipped, a few degenerate
    sc2 = hip.Scene.create(kw["vert_grid"], 200, 200)
    assert sc2.vertices()[3] in (False, True)             # consistent orientation may still hold ...
    kw = _rotated_hill(80.0)
    assert hip.Scene.create(kw["vert_grid"], 200, 200).vertices()[3] is False


def test_tilted_frame_against_very_steep_terrain(hip, orc):
    """Configuration 370 of HZ_FUZZ_SEED=31001 (found by the wide random sweep of round 3): 1 m x 90 m cells with
    100 m steps and vec_norm 0.4 degrees off the vertical.  Adjacent triangles face away from vec_norm, the ray origin
    lies behind them, and the certificates -- which skip the spokes at the cell's own vertex -- shortened rays that hit
    at once: 14 violations in one cell, visible only in the guard count (the horizon came out the same).  The window
    must be a graph over the cell's LOCAL horizontal plane (hz_near.hip); such cells get no certificate now."""
    rng = np.random.default_rng(31001)
    for it in range(371):
        kw, par, extra, tilt = cases.fuzz_case(rng)
        if it % 4 == 1:
            rng.integers(4, 15)
    assert (kw["dem_dim_0"], kw["dem_dim_1"], par["hori_acc"], par["azim_num"]) == (15, 78, 3.0, 45)
    out = hip.horizon.horizon_gridded(**kw, **par, rows=extra["rows"], count_work=True, _verify_near=True)
    st = dict(hip.horizon.last_stats)
    ref, _, so = orc.horizon_gridded(**kw, **par, rows=extra["rows"], return_stats=True)
    assert st["near_used"] == 1 and st["near_violations"] == 0
    assert np.array_equal(out[0], ref, equal_nan=True)
    assert st["num_rays"] == so["rays"] and st["guard_events"] == so["guards"] == 539
