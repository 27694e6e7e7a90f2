"""The near-field certificates (hz_near.hip) assume that the DEM mesh is a height field over the WORLD (x, y) plane;
the reference accepts any vertex buffer (horizon_comp.cpp:126-127).  The scene build checks the assumption per triangle
(HZ_BLOB_HEIGHT_FIELD: every DEM triangle projects onto (x, y) with the same orientation and a 2-D area above its own
rounding -- steepness does not matter).  Where some quads violate it (or an outer TIN is present) their (x, y) footprints
are marked in a coarse bitmap (HZ_BLOB_BAD_MAP) and only the cells near them lose their certificates (round 5; rounds
3-4 switched the whole scene off).  These tests feed meshes that violate the assumption."""
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


def test_rotated_frame_is_not_a_height_field_and_certificates_are_per_cell(hip, orc):
    kw = _rotated_hill(80.0)
    par = dict(dist_search=10.0, azim_num=36)
    h, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=1)
    st = dict(hip.horizon.last_stats)
    # the slopes that face away from the world z axis project flipped: not a height field; the cells near those quads run
    # without certificates, the others keep them -- and every shortened ray, traced again over its full length, agrees
    assert st["height_field"] == 0 and st["near_used"] == 1
    assert st["near_violations"] == 0 and st["near_verified"] == st["rays_shortened"]
    ref, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    assert np.array_equal(h, ref) and st["num_rays"] == so["rays"] and st["guard_events"] == so["guards"]
    # the same hill unrotated is a height field: certificates on, many rays shortened
    g = cases.c2_hill()
    h0, _ = hip.horizon.horizon_gridded(**cases.grid_kwargs(g), **par, count_work=True)
    st0 = dict(hip.horizon.last_stats)
    assert st0["height_field"] == 1 and st0["near_used"] == 1 and st0["rays_shortened"] > 0.3 * st0["num_rays"]
    assert st["rays_shortened"] < st0["rays_shortened"]
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


def test_folded_mesh_certificates_would_drop_hits_and_are_refused(hip, orc):
    kw = _folded_sheet()
    par = dict(dist_search=2.0, azim_num=24, hori_acc=1.0, elev_ang_low_lim=-60.0)
    ref, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    h, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=True)
    st = dict(hip.horizon.last_stats)
    # one of the two sheets projects against the other: its quads cover the whole (x, y) footprint of the mesh in the
    # scene's bad-quad bitmap, so EVERY cell is refused its certificate
    assert st["height_field"] == 0 and st["rays_shortened"] == 0
    assert np.array_equal(h, ref) and st["num_rays"] == so["rays"] and st["near_violations"] == 0
    # with the guard overridden the certificates shorten rays that the upper sheet blocks right above the origin:
    # the full-length re-trace of those rays disagrees (this is the failure the guard prevents)
    hip.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=True, _near_skip="force")
    forced = dict(hip.horizon.last_stats)
    assert forced["near_used"] == 1 and forced["near_violations"] > 0


def _tile_with_defects(n=3601):
    """The config-3 tile with what real DEMs bring along: 20 rectangular blocks raised or lowered by 0.3 ... 30 km (vertical
    steps along their rims; slopes of up to 1400 : 1), a 40 x 40 NoData hole at -32768 m, and five vertices whose (x, y)
    is displaced by 1.5 cells, so that the quads around them fold over their neighbours (triangles that project flipped)."""
    g = synth.fractal_tile(n=n, offset=16)
    rng = np.random.default_rng(505)
    z = g["z"].copy()
    steps = []
    for k in range(20):
        i0, j0 = int(rng.integers(100, n - 300)), int(rng.integers(100, n - 300))
        h, w = int(rng.integers(3, 120)), int(rng.integers(3, 120))
        dz = float(rng.choice([300.0, -300.0, 1000.0, 3000.0, 30000.0]))
        z[i0:i0 + h, j0:j0 + w] += np.float32(dz)
        steps.append((i0, j0, h, w, dz))
    hole = (1200, 2400, 40, 40)
    z[hole[0]:hole[0] + 40, hole[1]:hole[1] + 40] = np.float32(-32768.0)
    xx, yy = np.meshgrid(g["x"], g["y"])
    folds = []
    for k in range(5):
        i, j = int(rng.integers(200, n - 200)), int(rng.integers(200, n - 200))
        xx[i, j] += np.float32(1.5 * 21.44); yy[i, j] -= np.float32(1.5 * 30.87)
        folds.append((i, j))
    kw = cases.grid_kwargs(g)
    kw["vert_grid"] = synth.pack_vertices(xx, yy, z)
    return kw, dict(steps=steps, hole=hole, folds=folds, x=g["x"], y=g["y"], z=z)


def test_per_cell_guard_on_the_c3_tile_with_steps_a_nodata_hole_and_folds(hip, orc):
    """one near-vertical or flipped triangle used to switch the certificates off for all 12.7 M cells.
    Steps and NoData holes on a regular (x, y) grid ARE a height field (only the projection counts); the folded quads are
    not, and cost the certificates of the cells around them only."""
    kw, info = _tile_with_defects()
    in0 = in1 = 3569
    sc = hip.Scene.create(kw["vert_grid"], 3601, 3601)
    assert sc.vertices()[3] is False                      # the five folds
    vec_tilt = np.zeros((in0, in1, 3), np.float32); vec_tilt[:, :, 2] = 1.0
    par = dict(dist_search=50.0, azim_num=360)
    # the whole tile, every shortened ray traced a second time over its full length
    out = hip.horizon.horizon_gridded(**kw, **par, scene=sc, svf_vec_tilt=vec_tilt, svf_only=True, count_work=True, _verify_near=1)
    st = dict(hip.horizon.last_stats)
    assert st["near_used"] == 1 and st["height_field"] == 0
    assert st["near_violations"] == 0 and st["near_verified"] == st["rays_shortened"]
    assert st["rays_shortened"] >= 0.78 * st["num_rays"], st["rays_shortened"] / st["num_rays"]
    # bands through the defects: a fold, the rim of the NoData hole, the rim of a 30 km and of a 300 m step
    big = [s_ for s_ in info["steps"] if abs(s_[4]) == 30000.0][0]
    small = [s_ for s_ in info["steps"] if abs(s_[4]) == 300.0][0]
    bands = [info["folds"][0][0] - 16 - 1, info["hole"][0] - 16 - 1, big[0] - 16 - 1, small[0] + small[2] - 16 - 1]
    for r0 in bands:
        # (the band as an inner domain of its own: two rows of frames, offset_0 moved -- no 18 GB host array)
        kb = dict(kw, vec_norm=np.ascontiguousarray(kw["vec_norm"][r0:r0 + 2]), vec_north=np.ascontiguousarray(kw["vec_north"][r0:r0 + 2]),
                  offset_0=16 + r0)
        h, _ = hip.horizon.horizon_gridded(**kb, **par, scene=sc, count_work=True, _verify_near=1)
        s = dict(hip.horizon.last_stats)
        ref, _, so = orc.horizon_gridded(**kb, **par, return_stats=True)
        assert np.array_equal(h, ref), r0
        assert (s["num_rays"], s["guard_events"]) == (so["rays"], so["guards"]) and s["near_violations"] == 0, r0
        assert s["near_used"] == 1 and s["rays_shortened"] > 0
    # without the folds the same tile is a height field again, whatever the steps' slopes
    kw2 = dict(kw)
    xx, yy = np.meshgrid(info["x"], info["y"])
    kw2["vert_grid"] = synth.pack_vertices(xx, yy, info["z"])
    assert hip.Scene.create(kw2["vert_grid"], 3601, 3601).vertices()[3] is True


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


def test_cell_beside_a_spike_with_a_frame_that_is_not_quite_orthonormal(hip, orc):
    """Round 6, found by the long adversarial sweep of seed 64003 (configurations 734 and 1377; scripts/r06/diag_64003.py): a 300 m
    spike on a 1 m grid next to the cell, `vec_norm` 1.5e-5 too long and 3e-5 off the right angle with `vec_north`.  In the
    coordinates of the certificate pre-pass the ray leaves the half-plane H_k of its table azimuth by |n.t| per unit of height --
    9 mm beside the spike's face, i.e. 2.7 m of height on a face of slope 300 -- and the certificate let two rays start behind a
    face they hit: ONE horizon value per configuration came out one search step low (the oracle and the run without certificates
    agreed with each other).  hz_near.hip now widens the crossing interval by that offset (DESIGN_CERTIFICATES.md).  Replayed from
    the generator: production path == oracle, and every shortened ray re-traced over its full length takes the same decision."""
    rng = np.random.default_rng(64003 + 7)
    for it in range(1378):
        kw, par, desc = cases.adversarial_near_case(rng)
        if it not in (734, 1377):
            continue
        ho, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
        h, _ = hip.horizon.horizon_gridded(**kw, **par)
        st = dict(hip.horizon.last_stats)
        assert np.array_equal(h, ho, equal_nan=True), (it, desc, np.argwhere(h != ho)[:4].tolist())
        assert st["num_rays"] == so["rays"] and st["guard_events"] == so["guards"]
        hv, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=1)
        sv = dict(hip.horizon.last_stats)
        assert np.array_equal(hv, ho, equal_nan=True) and sv["near_violations"] == 0 and sv["near_verified"] == sv["rays_shortened"] > 0


def test_spikes_beside_cells_with_frames_skewed_up_to_the_refusal_threshold(hip, orc):
    """The worst case of rows F' / C' of DESIGN_CERTIFICATES.md, aimed at directly (the seeded generator only draws skews of 3e-5 and
    2e-3): 1 m grids with 300 m and 1000 m spikes, frames rotated about the vertical by random angles, `vec_north` up to 9.5e-5 off the
    right angle with `vec_norm` in either direction and `vec_norm` up to 5e-5 too long or too short (the frame check refuses at
    1e-4), ray origins 0.01 ... 20 m above the ground.  Production path == oracle; every shortened ray re-traced over its full length
    takes the same decision; and the certificates are still in use (a useful share of the rays is shortened)."""
    rng = np.random.default_rng(6401)
    shortened = 0
    for it in range(40):
        n0, n1 = int(rng.integers(11, 17)), int(rng.integers(11, 17))
        z = np.full((n0, n1), 100.0)
        for _ in range(int(rng.integers(1, 4))):
            z[int(rng.integers(2, n0 - 2)), int(rng.integers(2, n1 - 2))] += float(rng.choice([300.0, 1000.0, 30.0]))
        x = np.arange(n1, dtype=np.float32)
        y = (n0 - 1 - np.arange(n0)).astype(np.float32)
        xx, yy = np.meshgrid(x, y)
        off = 2
        in0, in1 = n0 - 2 * off, n1 - 2 * off
        rot = rng.uniform(0.0, 2.0 * np.pi, (in0, in1, 1))
        nrm = np.zeros((in0, in1, 3)); nrm[..., 2] = 1.0
        north = np.concatenate([np.sin(rot), np.cos(rot), np.zeros_like(rot)], axis=2)
        s_nt = rng.uniform(2.0e-5, 9.5e-5, (in0, in1, 1)) * rng.choice([-1.0, 1.0], (in0, in1, 1))
        s_len = rng.uniform(-5.0e-5, 5.0e-5, (in0, in1, 1))
        north = north + s_nt * nrm
        nrm = nrm * (1.0 + s_len)
        kw = dict(vert_grid=synth.pack_vertices(xx, yy, z.astype(np.float32)), dem_dim_0=n0, dem_dim_1=n1,
                  vec_norm=np.ascontiguousarray(nrm, np.float32), vec_north=np.ascontiguousarray(north, np.float32),
                  offset_0=off, offset_1=off)
        par = dict(dist_search=float(rng.choice([0.004, 0.012])), azim_num=360, hori_acc=float(rng.choice([0.1, 0.25])),
                   ray_algorithm=str(rng.choice(["guess_constant", "binary_search"])), elev_ang_low_lim=float(rng.choice([-45.0, -89.98])),
                   ray_org_elev=float(rng.choice([0.01, 2.0, 20.0])))
        ho, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
        h, _ = hip.horizon.horizon_gridded(**kw, **par)
        st = dict(hip.horizon.last_stats)
        assert np.array_equal(h, ho, equal_nan=True), (it, par, np.argwhere(h != ho)[:4].tolist())
        assert st["num_rays"] == so["rays"] and st["guard_events"] == so["guards"], (it, par)
        hv, _ = hip.horizon.horizon_gridded(**kw, **par, count_work=True, _verify_near=1)
        sv = dict(hip.horizon.last_stats)
        assert np.array_equal(hv, ho, equal_nan=True) and sv["near_violations"] == 0, (it, par, sv["near_violations"])
        shortened += int(sv["rays_shortened"] > 0)
    assert shortened >= 20
