"""BASELINE.json config 4 at its workload: shadow mask and direct-shortwave correction of the full 3601 x 3601
tile (3569^2 inner cells) for the 144 sun positions of one day (`synth.sun_positions`), with and without
atmospheric refraction (shadow_comp.cpp:386-605; usage examples/shadow/gridded_curved_DEM_SRTM.py:192-205).
Outputs stay in HBM (torch tensors through the C ABI); four row bands x eight sun positions are compared with
the CPU oracle bit for bit, size-independent properties are checked on everything."""
import json

import numpy as np
import pytest

from horayzon_amd import synth
from tests import cases

pytestmark = pytest.mark.gpu

BANDS = (0, 1203, 2310, 3561)          # first rows of the 8-row bands (both tile edges included)
SUNS = (30, 40, 52, 66, 72, 85, 100, 112)   # around sunrise, morning, noon, afternoon, sunset (alt -1 .. 67 deg)


@pytest.mark.parametrize("refrac", (False, True))
def test_c4_full_tile_144_sun_positions(hip, orc, refrac):
    torch = pytest.importorskip("torch")
    n, off = 3601, 16
    g = synth.fractal_tile(n=n, offset=off)
    in0 = in1 = n - 2 * off
    vec_tilt, enl = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], off)
    vec_norm, _ = synth.planar_frames(in0, in1)
    elev = np.ascontiguousarray(g["z"][off:off + in0, off:off + in1], np.float32)
    mask = np.ones((in0, in1), np.uint8)
    mask[100:140, 200:900] = 0                       # a masked patch: code 3 / fill value
    suns, alt, az = synth.sun_positions(num=144)
    assert (alt < np.deg2rad(-1.0)).any() and (alt > np.deg2rad(60.0)).any()   # night positions are submitted too
    t = hip.shadow.Terrain()
    t.initialise(g["vert_grid"], n, n, off, off, vec_tilt, vec_norm, enl, elev, mask, refrac_cor=refrac,
                 sw_dir_cor_fill=-7.0)
    dev = "cuda:0"
    d_sh = torch.full((144, in0, in1), 255, dtype=torch.uint8, device=dev)
    t.shadow_batch(suns, d_sh)
    st_sh = dict(t.last_stats)
    d_sw = torch.full((144, in0, in1), float("nan"), dtype=torch.float32, device=dev)
    t.sw_dir_cor_batch(suns, d_sw)
    st_sw = dict(t.last_stats)
    torch.cuda.synchronize()

    # ---- properties on all 144 x 12.7 M values -------------------------------------------------------
    d_mask = torch.from_numpy(mask).to(dev).bool()
    assert int(d_sh.max().item()) <= 3                                         # codes within {0, 1, 2, 3}
    assert bool((d_sh[:, ~d_mask] == 3).all().item()) and not bool((d_sh[:, d_mask] == 3).any().item())
    assert bool((d_sw[:, ~d_mask] == -7.0).all().item())
    assert not bool(torch.isnan(d_sw).any().item()) and float(d_sw[:, d_mask].min().item()) >= 0.0
    # deep night: the sun is below every horizon; only cells whose downward ray leaves the DEM unobstructed (tile rim,
    # summits looking out over the edge) can still count as lit -- the reference has no test for a sun below the horizon
    night = np.flatnonzero(alt < np.deg2rad(-15.0))
    assert len(night) > 10
    for s in night[::5]:
        assert float((d_sh[int(s)][d_mask] == 0).float().mean().item()) < 0.02
    lit = (d_sh == 0)
    # shadow and correction agree: a cell without direct light has correction 0; a lit cell a positive one
    # (the two differ only where the sun is within (90 - ang_max) degrees of the tilt plane)
    blocked = (d_sh == 2)
    assert float(d_sw[blocked].abs().max().item()) == 0.0
    noon = int(np.argmax(alt))
    assert float(lit[noon][d_mask].float().mean().item()) > 0.9
    assert float((d_sw[noon][lit[noon]] > 0).float().mean().item()) > 0.999
    frac_lit = [float(lit[s][d_mask].float().mean().item()) for s in (40, 52, 72)]
    assert frac_lit[0] < frac_lit[1] < frac_lit[2]                             # the morning fills with light
    # ray counts as the reference would count them: one ray per cell that passes the self-shading test
    assert st_sh["num_rays"] == int((d_sh != 1)[:, d_mask].sum().item())

    # ---- oracle, bit for bit, on 4 bands x 8 sun positions --------------------------------------------
    checked = 0
    for rb in BANDS:
        sl = slice(rb, rb + 8)
        tc = orc.Terrain()
        tc.initialise(g["vert_grid"], n, n, off + rb, off, np.ascontiguousarray(vec_tilt[sl]),
                      np.ascontiguousarray(vec_norm[sl]), np.ascontiguousarray(enl[sl]), np.ascontiguousarray(elev[sl]),
                      np.ascontiguousarray(mask[sl]), refrac_cor=refrac, sw_dir_cor_fill=-7.0)
        for s in SUNS:
            a = np.empty((8, in1), np.uint8); f = np.empty((8, in1), np.float32)
            tc.shadow(suns[s], a); tc.sw_dir_cor(suns[s], f)
            assert np.array_equal(d_sh[s, sl].cpu().numpy(), a), (rb, s)
            assert np.array_equal(d_sw[s, sl].cpu().numpy(), f), (rb, s)
            checked += a.size
    print(json.dumps({"refrac_cor": refrac, "cells": int(mask.sum()), "sun_positions": 144,
                      "shadow_kernel_s": st_sh["t_kernel_s"], "shadow_ms_per_position": 1e3 * st_sh["t_kernel_s"] / 144,
                      "shadow_mray_per_s": st_sh["num_rays"] / st_sh["t_kernel_s"] / 1e6,
                      "sw_dir_cor_kernel_s": st_sw["t_kernel_s"], "cells_checked_vs_oracle": checked}))


def test_overflowed_shadow_rays_are_traced_again_with_the_level_stack(hip, orc):
    """Round 4: the shadow kernel runs the fast stack (19 entries); a ray that runs out of entries is traced again with the
    one-entry-per-level discipline in its lane's own LDS column.  With 6 entries (hz_debug_set("shadow_fast_cap", 6)) that
    retry is the common case: the shadow / sw_dir_cor parity tests must stay bit-identical."""
    from horayzon_amd import _lib
    from tests import test_gpu_parity
    _lib.check(_lib.lib().hz_debug_set(b"shadow_fast_cap", 6))
    try:
        for refrac in (False, True):
            test_gpu_parity.test_shadow_and_sw_dir_cor(hip, orc, refrac)
    finally:
        _lib.check(_lib.lib().hz_debug_set(b"shadow_fast_cap", -1))
