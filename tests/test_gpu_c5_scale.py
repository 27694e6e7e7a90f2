"""Config 5 AT ITS WORKLOAD on one GPU: the 4 x 4 mosaic as a 14401 x 14401 synthetic tile (207 M vertices,
415 M triangles, 17.9 GB scene), 360 azimuths, the whole 206 M-cell inner domain.

1. `bench.py --workload c5` as the driver runs it (one rank through the sharded code path: scene broadcast out of the
   blob allocation, dist.sharded_rows, slab-local inputs made on the device, SVF fused, chunked horizon): the whole job,
   with rows of the gathered SVF dumped for the check below.
2. In this process: the same mosaic, the LBVH built again, a slab of rows with the horizon resident -- one row bit for bit
   against the CPU oracle -- and the dumped SVF rows (both edges, the middle and one more) against the SVF of the
   oracle's horizon for those rows (<= 1e-5, the north-star bound).
Prints the measured figures (run with -s to see them)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from horayzon_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, OFF, A = 14401, 16, 360
IN = N - 2 * OFF
ROWS = (0, 3001, IN // 2, IN - 1)


@pytest.fixture(scope="module")
def c5_job(tmp_path_factory):
    """The full config-5 job through bench.py (a subprocess, so that its 17.9 GB scene and host arrays are gone before
    this process builds its own)."""
    dump = str(tmp_path_factory.mktemp("c5") / "svf_rows.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--steps", "1", "--warmup", "1",
           "--dump-svf-rows", ",".join(str(r) for r in ROWS), "--dump-path", dump]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1]), np.load(dump)


def test_c5_full_job_through_sharded_rows(c5_job):
    d, svf_rows = c5_job
    c = d["config"]
    print(json.dumps({k: c[k] for k in ("scene_bcast_s", "bvh_build_s", "kernel_s_rank0", "svf_kernel_s_rank0",
                                        "near_prepass_s_rank0", "stack_fallbacks_rank0", "stack_redo_blocks_rank0",
                                        "guard_events_rank0", "guard_cells_rank0")} | {"cells_per_s": d["value"],
                                                                                         "ms_per_step": d["ms_per_step"]}))
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["scaling"] == "strong"
    assert abs(d["value"] - IN * IN / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]       # all 206 M cells, whole job
    assert d["value"] >= 5.3e6, d["value"]
    assert c["gathered_svf_finite"] is True and c["slabs"] == [[0, IN]]
    assert c["scene_bcast_s"] < 0.05 and c["scene_bcast_zero_copy"] is True and c["scene_bytes"] > 15e9
    # the deep tree overflows the fast stack in a few places only: blocks are repeated one by one, never whole launches
    assert c["stack_fallbacks_rank0"] <= 40 and c["stack_redo_blocks_rank0"] <= 20000
    assert svf_rows.shape == (len(ROWS), IN) and np.isfinite(svf_rows).all()
    assert 0.2 < svf_rows.min() and svf_rows.max() <= 1.0 + 1e-5


def test_c5_rows_against_the_oracle(hip, orc, c5_job):
    torch = pytest.importorskip("torch")
    from horayzon_amd import _lib
    _, svf_rows = c5_job
    rows = 32
    g = synth.fractal_tile(n=N, offset=OFF)
    sc = hip.Scene.create(g["vert_grid"], N, N)
    assert sc.stats["bvh_height"] <= 16 and sc.stats["scene_bytes"] > 15e9
    dev = "cuda:0"
    rb = IN // 2
    # slab-local inputs: this caller holds the 32 rows only
    d_norm = torch.zeros((rows, IN, 3), dtype=torch.float32, device=dev); d_norm[..., 2] = 1.0
    d_north = torch.zeros((rows, IN, 3), dtype=torch.float32, device=dev); d_north[..., 1] = 1.0
    d_mask = torch.ones((rows, IN), dtype=torch.uint8, device=dev)
    d_hori = torch.empty((rows, IN, A), dtype=torch.float32, device=dev)
    opts = _lib.hz_opts(); opts.device = 0; opts.top_nodes = -1; opts.regroup = -1
    opts.row_begin, opts.row_end = rb, rb + rows
    opts.hori_is_slab = 1; opts.inputs_are_slab = 1
    st = _lib.hz_stats()
    for _ in range(2):
        st = _lib.hz_stats()
        _lib.check(_lib.lib().hz_horizon_gridded_scene(
            sc._h, d_norm.data_ptr(), d_north.data_ptr(), OFF, OFF, d_hori.data_ptr(), IN, IN,
            A, 50.0, 0.25, b"guess_constant", -15.0, d_mask.data_ptr(), 0.0, 0.01, C.byref(opts), C.byref(st)))
    assert st.num_cells == rows * IN and st.guard_events == 0 and st.height_field == 1 and st.near_used == 1
    mid = d_hori[:1].cpu().numpy()
    # four 32-row slabs of the mosaic (both rims) with every shortened ray traced a second time
    near = dict(rays=0, rays_shortened=0, retraced=0, violations=0)
    for b in (0, 4801, 9613, IN - rows):
        o2 = _lib.hz_opts(); o2.device = 0; o2.top_nodes = -1; o2.regroup = -1
        o2.row_begin, o2.row_end = b, b + rows
        o2.hori_is_slab = 1; o2.inputs_are_slab = 1; o2.count_work = 1; o2.verify_near = 1
        s2 = _lib.hz_stats()
        _lib.check(_lib.lib().hz_horizon_gridded_scene(
            sc._h, d_norm.data_ptr(), d_north.data_ptr(), OFF, OFF, d_hori.data_ptr(), IN, IN,
            A, 50.0, 0.25, b"guess_constant", -15.0, d_mask.data_ptr(), 0.0, 0.01, C.byref(o2), C.byref(s2)))
        assert s2.near_used == 1 and s2.near_violations == 0, (b, s2.near_violations)
        assert s2.near_verified == s2.rays_shortened and s2.rays_shortened > 0.6 * s2.num_rays
        near["rays"] += s2.num_rays; near["rays_shortened"] += s2.rays_shortened
        near["retraced"] += s2.near_verified; near["violations"] += s2.near_violations
    from tests.test_gpu_fullsize import _log_r04
    _log_r04("c5_mosaic_4x32_rows", dict(near, shortened_fraction=near["rays_shortened"] / near["rays"]))
    del d_hori, sc
    torch.cuda.empty_cache()
    kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
    vec_tilt, _ = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], OFF)
    worst = 0.0
    for k, r in enumerate(ROWS):
        ref, azim, so = orc.horizon_gridded(**kw, dist_search=50.0, azim_num=A, rows=(r, r + 1), slab_only=True,
                                            return_stats=True)
        if r == rb:
            assert np.array_equal(mid, ref)                          # horizon bit-identical at 207 M vertices
        svf_ref = orc.sky_view_factor(azim, ref, np.ascontiguousarray(vec_tilt[r:r + 1]))
        err = float(np.abs(svf_rows[k] - svf_ref[0]).max())
        worst = max(worst, err)
        assert err <= 1.0e-5, (r, err)                               # the bench job's SVF for this row
    print(json.dumps({"n": N, "slab_rows": rows, "kernel_s": st.t_kernel_s, "cells_per_s": st.num_cells / st.t_kernel_s,
                      "mray_per_s": st.num_rays / st.t_kernel_s / 1e6, "svf_rows_checked": list(ROWS),
                      "svf_max_abs_err": worst}))
