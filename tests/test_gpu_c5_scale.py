"""Config 5 scale on ONE GPU: the 4 x 4 mosaic as a 14401 x 14401 synthetic tile (207 M vertices,
415 M triangles, 17.9 GB scene).  Builds the LBVH, computes a slab of rows at 360 azimuths with
all I/O resident in HBM and checks one row bit for bit against the CPU oracle.  Prints the
measured figures (run with -s to see them; recorded in profiles/r01/c5_scale_probe.json)."""
import ctypes as C
import json

import numpy as np
import pytest

from horayzon_amd import synth

pytestmark = pytest.mark.gpu


def test_c5_mosaic_scene_and_row_parity(hip, orc):
    torch = pytest.importorskip("torch")
    from horayzon_amd import _lib
    n, off, A, rows = 14401, 16, 360, 32
    g = synth.fractal_tile(n=n, offset=off)
    in0 = in1 = n - 2 * off
    sc = hip.Scene.create(g["vert_grid"], n, n)
    assert sc.stats["bvh_height"] <= 16 and sc.stats["scene_bytes"] > 15e9
    dev = "cuda:0"
    d_norm = torch.zeros((in0, in1, 3), dtype=torch.float32, device=dev); d_norm[..., 2] = 1.0
    d_north = torch.zeros((in0, in1, 3), dtype=torch.float32, device=dev); d_north[..., 1] = 1.0
    d_mask = torch.ones((in0, in1), dtype=torch.uint8, device=dev)
    d_hori = torch.empty((rows, in1, A), dtype=torch.float32, device=dev)
    rb = in0 // 2
    opts = _lib.hz_opts(); opts.device = 0; opts.top_nodes = -1; opts.regroup = -1
    opts.row_begin, opts.row_end = rb, rb + rows
    opts.hori_is_slab = 1              # d_hori holds only the slab (no address outside the allocation is formed)
    st = _lib.hz_stats()
    for _ in range(2):
        st = _lib.hz_stats()
        _lib.check(_lib.lib().hz_horizon_gridded_scene(
            sc._h, d_norm.data_ptr(), d_north.data_ptr(), off, off, d_hori.data_ptr(), in0, in1,
            A, 50.0, 0.25, b"guess_constant", -15.0, d_mask.data_ptr(), 0.0, 0.01, C.byref(opts), C.byref(st)))
    assert st.num_cells == rows * in1 and st.guard_events == 0
    kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
    ref, _, so = orc.horizon_gridded(**kw, dist_search=50.0, azim_num=A, rows=(rb, rb + 1), slab_only=True,
                                     return_stats=True)
    got = d_hori[:1].cpu().numpy()
    assert np.array_equal(got, ref)                                  # bit-identical at 207 M vertices
    print(json.dumps({"n": n, "bvh_build_s": sc.stats["t_bvh_s"], "scene_bytes": sc.stats["scene_bytes"],
                      "height": sc.stats["bvh_height"], "slab_rows": rows, "kernel_s": st.t_kernel_s,
                      "cells_per_s": st.num_cells / st.t_kernel_s, "mray_per_s": st.num_rays / st.t_kernel_s / 1e6}))
