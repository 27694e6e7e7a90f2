"""How far can Embree's evaluation of the triangle test sit from the oracle's?  

The reference's hit decisions are Embree's (horizon_comp.cpp:106 ROBUST flag, :258 rtcOccluded1; shadow_comp.cpp:466,
:576) and Embree cannot be built or installed in this image, so the question cannot be answered by comparison.  It can
be BOUNDED: this script runs the workloads of the parity tests through the CPU oracle with the triangle test evaluated
the ways Embree plausibly evaluates it (oracle/hz_oracle.c, "SENSITIVITY VARIANTS"):

  embree_fma_rcp    the robust Pluecker test with FMA-contracted cross / dot products, `stable_triangle_normal`,
                    and the depth test t = rcp(den) * T with a Newton-refined hardware reciprocal
  moeller_trumbore  the classic Moeller-Trumbore test BASELINE.json's north_star names

For every workload and variant it reports
  ray_flips_per_1e6   rays of the SHIPPED search whose hit decision the variant changes (same ray, both tests)
  out_mismatch_frac   fraction of output values (horizon angles / shadow codes) that differ when the whole computation
                      runs with the variant (a flipped ray moves the search, so this is not the same number)
  out_max_abs_diff    largest horizon difference [rad] (bounded by the table logic: one flip moves a result by one
                      search step)
Round 6 adds a third column, "published_test": the whole computation with the two clauses the contract gained in round 5
switched off (orc.set_den_noise(0): Embree's published `den != 0`; orc.set_box_start(0): box tests over [0, tfar]) -- the
published robust test against the shipped one (tests/test_oracle.py::test_round5_contract_clauses_move_no_baseline_result is
the committed form of the same comparison).  Whole-run figures only (the per-ray comparison hook knows the two variants above).
CPU only (OpenMP oracle); writes profiles/r06/embree_sensitivity.json (round 3: profiles/r03/).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from horayzon_amd import synth          # noqa: E402  (synthetic inputs only: no GPU code is touched)
from oracle import oracle as orc        # noqa: E402
from tests import cases                 # noqa: E402

VARIANTS = ("embree_fma_rcp", "moeller_trumbore")


def dem(z, dx=30.0, dy=30.0, offset=2):
    n0, n1 = z.shape
    x = (np.arange(n1) * dx).astype(np.float32)
    y = ((n0 - 1 - np.arange(n0)) * dy).astype(np.float32)
    xx, yy = np.meshgrid(x, y)
    vec_norm, vec_north = synth.planar_frames(n0 - 2 * offset, n1 - 2 * offset)
    return dict(vert_grid=synth.pack_vertices(xx, yy, z.astype(np.float32)), dem_dim_0=n0, dem_dim_1=n1,
                vec_norm=vec_norm, vec_north=vec_north, offset_0=offset, offset_1=offset)


def horizon_case(name, kw, **par):
    orc.set_tri_mode("plain"); orc.set_tri_compare(None)
    t0 = time.time()
    base, _, so = orc.horizon_gridded(**kw, **par, return_stats=True)
    out = {"workload": name, "values": int(base.size), "rays": so["rays"], "variants": {}}
    for v in VARIANTS:
        orc.set_tri_compare(v)
        again, _ = orc.horizon_gridded(**kw, **par)
        n, flips = orc.tri_compare_counts()
        orc.set_tri_compare(None)
        assert np.array_equal(again, base, equal_nan=True) and n == so["rays"]     # the comparison does not disturb the run
        orc.set_tri_mode(v)
        var, _, sv = orc.horizon_gridded(**kw, **par, return_stats=True)
        orc.set_tri_mode("plain")
        diff = np.abs(var - base)
        out["variants"][v] = {"rays_compared": n, "ray_flips": flips, "ray_flips_per_1e6": 1e6 * flips / max(n, 1),
                              "out_mismatch": int((diff > 0).sum()), "out_mismatch_frac": float((diff > 0).mean()),
                              "out_max_abs_diff_rad": float(diff.max()), "rays_of_variant_run": sv["rays"]}
    try:      # the published test: den != 0, box tests from 0 (the contract of rounds 1-4)
        orc.set_den_noise(0.0); orc.set_box_start(0.0)
        var, _, sv = orc.horizon_gridded(**kw, **par, return_stats=True)
    finally:
        orc.set_den_noise(); orc.set_box_start()
    diff = np.abs(var - base)
    out["variants"]["published_test"] = {"out_mismatch": int((diff > 0).sum()), "out_mismatch_frac": float((diff > 0).mean()),
                                         "out_max_abs_diff_rad": float(diff.max()), "rays_of_variant_run": sv["rays"],
                                         "guards_of_variant_run": sv["guards"], "guards": so["guards"]}
    out["seconds"] = time.time() - t0
    print(json.dumps(out), flush=True)
    return out


def shadow_case(name, g, n, off, bands, suns, refrac):
    in0 = in1 = n - 2 * off
    vec_tilt, enl = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], off)
    vec_norm, _ = synth.planar_frames(in0, in1)
    elev = np.ascontiguousarray(g["z"][off:off + in0, off:off + in1])
    mask = np.ones((in0, in1), np.uint8)
    out = {"workload": name, "values": 0, "rays": 0,
           "variants": {v: {"rays_compared": 0, "ray_flips": 0, "out_mismatch": 0, "sw_dir_cor_mismatch": 0} for v in VARIANTS + ("published_test",)}}
    t0 = time.time()
    for rb in bands:
        sl = slice(rb, rb + 8)
        tc = orc.Terrain()
        tc.initialise(g["vert_grid"], n, n, off + rb, off, np.ascontiguousarray(vec_tilt[sl]), np.ascontiguousarray(vec_norm[sl]),
                      np.ascontiguousarray(enl[sl]), np.ascontiguousarray(elev[sl]), np.ascontiguousarray(mask[sl]),
                      refrac_cor=refrac, sw_dir_cor_fill=-7.0)
        for s in suns:
            a = np.empty((8, in1), np.uint8); f = np.empty((8, in1), np.float32)
            orc.set_tri_mode("plain"); orc.set_tri_compare(None)
            tc.shadow(s, a); tc.sw_dir_cor(s, f)
            out["values"] += a.size; out["rays"] += tc.rays
            for v in VARIANTS:
                b = np.empty_like(a); h = np.empty_like(f)
                orc.set_tri_compare(v); tc.shadow(s, b)
                nn, fl = orc.tri_compare_counts(); orc.set_tri_compare(None)
                assert np.array_equal(a, b)
                orc.set_tri_mode(v); tc.shadow(s, b); tc.sw_dir_cor(s, h); orc.set_tri_mode("plain")
                r = out["variants"][v]
                r["rays_compared"] += nn; r["ray_flips"] += fl
                r["out_mismatch"] += int((a != b).sum()); r["sw_dir_cor_mismatch"] += int((f != h).sum())
            try:
                b = np.empty_like(a); h = np.empty_like(f)
                orc.set_den_noise(0.0); orc.set_box_start(0.0)
                tc.shadow(s, b); tc.sw_dir_cor(s, h)
            finally:
                orc.set_den_noise(); orc.set_box_start()
            r = out["variants"]["published_test"]
            r["out_mismatch"] += int((a != b).sum()); r["sw_dir_cor_mismatch"] += int((f != h).sum())
    for v in VARIANTS + ("published_test",):
        r = out["variants"][v]
        r["ray_flips_per_1e6"] = 1e6 * r["ray_flips"] / max(r["rays_compared"], 1)
        r["out_mismatch_frac"] = r["out_mismatch"] / max(out["values"], 1)
    out["seconds"] = time.time() - t0
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c3-rows", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "embree_sensitivity.json"))
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    res = {"threads": orc.num_threads(), "variants": list(VARIANTS), "cases": []}
    # ---- config 2: the 200 x 200 hill, three algorithms
    g = cases.c2_hill()
    for alg in cases.ALGS:
        res["cases"].append(horizon_case("c2_hill_" + alg, cases.grid_kwargs(g), dist_search=10.0, azim_num=36, ray_algorithm=alg))
    # ---- the five edge-case DEMs of test_degenerate_terrain_shapes
    yy, xx = np.mgrid[0:48, 0:52]
    rng = np.random.default_rng(5)
    strip = 200.0 * rng.random((3, 400))
    spike = np.zeros((33, 35)); spike[16, 17] = 500.0
    for name, kw, par in (
            ("flat", dem(np.full((40, 44), 250.0)), dict(dist_search=2.0, azim_num=16, elev_ang_low_lim=-30.0)),
            ("terraces", dem(100.0 * ((xx // 6) % 4) + 50.0 * ((yy // 5) % 3)), dict(dist_search=2.0, azim_num=24, elev_ang_low_lim=-80.0)),
            ("spike", dem(spike), dict(dist_search=2.0, azim_num=32, elev_ang_low_lim=-30.0)),
            ("strip_3x400", dem(strip, offset=0), dict(dist_search=20.0, azim_num=12, elev_ang_low_lim=-89.98)),
            ("strip_400x3", dem(np.ascontiguousarray(strip.T), offset=0), dict(dist_search=20.0, azim_num=12, elev_ang_low_lim=-89.98))):
        res["cases"].append(horizon_case("edge_" + name, kw, **par))
    # ---- rough terrain with tilted frames and large coordinates (curved-DEM like)
    gt = cases.rough_terrain(80, 70, seed=8, offset=5, relief=700.0, tilt_frames=True, origin=(2.6e6, 1.2e6))
    res["cases"].append(horizon_case("rough_tilted_large_coords", cases.grid_kwargs(gt), dist_search=3.0, azim_num=45, hori_acc=0.1,
                                     elev_ang_low_lim=-45.0, ray_algorithm="binary_search"))
    if not args.quick:
        # ---- config 3: the middle band of the 3601^2 tile (the rows test_c3_* checks against the oracle)
        n, off = 3601, 16
        g3 = synth.fractal_tile(n=n, offset=off)
        kw3 = {k: g3[k] for k in cases.GRID_KEYS}
        rb = 1777
        res["cases"].append(horizon_case("c3_tile_rows_%d_%d" % (rb, rb + args.c3_rows), kw3, dist_search=50.0, azim_num=360,
                                         rows=(rb, rb + args.c3_rows), slab_only=True))
        # ---- config 4: shadow bands x sun positions of test_gpu_c4_shadow
        suns, alt, _ = synth.sun_positions(num=144)
        sel = [suns[s] for s in (36, 44, 52, 60, 72, 84, 96, 104)]
        for refrac in (False, True):
            res["cases"].append(shadow_case("c4_shadow_bands_refrac%d" % int(refrac), g3, n, off, (0, 1200, 2400, 3560), sel, refrac))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    # summary table
    for c in res["cases"]:
        for v in VARIANTS + ("published_test",):
            r = c["variants"][v]
            print("%-34s %-17s rays %11d  flips/1e6 %9.3f  output mismatch %.3e%s" % (
                c["workload"], v, r.get("rays_compared", 0), r.get("ray_flips_per_1e6", float("nan")), r["out_mismatch_frac"],
                ("  max |d hori| %.2e rad" % r["out_max_abs_diff_rad"]) if "out_max_abs_diff_rad" in r else ""))


if __name__ == "__main__":
    main()
