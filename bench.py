#!/usr/bin/env python
"""bench.py -- horizon_gridded throughput on MI355X (BASELINE.json metric).

Workload (N = 1): BASELINE.json config 3 -- 3601 x 3601 synthetic SRTM-like tile
(1 arc-second spacing at 46 deg N, seeded fractal, integer metres), 16-cell ring,
360 azimuth sectors, guess_constant, dist_search 50 km, hori_acc 0.25 deg, with the
horizon array AND the fused sky view factor written to HBM.

A "step" is one pass of the hot path over one batch of grid cells: a slab of
`--rows-per-step` inner-domain rows (default 512 x 3569 cells x 360 azimuths = 1.83 M cells,
6.6e8 output values, ~1.4e9 rays) against the full-tile LBVH.  Consecutive steps take
consecutive slabs of the tile (wrapping around), so the default K = 7 steps cover the tile
once.  (Slabs much smaller than the GPU's resident capacity of 262 k cells leave a tail:
128-row steps run 17 % slower per cell.)  Inputs (scene blob, per-cell frames,
mask, tilt) are resident in HBM before the timed region and outputs stay in HBM; the
library is called through its C ABI with device pointers.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): rank 0 builds the scene
and broadcasts the blob over xGMI once (set-up, untimed -- like the BVH build); every rank
then processes its own slabs with no data-path collective (weak scaling: per-GPU work is
fixed); the per-rank SVF slabs are gathered at the end of the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows-per-step", type=int, default=512)
    ap.add_argument("--tile", type=int, default=3601)
    ap.add_argument("--azim", type=int, default=360)
    ap.add_argument("--dist-search", type=float, default=50.0)
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the tile the CPU baseline computes (0: cores / 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count", action="store_true", help="skip the counter pass (roofline.achieved becomes I/O only)")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    import horayzon_amd as hz
    from horayzon_amd import _lib, synth
    from horayzon_amd.dist import broadcast_scene, gather_rows
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    # HZ_FORCE_DIST=1 runs the multi-rank code path (RCCL broadcast of the scene, gather, all-reduce)
    # even with a single rank -- used to exercise it on a 1-GPU box
    use_dist = world > 1 or bool(os.environ.get("HZ_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(dev))
    L = _lib.lib()

    # ---- synthetic tile + scene (set-up, untimed) ----------------------------------------
    n, off, A = args.tile, 16, args.azim
    g = synth.fractal_tile(n=n, offset=off)
    in0 = in1 = n - 2 * off
    t0 = time.time()
    scene = hz.Scene.create(g["vert_grid"], n, n, device=local_rank) if rank == 0 else None
    t_build = time.time() - t0
    scene_stats = scene.stats if scene is not None else None
    t_bcast = 0.0
    if use_dist:
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.time()
        scene = broadcast_scene(scene, local_rank, src=0)
        torch.cuda.synchronize(); dist.barrier()
        t_bcast = time.time() - t0
    blob_ptr, blob_bytes = scene.blob()

    # per-cell inputs resident in HBM
    vec_tilt_h, _ = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], off)
    d_norm = torch.zeros((in0, in1, 3), dtype=torch.float32, device=dev); d_norm[..., 2] = 1.0
    d_north = torch.zeros((in0, in1, 3), dtype=torch.float32, device=dev); d_north[..., 1] = 1.0
    d_mask = torch.ones((in0, in1), dtype=torch.uint8, device=dev)
    d_tilt = torch.from_numpy(vec_tilt_h).to(dev)
    rps = args.rows_per_step
    n_slabs = in0 // rps
    d_hori = torch.empty((rps, in1, A), dtype=torch.float32, device=dev)       # reused slab buffer
    d_svf = torch.full((in0, in1), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    opts = _lib.hz_opts()
    opts.device = local_rank
    opts.top_nodes = -1
    opts.regroup = -1
    opts.vec_tilt = d_tilt.data_ptr()
    stats = _lib.hz_stats()

    def step(s, count=False):
        # weak scaling: rank r walks the slabs of the tile starting at slab r, so with the default 6 steps
        # every rank computes each of the 6 slabs exactly once (identical work per GPU for any N)
        slab = (s + rank) % n_slabs
        rb = slab * rps
        opts.row_begin, opts.row_end = rb, rb + rps
        opts.count_work = int(count)
        opts.svf = d_svf.data_ptr()
        # the library indexes hori by global cell; hand it the slab buffer shifted back by rb rows
        hori_ptr = d_hori.data_ptr() - 4 * rb * in1 * A
        rc = L.hz_horizon_gridded_scene(scene._h, d_norm.data_ptr(), d_north.data_ptr(), off, off,
                                        hori_ptr, in0, in1, A, args.dist_search, 0.25, b"guess_constant",
                                        -15.0, d_mask.data_ptr(), 0.0, 0.01, C.byref(opts), C.byref(stats))
        _lib.check(rc)
        return rb

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- counter pass (untimed): BVH nodes / triangle tests per ray for the roofline -------
    nodes_per_ray = tris_per_ray = None
    if rank == 0 and not args.no_count:
        c = _lib.hz_stats()
        opts.count_work = 1
        saved = stats
        stats = c
        step(n_slabs // 2, count=True)
        stats = saved
        nodes_per_ray = c.nodes_visited / max(c.num_rays, 1)
        tris_per_ray = c.tris_tested / max(c.num_rays, 1)

    for w in range(args.warmup):
        step(w)
    barrier()
    stats = _lib.hz_stats()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s)
    svf_full = None
    if use_dist:    # final gather of the per-rank SVF rows touched in the last step (4 B / cell)
        rb = (args.warmup + args.steps - 1 + rank) % n_slabs * rps
        gather_rows(d_svf[rb:rb + rps], [(0, rps)] * world, dst=0)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tsum = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(t.item())
        imbalance = elapsed / (float(tsum.item()) / world)      # slowest rank / mean rank
        tot = torch.tensor([stats.num_rays, stats.num_cells], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rays_total, cells_total = float(tot[0].item()), float(tot[1].item())
    else:
        rays_total, cells_total = float(stats.num_rays), float(stats.num_cells)
        imbalance = 1.0

    if rank == 0:
        # device-copy microbenchmark: what a plain HBM stream reaches on this box
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src); torch.cuda.synchronize()
        tc0 = time.perf_counter()
        for _ in range(10):
            dst.copy_(src)
        torch.cuda.synchronize()
        copy_gbs = 10 * 2.0 * src.numel() / (time.perf_counter() - tc0) / 1e9
        del src, dst
        k_launch_s = stats.t_kernel_s / max(args.steps, 1)        # HIP events on the kernel's stream
        rays_launch = stats.num_rays / max(args.steps, 1)
        cells_launch = stats.num_cells / max(args.steps, 1)
        # algorithmic bytes per launch: per-cell I/O + BVH traversal
        b_io = (12 + 12 + 12 + 1 + 12 + 4) * cells_launch + 4.0 * A * cells_launch
        b_trav = 0.0
        if nodes_per_ray is not None:
            b_trav = rays_launch * (nodes_per_ray * 64.0 + tris_per_ray * 24.0)
        achieved = (b_io + b_trav) / k_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("rows_per_step") == rps and tj.get("tile") == n and tj.get("azim") == A:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "grid_cells_per_s (horizon_gridded, 360 azimuths, 3601^2 SRTM-like tile)",
            "value": cells_total / elapsed,
            "unit": "cells/s",
            "mray_per_s": rays_total / elapsed / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "horizon_gridded guess_constant + fused SVF, %dx%d synthetic SRTM-like tile, "
                                   "%d azimuths, dist_search %g km, slab of %d rows per step"
                                   % (n, n, A, args.dist_search, rps),
                       "cells_per_step": int(cells_launch), "rays_per_cell_azimuth": rays_launch / max(cells_launch * A, 1),
                       "parallelism": "row-slab shard x%d, scene broadcast once" % world,
                       "bvh_build_s": scene_stats["t_bvh_s"] if scene_stats else None,
                       "scene_bytes": int(blob_bytes), "scene_bcast_s": t_bcast, "scene_create_wall_s": t_build,
                       "load_imbalance_max_over_mean": imbalance, "stack_retries": int(stats.stack_retries)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "hz::k_horizon<2,false,true,false>", "kernel_ms_per_launch": 1e3 * k_launch_s,
                         "alg_bytes_per_launch": b_io + b_trav, "nodes_per_ray": nodes_per_ray,
                         "tris_per_ray": tris_per_ray, "mray_per_s_kernel": rays_launch / k_launch_s / 1e6,
                         "svf_kernel_ms_per_launch": 1e3 * stats.t_svf_s / max(args.steps, 1),
                         "device_copy_gbs": copy_gbs},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g, args, A)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(g, args, A):
    """The CPU oracle (a port of the reference's algorithm with its own BVH -- NOT Embree)
    timed on this host's cores on a bounded sample: a few rows from the middle of the tile."""
    from oracle import oracle as orc
    kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
    in0 = g["vec_norm"].shape[0]
    if args.cpu_rows <= 0:      # about 10-30 s of CPU work on the host's cores
        args.cpu_rows = max(4, orc.num_threads() // 4)
    rb = in0 // 2
    _, _, st = orc.horizon_gridded(**kw, dist_search=args.dist_search, azim_num=A, rows=(rb, rb + args.cpu_rows),
                                   slab_only=True, return_stats=True)
    cells = args.cpu_rows * g["vec_norm"].shape[1]
    return {"value": cells / st["t_rays_s"], "unit": "cells/s", "cores": orc.num_threads(), "kind": "port",
            "mray_per_s": st["rays"] / st["t_rays_s"] / 1e6, "bvh_build_s": st["t_build_s"],
            "sample": "%d rows x %d cells x %d azimuths from the middle of the same tile, ray loop only "
                      "(%.1f s); CPU restatement with its own BVH, OpenMP over cells -- not Embree"
                      % (args.cpu_rows, g["vec_norm"].shape[1], A, st["t_rays_s"])}


if __name__ == "__main__":
    main()
