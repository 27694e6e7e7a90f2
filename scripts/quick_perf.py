"""Quick performance probe: central window of the 3601^2 synthetic tile, 360 azimuths."""
import argparse
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import horayzon_amd as hz
from horayzon_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=3601)
ap.add_argument("--win", type=int, default=512)
ap.add_argument("--azim", type=int, default=360)
ap.add_argument("--dist", type=float, default=50.0)
ap.add_argument("--alg", default="guess_constant")
ap.add_argument("--count", action="store_true")
ap.add_argument("--count-all", action="store_true")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--top", type=int, default=-1)
ap.add_argument("--regroup", type=int, default=-1)
ap.add_argument("--hit-cache", type=int, default=1)
ap.add_argument("--stack", type=int, default=0)
ap.add_argument("--no-near", action="store_true", help="switch the near-field certificates off")
ap.add_argument("--verify-near", action="store_true")
ap.add_argument("--left", type=lambda x: int(x, 0), default=0, help="hz_opts.left_min: byte l = hand-over threshold of level l (0: default, -1: off)")
ap.add_argument("--pgrid", type=int, default=0, help="hz_opts.persist_grid")
ap.add_argument("--tune", type=lambda x: int(x, 0), default=0, help="hz_opts.left_tune")
ap.add_argument("--verify-sample", type=int, default=0, help="production kernel with the sampled certificate check: one of every N shortened rays")
args = ap.parse_args()

t = time.time()
g = synth.fractal_tile(n=args.n, offset=16)
print("synth %.1fs" % (time.time() - t), flush=True)
n, w = args.n, args.win
off = (n - w) // 2
vec_norm, vec_north = synth.planar_frames(w, w)
t = time.time()
hz.horizon.schedule_overrides["left_tune"] = args.tune
sc = hz.Scene.create(g["vert_grid"], n, n)
print("scene create %.2fs" % (time.time() - t), json.dumps(sc.stats), flush=True)
for rep in range(args.reps):
    t = time.time()
    hori, azim = hz.horizon.horizon_gridded(g["vert_grid"], n, n, vec_norm, vec_north, off, off,
                                            args.dist, azim_num=args.azim, ray_algorithm=args.alg,
                                            scene=sc, _top_nodes=args.top, _regroup=args.regroup, _hit_cache=args.hit_cache, _near_skip=not args.no_near, _level_stack=(-args.stack if args.stack > 0 else False), _verify_near=(args.verify_sample if args.verify_sample else args.verify_near), _left_min=args.left, _persist_grid=args.pgrid, count_work=args.count_all or (args.count and rep == args.reps - 1))
    st = hz.horizon.last_stats
    print("rep %d wall %.2fs kernel %.3fs rays %d rays/(cell*az) %.2f Mray/s %.1f cells/s %.0f nodes/ray %.1f tris/ray %.1f"
          % (rep, time.time() - t, st["t_kernel_s"], st["num_rays"], st["num_rays"] / (w * w * args.azim),
             st["num_rays"] / st["t_kernel_s"] / 1e6, w * w / st["t_kernel_s"],
             st["nodes_visited"] / max(st["num_rays"], 1), st["tris_tested"] / max(st["num_rays"], 1)), flush=True)
    print("      stack redo blocks %d  fallbacks %d  left %.4fs cells %d again %d scratch %.0f MB" % (st.get("stack_redo_blocks", -1), st.get("stack_fallbacks", -1), st.get("t_left_s", 0.0), st.get("left_cells", 0), st.get("left_again", 0), st.get("scratch_bytes", 0) / 1e6), flush=True)
    print("      near pre-pass %.4fs  rays shortened %.3f  violations %d  re-traced %d" % (st["t_near_s"], st["rays_shortened"] / max(st["num_rays"], 1), st["near_violations"], st["near_verified"]), flush=True)
if args.count or args.count_all:
    print("SIMT efficiency: node step %.3f  leaf step %.3f  refill %.3f   (wave iters: node %.3g leaf %.3g refill %.3g)"
          % (st["nodes_visited"] / max(64 * st["wave_node_iters"], 1), st["tris_tested"] / 2 / max(64 * st["wave_leaf_iters"], 1),
             st["num_rays"] / max(64 * st["wave_refills"], 1), st["wave_node_iters"], st["wave_leaf_iters"], st["wave_refills"]))
print("hori range deg", np.rad2deg(np.nanmin(hori)), np.rad2deg(np.nanmax(hori)))
