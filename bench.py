#!/usr/bin/env python
"""bench.py -- horizon_gridded throughput on MI355X (BASELINE.json metric).

--workload c3 (default; the N = 1 headline): BASELINE.json config 3 -- 3601 x 3601 synthetic
SRTM-like tile (1 arc-second spacing at 46 deg N, seeded fractal, integer metres), 16-cell ring,
360 azimuth sectors, guess_constant, dist_search 50 km, hori_acc 0.25 deg, with the horizon array AND
the fused sky view factor written to HBM.  A "step" is one pass of the hot path over one batch of
grid cells: by default the whole inner domain of the tile in ONE launch (3569 x 3569 cells x 360 azimuths =
12.7 M cells, 4.6e9 output values = 18.3 GB, ~9.9e9 rays) against the full-tile LBVH; `--rows-per-step R` makes
a step a slab of R inner-domain rows instead (consecutive steps take consecutive slabs, wrapping; the last slab
is ragged).  One launch per tile is the default because every launch ends with a tail in which the last
workgroups run on a draining GPU (a lane owns one cell for all 360 azimuths, ~50 ms): 7 slabs of 512 rows pay it
7 times (5.63 M cells/s against 6.06 M for the whole tile, same kernel, same box).  Inputs (scene blob, per-cell
frames, mask, tilt) are resident in HBM before the timed region and outputs stay in HBM; the library is called
through its C ABI with device pointers (slab-local output buffers, opts.hori_is_slab).
N > 1 (one rank per GPU over RCCL; `python bench.py --gpus N` starts the N ranks itself when it is not already running
under torch.distributed.run, and fails loudly when the node has fewer than N GPUs): STRONG scaling of the same tile --
rank 0 builds the scene and broadcasts it over xGMI once (set-up, untimed like the BVH build; `--bcast blob` sends the
finished blob, `--bcast verts` the 12 V bytes of vertices and every rank rebuilds the LBVH), the inner rows are split
into one contiguous slab per rank (dist.row_slabs through dist.sharded_rows: the grid-cell shard SURVEY 8e names), every
rank traces its slab into its own resident horizon + SVF buffers (no data-path collective) and the SVF slabs are
gathered on rank 0 inside the timed region.  N = 1 is the plain one-launch line above, so the N = 1, 2, 4, 8 values form
one strong-scaling curve.  (`--scaling weak`: the old replica mode -- every rank computes the whole tile.)

--workload c5 (BASELINE.json config 5, strong scaling): the 4 x 4 mosaic (14401 x 14401, 206 M cells),
SVF-fused (the 298 GB horizon is never materialised), inner rows split by dist.row_slabs over WORLD_SIZE
ranks after ONE broadcast of the scene blob; each rank computes its slab (chunked inside the library),
the SVF is gathered on rank 0.  The timed region is the whole sharded job; `--steps` repeats it.
Reports cells/s, scene_bcast_s and the measured load imbalance (slowest rank / mean rank).

c5 is built for the first real 8-GPU run: only rank 0 synthesises the mosaic; the blob is broadcast straight out of
its allocation (no second copy); every rank derives the per-cell inputs of ITS slab on the device from the vertex
array inside the blob and passes them slab-local (opts.inputs_are_slab); the slabs are balanced by COST -- a sampled
pre-pass of one-row counting launches, split over the ranks (dist.estimate_row_cost) -- and both the predicted and the
measured load imbalance are reported.  `--emulate-ranks R` (one GPU) computes the R slabs of such a partition one
after the other and reports their times: the load balance an R-GPU run would see, measured without R GPUs.

--workload c4 (BASELINE.json config 4): Terrain.shadow over 144 diurnal sun positions of the c3 tile, outputs resident
in HBM; a step = one pass over all sun positions; its own roofline (shadow kernel).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NODE_BYTES = 32.0       # one 4-wide LBVH node (hz_common.h: Node; 64 B until round 3, 48 B early in round 4): the bytes a node visit reads
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 achievable
# The counter files under profiles/ (traffic.json, valu_model.json, valu_class_mix.json) describe MACHINE CODE: they carry the hash of
# the normalised gfx950 assembly of the traversal kernels (scripts/kernel_asm.py), which the build leaves in horayzon_amd/kernel_asm.sha.

# wave-level VALU instructions per wave iteration of k_horizon<guess_constant> (calibrated against
# SQ_INSTS_VALU of the PMC pass, profiles/<round>/valu_model.json)
VALU_MODEL_DEFAULT = {"node_iter": 147.0, "leaf_iter": 218.0, "refill_iter": 160.0, "per_cell": 0.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=("c3", "c4", "c5"), default="c3")
    ap.add_argument("--scaling", choices=("auto", "strong", "weak"), default="auto",
                    help="c3 with several ranks: strong (default) = one row slab of the tile per rank; weak = every rank "
                         "computes the whole tile (replicas)")
    ap.add_argument("--bcast", choices=("blob", "verts"), default="blob",
                    help="several ranks: broadcast the finished scene blob (vertices + LBVH), or only the 12 V bytes of "
                         "vertices and rebuild the LBVH on every rank")
    ap.add_argument("--no-e2e", action="store_true", help="c3: skip the untimed NumPy-in / NumPy-out call of horizon_gridded")
    ap.add_argument("--no-c5-extra", action="store_true",
                    help="c3, one rank: skip the config-5 job (14401^2 mosaic, ~2 min incl. synthesis) among the untimed extras")
    ap.add_argument("--c5-tile", type=int, default=14401, help="--gpus N > 1: DEM size of the config-5 job run over the same ranks after the c3 steps (extras.c5)")
    ap.add_argument("--dump-c5-path", default="", help="--gpus N > 1: .npy file for the gathered SVF of the extras.c5 job (tests)")
    ap.add_argument("--no-extras", action="store_true",
                    help="c3, one rank: skip the untimed extras (whole-tile binary_search / discrete_sampling, the curved tile, c4)")
    ap.add_argument("--suns", type=int, default=144, help="c4: sun positions per step")
    ap.add_argument("--locations", type=int, default=1000000, help="extras: number of random locations of the horizon_locations line")
    ap.add_argument("--refrac", type=int, default=0, help="c4: atmospheric refraction on (1) / off (0)")
    ap.add_argument("--which", choices=("shadow", "sw_dir_cor"), default="shadow", help="c4: output kind")
    ap.add_argument("--balance", choices=("cost", "cells"), default="cells",
                    help="c5: row slabs balanced by cell count (default) or by the sampled cost pre-pass (measured on the "
                         "synthetic mosaic, 8 emulated ranks: 1.028 against 1.033 max/mean, less than the pre-pass costs)")
    ap.add_argument("--plain-fraction", type=float, default=0.0,
                    help="sharded runs: scale the relief of this share of the DEM's rows (the northern ones) down to 3 %% -- an "
                         "inhomogeneous DEM on which balancing the row slabs by cell count and by sampled cost differ")
    ap.add_argument("--cost-samples", type=int, default=0, help="c5: probe rows of the cost pre-pass (0: max(16, 4 x ranks))")
    ap.add_argument("--emulate-ranks", type=int, default=0, help="c5, one GPU: time the slabs of an R-rank partition one by one")
    ap.add_argument("--dump-svf-rows", default="", help="sharded runs: comma separated inner-domain rows of the gathered SVF to save ('all': every row)")
    ap.add_argument("--dump-path", default="", help="sharded runs: .npy file for --dump-svf-rows")
    ap.add_argument("--rows-per-step", type=int, default=0, help="inner-domain rows per step (0: the whole tile in one launch)")
    ap.add_argument("--tile", type=int, default=None, help="DEM size (3601 for c3, 14401 for c5)")
    ap.add_argument("--azim", type=int, default=360)
    ap.add_argument("--dist-search", type=float, default=50.0)
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the tile the CPU baseline computes (0: cores / 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count", action="store_true", help="skip the counter pass (roofline is I/O only)")
    ap.add_argument("--no-peaks", action="store_true", help="skip the machine calibration kernels")
    return ap.parse_args()


def kernel_asm_sha():
    """Hash of the traversal kernels' device assembly as built (horayzon_amd/kernel_asm.sha, written by the Makefile next to the
    library); computed with hipcc when that file is missing; "unknown" without either."""
    try:
        with open(os.path.join(ROOT, "horayzon_amd", "kernel_asm.sha")) as f:
            v = f.read().strip()
        if len(v) == 64:
            return v
    except OSError:
        pass
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import kernel_asm
        return kernel_asm.sha(ROOT)
    except Exception:
        return "unknown"


def load_valu_model():
    """(instructions per wave iteration, fast-class share per section, notes): profiles/valu_model.json and
    profiles/valu_class_mix.json when they were measured for the present kernel sources, else the built-in constants."""
    model, mix = dict(VALU_MODEL_DEFAULT), dict(CLASS_MIX_DEFAULT)
    notes = {"valu_model": "built-in constants", "class_mix_note": "built-in class mix"}
    sha = kernel_asm_sha()
    try:
        mj = json.load(open(os.path.join(ROOT, "profiles", "valu_model.json")))
        if mj.get("kernel_asm_sha") == sha:
            model.update({k: mj[k] for k in model if k in mj})
            if "shadow_setup_winst_per_64_cells" in mj:
                model["shadow_setup"] = mj["shadow_setup_winst_per_64_cells"]
            notes["valu_model"] = "profiles/valu_model.json (calibrated on SQ_INSTS_VALU of this machine code)"
        else:
            notes["valu_model"] = "built-in constants (profiles/valu_model.json was measured for other machine code)"
    except Exception:
        pass
    try:
        cj = json.load(open(os.path.join(ROOT, "profiles", "valu_class_mix.json")))
        if cj.get("kernel_asm_sha") == sha:
            mix = {k: cj[k]["fast_fraction"] for k in mix if k in cj}
            notes["class_mix_note"] = "profiles/valu_class_mix.json (scripts/isa_class_mix.py, this machine code)"
    except Exception:
        pass
    return model, mix, notes


def machine_peaks(L, dev_index):
    """VALU issue ceiling (wave-level instructions / s, all SIMDs) and float4 copy bandwidth, measured here."""
    best, clk, simds = 0.0, C.c_double(0), C.c_int(0)
    for w in (1, 2):         # bursts of 3.5 / 7 ms at 8 waves per SIMD (longer pure-FMA runs are power-limited)
        r = C.c_double(0)
        if L.hz_debug_valu_peak(dev_index, 0, w, C.byref(r), C.byref(clk), C.byref(simds)) == 0:
            best = max(best, r.value)
    g = C.c_double(0)
    L.hz_debug_copy_peak(dev_index, 1 << 30, C.byref(g))
    # the ceiling: one wave64 VALU instruction per SIMD every 4 cycles at the engine clock; the FMA burst confirms
    # it to 1.5 % (4.06 cycles) and is reported next to it
    return {"valu_winst_per_s": clk.value * 1e9 / 4.0 * simds.value, "valu_winst_per_s_measured": best * simds.value,
            "simds": simds.value, "clock_ghz": clk.value,
            "cycles_per_wave_inst_measured": (clk.value * 1e9 / best) if best else None, "copy_gbs": g.value}


def inst_class_rates(L, dev_index):
    """Cycles per wave64 VALU instruction per SIMD of the two issue classes of gfx950, measured here (hz_bench.hip:
    k_inst_rate; selectors 17 / 19 / 11 = v_fma_f32 with sources in different VGPR banks, v_add_f32, v_mul_f32;
    1 / 3 / 15 / 9 = v_cvt_f32_ubyte0, v_perm_b32, v_min_f32, v_cmp_le_f32).  Two passes, the faster one counts (the first
    also warms the clocks)."""
    def rate(op):
        best = 1e9
        for _ in range(2):
            r = C.c_double(0)
            if L.hz_debug_inst_rate(dev_index, op, C.byref(r)) == 0 and r.value > 0:
                best = min(best, r.value)
        return best
    fast = [rate(op) for op in (17, 19, 11)]
    slow = [rate(op) for op in (1, 3, 15, 9)]
    # Each class is priced at the MEAN of its samples.  Round 4 priced it at the fastest sample so that the
    # model stayed <= 1 -- a yardstick re-chosen when it is exceeded is not a ceiling.  The model is reported RAW
    # (frac_model_raw, may exceed 1: then its instruction mix or its rates are off by that much) next to the counter-only
    # floor frac_valu_counter_floor (SQ_INSTS_VALU x 2 cycles, the SIMD-32 issue rate of the guide).
    return {"fast_cycles": sum(fast) / len(fast), "slow_cycles": sum(slow) / len(slow),
            "fast_cycles_min": min(fast), "slow_cycles_min": min(slow),
            "fast_same_bank_cycles": rate(0), "fast_samples": fast, "slow_samples": slow}


SHADOW_SETUP_WINST = (485.0, 1100.0)    # wave-level VALU instructions per 64 cells handed out: refraction off (calibrated on
                                        # SQ_INSTS_VALU, profiles/r03/pmc_shadow_refill.json) / on (estimate)
CLASS_MIX_DEFAULT = {"node_step": 0.44, "leaf_step": 0.88, "refill_and_loop_overhead": 0.62}   # fast-class share (ISA count)


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks (one per GPU) and pass their output
    through; rank 0 prints the JSON line."""
    import socket
    import subprocess
    import torch
    backend = os.environ.get("HZ_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    if have < args.gpus and backend != "gloo":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node (RCCL needs one GPU per rank; "
                         "HZ_DIST_BACKEND=gloo lets several ranks share a GPU for testing)" % (args.gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    backend = os.environ.get("HZ_DIST_BACKEND", "nccl")
    # (several ranks may share a GPU when the multi-rank code path is exercised on a box with fewer GPUs than ranks:
    #  HZ_DIST_BACKEND=gloo, since RCCL refuses two ranks on one device)
    if world > torch.cuda.device_count() and backend != "gloo":
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (one GPU per rank over RCCL)" % (world, torch.cuda.device_count()))
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # HZ_FORCE_DIST=1 runs the multi-rank code path (RCCL broadcast of the scene, gather, all-reduce)
    # even with a single rank -- used to exercise it on a 1-GPU box
    # (config 5 always runs the sharded code path: one rank is simply the N = 1 point of its scaling curve)
    use_dist = world > 1 or bool(os.environ.get("HZ_FORCE_DIST")) or args.workload == "c5"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda:%d" % local_rank))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()          # n_gpus of the line = the ranks the process group actually formed
    ctx = dict(args=args, world=world, rank=rank, local_rank=local_rank, use_dist=use_dist,
               dev="cuda:%d" % local_rank)
    strong_c3 = args.workload == "c3" and use_dist and args.scaling != "weak"
    fn = {"c3": (lambda c: run_sharded(c, "c3")) if strong_c3 else run_c3, "c4": run_c4,
          "c5": lambda c: run_sharded(c, "c5")}[args.workload]
    out = fn(ctx)
    if strong_c3 and world > 1 and not args.no_extras and not args.no_c5_extra:
        # BASELINE config 5 (the 4 x 4 mosaic, SVF-fused) over the SAME ranks: the config north_star names for 1 / 2 / 4 / 8 scaling.
        # `value` stays the c3 shard (its N = 1 point is the plain bench line); the c5 job is one untimed-by-the-contract step
        # with its own clock (barrier + synchronize on both sides, max over ranks: run_sharded).
        import copy
        a5 = copy.copy(args)
        a5.steps, a5.warmup, a5.tile, a5.emulate_ranks = 1, 1, args.c5_tile, 0
        a5.dump_svf_rows, a5.dump_path = ("all", args.dump_c5_path) if args.dump_c5_path else ("", "")
        torch.cuda.empty_cache()
        t5 = time.perf_counter()
        o5 = run_sharded(dict(ctx, args=a5), "c5")
        if rank == 0 and out is not None:
            c = o5["config"]
            out.setdefault("extras", {})["c5"] = {
                "metric": o5["metric"], "cells_per_s": o5["value"], "mray_per_s": o5["mray_per_s"], "n_gpus": o5["n_gpus"],
                "scaling": o5["scaling"], "job_s": o5["ms_per_step"] * 1e-3, "job_s_incl_bcast": c["job_s_incl_bcast"],
                "cells_per_s_incl_bcast": c["cells_per_s_incl_bcast"], "scene_bcast_s": c["scene_bcast_s"],
                "scene_bcast_bytes": c["scene_bcast_bytes"], "scene_bytes": c["scene_bytes"], "bvh_build_s": c["bvh_build_s"],
                "slabs": c["slabs"], "t_ranks_s": c["t_ranks_s"], "load_imbalance_max_over_mean": c["load_imbalance_max_over_mean"],
                "load_imbalance_predicted": c["load_imbalance_predicted"] if c["load_imbalance_predicted"] is not None else
                    (max(e - b for b, e in c["slabs"]) * len(c["slabs"]) / max(sum(e - b for b, e in c["slabs"]), 1) if c["slabs"] else None),
                "kernel_s_rank0": c["kernel_s_rank0"],
                "gathered_svf_finite": c["gathered_svf_finite"], "workload": c["workload"],
                "wall_s_incl_synthesis": time.perf_counter() - t5,
                "note": "BASELINE config 5 over the same %d ranks, after the c3 steps: one step = the whole inner domain, row slabs "
                        "balanced by %s; outside `value`" % (world, "sampled cost" if args.balance == "cost" else "cell count")}
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio: flush that first so that the JSON line is the last line
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        if args.gpus != world:
            out["config"]["gpus_flag"] = args.gpus        # the flag and the launcher disagreed: n_gpus is what ran
        print(json.dumps(out), flush=True)


def make_scene(ctx, g, n):
    """Rank 0 builds the scene; with several ranks it is broadcast once (RCCL over xGMI): the finished blob straight out of
    its allocation (`--bcast blob`), or the vertex array alone, after which every rank builds its own LBVH (`--bcast verts`:
    12 V bytes instead of ~85 V on the links, one 0.02 - 0.2 s build per rank).  Returns (scene, stats of the build on rank 0,
    wall seconds of the build on rank 0, wall seconds of broadcast [+ per-rank rebuild], bytes broadcast)."""
    import torch
    import torch.distributed as dist
    import horayzon_amd as hz
    from horayzon_amd.dist import broadcast_scene, broadcast_blob
    args = ctx["args"]
    verts_mode = ctx["use_dist"] and args.bcast == "verts"
    t0 = time.time()
    scene = None
    if ctx["rank"] == 0 and not verts_mode:                                  # g: rank 0 only
        scene = hz.Scene.create(g["vert_grid"], n, n, device=ctx["local_rank"])
    t_build = time.time() - t0
    scene_stats = scene.stats if scene is not None else None
    t_bcast, n_bytes = 0.0, 0
    if ctx["use_dist"]:
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.time()
        if verts_mode:
            gloo = dist.get_backend() == "gloo"
            buf = None
            if ctx["rank"] == 0:
                buf = torch.from_numpy(g["vert_grid"]).view(torch.uint8)
                buf = buf if gloo else buf.to(ctx["dev"])
            buf = broadcast_blob(buf, buf.numel() if buf is not None else 0, "cpu" if gloo else ctx["dev"], src=0)
            n_bytes = int(buf.numel())
            tb = time.time()
            scene = hz.Scene.create(buf, n, n, device=ctx["local_rank"])   # host (gloo) or device pointer: used in place
            torch.cuda.synchronize()
            if ctx["rank"] == 0:
                scene_stats, t_build = scene.stats, time.time() - tb
            del buf
        else:
            scene = broadcast_scene(scene, ctx["local_rank"], src=0)
            n_bytes = int(scene.blob()[1])
        torch.cuda.synchronize(); dist.barrier()
        t_bcast = time.time() - t0
    ctx["bcast_bytes"] = n_bytes
    return scene, scene_stats, t_build, t_bcast


def barrier(ctx):
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize()
    if ctx["use_dist"]:
        dist.barrier()
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------
# config 3: slab steps on the 3601^2 tile
# ------------------------------------------------------------------------------------------------
def run_c3(ctx):
    import torch
    import torch.distributed as dist
    from horayzon_amd import _lib, synth
    from horayzon_amd.dist import gather_rows
    args, rank, world, dev = ctx["args"], ctx["rank"], ctx["world"], ctx["dev"]
    L = _lib.lib()
    n, off, A = args.tile or 3601, 16, args.azim
    g = synth.fractal_tile(n=n, offset=off)
    in0 = in1 = n - 2 * off
    rps = args.rows_per_step if 0 < args.rows_per_step < in0 else in0
    scene, scene_stats, t_build, t_bcast = make_scene(ctx, g, n)
    blob_ptr, blob_bytes = scene.blob()
    n_slabs = (in0 + rps - 1) // rps                 # 1 by default; 7 with --rows-per-step 512 (six of 512 rows + 497)
    steps = args.steps if args.steps is not None else max(n_slabs, 3)
    warmup = args.warmup if args.warmup is not None else 1

    # per-cell inputs resident in HBM
    vec_tilt_h, _ = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], off)
    d_norm = torch.zeros((in0, in1, 3), dtype=torch.float32, device=dev); d_norm[..., 2] = 1.0
    d_north = torch.zeros((in0, in1, 3), dtype=torch.float32, device=dev); d_north[..., 1] = 1.0
    d_mask = torch.ones((in0, in1), dtype=torch.uint8, device=dev)
    d_tilt = torch.from_numpy(vec_tilt_h).to(dev)
    d_hori = torch.empty((rps, in1, A), dtype=torch.float32, device=dev)       # reused slab buffer
    d_svf = torch.full((in0, in1), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    opts = _lib.hz_opts()
    opts.device = ctx["local_rank"]
    opts.top_nodes = -1
    opts.regroup = -1
    opts.vec_tilt = d_tilt.data_ptr()
    opts.hori_is_slab = 1                     # d_hori and the svf pointer below address the slab's first row
    stats = _lib.hz_stats()

    def slab_of(v):
        s = v % n_slabs
        return s * rps, min(s * rps + rps, in0)

    def step(v, st, count=False):
        rb, re = slab_of(v)
        opts.row_begin, opts.row_end = rb, re
        opts.count_work = int(count)
        opts.svf = d_svf.data_ptr() + 4 * rb * in1
        rc = L.hz_horizon_gridded_scene(scene._h, d_norm.data_ptr(), d_north.data_ptr(), off, off,
                                        d_hori.data_ptr(), in0, in1, A, args.dist_search, 0.25, b"guess_constant",
                                        -15.0, d_mask.data_ptr(), 0.0, 0.01, C.byref(opts), C.byref(st))
        _lib.check(rc)
        return rb, re

    # ---- counter pass (untimed): wave-level work of one slab for the roofline ----------------------
    cw = None
    if rank == 0 and not args.no_count:
        cw = _lib.hz_stats()
        step(n_slabs // 2, cw, count=True)
    peaks = machine_peaks(L, ctx["local_rank"]) if (rank == 0 and not args.no_peaks) else None
    if peaks is not None:
        peaks["class_rates"] = inst_class_rates(L, ctx["local_rank"])

    # weak scaling: rank r takes steps r K ... r K + K - 1 of the slab sequence (per-GPU work fixed)
    base = rank * steps
    for w in range(warmup):
        step(base + w, _lib.hz_stats())
    barrier(ctx)
    stats = _lib.hz_stats()
    t0 = time.perf_counter()
    for s in range(steps):
        rb, re = step(base + warmup + s, stats)
    if ctx["use_dist"] and steps > 0:    # final gather of the SVF rows of every rank's last step (4 B / cell)
        last = torch.zeros((rps, in1), dtype=torch.float32, device=dev)      # (ranks may sit on slabs of different length)
        last[:re - rb] = d_svf[rb:re]
        gather_rows(last, [(0, rps)] * world, dst=0)
    barrier(ctx)
    elapsed = time.perf_counter() - t0
    if ctx["use_dist"]:
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
    if rank != 0:
        return None

    k_launch_s = stats.t_kernel_s / max(steps, 1)        # HIP events on the kernel's stream
    rays_launch = stats.num_rays / max(steps, 1)
    cells_launch = stats.num_cells / max(steps, 1)
    out = {
        "metric": "grid_cells_per_s (horizon_gridded, 360 azimuths, 3601^2 SRTM-like tile)",
        "value": cells_total / elapsed,
        "unit": "cells/s",
        "mray_per_s": rays_total / elapsed / 1e6,
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / max(steps, 1),
        "higher_is_better": True, "scaling": "weak" if (world > 1 or args.scaling == "weak") else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "c3: horizon_gridded guess_constant + fused SVF, %dx%d synthetic SRTM-like tile, "
                               "%d azimuths, dist_search %g km, %s"
                               % (n, n, A, args.dist_search,
                                  "the whole inner domain (%d rows) per step, one launch" % in0 if n_slabs == 1 else
                                  "slabs of <= %d rows per step (%d slabs cover the tile)" % (rps, n_slabs)),
                   "rows_per_step": rps, "cells_per_step": cells_launch, "rays_per_cell_azimuth": rays_launch / max(cells_launch * A, 1),
                   "parallelism": ("one rank: the N = 1 point of the strong-scaling curve (N > 1: one row slab of this tile per rank)"
                                   if world == 1 and args.scaling != "weak" else
                                   "weak scaling x%d (--scaling weak, replicas): same step sequence, rank r starts at step r K; "
                                   "scene broadcast once" % world),
                   "bvh_build_s": scene_stats["t_bvh_s"] if scene_stats else None,
                   "scene_bytes": int(blob_bytes), "scene_bcast_s": t_bcast, "scene_create_wall_s": t_build,
                   # HBM scratch of a step besides the scene and the resident inputs / outputs: near-field certificates ((2 A + 4) B per cell
                   # of the launch), records of the handed-over cells + their sort buffers (hz_stats.scratch_bytes)
                   "scratch_bytes": int(stats.scratch_bytes), "leftover_cells_per_step": stats.left_cells / max(steps, 1),
                   "leftover_groups_repeated": int(stats.left_redo_groups),
                   "load_imbalance_max_over_mean": imbalance, "near_prepass_ms_per_step": 1e3 * stats.t_near_s / max(steps, 1),
                   # the reference's searches do not terminate where these fire (horizon_comp.cpp:474-488, README.md:209-213):
                   # rim cells whose rays leave the DEM below the lowest table angle; this library stops them and counts
                   "guard_events_per_step": stats.guard_events / max(steps, 1),
                   "guard_cells_per_step": stats.guard_cells / max(steps, 1),
                   "stack_fallbacks": int(stats.stack_fallbacks), "stack_redo_blocks": int(stats.stack_redo_blocks),
                   "height_field": int(stats.height_field), "near_certificates_used": int(stats.near_used),
                   "device": torch.cuda.get_device_name(ctx["local_rank"]), "rocm": getattr(torch.version, "hip", None)},
        "roofline": roofline(args, stats, steps, cw, peaks, A, n, rps),
    }
    # the committed counter files were measured on one box with one ROCm: say so when this run is on another
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj.get("rocm") and (tj.get("rocm") != out["config"]["rocm"] or tj.get("device") != out["config"]["device"]):
            out["roofline"]["traffic_note"] += " -- measured with ROCm %s on %s, this run: %s on %s" % (
                tj.get("rocm"), tj.get("device"), out["config"]["rocm"], out["config"]["device"])
    except Exception:
        pass
    if args.dump_path and args.dump_svf_rows == "all" and n_slabs == 1:
        np.save(args.dump_path, d_svf.cpu().numpy())
    if world == 1 and not args.no_extras and n_slabs == 1:
        out["extras"] = c3_extras(ctx, L, scene, step_args=dict(d_norm=d_norm, d_north=d_north, d_mask=d_mask, d_tilt=d_tilt,
                                                                 d_hori=d_hori, d_svf=d_svf, in0=in0, in1=in1, off=off, A=A, n=n),
                                  peaks=peaks, g=g)
    if world == 1 and not args.no_extras and not args.no_c5_extra and n_slabs == 1 and n == 3601:
        del d_hori
        torch.cuda.empty_cache()
        out["extras"]["c5"] = c5_extra()
        d_hori = None
    if world == 1 and not args.no_e2e:
        # free the resident buffers of the timed region first: the drop-in call allocates its own
        del d_svf, d_tilt, d_norm, d_north, d_mask
        d_hori = None
        torch.cuda.empty_cache()
        out["config"]["e2e_numpy_call"] = e2e_numpy_call(g, vec_tilt_h, args, A)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(g, args, A)
    return out


def curved_tile_device(L, torch, n, off, dev_index, seed_elev):
    """The CURVED variant of config 3 (examples/horizon/gridded_curved_DEM.py: lon / lat / elevation on the WGS84 ellipsoid
    -> ECEF -> ENU; per-cell normal and north vectors are NOT axis aligned), prepared entirely on the device through the
    library's input-side entry points (hz_lonlat2ecef ... hz_slope_plane_meth)."""
    import horayzon_amd as hz
    from horayzon_amd import _lib
    dev = "cuda:%d" % dev_index
    lon = 8.0 + np.arange(n) / 3600.0
    lat = 47.0 - np.arange(n) / 3600.0
    elev = torch.from_numpy(seed_elev).to(dev)
    lon2 = torch.from_numpy(lon).to(dev)[None, :].expand(n, n).contiguous()
    lat2 = torch.from_numpy(lat).to(dev)[:, None].expand(n, n).contiguous()
    nn = n * n
    f64 = lambda: torch.empty(nn, dtype=torch.float64, device=dev)
    f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    X, Y, Z = f64(), f64(), f64()
    _lib.check(L.hz_lonlat2ecef(lon2.data_ptr(), lat2.data_ptr(), elev.data_ptr(), nn, 2, X.data_ptr(), Y.data_ptr(), Z.data_ptr(), dev_index))
    lon_or, lat_or = float(lon.mean()), float(lat.mean())
    xe, ye, ze = f32(n, n), f32(n, n), f32(n, n)
    _lib.check(L.hz_ecef2enu(X.data_ptr(), Y.data_ptr(), Z.data_ptr(), nn, lon_or, lat_or, 2, xe.data_ptr(), ye.data_ptr(), ze.data_ptr(), dev_index))
    vert_grid = hz.auxiliary.rearrange_pad_buffer(xe, ye, ze)
    sl = (slice(off, n - off), slice(off, n - off))
    in0 = n - 2 * off
    lon_i, lat_i = lon2[sl].contiguous(), lat2[sl].contiguous()
    Xi, Yi, Zi = (a.view(n, n)[sl].contiguous() for a in (X, Y, Z))
    vn_e, vno_e, vec_norm, vec_north = f32(in0, in0, 3), f32(in0, in0, 3), f32(in0, in0, 3), f32(in0, in0, 3)
    m = in0 * in0
    _lib.check(L.hz_surf_norm(lon_i.data_ptr(), lat_i.data_ptr(), m, vn_e.data_ptr(), dev_index))
    _lib.check(L.hz_north_dir(Xi.data_ptr(), Yi.data_ptr(), Zi.data_ptr(), vn_e.data_ptr(), m, 2, vno_e.data_ptr(), dev_index))
    _lib.check(L.hz_ecef2enu_vector(vn_e.data_ptr(), m, lon_or, lat_or, 2, vec_norm.data_ptr(), dev_index))
    _lib.check(L.hz_ecef2enu_vector(vno_e.data_ptr(), m, lon_or, lat_or, 2, vec_north.data_ptr(), dev_index))
    tilt_full = f32(n, n, 3)
    _lib.check(L.hz_slope_plane_meth(xe.data_ptr(), ye.data_ptr(), ze.data_ptr(), n, n, None, 0, tilt_full.data_ptr(), dev_index))
    vec_tilt = tilt_full[sl].contiguous()
    return vert_grid, vec_norm, vec_north, vec_tilt


def c3_extras(ctx, L, scene, step_args, peaks, g):
    """UNTIMED additions to the one driver-run line (none of this is part of `value`): the other two
    search algorithms over the WHOLE tile (horizon_comp.cpp:302-381), the curved variant of the tile with device-prepared
    frames (SURVEY 8d names both variants) and the config-4 step (shadow masks of 144 sun positions, with and without
    atmospheric refraction).  Each with its own counter pass on a slab where a VALU figure is given."""
    import torch
    import horayzon_amd as hz
    from horayzon_amd import _lib, synth
    args, dev = ctx["args"], ctx["dev"]
    a = step_args
    in0, in1, off, A, n = a["in0"], a["in1"], a["off"], a["A"], a["n"]
    res = {"note": "untimed extras, outside `value`; same scene, same resident buffers, one launch over the whole inner domain each"}

    def whole_tile(sc, norm, north, tilt, alg, rows=None, count=False):
        opts = _lib.hz_opts()
        opts.device = ctx["local_rank"]; opts.top_nodes = -1; opts.regroup = -1
        opts.vec_tilt = tilt.data_ptr(); opts.hori_is_slab = 1
        rb, re = rows if rows else (0, in0)
        opts.row_begin, opts.row_end = rb, re
        opts.count_work = int(count)
        opts.svf = a["d_svf"].data_ptr() + 4 * rb * in1
        st = _lib.hz_stats()
        _lib.check(L.hz_horizon_gridded_scene(sc._h, norm.data_ptr(), north.data_ptr(), off, off, a["d_hori"].data_ptr(), in0, in1, A,
                                              args.dist_search, 0.25, alg.encode(), -15.0, a["d_mask"].data_ptr(), 0.0, 0.01,
                                              C.byref(opts), C.byref(st)))
        return st

    def line(st, cw):
        d = {"cells_per_s": st.num_cells / st.t_kernel_s, "kernel_ms": 1e3 * st.t_kernel_s,
             "cells_per_s_with_prepass_and_svf": st.num_cells / (st.t_kernel_s + st.t_near_s + st.t_svf_s),
             "rays_per_cell_azimuth": st.num_rays / max(st.num_cells * A, 1), "mray_per_s": st.num_rays / st.t_kernel_s / 1e6,
             "guard_events": int(st.guard_events), "near_certificates_used": int(st.near_used),
             "stack_redo_blocks": int(st.stack_redo_blocks)}
        if cw is not None:
            r = roofline(args, st, 1, cw, peaks, A, n, in0)
            d.update({k: r.get(k) for k in ("frac_8d_hbm_model", "frac_model_raw", "frac_valu_counter_floor", "frac_uniform_4_cycle",
                                            "nodes_per_ray", "tris_per_ray", "lane_utilisation_node_leaf_steps")})
            d["frac_note"] = ("as roofline.* of the headline kernel; wave-iteration counts from a counting launch over the "
                              "middle 256 rows, per-iteration instruction constants of the guess_constant calibration")
        return d

    mid = (in0 // 2 - 128, in0 // 2 + 128) if in0 > 512 else None
    for alg in ("binary_search", "discrete_sampling"):           # horizon_comp.cpp:302-381
        cw = whole_tile(scene, a["d_norm"], a["d_north"], a["d_tilt"], alg, rows=mid, count=True) if not args.no_count else None
        st = whole_tile(scene, a["d_norm"], a["d_north"], a["d_tilt"], alg)
        res[alg] = line(st, cw)
    # ---- curved variant: frames that are not axis aligned, everything prepared on the device --------------------------
    t0 = time.perf_counter()
    vert_grid, c_norm, c_north, c_tilt = curved_tile_device(L, torch, n, off, ctx["local_rank"], np.ascontiguousarray(g["z"], np.float32))
    torch.cuda.synchronize()
    t_prep = time.perf_counter() - t0
    sc_c = hz.Scene.create(vert_grid, n, n, device=ctx["local_rank"])
    cw = whole_tile(sc_c, c_norm, c_north, c_tilt, "guess_constant", rows=mid, count=True) if not args.no_count else None
    whole_tile(sc_c, c_norm, c_north, c_tilt, "guess_constant")
    st = whole_tile(sc_c, c_norm, c_north, c_tilt, "guess_constant")
    res["curved_c3_guess_constant"] = dict(line(st, cw), input_prep_on_device_s=t_prep, bvh_build_s=sc_c.stats["t_bvh_s"],
                                           height_field=int(st.height_field),
                                           note="3601^2 one-arc-second tile on the WGS84 ellipsoid (lon 8..9 E, lat 46..47 N, the "
                                                "same synthetic elevations), ENU frame at the tile centre; vec_norm / vec_north / "
                                                "vec_tilt from hz_surf_norm / hz_north_dir / hz_slope_plane_meth on the device")
    del sc_c, vert_grid, c_norm, c_north, c_tilt
    # ---- config 4 on the same scene -------------------------------------------------------------------------------------
    S = args.suns
    vec_tilt_h, enl = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], off)
    elev = np.ascontiguousarray(g["z"][off:off + in0, off:off + in1])
    suns, _, _ = synth.sun_positions(num=S)
    c4 = {}
    for refrac in (False, True):
        terrain = hz.shadow.Terrain(device=ctx["local_rank"])
        terrain.initialise(g["vert_grid"], n, n, off, off, vec_tilt_h, g["vec_norm"], enl, elev, np.ones((in0, in1), np.uint8),
                           refrac_cor=refrac, scene=scene)
        for which, fn, dt in (("shadow", terrain.shadow_batch, torch.uint8), ("sw_dir_cor", terrain.sw_dir_cor_batch, torch.float32)):
            o = torch.empty((S, in0, in1), dtype=dt, device=dev)
            cw4 = None
            if not args.no_count:       # counter pass (untimed): node visits / triangle tests / wave iterations of all positions
                terrain.count_work(True)
                fn(suns, o)
                cw4 = dict(terrain.last_stats)
                terrain.count_work(False)
            fn(suns, o)
            fn(suns, o)
            ks = terrain.last_stats["t_kernel_s"]
            r4 = c4_roofline(n, S, in0 * in1, which == "shadow", ks, cw4, refrac, peaks, (peaks or {}).get("class_rates"))
            c4["%s_refrac_%d" % (which, int(refrac))] = {"ms_per_sun_position": 1e3 * ks / S, "kernel_ms_per_step": 1e3 * ks,
                                                        "cells_per_s": S * in0 * in1 / ks,
                                                        "mray_per_s": terrain.last_stats["num_rays"] / ks / 1e6,
                                                        "roofline": {k: r4.get(k) for k in (
                                                            "bound", "achieved", "peak", "unit", "frac", "frac_8d_hbm_model", "binding_resource",
                                                            "frac_valu_counter_floor", "frac_model_raw", "frac_uniform_4_cycle", "nodes_per_ray",
                                                            "tris_per_ray", "lane_utilisation_node_leaf_steps", "valu_winst_per_step_model")}}
            del o
        del terrain
    c4["note"] = "Terrain.shadow_batch / sw_dir_cor_batch over %d diurnal sun positions in one launch, outputs resident in HBM" % S
    res["c4"] = c4
    # ---- what a strong-scaling run of this tile over N GPUs can reach: the N row slabs of `--gpus N`, one after the other on
    #      this GPU (each alone, exactly as on its own rank: certificates + kernel + SVF of the slab, GPU event times) ----------
    try:
        from horayzon_amd.dist import row_slabs
        def slab_ms(b, e):
            st = whole_tile(scene, a["d_norm"], a["d_north"], a["d_tilt"], "guess_constant", rows=(b, e))
            return 1e3 * (st.t_kernel_s + st.t_near_s + st.t_svf_s)
        whole_ms = slab_ms(0, in0)
        whole_ms = min(whole_ms, slab_ms(0, in0))
        pred = {"whole_tile_ms": whole_ms}
        for N in (2, 4, 8):
            ts = [slab_ms(b, e) for b, e in row_slabs(in0, N)]
            pred["n%d" % N] = {"slab_ms": [round(t, 2) for t in ts], "predicted_efficiency": whole_ms / (N * max(ts)),
                               "predicted_cells_per_s": in0 * in1 / (1e-3 * max(ts))}
        pred["note"] = ("one-GPU emulation of `bench.py --gpus N` (strong scaling of this tile): efficiency = whole-tile time / (N x the "
                        "slowest slab), compute only (certificate pre-pass + kernel + SVF); a real run adds the scene broadcast "
                        "(job_s_incl_bcast) and the SVF gather.  The loss is the launch tail: a lane owns its cell for all 360 azimuths, "
                        "a wave lives ~33 ms, and a 1/8-tile launch is only ~5 wave generations deep (DESIGN.md section 8)")
        res["scaling_prediction"] = pred
    except Exception as e:
        res["scaling_prediction"] = {"error": repr(e)[:300]}
    # ---- horizon_locations on the same scene (horizon_comp.cpp:828-1094) -------------------------------
    try:
        res["locations"] = locations_extra(hz, scene, g, n, args, ctx["local_rank"])
    except Exception as e:          # the extras never fail the headline line
        res["locations"] = {"error": repr(e)[:300]}
    return res


def locations_extra(hz, scene, g, n, args, device):
    """UNTIMED extra: horayzon.horizon.horizon_locations for random locations scattered over the config-3 tile (a few metres
    above / below the surface, so the snap onto the mesh is exercised), default algorithm binary_search, default lower limit
    -89.98 deg, 360 azimuths; with and without the distance-to-horizon output (closest-hit queries: a plain per-lane loop, a
    tenth of the locations).  NumPy in / NumPy out: kernel seconds come from hz_stats."""
    rng = np.random.default_rng(5)
    out = {}
    for key, m, dist_out in (("binary_search", args.locations, False), ("binary_search_with_distance", max(args.locations // 10, 1), True)):
        ci = rng.integers(40, n - 40, m); cj = rng.integers(40, n - 40, m)
        coords = np.stack([g["x"][cj] + rng.uniform(-8.0, 8.0, m), g["y"][ci] + rng.uniform(-8.0, 8.0, m),
                           g["z"][ci, cj] + rng.uniform(-30.0, 60.0, m)], axis=1).astype(np.float32)
        vn = np.zeros((m, 3), np.float32); vn[:, 2] = 1.0
        vo = np.zeros((m, 3), np.float32); vo[:, 1] = 1.0
        r = hz.horizon.horizon_locations(g["vert_grid"], n, n, coords, vn, vo, args.dist_search, azim_num=360,
                                         hori_dist_out=dist_out, device=device, scene=scene)
        st = hz.horizon.last_stats
        ks = st["t_kernel_s"]
        out[key] = {"locations": int(m), "kernel_s": ks, "locations_per_s": m / ks, "mray_per_s": st["num_rays"] / ks / 1e6,
                    "rays_per_location_azimuth": st["num_rays"] / (m * 360.0), "on_the_mesh": int(st["num_cells"]),
                    "finite": bool(np.isfinite(r[0]).all())}
    out["note"] = ("untimed, outside `value`: random locations over the 3601^2 tile, 360 azimuths, binary_search, elev_ang_low_lim -89.98; "
                   "kernel seconds of hz_stats (snap onto the mesh included)")
    return out


def c5_extra():
    """UNTIMED extra: BASELINE config 5 (the 14401^2 mosaic, SVF-fused, through dist.sharded_rows with one rank) as its own
    process -- `python bench.py --workload c5` -- so that the driver-run line carries a config-5 figure too."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    t0 = time.perf_counter()
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "c5", "--steps", "1", "--warmup", "1"],
                           env=env, capture_output=True, text=True, timeout=900)
        d = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:          # the extras never fail the headline line
        return {"error": repr(e)[:300]}
    c = d["config"]
    return {"metric": d["metric"], "cells_per_s": d["value"], "job_s": d["ms_per_step"] * 1e-3, "mray_per_s": d["mray_per_s"],
            "n_gpus": d["n_gpus"], "scaling": d["scaling"], "kernel_s": c["kernel_s_rank0"], "near_prepass_s": c["near_prepass_s_rank0"],
            "svf_kernel_s": c["svf_kernel_s_rank0"], "bvh_build_s": c["bvh_build_s"], "scene_bytes": c["scene_bytes"],
            "stack_redo_blocks": c["stack_redo_blocks_rank0"], "gathered_svf_finite": c["gathered_svf_finite"],
            "wall_s_incl_synthesis": time.perf_counter() - t0,
            "note": "subprocess `bench.py --workload c5 --steps 1 --warmup 1`: whole 206.5 M-cell inner domain, one rank"}


def e2e_numpy_call(g, vec_tilt, args, A):
    """UNTIMED (not part of `value`): the drop-in call as a user of the reference makes it -- NumPy in, NumPy out
    (horizon.pyx:170-197): vertices and per-cell inputs uploaded, BVH built, horizon traced in chunks that are
    copied into the caller's 18 GB array behind the next chunk's kernel (the array is page-locked chunk by chunk on a
    helper thread so that those copies are DMA, HostPinner in hz_api.hip), SVF fused."""
    import horayzon_amd as hz
    kw = {k: g[k] for k in ("vert_grid", "dem_dim_0", "dem_dim_1", "vec_norm", "vec_north", "offset_0", "offset_1")}
    t0 = time.perf_counter()
    hori, azim, svf = hz.horizon.horizon_gridded(**kw, dist_search=args.dist_search, azim_num=A, svf_vec_tilt=vec_tilt)
    wall = time.perf_counter() - t0
    st = hz.horizon.last_stats
    ok = bool(np.isfinite(hori[::97, ::89]).all() and np.isfinite(svf).all())
    return {"wall_s": wall, "lib_total_s": st["t_total_s"], "h2d_s": st["t_h2d_s"], "bvh_build_s": st["t_bvh_s"],
            "kernel_s": st["t_kernel_s"], "near_prepass_s": st["t_near_s"], "svf_s": st["t_svf_s"],
            "d2h_tail_s": st["t_d2h_s"], "out_gb": hori.nbytes / 1e9, "cells_per_s_e2e": st["num_cells"] / wall,
            "finite": ok,
            "note": "untimed, outside `value`: NumPy in / NumPy out through horayzon.horizon.horizon_gridded; "
                    "d2h_tail_s is the part of the device-to-host copy that did not hide behind the kernels"}


def roofline(args, stats, steps, cw, peaks, A, n, rps):
    """The bound of k_horizon is the VALU port: achieved = the SIMD cycles its instructions need
    at the measured issue rates of the two instruction classes (wave-iteration counters of the COUNT instantiation x
    the calibrated instructions per iteration x the static class mix), peak = the SIMD cycles of the launch.  The HBM
    pair (algorithmic bytes -- mostly cache-served node re-reads -- and the counter-measured traffic) is reported
    next to it."""
    k_launch_s = stats.t_kernel_s / max(steps, 1)
    rays_launch = stats.num_rays / max(steps, 1)
    cells_launch = stats.num_cells / max(steps, 1)
    b_io = (12 + 12 + 12 + 1 + 12 + 4) * cells_launch + 4.0 * A * cells_launch
    left_s = getattr(stats, "t_left_s", 0.0) / max(steps, 1)
    r = {"kernel": "hz::k_horizon<2,false,true,false,false,false> (guess_constant, staged output, fast stack discipline, persistent waves) + its follow-up "
                   "launch hz::k_horizon<2,false,true,false,false,true> (the cells that blocks handed over when <= 36 of their 64 were unfinished, "
                   "sorted by azimuths left and position: k_left_keys + radix sort, included in kernel_ms_leftover_launch)",
         "kernel_ms_per_launch": 1e3 * k_launch_s,
         "kernel_ms_production_launch": 1e3 * (k_launch_s - left_s), "kernel_ms_leftover_launch": 1e3 * left_s,
         "leftover_cells_per_launch": getattr(stats, "left_cells", 0) / max(steps, 1),
         "mray_per_s_kernel": rays_launch / k_launch_s / 1e6 if k_launch_s else None,
         "svf_kernel_ms_per_launch": 1e3 * stats.t_svf_s / max(steps, 1)}
    model, mix0, mnotes = load_valu_model()
    sha = kernel_asm_sha()
    r["valu_model"] = mnotes["valu_model"]
    b_trav = 0.0
    if cw is not None and cw.num_rays:
        # the counter pass ran one slab: scale its wave-level counts to a mean launch by the ray count
        scale = rays_launch / cw.num_rays
        winst = scale * (cw.wave_node_iters * model["node_iter"] + cw.wave_leaf_iters * model["leaf_iter"]
                         + cw.wave_refills * model["refill_iter"]) + model["per_cell"] * cells_launch / 64.0
        nodes_per_ray = cw.nodes_visited / cw.num_rays
        tris_per_ray = cw.tris_tested / cw.num_rays
        b_trav = rays_launch * (nodes_per_ray * NODE_BYTES + tris_per_ray * 24.0)      # SURVEY 8(d) with this round's node size
        lanes = (cw.nodes_visited + cw.tris_tested / 2.0) / max(64.0 * (cw.wave_node_iters + cw.wave_leaf_iters), 1.0)
        r.update({"nodes_per_ray": nodes_per_ray, "tris_per_ray": tris_per_ray,
                  "valu_winst_per_launch": winst, "valu_model_constants": model, "lane_utilisation_node_leaf_steps": lanes})
        if peaks and peaks["valu_winst_per_s"]:
            # The VALU port as the bound.  gfx950 issues wave64 VALU instructions in two classes (class_rates, measured
            # above): fast (FP32 fma / mul / add, moves, logic, integer add: ~2.4 cycles per SIMD when the VGPR sources
            # sit in different banks, ~4.1 when they collide) and slow (conversions, v_perm, min / max, compares, ...:
            # ~4.15).  achieved = the SIMD cycles the launch's instructions need at the conflict-free rates (wave-
            # iteration counters x calibrated instructions per iteration x static class mix of the section), peak = the
            # SIMD cycles the launch had.  `frac` is therefore a LOWER bound of the VALU-busy share; with every fast
            # instruction colliding it is frac_uniform_4_cycle (the round-1/2 model).
            mix, mix_note = mix0, mnotes["class_mix_note"]
            cr = peaks.get("class_rates") or {"fast_cycles": 2.4, "slow_cycles": 4.15}
            cyc = lambda f: f * cr["fast_cycles"] + (1.0 - f) * cr["slow_cycles"]
            cycles_need = scale * (cw.wave_node_iters * model["node_iter"] * cyc(mix["node_step"])
                                   + cw.wave_leaf_iters * model["leaf_iter"] * cyc(mix["leaf_step"])
                                   + cw.wave_refills * model["refill_iter"] * cyc(mix["refill_and_loop_overhead"]))
            cycles_have = peaks["simds"] * peaks["clock_ghz"] * 1e9 * k_launch_s
            r["valu"] = {"binding": True, "achieved": cycles_need / k_launch_s / 1e9,
                         "peak": peaks["simds"] * peaks["clock_ghz"], "unit": "G SIMD-cycles/s (VALU busy)",
                         "frac_model_raw": cycles_need / cycles_have,
                         "frac_valu_counter_floor": 2.0 * winst / cycles_have,
                         "frac_uniform_4_cycle": winst / k_launch_s / peaks["valu_winst_per_s"],
                         "valu_winst_per_s": winst / k_launch_s,
                         "class_rates_cycles_per_wave_inst": cr, "class_mix_fast_fraction": mix, "class_mix_note": mix_note,
                         "note": "frac_model_raw = SIMD cycles the launch's VALU instructions need at the MEAN measured issue rate of "
                                 "their class (wave-iteration counters x calibrated instructions per iteration x static class mix) / "
                                 "SIMD cycles of the launch -- a model, reported raw even above 1; frac_valu_counter_floor = wave-level "
                                 "VALU instructions (calibrated on SQ_INSTS_VALU) x 2 cycles / SIMD cycles: what the counters alone "
                                 "guarantee.  The truth lies between the two.  A 3.5 ms burst of v_fma_f32 chains with scalar "
                                 "operands: %.3f cycles per instruction" % (peaks["cycles_per_wave_inst_measured"] or 0.0),
                         "clock_ghz": peaks["clock_ghz"], "simds": peaks["simds"]}
            for k in ("frac_model_raw", "frac_valu_counter_floor", "frac_uniform_4_cycle", "valu_winst_per_s"):
                r[k] = r["valu"][k]
            r["class_mix_fast_fraction"] = mix
    alg = (b_io + b_trav) / k_launch_s / 1e9 if k_launch_s else None
    traffic, tnote = None, "profiles/traffic.json missing"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("kernel_asm_sha") != sha:
                tnote = "profiles/traffic.json was measured for other machine code (stale): not reported"
            elif tj.get("rows_per_step") == rps and tj.get("tile") == n and tj.get("azim") == A:
                traffic, tnote = tj.get("hbm_bytes_per_launch"), "rocprofv3 PMC passes of this kernel (profiles/traffic.json)"
        except Exception:
            pass
    r.update({"traffic": traffic, "traffic_note": tnote,
              "hbm": {"peak_gbs": HBM_PEAK_GBS, "copy_kernel_gbs": peaks["copy_gbs"] if peaks else None,
                      "alg_bytes_per_launch": b_io + b_trav, "alg_gbs_cache_served": alg,
                      "alg_frac_of_peak_cache_served": alg / HBM_PEAK_GBS if alg else None,
                      "hbm_counter_gbs": traffic / k_launch_s / 1e9 if traffic else None,
                      "hbm_frac": traffic / k_launch_s / 1e9 / HBM_PEAK_GBS if traffic else None}})
    # The contract fields follow SURVEY 8(d) as written: bound "hbm", achieved = algorithmic bytes (B_io + rays x (nodes x 32 B +
    # triangles x 24 B)) / kernel time, peak = 8 TB/s.  ~99.9 % of those bytes are node re-reads served by L1 / L2: the figure is
    # NOT an HBM utilisation (hbm.hbm_frac, from the PMC counters, is) and the resource that binds is the VALU port (valu.*).
    r.update({"bound": "hbm", "achieved": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s (algorithmic, cache-served)",
              "frac": alg / HBM_PEAK_GBS if alg else None})
    r["frac_8d_hbm_model"] = r["frac"]
    # (algorithmic bytes are cache-served: the figure CAN exceed 1 -- said explicitly when it does, never capped)
    r["frac_exceeds_1_cache_served"] = bool(r["frac"] is not None and r["frac"] > 1.0)
    r["binding_resource"] = "valu_issue" if "valu" in r else "unknown (no counter pass)"
    r["frac_is"] = ("SURVEY 8(d): algorithmic bytes / kernel time / HBM peak -- cache-served, not HBM traffic; hbm.hbm_frac is the "
                    "counter-measured HBM utilisation; the kernel is VALU-issue bound: valu.frac_model_raw (class model, mean-priced, "
                    "raw) and valu.frac_valu_counter_floor (counter-only lower bound)")
    return r


# ------------------------------------------------------------------------------------------------
# row-sharded jobs: config 3 on N > 1 ranks (strong scaling of the headline tile) and config 5 (the 14401^2 mosaic)
# ------------------------------------------------------------------------------------------------
def _tilt_rows_device(verts, off, in1, b, e):
    """vec_tilt of inner-domain rows [b, e) from the vertex array inside the blob, on the device: the same centred
    differences (float64) as synth.tilt_from_planar_dem -- a stand-in for topo_param.slope_plane_meth."""
    import torch
    z = verts[b + off - 1:e + off + 1, off - 1:off + in1 + 1, 2].double()
    x = verts[0, off - 1:off + in1 + 1, 0].double()
    y = verts[b + off - 1:e + off + 1, 0, 1].double()
    dzdx = (z[1:-1, 2:] - z[1:-1, :-2]) / (x[2:] - x[:-2])[None, :]
    dzdy = (z[2:, 1:-1] - z[:-2, 1:-1]) / (y[2:] - y[:-2])[:, None]
    nrm = torch.sqrt(dzdx * dzdx + dzdy * dzdy + 1.0)
    t = torch.stack([-dzdx / nrm, -dzdy / nrm, 1.0 / nrm], dim=2).float()
    t = t / torch.sqrt((t.double() ** 2).sum(dim=2, keepdim=True)).float()
    return t.contiguous()


def run_sharded(ctx, kind):
    """Row-sharded job over the ranks: `kind` "c5" = the 14401^2 mosaic, SVF-fused, horizon never
    materialised; "c3" = the 3601^2 headline tile with the horizon array AND the SVF written to HBM (each rank into its own
    resident slab buffers) -- the N > 1 points of the c3 strong-scaling curve.  One step = one pass over the whole inner
    domain: dist.sharded_rows (slabs -> compute -> timing exchange -> gather of the SVF on rank 0)."""
    import torch
    import torch.distributed as dist
    from horayzon_amd import _lib, synth
    from horayzon_amd import dist as _dist_mod
    from horayzon_amd.dist import sharded_rows, row_slabs, estimate_row_cost, predicted_imbalance
    args, rank, world, dev = ctx["args"], ctx["rank"], ctx["world"], ctx["dev"]
    L = _lib.lib()
    c3 = kind == "c3"
    n, off, A = args.tile or (3601 if c3 else 14401), 16, args.azim
    steps = args.steps if args.steps is not None else (3 if c3 else 1)
    warmup = args.warmup if args.warmup is not None else 1
    in0 = in1 = n - 2 * off
    # only the building rank synthesises the DEM; everybody else receives vertices (+ LBVH) in the one broadcast
    g = synth.fractal_tile(n=n, offset=off, plain_fraction=args.plain_fraction) if rank == 0 else None
    scene, scene_stats, t_build, t_bcast = make_scene(ctx, g, n)
    del g
    blob_ptr, blob_bytes = scene.blob()
    vptr, d0, d1, height_field = scene.vertices()
    verts = _dist_mod.device_bytes_or_copy(vptr, d0 * d1 * 12, ctx["local_rank"], owner=scene)[0].view(torch.float32).view(d0, d1, 3)
    stats = _lib.hz_stats()
    cache = {}            # c3: the slab's inputs and output buffers stay resident across the steps (as in the N = 1 line)

    def slab_inputs(b, e):
        """Per-cell inputs of rows [b, e) only, made on this rank's GPU (planar frames; tilt from the blob's vertices)."""
        if c3 and cache.get("rows") == (b, e):
            return cache["inputs"]
        norm = torch.zeros((e - b, in1, 3), dtype=torch.float32, device=dev); norm[..., 2] = 1.0
        north = torch.zeros((e - b, in1, 3), dtype=torch.float32, device=dev); north[..., 1] = 1.0
        mask = torch.ones((e - b, in1), dtype=torch.uint8, device=dev)
        res = (norm, north, mask, _tilt_rows_device(verts, off, in1, b, e))
        if c3:
            cache["rows"], cache["inputs"] = (b, e), res
            cache["hori"] = torch.empty((e - b, in1, A), dtype=torch.float32, device=dev)
            cache["svf"] = torch.empty((e - b, in1), dtype=torch.float32, device=dev)
        return res

    def run_slab(b, e, st, azim=A, count=False, probe=False, mask_override=None):
        if e <= b:
            return torch.full((0, in1), float("nan"), dtype=torch.float32, device=dev)
        norm, north, mask, tilt = slab_inputs(b, e)
        if mask_override is not None:
            mask = mask_override
        materialise = c3 and not probe and cache.get("rows") == (b, e)
        svf = cache["svf"] if materialise else torch.empty((e - b, in1), dtype=torch.float32, device=dev)
        svf.fill_(float("nan"))
        opts = _lib.hz_opts()
        opts.device = ctx["local_rank"]
        opts.top_nodes = -1; opts.regroup = -1
        opts.vec_tilt = tilt.data_ptr()
        opts.svf = svf.data_ptr()
        opts.hori_is_slab = 1; opts.inputs_are_slab = 1
        opts.skip_hori = 0 if materialise else 1     # c5 / probes: the horizon lives in a bounded device buffer, chunk by chunk
        opts.count_work = int(count)
        opts.row_begin, opts.row_end = b, e
        rc = L.hz_horizon_gridded_scene(scene._h, norm.data_ptr(), north.data_ptr(), off, off,
                                        cache["hori"].data_ptr() if materialise else None, in0, in1, azim,
                                        args.dist_search, 0.25, b"guess_constant", -15.0, mask.data_ptr(), 0.0,
                                        0.01, C.byref(opts), C.byref(st))
        _lib.check(rc)
        return svf

    def compute(b, e):
        return run_slab(b, e, stats)

    # cost of a row: the wave-level VALU work (calibrated instructions per wave iteration x the wave-iteration counters
    # of the counting instantiation) of the 16-row tile row that holds it, measured on every 8th 16 x 16 tile of that tile
    # row with ALL azimuths.  Full 8 x 8 blocks per wave, as in the real launch: the SIMT cost is what is measured, not
    # lane-level ray / node counts (round 3: one-row probes with lane counts predicted the slab times WORSE than the plain
    # cell count).  Round 3 probed every cell with an eighth of the azimuths instead: guess_constant then starts every
    # search 8 sectors away from its last result, which overprices rough terrain against smooth terrain -- on the half-plain
    # DEM of --plain-fraction it predicted the wrong half to be the expensive one (profiles/r04/).
    a_probe = A
    probe_s = [0.0]
    tile_cols = (in1 + 15) // 16

    def probe(row):
        t0 = time.perf_counter()
        st = _lib.hz_stats()
        rb = min(row // 16 * 16, max(in0 - 16, 0))
        re = min(rb + 16, in0)
        pm = torch.zeros((re - rb, tile_cols, 16), dtype=torch.uint8, device=dev)
        pm[:, (rb // 16) % 8::8, :] = 1                 # every 8th tile of this tile row, staggered from row to row
        pm = pm.reshape(re - rb, tile_cols * 16)[:, :in1].contiguous()
        saved = dict(cache); cache.clear()             # probe rows are not the slab: do not disturb the resident buffers
        run_slab(rb, re, st, azim=a_probe, count=True, probe=True, mask_override=pm)
        cache.clear(); cache.update(saved)
        probe_s[0] += time.perf_counter() - t0
        m = VALU_MODEL_DEFAULT
        w = m["node_iter"] * st.wave_node_iters + m["leaf_iter"] * st.wave_leaf_iters + m["refill_iter"] * st.wave_refills
        return w * in1 / max(st.num_cells, 1)          # cost of one row of in1 cells

    n_samples = args.cost_samples or max(32, 8 * max(world, args.emulate_ranks))

    cost, t_cost = None, 0.0
    balance_cost = args.balance == "cost" and (world > 1 or os.environ.get("HZ_FORCE_COST"))
    if c3 and balance_cost:      # c3: the partition is part of the set-up (the resident slab buffers depend on it)
        tc0 = time.perf_counter()
        cost = estimate_row_cost(in0, probe, samples=n_samples)
        t_cost = time.perf_counter() - tc0
    slab0 = row_slabs(in0, world, cost)[rank]
    # counter pass and machine calibration for the roofline (rank 0, untimed): c3 only
    cw = peaks = None
    if c3:
        if slab0[1] > slab0[0]:
            slab_inputs(*slab0)
        if rank == 0 and not args.no_count and slab0[1] > slab0[0]:
            cw = _lib.hz_stats()
            run_slab(slab0[0], slab0[1], cw, count=True)
        if rank == 0 and not args.no_peaks:
            peaks = machine_peaks(L, ctx["local_rank"])
            peaks["class_rates"] = inst_class_rates(L, ctx["local_rank"])
    for w in range(warmup):      # c3: one pass over this rank's slab; c5: a short slab of its rows (clocks, allocator)
        run_slab(slab0[0], slab0[1] if c3 else min(slab0[0] + 64, slab0[1]), _lib.hz_stats())
    if args.emulate_ranks > 1:
        return emulate_ranks(ctx, in0, in1, run_slab, probe, n_samples, blob_bytes)
    # the collectives of a step once before the clock starts (RCCL sets its point-to-point channels up on first use)
    _slabs_w = row_slabs(in0, world, cost)
    _dist_mod.gather_rows(torch.zeros((_slabs_w[rank][1] - _slabs_w[rank][0], 1), dtype=torch.float32, device=dev), _slabs_w, dst=0)
    _tw = _dist_mod._host_staged(torch.zeros(1, dtype=torch.float64, device=dev), None)
    dist.all_gather([torch.empty_like(_tw) for _ in range(world)], _tw)
    barrier(ctx)
    t0 = time.perf_counter()
    res = None
    for s in range(steps):
        probe_s[0] = 0.0
        if balance_cost and not c3:
            tc0 = time.perf_counter()
            cost = estimate_row_cost(in0, probe, samples=n_samples)
            t_cost = time.perf_counter() - tc0
        res = sharded_rows(in0, compute, sync=torch.cuda.synchronize, dst=0, cost=cost)
    barrier(ctx)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    tot = torch.tensor([stats.num_rays, stats.num_cells], dtype=torch.float64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    rays_total, cells_total = float(tot[0].item()), float(tot[1].item())
    if rank != 0:
        return None
    svf_ok = None
    if res is not None and res["full"] is not None:
        svf_ok = bool(torch.isfinite(res["full"]).all().item()) and tuple(res["full"].shape) == (in0, in1)
        if args.dump_svf_rows and args.dump_path:
            if args.dump_svf_rows == "all":
                np.save(args.dump_path, res["full"].cpu().numpy())
            else:
                rows = [int(r) for r in args.dump_svf_rows.split(",")]
                np.save(args.dump_path, res["full"][rows].cpu().numpy())
    step_s = elapsed / max(steps, 1)
    # c3, N > 1: what the one-GPU emulation predicts for this partition, next to what was just measured -- rank 0 computes the
    # whole tile and then every rank's slab on ITS GPU, one after the other, each alone as on its own rank (untimed; the
    # other ranks are done).  predicted_efficiency = whole / (N x slowest slab); measured_efficiency = whole / (N x step).
    prediction = None
    if c3 and world > 1 and not args.no_extras:
        try:
            def alone_ms(b, e):
                cache.clear(); slab_inputs(b, e)
                best = None
                for _ in range(2):
                    st = _lib.hz_stats()
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    run_slab(b, e, st)
                    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t1)
                    best = dt if best is None else min(best, dt)
                return best
            slabs_now = row_slabs(in0, world, cost)
            ts = [alone_ms(b, e) for b, e in slabs_now]
            whole_ms = alone_ms(0, in0)
            cache.clear()
            prediction = {"whole_tile_ms_one_gpu": whole_ms, "slab_ms_one_gpu": [round(t, 2) for t in ts],
                          "predicted_efficiency": whole_ms / (world * max(ts)),
                          "measured_efficiency_same_run": whole_ms / (world * 1e3 * step_s),
                          "note": "wall times of hz_horizon_gridded_scene on rank 0's GPU, slabs one by one (compute only: certificates + "
                                  "kernel + SVF); measured_efficiency uses this run's ms_per_step (incl. the SVF gather)"}
        except Exception as ex:
            prediction = {"error": repr(ex)[:300]}
    config = {
        "predicted_efficiency": prediction["predicted_efficiency"] if prediction and "predicted_efficiency" in prediction else None,
        "scaling_prediction": prediction,
        "parallelism": "strong scaling: row slabs over %d ranks (dist.row_slabs, balanced by %s), scene broadcast once (%s), "
                       "slab-local inputs made on each rank's GPU, %sSVF gathered on rank 0"
                       % (world, "sampled cost" if cost is not None else "cell count",
                          "vertices only, LBVH rebuilt per rank" if args.bcast == "verts" and ctx["use_dist"] else
                          "out of the blob allocation",
                          "horizon written by every rank into its own resident slab, " if c3 else ""),
        "slabs": res["slabs"] if res else None, "t_ranks_s": res["t_ranks"] if res else None,
        "load_imbalance_max_over_mean": res["imbalance"] if res else None,
        "load_imbalance_predicted": res["imbalance_predicted"] if res else None,
        "load_imbalance_predicted_if_balanced_by_cells":
            predicted_imbalance(row_slabs(in0, world), cost) if cost is not None else None,
        "cost_prepass_s": t_cost, "cost_prepass_probe_rows": n_samples if cost is not None else 0,
        "cost_prepass_probe_azimuths": a_probe,
        "bvh_build_s": scene_stats["t_bvh_s"] if scene_stats else None, "scene_bytes": int(blob_bytes),
        "scene_bcast": args.bcast if ctx["use_dist"] else None, "scene_bcast_bytes": int(ctx.get("bcast_bytes", 0)),
        "scene_bcast_s": t_bcast, "scene_bcast_zero_copy": _dist_mod.last_broadcast_zero_copy, "scene_create_wall_s": t_build,
        # the job as a user would pay for it once: scene broadcast (+ per-rank rebuild with --bcast verts) + one step
        "job_s_incl_bcast": t_bcast + step_s,
        "cells_per_s_incl_bcast": cells_total / max(steps, 1) / (t_bcast + step_s),
        "kernel_s_rank0": stats.t_kernel_s,
        "svf_kernel_s_rank0": stats.t_svf_s, "near_prepass_s_rank0": stats.t_near_s,
        "stack_fallbacks_rank0": int(stats.stack_fallbacks), "stack_redo_blocks_rank0": int(stats.stack_redo_blocks),
        "guard_events_rank0": int(stats.guard_events), "guard_cells_rank0": int(stats.guard_cells),
        "height_field": int(height_field), "gathered_svf_finite": svf_ok}
    if c3:
        my_rows = slab0[1] - slab0[0]
        config["workload"] = ("c3: horizon_gridded guess_constant + fused SVF, %dx%d synthetic SRTM-like tile, %d azimuths, "
                              "dist_search %g km, the whole inner domain (%d rows) per step, split into %d row slab(s)"
                              % (n, n, A, args.dist_search, in0, world))
        config["cells_per_step"] = cells_total / max(steps, 1)
        config["rays_per_cell_azimuth"] = rays_total / max(cells_total * A, 1)
        roof = roofline(args, stats, steps, cw, peaks, A, n, my_rows)
        roof["note"] = "rank 0's slab (%d rows); the kernel's N = 1 roofline is the plain `bench.py` line" % my_rows
        metric = "grid_cells_per_s (horizon_gridded, 360 azimuths, 3601^2 SRTM-like tile)"
    else:
        config["workload"] = ("c5: horizon_gridded guess_constant, SVF-fused (horizon never materialised), %dx%d synthetic "
                              "mosaic, %d azimuths, dist_search %g km; one step = the whole inner domain (%d x %d cells)"
                              % (n, n, A, args.dist_search, in0, in1))
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "binding_resource": "valu_issue",
                "traffic": None, "note": "see the c3 line: same kernel; c5 reports scaling, not the kernel roofline"}
        metric = "grid_cells_per_s (horizon_gridded + SVF, 360 azimuths, 4x4 mosaic of 3601^2 SRTM-like tiles)"
    return {
        "metric": metric,
        "value": cells_total / elapsed, "unit": "cells/s", "mray_per_s": rays_total / elapsed / 1e6,
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * step_s,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": config.pop("workload"), **config},
        "roofline": roof,
    }


def emulate_ranks(ctx, in0, in1, run_slab, probe, n_samples, blob_bytes):
    """One GPU: the R slabs of a cost-balanced and of a cell-count-balanced R-rank partition, computed one after the
    other and timed -- max / mean of those times is the load imbalance an R-GPU run of this job would see (each slab
    runs alone on the GPU, exactly as on its own rank)."""
    import torch
    from horayzon_amd import _lib
    from horayzon_amd.dist import row_slabs, estimate_row_cost, predicted_imbalance
    R = ctx["args"].emulate_ranks
    t0 = time.perf_counter()
    cost = estimate_row_cost(in0, probe, samples=n_samples)
    t_cost = time.perf_counter() - t0
    out = {"emulated_ranks": R, "cost_prepass_s_one_rank_doing_all_probes": t_cost, "probe_rows": n_samples}
    total_cells, total_s = 0, 0.0
    for name, slabs in (("cost", row_slabs(in0, R, cost)), ("cells", row_slabs(in0, R))):
        ts = []
        for b, e in slabs:
            st = _lib.hz_stats()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            run_slab(b, e, st)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
            total_cells += st.num_cells; total_s += ts[-1]
        out[name] = {"slabs": slabs, "t_slab_s": ts, "imbalance_measured": max(ts) / (sum(ts) / len(ts)),
                     "imbalance_predicted": predicted_imbalance(slabs, cost), "job_s_if_parallel": max(ts)}
    return {"metric": "load balance of an emulated %d-rank c5 partition (one GPU, slabs timed one by one)" % R,
            "value": total_cells / total_s, "unit": "cells/s", "n_gpus": 1, "steps": 1, "warmup": 0,
            "ms_per_step": 1e3 * total_s, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": {"workload": "c5 --emulate-ranks %d" % R, **out,
                                                            "scene_bytes": int(blob_bytes)},
            "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "binding_resource": "valu_issue",
                         "frac": None, "traffic": None, "note": "load-balance probe, see the c3 line for the kernel"}}


def c4_roofline(n, S, cells, shadow, k_step, cw, refrac, peaks, cr):
    """Section 8(d) figures of one step (S sun positions, one launch) of k_shadow_refill: algorithmic bytes / kernel time / HBM peak
    (`frac`, cache-served), and -- with a counting pass `cw` (and the machine calibration `peaks`, `cr`) -- the VALU-issue
    bracket that actually binds (counter floor, class model), as for k_horizon.  shadow_comp.cpp:386-605."""
    V = n * n
    out_b = 1 if shadow else 4
    # SURVEY 8(d) "shadow bytes": 12 V + 33 C once at initialise, 1 C (shadow) / 4 C (sw_dir_cor) per sun position;
    # B_trav = rays x (node visits x 32 B + triangle tests x 24 B) from the counting pass -- served by the caches
    b_io = S * out_b * cells + (12.0 * V + 33.0 * cells)
    b_trav = (cw["nodes_visited"] * NODE_BYTES + cw["tris_tested"] * 24.0) if cw else 0.0
    alg = (b_io + b_trav) / k_step / 1e9 if k_step else None
    roof = {"kernel": "hz::k_shadow_refill<false> (all sun positions of a step in one launch, lane refill)",
            "kernel_ms_per_step": 1e3 * k_step, "traffic": None,
            "hbm": {"peak_gbs": HBM_PEAK_GBS, "alg_bytes_per_step": b_io + b_trav, "io_bytes_per_step": b_io,
                    "alg_gbs_cache_served": alg,
                    "note": "algorithmic bytes (SURVEY 8d) are almost all node re-reads served by L1 / L2: the figure may exceed "
                            "the HBM peak and is not an HBM utilisation; the compulsory bytes are io_bytes_per_step"}}
    if cw:
        # VALU port as the bound, as for k_horizon: the traversal is the same hz_trace (147 / 218 wave instructions per node /
        # leaf step, class mix of those sections); the per-cell set-up (ray, self-shading test, refraction) is priced with
        # SETUP wave instructions per 64 cells handed out (calibrated on SQ_INSTS_VALU, profiles/r03/pmc_shadow_refill.json)
        n_it, l_it = cw["wave_node_iters"], cw["wave_leaf_iters"]
        rounds = S * cells / 64.0
        vm, vmix, vnotes = load_valu_model()
        # (calibrated with the traversal constants on the shadow kernel's own SQ_INSTS_VALU when the profiles are current;
        #  refraction adds the round-3 difference of the two built-in figures)
        setup = vm.pop("shadow_setup", SHADOW_SETUP_WINST[0]) + (SHADOW_SETUP_WINST[1] - SHADOW_SETUP_WINST[0]) * int(bool(refrac))
        winst = vm["node_iter"] * n_it + vm["leaf_iter"] * l_it + setup * rounds
        roof.update({"nodes_per_ray": cw["nodes_visited"] / max(cw["num_rays"], 1), "tris_per_ray": cw["tris_tested"] / max(cw["num_rays"], 1),
                     "wave_node_iters": n_it, "wave_leaf_iters": l_it,
                     "lane_utilisation_node_leaf_steps": (cw["nodes_visited"] + cw["tris_tested"] / 2.0) / max(64.0 * (n_it + l_it), 1.0),
                     "valu_winst_per_step_model": winst})
        if peaks:
            cr = cr or {"fast_cycles": 2.4, "slow_cycles": 4.15}
            cyc = lambda f: f * cr["fast_cycles"] + (1.0 - f) * cr["slow_cycles"]
            need = (vm["node_iter"] * n_it * cyc(vmix["node_step"])
                    + vm["leaf_iter"] * l_it * cyc(vmix["leaf_step"]) + setup * rounds * cyc(0.7))
            have = peaks["simds"] * peaks["clock_ghz"] * 1e9 * k_step
            roof["valu"] = {"binding": True, "achieved": need / k_step / 1e9, "peak": peaks["simds"] * peaks["clock_ghz"],
                            "unit": "G SIMD-cycles/s (VALU busy)", "frac_model_raw": need / have,
                            "frac_valu_counter_floor": 2.0 * winst / have, "frac_uniform_4_cycle": 4.0 * winst / have,
                            "class_rates_cycles_per_wave_inst": cr, "valu_model": vnotes["valu_model"], "valu_model_constants": vm,
                            "class_mix_fast_fraction": vmix,
                            "note": "as for k_horizon (c3 line): frac_model_raw = the class model at mean-priced issue rates, reported "
                                    "raw (no cap: above 1 means the model's mix or rates are off by that much); "
                                    "frac_valu_counter_floor = wave-level VALU instructions x 2 cycles / SIMD cycles of the launch"}
            for k in ("frac_model_raw", "frac_valu_counter_floor", "frac_uniform_4_cycle"):
                roof[k] = roof["valu"][k]
    # contract fields as SURVEY 8(d) writes them (algorithmic bytes, cache-served; see hbm.note); the binding resource is the VALU port
    roof.update({"bound": "hbm", "achieved": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s (algorithmic, cache-served, see hbm.note)",
                 "frac": alg / HBM_PEAK_GBS if alg else None, "frac_8d_hbm_model": alg / HBM_PEAK_GBS if alg else None,
                 "frac_exceeds_1_cache_served": bool(alg and alg / HBM_PEAK_GBS > 1.0),
                 "binding_resource": "valu_issue" if "valu" in roof else "unknown (no counter pass)"})
    return roof


# ------------------------------------------------------------------------------------------------
# config 4: shadow mask over one day of sun positions on the c3 tile
# ------------------------------------------------------------------------------------------------
def run_c4(ctx):
    import torch
    import horayzon_amd as hz
    from horayzon_amd import _lib, synth
    args, rank, world, dev = ctx["args"], ctx["rank"], ctx["world"], ctx["dev"]
    n, off = args.tile or 3601, 16
    S = args.suns
    g = synth.fractal_tile(n=n, offset=off)
    in0 = in1 = n - 2 * off
    scene, scene_stats, t_build, t_bcast = make_scene(ctx, g, n)
    vec_tilt, enl = synth.tilt_from_planar_dem(g["x"], g["y"], g["z"], off)
    vec_norm, _ = synth.planar_frames(in0, in1)
    elev = np.ascontiguousarray(g["z"][off:off + in0, off:off + in1])
    mask = np.ones((in0, in1), np.uint8)
    suns, alt, _ = synth.sun_positions(num=S)
    terrain = hz.shadow.Terrain(device=ctx["local_rank"])
    terrain.initialise(g["vert_grid"], n, n, off, off, vec_tilt, vec_norm, enl, elev, mask,
                       refrac_cor=bool(args.refrac), scene=scene)
    shadow = args.which == "shadow"
    out = torch.empty((S, in0, in1), dtype=torch.uint8 if shadow else torch.float32, device=dev)   # resident output
    fn = terrain.shadow_batch if shadow else terrain.sw_dir_cor_batch
    steps = args.steps if args.steps is not None else 3
    warmup = args.warmup if args.warmup is not None else 1
    cw = None
    if rank == 0 and not args.no_count:          # counter pass (untimed): node visits / triangle tests of all positions
        terrain.count_work(True)
        fn(suns, out)
        cw = dict(terrain.last_stats)
        terrain.count_work(False)
    for w in range(warmup):
        fn(suns, out)
    barrier(ctx)
    t_kernel, rays = 0.0, 0
    t0 = time.perf_counter()
    for s in range(steps):
        fn(suns, out)
        t_kernel += terrain.last_stats["t_kernel_s"]; rays += terrain.last_stats["num_rays"]
    barrier(ctx)
    elapsed = time.perf_counter() - t0
    if ctx["use_dist"]:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    cells = in0 * in1
    k_step = t_kernel / max(steps, 1)
    codes = None
    if shadow:
        o = out[S // 2].cpu().numpy()
        codes = [float((o == c).mean()) for c in range(4)]
    peaks = machine_peaks(_lib.lib(), ctx["local_rank"]) if (cw and not args.no_peaks) else None
    cr = inst_class_rates(_lib.lib(), ctx["local_rank"]) if peaks else None
    roof = c4_roofline(n, S, cells, shadow, k_step, cw, bool(args.refrac), peaks, cr)
    res = {
        "metric": "grid_cells_per_s (Terrain.%s, %d sun positions, 3601^2 SRTM-like tile)" % (args.which, S),
        "value": world * steps * S * cells / elapsed, "unit": "cells/s",
        "mray_per_s": world * rays / elapsed / 1e6,
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / max(steps, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "c4: Terrain.%s over %d diurnal sun positions (lat 46 N, day 172), %dx%d synthetic tile, "
                               "refrac_cor=%s, outputs resident in HBM; a step = all %d positions"
                               % (args.which, S, n, n, bool(args.refrac), S),
                   "ms_per_sun_position": 1e3 * k_step / S, "rays_per_step": rays / max(steps, 1),
                   "sun_alt_deg_minmax": [float(np.rad2deg(alt.min())), float(np.rad2deg(alt.max()))],
                   "code_fractions_0123_at_noon": codes,
                   "bvh_build_s": scene_stats["t_bvh_s"] if scene_stats else None},
        "roofline": roof,
    }
    return res


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
    embree = None
    tpath = os.path.join(ROOT, "tests", "golden", "embree_timing.json")
    if os.path.exists(tpath):      # recorded by scripts/make_embree_fixtures.py where the reference is installed
        try:
            tj = json.load(open(tpath))
            c3 = tj.get("c3_tile") or {}
            if c3.get("cells_per_s"):
                embree = {"kind": "embree", "value": c3["cells_per_s"], "unit": "cells/s", "mray_per_s": c3.get("mray_per_s"),
                          "cores": (tj.get("environment") or {}).get("logical_cores"),
                          "recorded_on": tj.get("environment"), "reference_version": tj.get("reference_version"),
                          "sample": "%s rows of the same 3601^2 tile, the reference's own 'Ray tracing time' "
                                    "(horizon_comp.cpp:802-810); recorded elsewhere, NOT timed on this node" % c3.get("rows")}
        except Exception:
            embree = None
    base = _cpu_port_line(cells, st, args, g, A, orc)
    base["embree"] = embree if embree is not None else (
        "unavailable: Embree 4 / oneTBB are not installed on this node (and not in the build image); "
        "scripts/make_embree_fixtures.py records the reference's own timing into tests/golden/embree_timing.json where they are")
    return base


def _cpu_port_line(cells, st, args, g, A, orc):
    return {"value": cells / st["t_rays_s"], "unit": "cells/s", "cores": orc.num_threads(), "kind": "port",
            "mray_per_s": st["rays"] / st["t_rays_s"] / 1e6, "bvh_build_s": st["t_build_s"],
            "sample": "%d rows x %d cells x %d azimuths from the middle of the same tile, ray loop only "
                      "(%.1f s); CPU restatement with its own BVH, OpenMP over cells -- not Embree"
                      % (args.cpu_rows, g["vec_norm"].shape[1], A, st["t_rays_s"])}


if __name__ == "__main__":
    main()
