"""bench.py with TWO ranks on the one GPU of the test box: the multi-rank code path the driver launches on an 8-GPU
node (rank 0 builds and broadcasts the scene, every rank computes, SVF rows are gathered, times reduced over the ranks)
including everything only ranks > 0 execute.  RCCL refuses two ranks on one device, so the collectives run on gloo
(HZ_DIST_BACKEND; CUDA tensors are staged through host memory by horayzon_amd.dist) -- the bench code is the same."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(port, *args):
    env = dict(os.environ, HZ_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", *args]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])          # the JSON line is the last line of stdout


def test_config3_weak_scaling_path_with_two_ranks():
    d = _run(29621, "--tile", "601", "--azim", "72", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    cells = 569 * 569
    assert d["config"]["cells_per_step"] == cells
    # whole-job aggregate: both ranks' cells over the slowest rank's time
    assert abs(d["value"] - 2 * 2 * cells / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    assert d["config"]["scene_bcast_s"] > 0 and d["config"]["load_imbalance_max_over_mean"] >= 1.0
    assert d["roofline"]["bound"] in ("valu_issue", "hbm") and d["roofline"]["kernel_ms_per_launch"] > 0


def test_config5_row_sharding_with_two_ranks():
    """Two ranks: only rank 0 synthesises the DEM, the blob is broadcast out of its own allocation, rank 1 derives its
    slab's inputs from the received vertices (slab-local, opts.inputs_are_slab), the slabs come from the sampled cost
    pre-pass (probe rows split over both ranks, joined by one all_reduce)."""
    d = _run(29622, "--workload", "c5", "--tile", "801", "--azim", "72", "--balance", "cost")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    c = d["config"]
    slabs = c["slabs"]
    assert slabs[0][0] == 0 and slabs[0][1] == slabs[1][0] and slabs[1][1] == 769 and 300 < slabs[0][1] < 470
    assert c["gathered_svf_finite"] is True and len(c["t_ranks_s"]) == 2
    assert 1.0 <= c["load_imbalance_max_over_mean"] < 2.0
    assert "sampled cost" in c["parallelism"] and c["cost_prepass_probe_rows"] >= 16 and c["cost_prepass_s"] > 0
    assert 1.0 <= c["load_imbalance_predicted"] <= 1.02          # the split hits the cost target to within a row
    assert c["load_imbalance_predicted_if_balanced_by_cells"] >= c["load_imbalance_predicted"] - 1e-9
    assert c["scene_bcast_zero_copy"] is True and c["height_field"] == 1


def test_config5_single_rank_broadcast_is_free_and_emulated_partition():
    """One rank through the same code: the 'broadcast' out of the blob allocation costs nothing; --emulate-ranks times
    the slabs of a 4-rank partition one by one (the load balance a 4-GPU run would see)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29623")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5", "--tile", "801", "--azim", "72"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["config"]["scene_bcast_s"] < 0.05 and d["config"]["gathered_svf_finite"] is True
    env["MASTER_PORT"] = "29624"
    p = subprocess.run(cmd + ["--emulate-ranks", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    e = json.loads(p.stdout.strip().splitlines()[-1])["config"]
    for kind in ("cost", "cells"):
        sl = e[kind]["slabs"]
        assert len(sl) == 4 and sl[0][0] == 0 and sl[-1][1] == 769 and all(sl[i][1] == sl[i + 1][0] for i in range(3))
        assert len(e[kind]["t_slab_s"]) == 4 and e[kind]["imbalance_measured"] >= 1.0
    assert e["cost"]["imbalance_predicted"] <= e["cells"]["imbalance_predicted"] + 1e-9


def test_config4_bench_line():
    """bench.py --workload c4 (small tile, 8 sun positions): the shadow line with its own roofline."""
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c4", "--tile", "601", "--suns", "8", "--steps", "2"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    cells = 569 * 569
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "cells/s" and "Terrain.shadow" in d["metric"]
    assert abs(d["value"] - 2 * 8 * cells / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and 0.0 < r["frac"] <= 1.0 and r["kernel_ms_per_step"] > 0
    assert r["nodes_per_ray"] > 1 and 0.0 < r["lane_utilisation_node_leaf_steps"] <= 1.0
