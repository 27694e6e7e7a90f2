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
    d = _run(29622, "--workload", "c5", "--tile", "801", "--azim", "72")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    slabs = d["config"]["slabs"]
    assert slabs[0][0] == 0 and slabs[0][1] == slabs[1][0] and slabs[1][1] == 769 and abs(slabs[0][1] - 384.5) <= 1
    assert d["config"]["gathered_svf_finite"] is True and len(d["config"]["t_ranks_s"]) == 2
    assert 1.0 <= d["config"]["load_imbalance_max_over_mean"] < 2.0
