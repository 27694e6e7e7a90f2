"""bench.py's command line without a GPU: it must refuse loudly (no CPU fallback, no fabricated line)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus", ["1", "2"])
def test_bench_needs_a_gpu(gpus):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by tests/test_gpu_bench_ranks.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs an MI355X" in (p.stderr + p.stdout) and '"metric"' not in p.stdout
