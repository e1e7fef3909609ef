"""bench.py under its N>1 launch contract: two ranks (gloo, both on GPU 0) must finish and rank 0 must print
the JSON line -- guards the collective legs of the benchmark (a rank-0-only DDP step deadlocks the job)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_completes(hip_lib):
    env = dict(os.environ, UD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--workload", "lidar", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"]
    assert "roofline" in out and "roofline_mfma" in out
