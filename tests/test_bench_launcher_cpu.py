"""`python bench.py --gpus N` must launch its own ranks (the driver may call it with or without torchrun) and shard the 256
chains of BASELINE config 3 over them (strong scaling).  No GPU here: `--dry-run` exercises the launcher, the per-rank slices
of the one position stream, and the gather -- everything around the engine call."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2])
def test_bench_self_launches_and_shards_config3(n):
    out = _run("--gpus", str(n))
    assert out["n_gpus"] == n and out["ranks_seen"] == n and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 256                      # 256 chains in TOTAL, however many GPUs
    assert ("%d per GPU" % (256 // n)) in out["config"]["workload"]
    assert out["steps"] == 2 and out["warmup"] == 1 and out["dry_run"] is True and out["verified_vs_single_gpu"] is True
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert key in out


def test_bench_weak_scaling_flag():
    out = _run("--gpus", "2", "--weak")
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 512
