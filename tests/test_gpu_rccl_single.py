"""RCCL executed on the one GPU a test box has: a world_size-1 `nccl` process group runs init, all-reduce, both forms of the
product's all-gather on device tensors and `sharding.run_sharded` through the real engine; `bench.py --gpus 1 --force-dist`
runs the benchmark's N > 1 code path (gather + max-over-ranks + bit-equality check) on backend "nccl".  Child processes: a
process group is process-global state."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_rccl_world_size_one_collectives_and_run_sharded():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_single.py")], capture_output=True, text=True, env=_env(),
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DONE" in p.stdout and p.stdout.count("OK ") == 5


def test_bench_single_gpu_through_rccl():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1",
                        "--layers", "2", "--no-cpu-baseline", "--no-strict", "--no-msa", "--no-roofline", "--native-gather"], capture_output=True,
                       text=True, env=_env(), timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["ranks_seen"] == 1 and out["n_gpus"] == 1
    assert out["verified_vs_single_gpu"] is True and out["value"] > 0
    assert out["native_gather_equal"] is True              # the C-ABI collective (pg_gather_tokens) returned the same tokens
