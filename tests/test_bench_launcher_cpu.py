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


def test_cpu_baseline_leg_and_traffic_file_selection():
    """The two parts of the bench line VERDICT r04 found wrong, runnable without a GPU: (1) `roofline.traffic` comes from the newest
    ESM-1b PMC file -- never from an ESM-MSA-1b one -- and carries the four per-layer GEMMs; (2) the CPU-baseline leg (BASELINE.md
    section 3: chains x iterations in torch CPU ops at the fastest thread count, config 1 in full, the per-position loop) runs and
    reports the fields the line promises -- here on ONE layer of the real widths and 2 chains, with no engine to check."""
    sys.path.insert(0, ROOT)
    import bench
    from protein_gibbs_sampler_amd import weights
    total, per, src = bench.measured_gemm_traffic()
    assert src and "_msa_" not in src and src.endswith("_hbm_traffic_pmc.json")
    assert set(per) == {"gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2"} and 1.5e9 < total < 4e9
    cfg = dict(weights.ESM1B_CONFIG)
    cfg["n_layers"] = 1
    sd = weights.synthetic_state_dict(cfg, seed=0, **bench.SYNTH_KW)
    out = bench.cpu_baseline(cfg, sd, None, None, 256, 64, 6, list(range(4, 24)), None, chains=2, iters=2, check_chains=1)
    assert out["kind"] == "port" and out["unit"] == "sampled positions/s" and out["value"] > 0
    assert out["cores"] in [int(k) for k in out["thread_sweep_s_per_layer"]] and out["cores"] <= out["host_logical_cpus"]
    assert "2 of 256 chains x 2 Gibbs iterations" in out["sample"]
    assert out["logit_check"]["torch_baseline_vs_checker_max_abs"] < 1e-3
    assert out["config1"]["cpu_ms_per_iter"] > 0 and out["reference_style_position_loop"]["us_per_position_one_core"] > 0
