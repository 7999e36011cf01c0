#!/usr/bin/env python3
"""What one GPU of BASELINE config 3 runs: 256 / N chains (N = 1, 2, 4, 8 -> 66048 ... 8448 token rows after padding).
(1) per-GEMM times at those row counts for the default dispatch and for each tile kernel forced, (2) ms per Gibbs iteration of the
whole engine at those shard sizes -- so the expected strong-scaling efficiency is known before an 8-GPU node is."""
import ctypes, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, epi, v, iters=200):
    ms = ctypes.c_double()
    rc = L.pg_dbg_gemm_bench(0, M, N, K, epi, v, iters, ctypes.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")
SHAPES = (("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2))
print("us per launch: default dispatch / 64^2 tiles / 128^2 tiles / 256^2 ping-pong / 256^2 16-wave")
for chains in (32, 64, 128, 256):
    M = (chains * 258 + 255) // 256 * 256
    row = []
    for name, N, K, epi in SHAPES:
        t = [run(M, N, K, epi, v) for v in (2, 6, 7, 20, 80)]
        row.append("%s %6.1f /%6.1f /%6.1f /%6.1f /%6.1f (%4.0f TF)" % ((name,) + tuple(t) + (2.0 * M * N * K / t[0] / 1e6,)))
    print("chains=%3d M=%5d | %s" % (chains, M, " | ".join(row)))
if "--engine" in sys.argv:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = None
    for chains in (256, 128, 64, 32):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--chains", str(chains), "--steps", "10", "--warmup", "2",
                              "--no-cpu-baseline", "--no-strict", "--no-msa"], capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print("bench failed for", chains, out.stderr[-500:]); continue
        j = json.loads(line[0])
        base = base or j["ms_per_step"]
        print("engine: %3d chains: %7.2f ms/iteration, %8.0f positions/s, GEMM %.0f TF; as 1/N of a strong-scaling job: efficiency %.2f"
              % (chains, j["ms_per_step"], j["value"], j.get("roofline", {}).get("achieved", 0), base / j["ms_per_step"] / (256 / chains)))
