#!/usr/bin/env python3
"""Per-tile efficiency of the tile-height ladder (gemm_ladder.hip) against the 256- / 192-row ping-pong kernel: each height forced
(PGIBBS_GEMM_LADDER=h, one process per height: the switch is read once) at the residual / fc1 shapes of a 1/8 ... 1/1 shard of
config 3.  Prints us per launch and the time per 256-row-round-equivalent of work on one CU (us x 256 CUs x 256 / (rows x tiles_n))."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import ctypes, sys
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
for a in sys.argv[1:]:
    M, N, K, epi = (int(v) for v in a.split(","))
    ms = ctypes.c_double()
    rc = L.pg_dbg_gemm_bench(0, M, N, K, epi, 2, 100, ctypes.byref(ms))
    print("%%d,%%d,%%d,%%d,%%.2f" %% (M, N, K, epi, ms.value * 1e3 if rc == 0 else float("nan")), flush=True)
""" % ROOT
SHAPES = [(M, N, K, e) for M in (8448, 16640, 33024, 66048) for (N, K, e) in ((1280, 1280, 2), (1280, 5120, 2), (5120, 1280, 1))]
MODES = [("pp256", dict(PGIBBS_GEMM_LADDER="0", PGIBBS_GEMM_T192="0")), ("pp192", dict(PGIBBS_GEMM_LADDER="0", PGIBBS_GEMM_T192="2"))] + \
        [("ppx%d" % h, dict(PGIBBS_GEMM_LADDER=str(h), PGIBBS_GEMM_T192="0")) for h in (160, 176, 208, 224, 240)] + \
        [("default", {})]
res = {}
for name, env in MODES:
    p = subprocess.run([sys.executable, "-c", CHILD] + ["%d,%d,%d,%d" % s for s in SHAPES], capture_output=True, text=True,
                       env=dict(os.environ, **env), timeout=1200)
    for ln in p.stdout.splitlines():
        M, N, K, e, us = ln.split(",")
        res[(name, int(M), int(N), int(K), int(e))] = float(us)
    if p.returncode:
        print(name, "failed:", p.stderr[-400:])
print("us per launch (epi 2 = residual read-modify-write, 1 = bias + GELU, 16-bit out); fc1's GELU epilogue exists at 208 / 224 / 240 only")
print("%-28s" % "shape" + "".join("%10s" % n for n, _ in MODES))
for s in SHAPES:
    print("%-28s" % ("M=%d N=%d K=%d e=%d" % s) + "".join("%10.1f" % res.get((n,) + s, float("nan")) for n, _ in MODES))
