#!/usr/bin/env python3
"""GEMM micro-benchmark on the ESM-MSA-1b shapes of BASELINE config 4 (M = 64 x 32 x 257 = 526336 token rows, d = 768)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
M = 526336
SHAPES = [("qkv", M, 2304, 768, 0), ("out", M, 768, 768, 2), ("fc1", M, 3072, 768, 1), ("fc2", M, 768, 3072, 2)]
variants = [int(v) for v in (sys.argv[1:] or ["2", "80"])]
for name, m, n, k, epi in SHAPES:
    row = []
    for v in variants:
        ms = ctypes.c_double()
        _lib.check(L.pg_dbg_gemm_bench(0, m, n, k, epi, v, 30, ctypes.byref(ms)))
        row.append("v%d %.3f ms %7.1f TF" % (v, ms.value, 2.0 * m * n * k / ms.value / 1e9))
    print("%-4s M=%d N=%d K=%d epi=%d | %s" % (name, m, n, k, epi, " | ".join(row)))
