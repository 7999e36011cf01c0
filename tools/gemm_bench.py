#!/usr/bin/env python3
"""GEMM micro-benchmark on the forward pass's shapes (M = 66048 tokens): TFLOP/s per kernel variant."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib  # noqa: E402

SHAPES = [("qkv", 66048, 3840, 1280, 0), ("out", 66048, 1280, 1280, 2), ("fc1", 66048, 5120, 1280, 1), ("fc2", 66048, 1280, 5120, 2)]
L = _lib.lib()
ITERS = int(os.environ.get("PGIBBS_BENCH_ITERS", "300"))
variants = [int(v) for v in (sys.argv[1:] or ["1", "2"])]
for rep in range(2):
    for name, M, N, K, epi in SHAPES:
        row = []
        for v in variants:
            ms = ctypes.c_double()
            _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, epi, v, ITERS, ctypes.byref(ms)))
            row.append("v%d %.3f ms %7.1f TF" % (v, ms.value, 2.0 * M * N * K / ms.value / 1e9))
        print("%-4s M=%d N=%d K=%d epi=%d | %s" % (name, M, N, K, epi, " | ".join(row)))
