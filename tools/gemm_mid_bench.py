#!/usr/bin/env python3
"""GEMM time (us) for small and medium token counts: default dispatch vs forced 64x64 / 128x128 tile kernels."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, epi, v):
    ms = ctypes.c_double()
    rc = L.pg_dbg_gemm_bench(0, M, N, K, epi, v, 500, ctypes.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")
for M in (16, 32, 64, 128, 192, 256, 512, 1024, 2048, 4096, 8192, 16384):
    row = []
    for name, N, K, epi in (("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2)):
        row.append("%s %6.1f /%6.1f /%6.1f" % (name, run(M, N, K, epi, 2), run(M, N, K, epi, 6), run(M, N, K, epi, 7)))
    print("M=%5d  (default / 64^2 / 128^2 us) | %s" % (M, " | ".join(row)))
