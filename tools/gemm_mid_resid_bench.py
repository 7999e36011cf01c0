#!/usr/bin/env python3
"""Residual GEMMs (out += x w^T + b) at mid-size token counts: default dispatch / 64x64 / 128x128 / 192x256 / 256x256 (ping-pong) tiles, us."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, v):
    ms = ctypes.c_double()
    rc = L.pg_dbg_gemm_bench(0, M, N, K, 2, v, 300, ctypes.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")
for N, K in ((1280, 1280), (1280, 5120), (768, 768), (768, 3072)):
    for M in (2048, 3072, 4096, 5120, 6144, 7168, 8448, 9728, 12288, 16640, 20480, 24832):
        t = [run(M, N, K, v) for v in (2, 6, 7, 8, 20)]
        best = min(range(1, 5), key=lambda i: t[i])
        print("N=%4d K=%4d M=%5d tiles256=%4d | default %7.1f | 64^2 %7.1f | 128^2 %7.1f | 192 %7.1f | 256pp %7.1f | best %s%s" % (
            N, K, M, (M // 256) * (N // 256), t[0], t[1], t[2], t[3], t[4], ("64^2", "128^2", "192", "256pp")[best - 1],
            "" if t[0] <= 1.03 * t[best] else "  <-- default %.0f %% slower" % (100 * (t[0] / t[best] - 1))))
