#!/usr/bin/env python3
"""Per-tile cost of prologue + one K-tile + epilogue (variant 33 = bf16, 34 = fp32 residual, 30-style none = 36/30) as a
function of how many CUs are busy: is a lone CU's epilogue any faster than a chip-wide one?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, v, iters=2000):
    ms = ctypes.c_double()
    _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, 0, v, iters, ctypes.byref(ms)))
    return ms.value * 1e3
for M, N in ((512, 1024), (512, 4096), (1024, 8192), (2048, 8192), (4096, 16384), (8192, 16384)):
    tiles = (M // 256) * (N // 256)
    full, nost, one = run(M, N, 1280, 22), run(M, N, 1280, 36), run(M, N, 1280, 33)
    print("tiles=%5d (%.2f rounds): K=1280 no-MFMA full %.1f us | staging only %.1f us | 1 K-tile + epilogue %.1f us" % (tiles, tiles / 256, full, nost, one))
