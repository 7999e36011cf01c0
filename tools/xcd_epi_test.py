#!/usr/bin/env python3
"""Is the 256^2 kernel's epilogue burst bound by the whole chip's HBM or by each XCD's own path?  Runs the QKV / out-proj
shapes with only XCD 0 (or XCDs 0-1) executing tiles and compares time per round of tiles with the full chip."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, epi, v, iters=200):
    ms = ctypes.c_double()
    _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, epi, v, iters, ctypes.byref(ms)))
    return ms.value
M = 66048
for name, N, K, epi, full_v, one_v in (("qkv bf16", 3840, 1280, 0, 20, 38), ("out-proj resid", 1280, 1280, 2, 69, 68)):
    run(M, N, K, epi, 2, 50)
    tiles = (M // 256) * (N // 256)
    full = run(M, N, K, epi, full_v)
    one = run(M, N, K, epi, one_v)
    noepi = run(M, N, K, 0, 30) if epi == 0 else None
    print("%-15s tiles %d: all XCDs %.3f ms (%.2f us per round of 256 tiles) | XCD 0 alone %.3f ms (%.2f us per round of 32 tiles)%s"
          % (name, tiles, full, full / (tiles / 256) * 1e3, one, one / (tiles / 8 / 32) * 1e3,
             " | no-epilogue all XCDs %.3f ms (%.2f us/round)" % (noepi, noepi / (tiles / 256) * 1e3) if noepi else ""))
    if epi == 0:
        two = run(M, N, K, epi, 39)
        print("                XCDs 0-1: %.3f ms (%.2f us per round of 64 tiles)" % (two, two / (tiles / 4 / 64) * 1e3))
