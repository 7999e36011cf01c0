#!/usr/bin/env python3
"""Cost of each epilogue on the four GEMM shapes at steady-state clocks (300 launches each)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, epi, v, iters=300):
    ms = ctypes.c_double()
    _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, epi, v, iters, ctypes.byref(ms)))
    return ms.value
M = 66048
for name, N, K in (("qkv", 3840, 1280), ("out", 1280, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120)):
    run(M, N, K, 0, 2, 100)
    none = run(M, N, K, 0, 30)
    row = ["none %.3f" % none]
    for epi, nm in ((0, "bf16"), (1, "bf16+gelu"), (3, "f32"), (2, "f32+resid")):
        row.append("%s %.3f" % (nm, run(M, N, K, epi, 2)))
    print("%-4s N=%d K=%d | %s" % (name, N, K, " | ".join(row)))
