#!/usr/bin/env python3
"""gemm_rowln.hip (ESM-MSA-1b out-projection with the next LayerNorm in its epilogue) against the residual GEMM + LayerNorm kernel it
replaces, on the token-row counts of configs 4 and 5 (pg_dbg_rowln_bench): ms per launch, the fused kernel's main loop alone and its
epilogue with four half-steps of main loop, and whether x / h are bit-identical between the two paths."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
print("%-10s %-5s | %8s %10s %12s | %9s %9s %9s | %s" % ("rows", "K", "fused", "main only", "4hs+epilogue", "resid GEMM", "LayerNorm", "sum", "max_diff (0 = same bits)"))
for M, K in ((526336, 768), (65664, 768), (262656, 768), (16448, 768), (526336, 3072)):
    ms = (ctypes.c_double * 5)()
    md = ctypes.c_double(-1)
    rc = L.pg_dbg_rowln_bench(0, M, K, 20, ms, ctypes.byref(md))
    if rc:
        print(M, K, "failed:", L.pg_last_error()); continue
    print("%-10d %-5d | %8.3f %10.3f %12.3f | %9.3f %9.3f %9.3f | %g" % (M, K, ms[0], ms[1], ms[2], ms[3], ms[4], ms[3] + ms[4], md.value), flush=True)
