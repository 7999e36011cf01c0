#!/usr/bin/env python3
"""Which tile kernel for the GEMMs of a few-chain job (8 ... 24 chains of L = 256: 2304 ... 6400 padded token rows)?  us per launch of the
four per-layer projections with every tile kernel forced (no K-splits: jobs above 2048 token rows keep the sequential k order)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, epi, v):
    ms = ctypes.c_double()
    rc = L.pg_dbg_gemm_bench(0, M, N, K, epi, v, 300, ctypes.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")
V = (("64^2", 6), ("128^2", 7), ("pp256", 20), ("w16-256", 80), ("pp192", 8))
print("us per launch: " + " / ".join(n for n, _ in V))
for M in (1280, 2304, 3328, 4352, 5376, 6400, 8448):
    row = []
    for name, N, K, epi in (("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2)):
        t = [run(M, N, K, epi, v) if not (v == 8 and epi != 2) else float("nan") for _, v in V]
        row.append("%s %s" % (name, " /".join("%6.1f" % x for x in t)))
    print("M=%5d | %s" % (M, " | ".join(row)), flush=True)
