#!/usr/bin/env python3
"""Residual-update GEMMs (out-projection, fc2; ESM-1b and ESM-MSA-1b shapes, whole batches and 1/8 shards): the 8-wave 256 x 256
kernel (variant 2 with PGIBBS_GEMM_RESID=pp) against the two-resident 256 x 128 kernel (variant 50) and its ablations
(54: no epilogue, 53: one half-step + epilogue)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PGIBBS_GEMM_RESID", "pp")
from protein_gibbs_sampler_amd import _lib  # noqa: E402

SHAPES = [("out", 66048, 1280, 1280), ("fc2", 66048, 1280, 5120), ("out/8", 8448, 1280, 1280), ("fc2/8", 8448, 1280, 5120),
          ("msa_out", 526336, 768, 768), ("msa_fc2", 526336, 768, 3072), ("cfg5_out", 65792, 768, 768)]
L = _lib.lib()
ITERS = int(os.environ.get("PGIBBS_BENCH_ITERS", "200"))
variants = [int(v) for v in (sys.argv[1:] or ["2", "50", "54", "53"])]
for rep in range(2):
    for name, M, N, K in SHAPES:
        row = []
        for v in variants:
            ms = ctypes.c_double()
            it = ITERS if M < 200000 else max(20, ITERS // 5)
            _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, 2, v, it, ctypes.byref(ms)))
            row.append("v%d %.1f us %6.0f TF" % (v, ms.value * 1e3, 2.0 * M * N * K / ms.value / 1e9))
        print("%-8s M=%d N=%d K=%d | %s" % (name, M, N, K, " | ".join(row)), flush=True)
