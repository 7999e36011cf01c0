#!/usr/bin/env python3
"""Runs one GEMM shape/variant a few times (for rocprofv3 --pmc runs).  usage: gemm_prof.py VARIANT [M N K EPI]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib  # noqa: E402

v = int(sys.argv[1])
M, N, K, epi = [int(x) for x in sys.argv[2:6]] if len(sys.argv) >= 6 else (66048, 3840, 1280, 0)
ms = ctypes.c_double()
_lib.check(_lib.lib().pg_dbg_gemm_bench(0, M, N, K, epi, v, 3, ctypes.byref(ms)))
print("variant %d: %.3f ms, %.1f TF" % (v, ms.value, 2.0 * M * N * K / ms.value / 1e9))
