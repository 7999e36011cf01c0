"""VERDICT r04 item 2, measured: ESM-1b's QKV projection fused with attention.  The fused kernel's best case -- whole sequences of
exactly 256 tokens, one head's [q | k | v] per 256 x 192 tile, 16 query blocks on 16 waves -- is gemm_colattn_kernel<16> as it
stands; this times it against the projection (256 x 256 tiles) + attention_kernel pair on the same operands (config-2 batch,
d = 1280, 20 heads), and at T = 128 / 64."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib

L = _lib.lib()
for B, T in ((256, 256), (256, 256), (512, 128), (1024, 64)):
    ms = (ctypes.c_double * 3)()
    diff = ctypes.c_double(-1)
    _lib.check(L.pg_dbg_qkv_attention_bench(0, B, T, 20, 20, ms, ctypes.byref(diff)))
    print("B=%4d T=%3d d=1280: fused %.1f us | projection %.1f + attention %.1f = %.1f us | fused - unfused = %+.1f us | max |ctx diff| %g"
          % (B, T, 1e3 * ms[0], 1e3 * ms[1], 1e3 * ms[2], 1e3 * (ms[1] + ms[2]), 1e3 * (ms[0] - ms[1] - ms[2]), diff.value), flush=True)
