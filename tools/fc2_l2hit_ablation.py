"""Is fc2's fabric over-read costing time?  The K = 5120 main loop with every operand DMA served by the L2
(ablation 17 of gemm_bf16_pp_kernel: all K-steps re-read K-tile 0, so nothing but the first step leaves the XCD) against the real
loop, same tile order, bf16 epilogue on both (the ablation is instantiated for that epilogue); and the K = 1280 shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
for rep in range(2):
    for name, M, N, K in (("fc2 shape", 66048, 1280, 5120), ("out shape", 66048, 1280, 1280), ("qkv shape", 66048, 3840, 1280)):
        t = []
        for v in (20, 37):
            ms = ctypes.c_double()
            _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, 0, v, 200, ctypes.byref(ms)))
            t.append(ms.value * 1e3)
        print("%s M=%d N=%d K=%d bf16 epilogue: real loop %.1f us (%.0f TF) | every DMA an L2 hit %.1f us (%.0f TF) | %+.1f %%"
              % (name, M, N, K, t[0], 2.0 * M * N * K / t[0] / 1e6, t[1], 2.0 * M * N * K / t[1] / 1e6, 100 * (t[1] / t[0] - 1)), flush=True)
