#!/usr/bin/env python3
"""Error map of a forced GEMM tile kernel (PGIBBS_GEMM=80: the 16-wave kernel for every 256-multiple shape) against numpy."""
import os, sys
os.environ.setdefault("PGIBBS_GEMM", "80")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from protein_gibbs_sampler_amd import _lib

def bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)

L = _lib.lib()
CASES = [(512, 256, 64, 0), (512, 256, 128, 0), (512, 512, 256, 0), (512, 512, 512, 0), (512, 512, 1280, 0),
         (256, 256, 256, 3), (2048, 1024, 320, 2), (256, 256, 256, 1), (256, 256, 256, 4)]
if os.environ["PGIBBS_GEMM"] not in ("80",):
    CASES = [c for c in CASES if c[3] == 3]        # ablation variants exist for the bf16 epilogue only
    CASES += [(256, 256, 320, 3), (512, 512, 1280, 3)]
for (M, N, K, epi) in CASES:
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))
    b = rng.standard_normal(N, dtype=np.float32)
    out = rng.standard_normal((M, N), dtype=np.float32)
    res0 = out.astype(np.float64)
    _lib.check(L.pg_dbg_gemm(0, 0, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, epi))
    ref = bf16(x).astype(np.float64) @ bf16(w).astype(np.float64).T + b
    if epi in (1, 4):
        from scipy.special import erf
        ref = 0.5 * ref * (1 + erf(ref * 0.7071067811865476))
    if epi == 2:
        ref = ref + res0
    err = np.abs(out - ref)
    bad = err > (0.03 if epi >= 3 else 2e-3) * max(1.0, np.abs(ref).max())
    print("M=%d N=%d K=%d epi=%d: max err %.3e, bad %d of %d" % (M, N, K, epi, err.max(), bad.sum(), bad.size))
    FAIL = globals().get("FAIL", 0) + int(bad.any())
    if bad.any():
        mm, nn = np.nonzero(bad)
        print("   bad m%%256: %s" % np.unique(mm % 256)[:40])
        print("   bad n%%256: %s" % np.unique(nn % 256)[:40])
        print("   first bad:", list(zip(mm[:8], nn[:8])), "err", err[mm[:8], nn[:8]])

sys.exit(1 if globals().get("FAIL", 0) else 0)
