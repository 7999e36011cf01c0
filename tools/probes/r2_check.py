#!/usr/bin/env python3
"""Two-resident residual GEMM (csrc/gemm_r2.hip) against the 8-wave 256 x 256 kernel and numpy: x += a w^T + b must be
BIT-IDENTICAL between the two kernels (same MFMA, same k order, bias before the residual) at shapes with and without tail
tiles.  One child process per kernel (the switch PGIBBS_GEMM_RESID is read once)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(66048, 1280, 1280), (66048, 1280, 320), (16384, 768, 768), (8448, 1280, 1280), (33024, 1280, 5120), (131072, 768, 192),
          (2048 * 17, 2048, 128)]

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import _lib
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(M + N + K)
a = rng.standard_normal((M, K), dtype=np.float32)
w = rng.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))
b = rng.standard_normal(N, dtype=np.float32) * 0.3
x = rng.standard_normal((M, N), dtype=np.float32) * 2 + 1.5
_lib.check(_lib.lib().pg_dbg_gemm(0, 0, _lib.ptr(a), _lib.ptr(w), _lib.ptr(b), _lib.ptr(x), M, N, K, 2))
np.save(sys.argv[4], x)
"""


def bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def main():
    bad = 0
    for M, N, K in SHAPES:
        outs = {}
        for mode in ("pp", "r2"):
            path = "/tmp/r2chk_%s.npy" % mode
            env = dict(os.environ, PGIBBS_GEMM_RESID=mode)
            p = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(M), str(N), str(K), path], env=env, capture_output=True, text=True)
            if p.returncode:
                print("FAIL child", mode, M, N, K, p.stderr[-2000:])
                bad += 1
                continue
            outs[mode] = np.load(path)
        if len(outs) < 2:
            continue
        same = np.array_equal(outs["pp"].view(np.uint32), outs["r2"].view(np.uint32))
        rng = np.random.default_rng(M + N + K)
        a = rng.standard_normal((M, K), dtype=np.float32)
        w = rng.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))
        b = rng.standard_normal(N, dtype=np.float32) * 0.3
        x = rng.standard_normal((M, N), dtype=np.float32) * 2 + 1.5
        rows = np.r_[0:64, M // 2:M // 2 + 64, M - 600:M]          # a sample of rows incl. the tail-tile region
        want = x[rows].astype(np.float64) + bf16(a[rows]).astype(np.float64) @ bf16(w).astype(np.float64).T + b
        err = np.abs(outs["r2"][rows] - want).max()
        nd = int((outs["pp"] != outs["r2"]).sum())
        print("M=%d N=%d K=%d: r2 == pp bitwise: %s (%d differ), max |r2 - float64| on %d rows = %.3e" % (M, N, K, same, nd, len(rows), err))
        if not same or err > 2e-3 * max(1.0, np.abs(want).max()):
            bad += 1
    print("OK" if not bad else "FAILED: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
