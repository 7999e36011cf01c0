"""LayerNorm inside the residual GEMMs (csrc/gemm_epilogue.h, EPI_F32_RESID_LN) and the 64 x 64 tail tiles that replace the
serial "peel" launches.

Kernel level (C ABI pg_dbg_gemm_resid_ln): x = resid + a w^T + b against numpy; h written by the launch itself (the workgroup that
completes a row panel normalises it) must equal, BIT FOR BIT, the stand-alone LayerNorm kernel on the same x -- at a shape
where the last m-panels go through tail tiles (66 048 rows: 256 + 2 panels), with several launches back to back (the arrival
counters reset themselves; who arrives last differs every time).  Engine level: the full ESM-1b / a small MSA model with the
fusion on and off give identical logits, and a chain's logits do not depend on its position in the batch.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from protein_gibbs_sampler_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("M,N,K,repeats", [(66048, 256, 128, 3), (66048, 1280, 128, 2), (16384, 768, 192, 2), (34048, 1280, 320, 1),
                                           (2048, 2048, 128, 1)])
def test_residual_gemm_normalises_its_finished_panels(M, N, K, repeats):
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) / np.float32(np.sqrt(K))
    b = rng.standard_normal(N, dtype=np.float32) * 0.3
    resid = rng.standard_normal((M, N), dtype=np.float32) * 2 + 1.5
    resid[:, ::7] *= 4
    gamma = (1 + 0.2 * rng.standard_normal(N)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(N)).astype(np.float32)
    x = resid.copy()
    h_fused = np.empty((M, N), dtype=np.float32)
    h_kernel = np.empty((M, N), dtype=np.float32)
    rc = _lib.lib().pg_dbg_gemm_resid_ln(0, _lib.ptr(a), _lib.ptr(w), _lib.ptr(b), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta),
                                         _lib.ptr(h_fused), _lib.ptr(h_kernel), M, N, K, 1e-5, repeats)
    if M == 34048:
        # 133 x 5 tiles: the XCDs' tile ranges do not start on group boundaries -> the launcher refuses to fuse (the engine
        # then runs GEMM + LayerNorm kernel, identical bits)
        assert rc == _lib.PG_ERR_UNSUPPORTED
        return
    _lib.check(rc)
    want_x = resid.astype(np.float64) + _bf16(a).astype(np.float64) @ _bf16(w).astype(np.float64).T + b
    assert np.abs(x - want_x).max() < 2e-3 * max(1.0, np.abs(want_x).max())
    assert np.array_equal(h_fused.view(np.uint32), h_kernel.view(np.uint32))        # same rows, same code: same bits
    x64 = x.astype(np.float64)
    ln = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * gamma + beta
    assert (np.abs(h_fused - ln) <= 1e-4 + np.abs(ln) * 2.0 ** -8).all()            # bf16 rounding of the fp32 LayerNorm


_CHILD = r"""
import sys, warnings, numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import models, weights
which = sys.argv[1]
if which == "esm":
    cfg = dict(weights.ESM1B_CONFIG)
    sd = weights.synthetic_state_dict(cfg, seed=11, std=0.025, embed_std=0.3, ln_jitter=0.1)
    rng = np.random.default_rng(21)
    tok = np.concatenate([np.zeros((3, 1), np.int64), rng.integers(4, 24, (3, 256)), np.full((3, 1), 2)], axis=1)
    tok[:, 5:60:4] = 32
    # 40 chains (10 320 token rows: big tiles + tail tiles): the three probe chains first and again at the END of the batch
    big = np.concatenate([tok, np.tile(tok[:1], (34, 1)), tok])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = models.ESM1b(state_dict=sd, config=cfg, precision="bf16").model.to("cuda:0")
    out = m.forward_logits(big)
    np.save(sys.argv[2], np.stack([out[:3], out[-3:]]))
else:
    cfg = dict(weights.MSA1B_CONFIG)
    sd = weights.synthetic_state_dict(cfg, seed=12, std=0.025, embed_std=0.3, ln_jitter=0.1)
    rng = np.random.default_rng(22)
    B, R, C = 6, 32, 257                                  # 49 344 token rows
    tok = np.concatenate([np.zeros((B, R, 1), np.int64), rng.integers(4, 24, (B, R, C - 1))], axis=2)
    tok[:, :, 7:90:5] = 32
    tok[-1] = tok[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = models.ESM_MSA1(state_dict=sd, config=cfg, precision="bf16").model.to("cuda:0")
    out = m.forward_logits(tok)
    np.save(sys.argv[2], np.stack([out[0], out[-1]]))
"""


@pytest.mark.parametrize("which", ["esm", "msa"])
def test_fused_layernorm_changes_nothing_in_the_engine(which, tmp_path):
    outs = {}
    for fuse in ("1", "0"):            # the switch is read at engine creation; a child process per setting keeps it honest
        env = dict(os.environ, PGIBBS_LN_FUSE=fuse)
        p = subprocess.run([sys.executable, "-c", _CHILD % ROOT, which, str(tmp_path / ("out%s.npy" % fuse))],
                           capture_output=True, text=True, env=env, timeout=1200)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        outs[fuse] = np.load(tmp_path / ("out%s.npy" % fuse))
    assert np.isfinite(outs["1"]).all()
    assert np.array_equal(outs["1"], outs["0"])                    # fused == LayerNorm kernel, bit for bit
    assert np.array_equal(outs["1"][0], outs["1"][1])              # and a chain / MSA does not care where in the batch it sits
