"""Kernel-level parity of the strict precision mode (PG_PREC_FP32, the mode that meets north_star's 1e-3 logit tolerance) against
float64 numpy on the UNROUNDED fp32 inputs -- no bf16 rounding is granted to the kernels here:

  * the projection: one bf16 MFMA GEMM over the K-concatenated split operands [xl | xh | xh] . [wh | wl | wh]^T (engine.h dense3),
  * attention: q, k, v and P split into bf16 (hi, lo) pairs, three MFMAs per product (csrc/attention_f32.hip) -- whole-sequence,
    multi-tile (online softmax), tied-row scores + apply, column layout,
  * the all-VALU fp32 attention kernels (PGIBBS_ATTN_F32=valu, read once per process -> child process) as an independent
    second implementation: both must agree with numpy to the same bound.

Outputs come back as hi + lo of the operand rows the kernels write (16 mantissa bits), so the bound is ~2^-16 relative.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from protein_gibbs_sampler_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 6e-5          # (hi, lo) bf16 pair = 16-17 mantissa bits on every operand; fp32 accumulation


@pytest.mark.parametrize("M,N,K,epi", [(513, 1280, 1280, 0), (300, 384, 256, 2), (40, 256, 64, 0), (256, 128, 192, 2),
                                       (2048, 2304, 128, 0), (8192 + 256, 1280, 320, 2), (1000, 768, 3072, 0),
                                       # 258 tiles of 256 x 256: one full round + two m-panels as 64 x 64 tail tiles of the fused kernel
                                       (66048, 256, 64, 0), (66048, 256, 128, 2)])
def test_strict_projection_gemm(M, N, K, epi):
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.0 / np.sqrt(K))
    b = rng.standard_normal(N, dtype=np.float32)
    out = rng.standard_normal((M, N), dtype=np.float32) * 3
    res0 = out.astype(np.float64)
    _lib.check(_lib.lib().pg_dbg_gemm(0, _lib.PG_PREC_FP32, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, epi))
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    if epi == 2:
        ref = ref + res0
    err = np.abs(out - ref).max()
    # bf16 operands alone would sit at ~4e-3 here
    assert err < 2e-5 * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("M,N,K", [(513, 1280, 256), (300, 256, 64), (2100, 512, 320), (66048, 256, 64)])     # the last: tail tiles too
def test_strict_fc1_fused_gelu_split_epilogue(M, N, K):
    """fc1 of the strict mode: GELU and the (hi, lo) split of the result happen in the GEMM's epilogue, which writes fc2's operand
    rows [lo | hi | hi] (the debug entry checks that the two hi copies agree and returns hi + lo).  Against float64 erf-GELU:
    the epilogue's GELU is a fit with 3.2e-6 absolute error, the pair carries 16 mantissa bits."""
    from scipy.special import erf
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.5 / np.sqrt(K))
    b = rng.standard_normal(N, dtype=np.float32)
    out = np.empty((M, N), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_gemm(0, _lib.PG_PREC_FP32, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, 5))
    z = x.astype(np.float64) @ w.astype(np.float64).T + b
    ref = 0.5 * z * (1.0 + erf(z / np.sqrt(2.0)))
    err = np.abs(out - ref).max()
    print("\nfused GELU-split epilogue %s: max err %.3e (max |gelu| %.2f)" % ((M, N, K), err, np.abs(ref).max()))
    # the projection itself is held to 2e-5 * max|z| above; the GELU fit adds 3.2e-6; bf16 operands alone would sit at ~4e-3
    assert err < 3e-5 * max(1.0, np.abs(z).max()), err


_FUSED_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import _lib
outs = []
for M, N, K, epi in ((8448, 1280, 320, 0), (8448 + 256, 1280, 128, 2), (16384, 2304, 64, 0)):
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.0 / np.sqrt(K))
    b = rng.standard_normal(N, dtype=np.float32)
    out = rng.standard_normal((M, N), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_gemm(0, _lib.PG_PREC_FP32, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, epi))
    outs.append(out)
np.savez(sys.argv[1], *outs)
"""


def test_fused_three_product_kernel_is_bit_identical_with_the_plain_gemm(tmp_path):
    """Shapes with >= 128 tiles of 256 x 256 take gemm_split3_w16_kernel (three MFMA products per staged operand block); with
    PGIBBS_SPLIT3_FUSED=0 the same operands go through the plain bf16 GEMM over K' = 3K.  Same k order per accumulator ->
    the outputs must be equal bit for bit (this is what keeps strict-mode shards of any size identical)."""
    res = {}
    for fused in ("1", "0"):
        f = str(tmp_path / ("fused%s.npz" % fused))
        env = dict(os.environ, PGIBBS_SPLIT3_FUSED=fused)
        p = subprocess.run([sys.executable, "-c", _FUSED_CHILD % ROOT, f], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        res[fused] = np.load(f)
    for k in res["1"].files:
        assert (res["1"][k] == res["0"][k]).all(), k


_NODUP_CHILD = """
import sys, warnings
import numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import models, weights
rng = np.random.default_rng(5)
outs = []
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cfg = weights.make_config(weights.ESM1B_CONFIG, n_layers=2)
    lm = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=3), config=cfg, precision="fp32").model.to("cuda:0")
    tok = np.concatenate([np.zeros((40, 1), np.int64), rng.integers(4, 24, (40, 256)), np.full((40, 1), 2)], axis=1)
    tok[:, 7::11] = 32
    outs.append(lm.forward_logits(tok))                      # 10 320 token rows: every projection on the fused kernel
    outs.append(lm.forward_logits(tok[:3]))                  # 774 rows: the plain kernels, which read the duplicate block
    cfg = weights.make_config(weights.MSA1B_CONFIG, n_layers=2)
    lm = models.ESM_MSA1(state_dict=weights.synthetic_state_dict(cfg, seed=4), config=cfg, precision="fp32").model.to("cuda:0")
    msa = np.concatenate([np.zeros((2, 64, 1), np.int64), rng.integers(4, 24, (2, 64, 256))], axis=2)
    msa[:, 0, 5::9] = 32
    outs.append(lm.forward_logits(msa))                      # 32 896 token rows
    outs.append(lm.forward_logits(msa[:1, :4, :40]))
np.savez(sys.argv[1], *[np.asarray(o) for o in outs])
"""


def test_skipping_the_duplicate_operand_block_changes_nothing(tmp_path):
    """A producer of split operand rows (LayerNorm, fc1's epilogue) leaves the duplicate hi block of every 32-column group unwritten
    when its consumer is the fused three-product kernel, which reads [lo | hi] only (Engine::dense3_wants_dup, EPI_SPLIT2_GELU);
    PGIBBS_SPLIT3_NODUP=0 writes all three blocks as before.  Strict-mode logits of both engines, big (fused consumers) and small
    (plain consumers, which do read the block -- after a big forward has left the buffers without it), must be equal bit for bit."""
    res = {}
    for nodup in ("1", "0"):
        f = str(tmp_path / ("nodup%s.npz" % nodup))
        env = dict(os.environ, PGIBBS_SPLIT3_NODUP=nodup)
        p = subprocess.run([sys.executable, "-c", _NODUP_CHILD % ROOT, f], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        res[nodup] = np.load(f)
    assert len(res["1"].files) == 4
    for k in res["1"].files:
        assert np.isfinite(res["1"][k]).all()
        assert (res["1"][k] == res["0"][k]).all(), k


def _attention_ref(qkv, d, H):
    B, T = qkv.shape[:2]
    r = qkv.astype(np.float64)
    q, k, v = (r[..., i * d:(i + 1) * d].reshape(B, T, H, 64).transpose(0, 2, 1, 3) for i in range(3))
    a = q @ k.transpose(0, 1, 3, 2)
    p = np.exp(a - a.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(B, T, d)


ATT_CASES = [(2, 27, 2), (3, 258, 2), (1, 16, 1), (2, 64, 1), (1, 65, 1), (2, 100, 3), (1, 160, 1), (1, 161, 2), (1, 513, 1), (1, 1024, 1)]


@pytest.mark.parametrize("B,T,H", ATT_CASES)
def test_strict_attention(B, T, H):
    rng = np.random.default_rng(T)
    d = H * 64
    qkv = rng.standard_normal((B, T, 3 * d), dtype=np.float32)
    qkv[..., :d] *= 0.35                                       # q is pre-scaled in the engine
    ctx = np.empty((B, T, d), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_attention(0, _lib.PG_PREC_FP32, _lib.ptr(qkv), _lib.ptr(ctx), B, T, H))
    ref = _attention_ref(qkv, d, H)
    err = np.abs(ctx - ref).max()
    assert err < REL * max(1.0, np.abs(ref).max()), err       # the bf16 kernel is held to 2.5e-2 on the same inputs


MSA_CASES = [(2, 3, 20, 2), (1, 32, 257, 2), (2, 5, 70, 1), (1, 8, 130, 3), (1, 4, 64, 2), (1, 2, 300, 1), (1, 16, 513, 1), (3, 9, 65, 2)]


def _msa_refs(qkv, B, R, C, H, scale):
    d = H * 64
    r = qkv.astype(np.float64)
    q, k, v = (r[..., i * d:(i + 1) * d].reshape(B, R, C, H, 64) for i in range(3))
    a = np.einsum("brihd,brjhd->bhij", q, k) * float(scale)
    p = np.exp(a - a.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    row = np.einsum("bhij,brjhd->brihd", p, v).reshape(B, R, C, d)
    a = np.einsum("bichd,bjchd->bhcij", q, k)
    p = np.exp(a - a.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    col = np.einsum("bhcij,bjchd->bichd", p, v).reshape(B, R, C, d)
    return row, col


@pytest.mark.parametrize("B,R,C,H", MSA_CASES)
def test_strict_msa_attention(B, R, C, H):
    rng = np.random.default_rng(C + R)
    d = H * 64
    qkv = rng.standard_normal((B, R, C, 3 * d), dtype=np.float32)
    qkv[..., :d] *= 0.35
    scale = np.float32(1.0 / np.sqrt(R))
    row_ref, col_ref = _msa_refs(qkv, B, R, C, H, scale)
    for which, ref in ((2, row_ref), (3, col_ref)):
        ctx = np.empty((B, R, C, d), dtype=np.float32)
        _lib.check(_lib.lib().pg_dbg_msa_attention(0, which, _lib.ptr(qkv), _lib.ptr(ctx), B, R, C, H, float(scale)))
        err = np.abs(ctx - ref).max()
        assert err < REL * max(1.0, np.abs(ref).max()), (which, err)


_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_gpu_strict_kernels as t
for c in t.ATT_CASES[:6]:
    t.test_strict_attention(*c)
for c in t.MSA_CASES[:5]:
    t.test_strict_msa_attention(*c)
print("valu kernels OK")
"""


def test_valu_attention_kernels_agree_too():
    """PGIBBS_ATTN_F32=valu selects the all-VALU fp32 kernels: an independent formulation (no MFMA, no operand split, thread per
    query) that must meet the same bound against numpy -- a cross-check that the bound above is not an artefact of the split."""
    env = dict(os.environ, PGIBBS_ATTN_F32="valu")
    p = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, env=env,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "valu kernels OK" in p.stdout
