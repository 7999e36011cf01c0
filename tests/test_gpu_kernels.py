"""Kernel-level parity on the MI355X, each kernel through its C-ABI debug entry point against numpy:
bf16 MFMA GEMM (+ fused epilogues), LayerNorm, fused attention.  Inputs are asymmetric/random so a
transposed operand or output cannot pass."""
import numpy as np
import pytest

from protein_gibbs_sampler_amd import _lib

pytestmark = pytest.mark.gpu


def _bf16(a):
    """round-to-nearest-even to bfloat16, returned as float32 (numpy restatement for the comparison)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x * 0.7071067811865476))


@pytest.mark.parametrize("M,N,K,epi", [(300, 256, 128, 0), (70, 128, 64, 0), (256, 384, 256, 1), (513, 1280, 1280, 0),
                                       (1000, 512, 5120, 1),
                                       # skinny (M <= 64) weight-streaming kernel
                                       (27, 1280, 1280, 0), (1, 128, 64, 0), (16, 3840, 1280, 1), (64, 1280, 5120, 0), (33, 256, 192, 1),
                                       (100, 1280, 1280, 0), (216, 5120, 1280, 1), (256, 768, 3072, 0), (129, 128, 64, 0),
                                       # epi 2 = residual read-modify-write (out += x w^T + b): 256^2, 128^2 and skinny kernels
                                       (513, 1280, 1280, 2), (700, 1280, 320, 2), (300, 384, 256, 2), (40, 1280, 5120, 2),
                                       (512, 1280, 320, 0), (512, 1280, 320, 1), (300, 256, 64, 2),
                                       # deep K, few tiles: split-K into scratch + fixed-order reduction (skinny, 64^2; K = 4*1280, 3*1024)
                                       (32, 1280, 5120, 2), (216, 1280, 5120, 2), (1000, 768, 3072, 2), (2048, 1280, 5120, 2), (48, 256, 2048, 2),
                                       # 128x128-tile kernel (>= 200 of its tiles, < 128 tiles of 256^2)
                                       (2048, 2304, 128, 2), (2048, 2304, 192, 4), (2304, 1792, 64, 0),
                                       # epi 3 / 4 = bf16 outputs (QKV / fc1 epilogues): one-shot 256^2, 128^2, skinny ...
                                       (513, 1280, 1280, 3), (700, 1280, 320, 4), (300, 384, 256, 3), (40, 1280, 5120, 4),
                                       # ... and >= 2 tiles per CU: the persistent kernel (ragged XCD shares, odd K-tile count)
                                       (8192, 4096, 256, 3), (8192 + 256, 4096 + 256, 320, 4), (16384, 2304, 128, 3)])
def test_gemm_bf16(M, N, K, epi):
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.0 / np.sqrt(K))
    x[:, 0] += np.arange(M, dtype=np.float32) * 0.01        # break any row/col symmetry
    b = rng.standard_normal(N, dtype=np.float32)
    out = rng.standard_normal((M, N), dtype=np.float32) * 3
    res0 = out.astype(np.float64)
    _lib.check(_lib.lib().pg_dbg_gemm(0, _lib.PG_PREC_BF16, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, epi))
    ref = _bf16(x).astype(np.float64) @ _bf16(w).astype(np.float64).T + b
    if epi in (1, 4):
        ref = _gelu(ref)
    if epi == 2:
        ref = ref + res0
    if epi >= 3:                                               # bf16 result: half an ulp (2^-9 relative) on top
        assert (np.abs(out - ref) <= 2e-3 * max(1.0, np.abs(ref).max()) + np.abs(ref) * 2.0 ** -8).all()
        assert (out == _bf16(out)).all()
        return
    err = np.abs(out - ref).max()
    assert err < 2e-3 * max(1.0, np.abs(ref).max()), err      # fp32 accumulation-order noise only


def test_gemm_dispatch_sweep():
    """Seeded sweep over the dispatch branches (weight streaming, 64^2, 128^2, 256^2 ping-pong, peeled panels, split-K), all
    epilogues: every branch must agree with numpy.  Shapes are drawn so that each branch is hit several times."""
    rng = np.random.default_rng(2024)
    cases = []
    for _ in range(10):                                   # small / medium M
        cases.append((int(rng.integers(1, 1400)), int(rng.choice([128, 256, 320, 768, 1280])), int(rng.choice([64, 192, 768, 2048, 3072])),
                      int(rng.choice([0, 1, 2, 3, 4]))))
    for panels, N in ((103, 1280), (129, 1280), (65, 2304), (52, 2560), (131, 768)):       # >= 128 tiles of 256^2, ragged last rounds
        cases.append((panels * 256 - int(rng.integers(0, 200)), N, int(rng.choice([128, 320])), int(rng.choice([0, 2, 3, 4]))))
    for M, N, K, epi in cases:
        x = rng.standard_normal((M, K), dtype=np.float32)
        w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.0 / np.sqrt(K))
        b = rng.standard_normal(N, dtype=np.float32)
        out = rng.standard_normal((M, N), dtype=np.float32)
        res0 = out.copy()
        _lib.check(_lib.lib().pg_dbg_gemm(0, _lib.PG_PREC_BF16, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, epi))
        ref = _bf16(x) @ _bf16(w).T + b
        if epi in (1, 4):
            ref = _gelu(ref.astype(np.float64)).astype(np.float32)
        if epi == 2:
            ref = ref + res0
        tol = 3e-3 * max(1.0, float(np.abs(ref).max())) + (np.abs(ref) * 2.0 ** -8 if epi >= 3 else 0)
        assert (np.abs(out - ref) <= tol).all(), (M, N, K, epi, float(np.abs(out - ref).max()))


@pytest.mark.parametrize("M,d", [(5, 128), (1000, 1280), (33, 768), (7, 256)])
def test_layernorm(M, d):
    rng = np.random.default_rng(d)
    x = (rng.standard_normal((M, d), dtype=np.float32) * 3 + 1).astype(np.float32)
    g = rng.standard_normal(d, dtype=np.float32)
    b = rng.standard_normal(d, dtype=np.float32)
    y = np.empty_like(x)
    _lib.check(_lib.lib().pg_dbg_layernorm(0, _lib.ptr(x), _lib.ptr(g), _lib.ptr(b), _lib.ptr(y), M, d, 1e-5))
    x64 = x.astype(np.float64)
    ref = (x64 - x64.mean(-1, keepdims=True)) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * g + b
    assert np.abs(y - ref).max() < 2e-5 * max(1, np.abs(ref).max())


@pytest.mark.parametrize("B,T,H", [(2, 27, 2), (3, 258, 2), (1, 16, 1), (2, 100, 3), (1, 200, 1), (1, 300, 2), (1, 513, 1),
                                   (1, 577, 1), (2, 700, 2), (1, 1024, 1),
                                   # 4 n + 1 query blocks (one wave alone in the last round) at every tile width of the ladder
                                   (2, 65, 1), (2, 129, 2), (1, 144, 1), (2, 193, 1), (1, 208, 2), (2, 257, 2), (1, 272, 1), (1, 321, 1),
                                   (1, 385, 2), (1, 449, 1), (1, 464, 1), (1, 528, 1)])
def test_attention(B, T, H):
    rng = np.random.default_rng(T)
    d = H * 64
    qkv = rng.standard_normal((B, T, 3 * d), dtype=np.float32)
    qkv[..., :d] *= 0.35                                       # q is pre-scaled in the engine
    ctx = np.empty((B, T, d), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_attention(0, _lib.PG_PREC_BF16, _lib.ptr(qkv), _lib.ptr(ctx), B, T, H))
    r = _bf16(qkv).astype(np.float64)
    q, k, v = (r[..., i * d:(i + 1) * d].reshape(B, T, H, 64).transpose(0, 2, 1, 3) for i in range(3))
    a = q @ k.transpose(0, 1, 3, 2)
    p = np.exp(a - a.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
    # P and the output are rounded to bf16 (8-bit mantissa): abs error ~ 2^-8 * |ctx|
    assert np.abs(ctx - ref).max() < 2.5e-2, np.abs(ctx - ref).max()
    assert np.abs(ctx - ref).mean() < 3e-3


_ATT_SPLIT_CHILD = """
import sys, numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import _lib
outs = []
for (B, T, H, prec) in [(3, 258, 2, 0), (33, 258, 20, 0), (1, 64, 1, 0), (7, 130, 3, 0), (40, 100, 20, 0), (2, 513, 2, 0), (5, 330, 4, 2), (27, 258, 20, 2)]:
    rng = np.random.default_rng(B * 1000 + T)
    d = H * 64
    qkv = rng.standard_normal((B, T, 3 * d), dtype=np.float32)
    qkv[..., :d] *= 0.35
    ctx = np.empty((B, T, d), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_attention(0, prec, _lib.ptr(qkv), _lib.ptr(ctx), B, T, H))
    outs.append(ctx)
np.savez(sys.argv[1], *outs)
"""


def test_attention_split_of_the_last_round_is_bit_identical(tmp_path):
    """Round 6: the (sequence, head) pairs of a launch's partial last round are split over 2-4 workgroups by query blocks
    (attention_kernel<.., SPLIT>: 640 pairs of a 32-chain shard = 512 whole + 128 x 4 quarter workgroups).  A query block's
    arithmetic does not depend on the workgroup that runs it: PGIBBS_ATTN_SPLIT=1 (default) and =0 give the same bits -- few pairs
    (everything split), more pairs than resident workgroups (33 x 20: 512 whole + 148 x 3), both operand flavours."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for sw in ("1", "0"):
        f = str(tmp_path / ("att_%s.npz" % sw))
        p = subprocess.run([sys.executable, "-c", _ATT_SPLIT_CHILD % root, f], capture_output=True, text=True,
                           env=dict(os.environ, PGIBBS_ATTN_SPLIT=sw), timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        res[sw] = np.load(f)
    assert len(res["0"].files) == 8
    for k in res["0"].files:
        assert np.isfinite(res["0"][k]).all() and np.abs(res["0"][k]).max() > 0
        assert (res["1"][k].view(np.uint32) == res["0"][k].view(np.uint32)).all(), k


_T192_CHILD = """
import sys, numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import _lib
outs = []
for (M, N, K, prec) in [(8448, 1280, 1280, _lib.PG_PREC_BF16), (8448, 1280, 5120, _lib.PG_PREC_BF16), (16640, 1280, 1280, _lib.PG_PREC_BF16),
                        (7168, 1280, 128, _lib.PG_PREC_BF16), (7424, 1280, 320, _lib.PG_PREC_BF16), (43008, 768, 768, _lib.PG_PREC_BF16),
                        (8448, 1280, 1280, _lib.PG_PREC_F16), (7168, 1280, 5120, _lib.PG_PREC_F16)]:      # the fp16-operand flavour too
    rng = np.random.default_rng(M + K)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.0 / np.sqrt(K))
    b = rng.standard_normal(N, dtype=np.float32)
    out = rng.standard_normal((M, N), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_gemm(0, prec, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, 2))
    outs.append(out[::7].copy())
np.savez(sys.argv[1], *outs)
"""


_LADDER_CHILD = """
import sys, numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import _lib
outs = []
# (live rows M, N, K, epilogue 2 = residual / 4 = bias + GELU with 16-bit output, precision); M is padded to 256 inside: the cases
# cover an exact fit (8448 = 48 x 176), a shifted last row panel (M_pad not a multiple of the height), live rows well below the padding,
# the shortest K (two K-tiles), the deep K of fc2 (tile grouping 2) and the fp16-operand flavour
for (M, N, K, epi, prec) in [(8256, 1280, 1280, 2, _lib.PG_PREC_BF16), (8256, 1280, 5120, 2, _lib.PG_PREC_BF16), (8256, 5120, 1280, 4, _lib.PG_PREC_BF16),
                             (16512, 1280, 1280, 2, _lib.PG_PREC_BF16), (3300, 2560, 128, 2, _lib.PG_PREC_BF16), (3300, 2560, 320, 4, _lib.PG_PREC_BF16),
                             (33024, 1280, 320, 2, _lib.PG_PREC_BF16), (8256, 1280, 1280, 2, _lib.PG_PREC_F16), (6000, 1536, 768, 4, _lib.PG_PREC_F16)]:
    rng = np.random.default_rng(M + K + epi)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.0 / np.sqrt(K))
    b = rng.standard_normal(N, dtype=np.float32)
    out = rng.standard_normal((M, N), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_gemm(0, prec, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, epi))
    outs.append(np.concatenate([out[::5], out[-300:]]).copy())          # a sample of all rows + the whole last row panels
np.savez(sys.argv[1], *outs)
"""


@pytest.mark.parametrize("height", [160, 176, 208, 224, 240])
def test_tile_height_ladder_is_bit_identical_with_256_row_tiles(tmp_path, height):
    """gemm_ladder.hip (round 6): the 8-wave kernel with (XJ0 + XJ1) x 16 token rows per tile, picked by launch_gemm_big when a
    height between 160 and 240 rows needs fewer round-equivalents than 256- / 192-row tiles (a 32-chain shard's out-projection /
    fc2: 235 tiles of 176 rows; its fc1: 740 tiles of 224).  PGIBBS_GEMM_LADDER=h forces height h wherever the epilogue has it
    (residual: all five; fc1's GELU epilogue: 208, 224, 240 -- the others then run the default dispatch), =0 forbids the ladder:
    equal bit for bit, incl. the row panels at the end (a last panel that would reach past the padded rows is shifted up and must
    not add to the residual rows of its neighbour twice)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for sw in (str(height), "0"):
        f = str(tmp_path / ("ladder_%s.npz" % sw))
        p = subprocess.run([sys.executable, "-c", _LADDER_CHILD % root, f], capture_output=True, text=True,
                           env=dict(os.environ, PGIBBS_GEMM_LADDER=sw, PGIBBS_GEMM_T192="0"), timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        res[sw] = np.load(f)
    assert len(res["0"].files) == 9
    for k in res["0"].files:
        assert np.isfinite(res["0"][k]).all()
        assert (res[str(height)][k] == res["0"][k]).all(), (height, k)


def test_192_row_tiles_are_bit_identical_with_256_row_tiles(tmp_path):
    """Residual GEMMs of mid-size batches run on 192 x 256 tiles when 256-row tiles would leave CUs idle (launch_gemm_big: a 32-chain
    shard's out-projection has 165 tiles of 256 rows for 256 CUs).  PGIBBS_GEMM_T192=2 forces them for every big residual GEMM --
    leftover rows of 0, 64 and 128 (64 x 64 tail tiles), both tile groupings, the shortest K -- =0 forbids them: equal bit for bit
    (same k order per accumulator), which is what keeps a shard's logits identical with the whole batch's."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for sw in ("2", "0"):
        f = str(tmp_path / ("t192_%s.npz" % sw))
        p = subprocess.run([sys.executable, "-c", _T192_CHILD % root, f], capture_output=True, text=True,
                           env=dict(os.environ, PGIBBS_GEMM_T192=sw), timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        res[sw] = np.load(f)
    assert len(res["2"].files) == 8
    for k in res["2"].files:
        assert np.isfinite(res["2"][k]).all()
        assert (res["2"][k] == res["0"][k]).all(), k
