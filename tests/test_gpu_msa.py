"""ESM-MSA-1b path on the MI355X against the CPU oracle: the two axial-attention kernels, the whole
forward (logits), and the Gibbs loops of ESM_MSA_sampler.generate / generate_single with every draw
replayed by the oracle from the logits the engine emitted."""
import random
import warnings

import numpy as np
import torch
import pytest

from oracle import draw as odraw
from oracle.msa_forward import MsaConfig, msa_forward, synthetic_msa_weights
from protein_gibbs_sampler_amd import _lib, esm_msa_sampler, models, weights
from test_gpu_kernels import _bf16

pytestmark = pytest.mark.gpu
BF16_TOL = 0.15


@pytest.mark.parametrize("B,R,C,H", [(2, 3, 20, 2), (1, 32, 257, 2), (2, 5, 70, 1), (1, 8, 130, 3), (1, 2, 300, 1),
                                     (1, 128, 513, 1), (1, 37, 100, 2), (40, 9, 65, 12),
                                     # width boundaries of the 4- / 9- / 8-wave instantiations, odd row counts in split-R
                                     (1, 4, 64, 2), (1, 9, 288, 1), (1, 64, 289, 1), (3, 16, 384, 2), (2, 7, 450, 2), (1, 13, 576, 1)])
def test_row_attention_kernel(B, R, C, H):
    rng = np.random.default_rng(C)
    d = H * 64
    qkv = rng.standard_normal((B, R, C, 3 * d), dtype=np.float32)
    scale = np.float32(0.125 / np.sqrt(R))
    ctx = np.empty((B, R, C, d), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_msa_attention(0, 0, _lib.ptr(qkv), _lib.ptr(ctx), B, R, C, H, float(scale)))
    r = _bf16(qkv).astype(np.float64)
    q, k, v = (r[..., i * d:(i + 1) * d].reshape(B, R, C, H, 64) for i in range(3))
    a = np.einsum("brihd,brjhd->bhij", q, k) * float(scale)
    p = np.exp(a - a.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = np.einsum("bhij,brjhd->brihd", p, v).reshape(B, R, C, d)
    assert np.abs(ctx - ref).max() < 2.5e-2 and np.abs(ctx - ref).mean() < 3e-3


@pytest.mark.parametrize("B,R,C,H", [(2, 3, 20, 2), (1, 32, 50, 2), (1, 128, 9, 1), (2, 1, 12, 1), (1, 17, 33, 3), (1, 600, 3, 1)])
def test_column_attention_kernel(B, R, C, H):
    rng = np.random.default_rng(R)
    d = H * 64
    qkv = rng.standard_normal((B, R, C, 3 * d), dtype=np.float32)
    qkv[..., :d] *= 0.35
    ctx = np.empty((B, R, C, d), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_msa_attention(0, 1, _lib.ptr(qkv), _lib.ptr(ctx), B, R, C, H, 1.0))
    r = _bf16(qkv).astype(np.float64)
    q, k, v = (r[..., i * d:(i + 1) * d].reshape(B, R, C, H, 64) for i in range(3))
    a = np.einsum("bichd,bjchd->bhcij", q, k)
    p = np.exp(a - a.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = np.einsum("bhcij,bjchd->bichd", p, v).reshape(B, R, C, d)
    assert np.abs(ctx - ref).max() < 2.5e-2 and np.abs(ctx - ref).mean() < 3e-3


CK = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=80, max_rows=16)


def _model(sd, ck=CK, precision="bf16"):
    cfg = weights.make_config(weights.MSA1B_CONFIG, d_model=ck["d_model"], n_layers=ck["n_layers"], d_ffn=ck["d_ffn"],
                              max_positions=ck["max_pos"], max_msa_rows=ck["max_rows"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return models.ESM_MSA1(state_dict=sd, config=cfg, precision=precision)


def test_msa_strict_mode_logits_within_1e3():
    ocfg = MsaConfig(**CK)
    sd = synthetic_msa_weights(ocfg, seed=4, std=0.08, embed_std=0.5, ln_jitter=0.1)
    m = _model(sd, precision="fp32").model.to("cuda:0")
    rng = np.random.default_rng(1)
    for (B, R, C) in [(2, 4, 21), (1, 7, 66), (3, 1, 10), (1, 3, 78)]:
        tok = rng.integers(4, 24, (B, R, C))
        tok[rng.random((B, R, C)) < 0.1] = 30
        tok[rng.random((B, R, C)) < 0.1] = 32
        tok[..., 0] = 0
        got = m.forward_logits(tok)
        want = msa_forward(sd, ocfg, tok)
        err = np.abs(got - want).max()
        print("\nMSA strict forward %s: max|engine - oracle| = %.3e (logit std %.2f)" % ((B, R, C), err, want.std()))
        assert err < 1e-3


def test_msa_forward_logits_vs_oracle():
    ocfg = MsaConfig(**CK)
    sd = synthetic_msa_weights(ocfg, seed=4, std=0.08, embed_std=0.5, ln_jitter=0.1)
    m = _model(sd).model.to("cuda:0")
    rng = np.random.default_rng(1)
    for (B, R, C) in [(2, 4, 21), (1, 7, 66), (3, 1, 10), (1, 16, 70)]:
        tok = rng.integers(4, 24, (B, R, C))
        tok[rng.random((B, R, C)) < 0.1] = 30
        tok[rng.random((B, R, C)) < 0.1] = 32
        tok[..., 0] = 0
        got = m.forward_logits(tok)
        want = msa_forward(sd, ocfg, tok)
        err = np.abs(got - want).max()
        print("\nMSA forward %s: max|engine - oracle| = %.3e (logit std %.2f), argmax agreement %.4f"
              % ((B, R, C), err, want.std(), (got.argmax(-1) == want.argmax(-1)).mean()))
        assert err < BF16_TOL


def test_msa_forward_wide_alignment():
    """C > 576 columns: row attention takes the fp32-scores path."""
    ck = dict(d_model=128, n_layers=1, n_heads=2, d_ffn=256, max_pos=700, max_rows=4)
    ocfg = MsaConfig(**ck)
    sd = synthetic_msa_weights(ocfg, seed=9, std=0.08, embed_std=0.5, ln_jitter=0.1)
    m = _model(sd, ck).model.to("cuda:0")
    rng = np.random.default_rng(2)
    tok = rng.integers(4, 24, (1, 2, 640))
    tok[..., 0] = 0
    got = m.forward_logits(tok)
    want = msa_forward(sd, ocfg, tok)
    assert np.abs(got - want).max() < BF16_TOL


def test_msa_generate_draws_replay_exactly():
    ocfg = MsaConfig(**CK)
    sd = synthetic_msa_weights(ocfg, seed=6, std=0.08, embed_std=0.5, ln_jitter=0.1)
    s = esm_msa_sampler.ESM_MSA_sampler(_model(sd), device="gpu")
    s.draw_seed, s.record = 7, True
    msa = ["MEPAATGQEAEECAHSGRGEAW", "MEP-ATGQEAEECAHSG-GEAW", "MKPAATGQ--EECAHSGRGEAV"]
    L, R, B, P, iters = len(msa[0]), 3, 2, 4, 3
    random.seed(11)
    out = s.generate(2 * R * B, msa, batch_size=B, num_iters=iters, num_positions=P, top_k=3, burnin=1, temperature=0.8,
                     show_progress_bar=False)
    assert len(out) == 2 * R * B and all(len(x) == L for x in out)
    random.seed(11)
    for rnd, run in enumerate(s.last_run):
        table = np.asarray([[[random.sample(range(1, L + 1), P) for _ in range(R)] for _ in range(B)] for _ in range(iters)])
        assert (run["table"] == table).all()
        tok = s.get_init_msa(msa, L, B).numpy().astype(np.int32)
        for it in range(iters):
            rows = run["sampled_logits"][it].reshape(-1, 33)
            rid = rnd * B * R + np.repeat(np.arange(B * R), P)
            want = odraw.draw_rows(rows, s.valid_aa_idx, 3, it < 1, 0.8, rid, it, np.tile(np.arange(P), B * R), 0, 7).reshape(B, R, P)
            assert (want == run["sampled_tokens"][it]).all()
            tin = tok.copy()
            for b in range(B):
                for r in range(R):
                    tin[b, r, table[it, b, r]] = 32
            ref = msa_forward(sd, ocfg, tin)
            ref_rows = np.stack([ref[b, r, table[it, b, r]] for b in range(B) for r in range(R)]).reshape(-1, 33)
            assert np.abs(rows - ref_rows).max() < BF16_TOL
            for b in range(B):
                for r in range(R):
                    tok[b, r, table[it, b, r]] = want[b, r]
        assert (tok == run["tokens"]).all()


@pytest.mark.parametrize("target_index", [0, -1, 1])
def test_msa_generate_single_native(target_index):
    ocfg = MsaConfig(**CK)
    sd = synthetic_msa_weights(ocfg, seed=8, std=0.08, embed_std=0.5, ln_jitter=0.1)
    s = esm_msa_sampler.ESM_MSA_sampler(_model(sd), device="cuda:0")
    s.draw_seed, s.record = 3, True
    msa = ["ACDEFGHIKLMNPQ", "ACDEFGHIKLMNPQ", "AC-EFGHIKLMNPV"]
    L, R = len(msa[0]), 3
    random.seed(2)
    out = s.generate_single(msa, steps=4, passes=2, burn_in=1, target_index=target_index, k=1, exclude_positions=[0, 5])
    run = s.last_run[0]
    # host control logic: shuffle + partition exactly as the reference (esm_msa_sampler.py:119-131)
    random.seed(2)
    positions = [p for p in range(1, L + 1) if p not in (1, 6)]
    steps = []
    for _ in range(2):
        random.shuffle(positions)
        steps += esm_msa_sampler.partition(positions, 4)
    tr = target_index % R
    tok = s.get_init_msa(msa, L, 1).numpy().astype(np.int32)
    for i, st in enumerate(steps):
        assert run["table"][i, 0, :len(st)].tolist() == st and (run["table"][i, 0, len(st):] == -1).all()
        tok[0, R - 1, st] = 32                                   # quirk Q2: row -1 is the one masked
        rows = run["sampled_logits"][i][:len(st)]
        ref = msa_forward(sd, ocfg, tok)[0, tr, st]
        assert np.abs(rows - ref).max() < BF16_TOL
        want = odraw.draw_rows(rows, s.valid_aa_idx, 1, i < 4, None, [tr] * len(st), i, np.arange(len(st)), 0, 3)
        assert (want == run["sampled_tokens"][i][:len(st)]).all()
        tok[0, tr, st] = want
    assert (tok == run["tokens"]).all()
    assert out == s.untokenize_batch(tok)[target_index]


def test_generate_single_batch_equals_serial_calls_small_msas():
    """Row e5 on small alignments (the few-token regime: the engine runs the batch's templates one after the other) and on a
    list with mixed shapes and exclusion lists: strings, recorded logits and tokens equal the serial generate_single calls."""
    ocfg = MsaConfig(**CK)
    sd = synthetic_msa_weights(ocfg, seed=8, std=0.08, embed_std=0.5, ln_jitter=0.1)
    s = esm_msa_sampler.ESM_MSA_sampler(_model(sd), device="cuda:0")
    s.record = True
    a = ["ACDEFGHIKLMNPQ", "ACDEFGHIKLMNPQ", "AC-EFGHIKLMNPV"]
    b = ["MEPAATGQEAEECA", "MEP-ATGQEAEECA", "MKPAATGQ--EECA"]
    c = ["ACDEFGHIKL", "ACDEFGHIKV"]
    msas, excl = [a, b, c, a, b], [[0, 5], None, [], [1], [2, 3, 4]]
    random.seed(2)
    torch.manual_seed(9)
    serial, runs = [], []
    for m, e in zip(msas, excl):
        serial.append(s.generate_single(m, steps=4, passes=2, burn_in=1, target_index=-1, k=2, exclude_positions=e))
        runs.append(s.last_run[0])
    state = random.getrandbits(32)
    random.seed(2)
    torch.manual_seed(9)
    batched = s.generate_single_batch(msas, steps=4, passes=2, burn_in=1, target_index=-1, k=2, exclude_positions=excl, max_batch=3)
    assert batched == serial and random.getrandbits(32) == state
    for j, (one, many) in enumerate(zip(runs, s.last_run)):
        P = one["table"].shape[-1]
        assert (many["table"][:, :, :P] == one["table"]).all() and (many["table"][:, :, P:] == -1).all()
        valid = one["table"][:, 0] >= 0
        assert np.array_equal(many["sampled_logits"][:, :P][valid], one["sampled_logits"][valid]), j
        assert (many["sampled_tokens"][:, :P][valid] == one["sampled_tokens"][valid]).all() and (many["tokens"] == one["tokens"]).all()


_COLFUSE_CHILD = r"""
import sys, warnings, numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import models, weights
prec = sys.argv[1]
cfg = weights.make_config(weights.MSA1B_CONFIG, n_layers=3)
sd = weights.synthetic_state_dict(cfg, seed=12, std=0.025, embed_std=0.3, ln_jitter=0.1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = models.ESM_MSA1(state_dict=sd, config=cfg, precision=prec).model.to("cuda:0")
rng = np.random.default_rng(22)
outs = {}
for (B, R, C) in [(3, 32, 41), (2, 64, 23), (1, 128, 37), (1, 256, 9), (2, 32, 257), (2, 24, 30)]:
    tok = rng.integers(4, 24, (B, R, C))
    tok[rng.random((B, R, C)) < 0.1] = 30
    tok[rng.random((B, R, C)) < 0.05] = 32
    tok[..., 0] = 0
    outs["%%dx%%dx%%d" %% (B, R, C)] = m.forward_logits(tok)
np.savez(sys.argv[2], **outs)
"""


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_fused_column_attention_is_bit_identical_with_the_unfused_path(precision, tmp_path):
    """gemm_colattn.hip (round 4): the column block's QKV projection and attention in one launch -- LayerNorm rows in column-major
    token order, q / k / v as bf16 planes in LDS, attention_kernel's arithmetic per (sequence, query block) -- must give the logits
    of the unfused path (projection kernel + attention kernel) BIT FOR BIT at every supported depth (32, 64, 128, 256), with row
    counts that are not multiples of 256 (pad sequences) and at config 4's width; depth 24 takes the unfused path either way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for fuse in ("1", "0"):
        out = tmp_path / ("colfuse%s.npz" % fuse)
        p = subprocess.run([sys.executable, "-c", _COLFUSE_CHILD % root, precision, str(out)], capture_output=True, text=True,
                           env=dict(os.environ, PGIBBS_MSA_COLFUSE=fuse), timeout=1200)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        res[fuse] = np.load(out)
    for k in res["1"].files:
        assert np.isfinite(res["1"][k]).all(), k
        assert np.array_equal(res["1"][k].view(np.uint32), res["0"][k].view(np.uint32)), k


_ROWLN_CHILD = """
import sys, warnings, numpy as np
sys.path.insert(0, %r)
from protein_gibbs_sampler_amd import models, weights
cfg = weights.make_config(weights.MSA1B_CONFIG, d_model=768, n_layers=2, d_ffn=1024, max_positions=600, max_msa_rows=128)
sd = weights.synthetic_state_dict(cfg, seed=3, std=0.03, embed_std=0.3, ln_jitter=0.1)
outs = []
for prec in ("bf16", "fp16"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = models.ESM_MSA1(state_dict=sd, config=cfg, precision=prec).model.to("cuda:0")
    # depth 32 / 64 / 128: the fused column block (LayerNorm rows in column-major token order); depth 5 / 37: ordinary order; row counts
    # that are no multiple of the 112-row tile, a last panel shifted up against the padded rows, several MSAs per call
    for (B, R, C) in [(1, 32, 257), (2, 5, 70), (3, 64, 100), (1, 37, 130), (1, 128, 513), (4, 32, 57)]:
        rng = np.random.default_rng(B * 100 + R + C)
        tok = rng.integers(4, 24, (B, R, C))
        tok[rng.random((B, R, C)) < 0.1] = 30
        tok[rng.random((B, R, C)) < 0.08] = 32
        tok[..., 0] = 0
        outs.append(m.forward_logits(tok.astype(np.int32))[:, ::3, ::5].copy())
np.savez(sys.argv[1], *outs)
"""


def test_full_row_out_projection_with_layernorm_epilogue_is_bit_identical(tmp_path):
    """gemm_rowln.hip (round 6): ESM-MSA-1b's attention out-projections as tiles of 112 token rows x all 768 columns whose epilogue
    adds the residual and normalises the finished rows (no LayerNorm launch, 2.4 GB less HBM traffic per out-projection at config
    4).  Same MFMA / k order, the ping-pong epilogue's x_old + (acc + bias), ln_row.h's LayerNorm on one wave per row: the logits
    are bit-identical with the two-launch path (PGIBBS_ROWLN=0), in both operand flavours, token-order and column-major h rows."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for sw in ("1", "0"):
        f = str(tmp_path / ("rowln_%s.npz" % sw))
        p = subprocess.run([sys.executable, "-c", _ROWLN_CHILD % root, f], capture_output=True, text=True,
                           env=dict(os.environ, PGIBBS_ROWLN=sw, PGIBBS_ROWLN_MIN_ROWS="0"), timeout=1200)
        assert p.returncode == 0, p.stderr[-3000:]
        res[sw] = np.load(f)
    assert len(res["0"].files) == 12
    for k in res["0"].files:
        assert np.isfinite(res["0"][k]).all() and res["0"][k].std() > 0.1
        assert (res["1"][k].view(np.uint32) == res["0"][k].view(np.uint32)).all(), k


def test_full_row_kernel_against_the_two_launches_on_raw_operands():
    """pg_dbg_rowln_bench: gemm_rowln.hip and residual GEMM + LayerNorm kernel applied once to the same synthetic operands (16 448 and
    65 664 token rows: a last row panel shifted against the padding, K = 768 and 256): x (fp32) and h (16-bit) equal bit for bit."""
    import ctypes
    L = _lib.lib()
    for M, K in ((16448, 768), (65664, 768), (5000, 256)):
        ms = (ctypes.c_double * 5)()
        md = ctypes.c_double(-1)
        _lib.check(L.pg_dbg_rowln_bench(0, M, K, 1, ms, ctypes.byref(md)))
        assert md.value == 0.0, (M, K, md.value)
