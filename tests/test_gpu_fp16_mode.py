"""PG_PREC_F16: the throughput mode's kernels compiled a second time with IEEE fp16 operands (csrc/pg_common.h, namespace
pg::opf16).  Kernel level: every GEMM dispatch branch, the fused attention and the two MSA attention blocks
through their C-ABI debug entries against float64 numpy on fp16-rounded inputs -- the tolerances are the bf16 tests' divided by
the 8x finer mantissa.  Engine level: a small ESM-1b and a small ESM-MSA-1b against the fp32 oracle (the fp16 engine must be
several times closer than the bf16 engine on the same input), and a Gibbs run whose every draw replays bit-exactly through
oracle.draw from the logits the fp16 engine emitted (positions, scatter, draw and write-back do not depend on the mode)."""
import random
import warnings

import numpy as np
import pytest

from oracle import draw as odraw
from oracle.esm_forward import EsmConfig, esm1b_forward
from protein_gibbs_sampler_amd import _lib, esm_sampler, models, weights

pytestmark = pytest.mark.gpu


def _f16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x * 0.7071067811865476))


@pytest.mark.parametrize("M,N,K,epi", [(300, 256, 128, 0), (513, 1280, 1280, 0), (1000, 512, 5120, 1), (27, 1280, 1280, 0), (16, 3840, 1280, 1),
                                       (513, 1280, 1280, 2), (40, 1280, 5120, 2), (2048, 1280, 5120, 2), (2048, 2304, 192, 4),
                                       (513, 1280, 1280, 3), (700, 1280, 320, 4), (8192 + 256, 4096 + 256, 320, 4), (16384, 2304, 128, 3),
                                       (66048, 1280, 128, 2), (34048, 1280, 320, 3)])
def test_gemm_f16_operands(M, N, K, epi):
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.0 / np.sqrt(K))
    x[:, 0] += np.arange(M, dtype=np.float32) * (0.01 if M < 4096 else 1e-4)
    b = rng.standard_normal(N, dtype=np.float32)
    out = rng.standard_normal((M, N), dtype=np.float32) * 3
    res0 = out.astype(np.float64)
    _lib.check(_lib.lib().pg_dbg_gemm(0, _lib.PG_PREC_F16, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, epi))
    rows = np.arange(M) if M <= 8448 else np.r_[0:300, M // 2:M // 2 + 300, M - 700:M]
    ref = _f16(x[rows]).astype(np.float64) @ _f16(w).astype(np.float64).T + b
    if epi in (1, 4):
        ref = _gelu(ref)
    if epi == 2:
        ref = ref + res0[rows]
    got = out[rows]
    if epi >= 3:                                               # fp16 result: half an ulp (2^-12 relative) on top
        assert (np.abs(got - ref) <= 2e-3 * max(1.0, np.abs(ref).max()) + np.abs(ref) * 2.0 ** -11).all()
        assert (got == _f16(got)).all()
        return
    assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())


def test_gemm_f16_is_closer_to_fp32_than_bf16():
    """The point of the mode: on the same fp32 operands the fp16-operand product is ~8x closer to the exact one."""
    rng = np.random.default_rng(5)
    M, N, K = 1024, 1280, 1280
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(0.025)
    b = np.zeros(N, dtype=np.float32)
    exact = x.astype(np.float64) @ w.astype(np.float64).T
    errs = {}
    for name, prec in (("bf16", _lib.PG_PREC_BF16), ("fp16", _lib.PG_PREC_F16)):
        out = np.zeros((M, N), dtype=np.float32)
        _lib.check(_lib.lib().pg_dbg_gemm(0, prec, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, 0))
        errs[name] = np.abs(out - exact).mean()
    print("\nmean |GEMM error| vs float64: bf16 operands %.3e, fp16 operands %.3e" % (errs["bf16"], errs["fp16"]))
    assert errs["fp16"] * 5 < errs["bf16"]


@pytest.mark.parametrize("B,T,H", [(2, 27, 2), (3, 258, 2), (1, 16, 1), (1, 300, 2), (1, 577, 1), (2, 700, 2)])
def test_attention_f16_operands(B, T, H):
    rng = np.random.default_rng(T)
    d = H * 64
    qkv = rng.standard_normal((B, T, 3 * d), dtype=np.float32)
    qkv[..., :d] *= 0.35
    ctx = np.empty((B, T, d), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_attention(0, _lib.PG_PREC_F16, _lib.ptr(qkv), _lib.ptr(ctx), B, T, H))
    r = _f16(qkv).astype(np.float64)
    q, k, v = (r[..., i * d:(i + 1) * d].reshape(B, T, H, 64).transpose(0, 2, 1, 3) for i in range(3))
    a = q @ k.transpose(0, 1, 3, 2)
    p = np.exp(a - a.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
    assert np.abs(ctx - ref).max() < 4e-3, np.abs(ctx - ref).max()      # bf16 kernel: 2.5e-2
    assert np.abs(ctx - ref).mean() < 4e-4                               # bf16 kernel: 3e-3
    assert (ctx == _f16(ctx)).all()


@pytest.mark.parametrize("which,B,R,C,H", [(4, 1, 8, 40, 2), (4, 2, 32, 257, 2), (4, 1, 128, 513, 1), (5, 1, 8, 40, 2), (5, 2, 32, 257, 2)])
def test_msa_attention_f16_operands(which, B, R, C, H):
    rng = np.random.default_rng(R * C + which)
    d = H * 64
    qkv = rng.standard_normal((B, R, C, 3 * d), dtype=np.float32)
    scale = 0.125 / np.sqrt(R)
    if which == 5:
        qkv[..., :d] *= 0.125
    ctx = np.empty((B, R, C, d), dtype=np.float32)
    _lib.check(_lib.lib().pg_dbg_msa_attention(0, which, _lib.ptr(qkv), _lib.ptr(ctx), B, R, C, H, float(scale)))
    r = _f16(qkv).astype(np.float64)
    q, k, v = (r[..., i * d:(i + 1) * d].reshape(B, R, C, H, 64) for i in range(3))
    if which == 4:      # tied row attention: one C x C map per (b, h) from the scores summed over the rows
        a = np.einsum("brihd,brjhd->bhij", q, k) * scale
        p = np.exp(a - a.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        ref = np.einsum("bhij,brjhd->brihd", p, v).reshape(B, R, C, d)
    else:               # column attention: over the R rows of every column
        a = np.einsum("bichd,bjchd->bchij", q, k)
        p = np.exp(a - a.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        ref = np.einsum("bchij,bjchd->bichd", p, v).reshape(B, R, C, d)
    err = np.abs(ctx - ref)
    assert err.max() < 6e-3 and err.mean() < 5e-4, (err.max(), err.mean())


def test_small_esm1b_fp16_engine_against_the_oracle_and_the_bf16_engine():
    cfg = weights.make_config(weights.ESM1B_CONFIG, d_model=256, n_layers=6, d_ffn=1024, max_positions=300)
    sd = weights.synthetic_state_dict(cfg, seed=5, std=0.05, embed_std=0.3, ln_jitter=0.1)
    ocfg = EsmConfig(d_model=256, n_layers=6, n_heads=4, d_ffn=1024, max_pos=300)
    rng = np.random.default_rng(3)
    tok = np.concatenate([np.zeros((5, 1), np.int64), rng.integers(4, 24, (5, 256)), np.full((5, 1), 2)], axis=1)
    tok[:, 4:200:9] = 32
    want = esm1b_forward(sd, ocfg, tok)
    errs = {}
    for prec in ("bf16", "fp16"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = models.ESM1b(state_dict=sd, config=cfg, precision=prec).model.to("cuda:0")
        got = m.forward_logits(tok)
        assert np.isfinite(got).all()
        errs[prec] = (np.abs(got - want).max(), np.abs(got - want).mean())
    print("\nsmall ESM-1b (logit std %.2f): max / mean |logit err|  bf16 %.3e / %.3e   fp16 %.3e / %.3e"
          % (want.std(), errs["bf16"][0], errs["bf16"][1], errs["fp16"][0], errs["fp16"][1]))
    assert errs["fp16"][1] * 4 < errs["bf16"][1] and errs["fp16"][0] * 3 < errs["bf16"][0]


def test_fp16_gibbs_run_draws_replay_through_the_oracle():
    cfg = weights.make_config(weights.ESM1B_CONFIG, d_model=256, n_layers=4, d_ffn=512, max_positions=128)
    sd = weights.synthetic_state_dict(cfg, seed=5, std=0.05, embed_std=0.3, ln_jitter=0.1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = models.ESM1b(state_dict=sd, config=cfg, precision="fp16")
    sampler = esm_sampler.ESM_sampler(model, device="cuda:0")
    sampler.draw_seed, sampler.record = 99, True
    random.seed(1)
    seed = "MEPAATGQEAEECAHSGRGEAWEEV"
    out = sampler.generate(6, seed, batch_size=6, num_iters=4, num_positions=5, top_k=3, burnin=2, temperature=0.9, show_progress_bar=False)
    assert len(out) == 6 and all(len(s) == 25 for s in out)
    run = sampler.last_run[0]
    random.seed(1)
    ref_table = np.asarray([[random.sample(range(1, 26), 5) for _ in range(6)] for _ in range(4)])
    assert (run["table"] == ref_table).all()
    for it in range(4):
        rows = run["sampled_logits"][it].reshape(-1, 33)
        toks = odraw.draw_rows(rows, sampler.valid_aa_idx, 3, it < 2, 0.9, np.repeat(np.arange(6), 5), it, np.tile(np.arange(5), 6), 0, 99)
        assert (toks == run["sampled_tokens"][it].reshape(-1)).all()


# ---- precision="auto": fp16 operands behind a range guard (ADVICE r04 pg_common.h:57) ------------------------
_AUTO_CFG = dict(d_model=256, n_layers=3, d_ffn=512, max_positions=128)


def _auto_setup(scale_fc1=1.0, poison=None):
    cfg = weights.make_config(weights.ESM1B_CONFIG, **_AUTO_CFG)
    sd = weights.synthetic_state_dict(cfg, seed=21, std=0.05, embed_std=0.3, ln_jitter=0.1)
    sd = {k: v.copy() for k, v in sd.items()}
    sd["layers.1.fc1.weight"] *= np.float32(scale_fc1)
    if poison:
        sd[poison][0] = 1e5
    tok = np.concatenate([np.zeros((3, 1)), np.random.default_rng(2).integers(4, 24, (3, 40)), np.full((3, 1), 2)], axis=1).astype(np.int32)
    tok[1, 5] = tok[2, 17] = 32
    return cfg, sd, tok


def test_auto_precision_is_fp16_on_ordinary_weights():
    """models.* default since round 5: fp16 operands (8x closer to the fp32 reference than bf16), no warning, probe passed."""
    cfg, sd, tok = _auto_setup()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lm = models.ESM1b(state_dict=sd, config=cfg).model.to("cuda:0")
        got = lm.forward_logits(tok)
    assert lm.auto and lm.precision_name == "fp16"
    want = esm1b_forward(sd, EsmConfig(d_model=256, n_layers=3, n_heads=4, d_ffn=512, max_pos=128), tok)
    bf = models.ESM1b(state_dict=sd, config=cfg, precision="bf16").model.to("cuda:0").forward_logits(tok)
    e16, ebf = np.abs(got - want).max(), np.abs(bf - want).max()
    print("\nauto (fp16) %.2e vs bf16 %.2e from the fp32 oracle" % (e16, ebf))
    assert e16 < ebf / 3


def test_auto_precision_weight_scan_selects_bf16():
    cfg, sd, tok = _auto_setup(poison="layers.0.fc2.bias")
    with pytest.warns(UserWarning, match="layers.0.fc2.bias.*does not fit IEEE fp16"):
        lm = models.ESM1b(state_dict=sd, config=cfg).model
    assert lm.precision_name == "bf16"
    assert np.isfinite(lm.to("cuda:0").forward_logits(tok)).all()


def test_auto_precision_probe_moves_an_overflowing_checkpoint_to_bf16():
    """fc1 weights that fit fp16 (max ~ 2e4) but drive GELU(fc1) past 65504: the probe forward at .to() sees non-finite logits,
    the engine is rebuilt with bf16 operands, ONE warning; results are those of a bf16 engine bit for bit."""
    cfg, sd, tok = _auto_setup(scale_fc1=1e5)
    assert np.abs(sd["layers.1.fc1.weight"]).max() < 65504
    with pytest.warns(UserWarning) as rec:
        lm = models.ESM1b(state_dict=sd, config=cfg).model.to("cuda:0")
        got = lm.forward_logits(tok)
    assert len([w for w in rec if "rebuilding it with bf16" in str(w.message)]) == 1
    assert lm.precision_name == "bf16" and np.isfinite(got).all()
    bf = models.ESM1b(state_dict=sd, config=cfg, precision="bf16").model.to("cuda:0").forward_logits(tok)
    assert np.array_equal(got.view(np.uint32), bf.view(np.uint32))


def test_auto_precision_falls_back_inside_a_call_and_restores_the_tokens(monkeypatch):
    """Without the probe (PGIBBS_F16_PROBE=0: stands for an overflow only some later input triggers) the Gibbs call itself reports
    PG_ERR_RANGE before it writes the caller's tokens; the wrapper restores the tokens and runs THAT CALL again on the bf16 engine
    of the same weights: same tokens and emitted logits as a bf16 engine, one warning.  Round 6: the object's mode
    does not change -- later calls are again tried in fp16 (and, here, fall back again, silently)."""
    monkeypatch.setenv("PGIBBS_F16_PROBE", "0")
    cfg, sd, tok = _auto_setup(scale_fc1=1e5)
    table = np.stack([np.stack([np.random.default_rng(9 + i).choice(np.arange(1, 41), 4, replace=False) for _ in range(3)]) for i in range(2)]).astype(np.int32)
    params = _lib.make_sample_params(True, 32, 0, float("inf"), 1.0, list(range(4, 24)), rng_seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lm = models.ESM1b(state_dict=sd, config=cfg).model.to("cuda:0")          # no probe, no warning yet
    assert lm.precision_name == "fp16"
    t_auto = tok.copy()
    with pytest.warns(UserWarning, match="run again with bf16 operands") as rec:
        lg_auto, _ = lm.gibbs_run(t_auto, table, params, want_logits=True)
    assert len(rec) == 1 and lm.precision_name == "fp16" and lm.fallbacks == 1 and lm.take_fell_back()
    ref = models.ESM1b(state_dict=sd, config=cfg, precision="bf16").model.to("cuda:0")
    t_ref = tok.copy()
    lg_ref, _ = ref.gibbs_run(t_ref, table, params, want_logits=True)
    assert np.array_equal(t_auto, t_ref) and np.array_equal(lg_auto.view(np.uint32), lg_ref.view(np.uint32))
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                           # announced once
        got = lm.forward_logits(tok)
    assert lm.fallbacks == 2 and np.array_equal(got.view(np.uint32), ref.forward_logits(tok).view(np.uint32))


def _msa_auto_setup():
    from oracle.msa_forward import MsaConfig, synthetic_msa_weights
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=640, max_rows=16)
    sd = synthetic_msa_weights(MsaConfig(**ck), seed=4, std=0.08, embed_std=0.5, ln_jitter=0.1)
    cfg = weights.make_config(weights.MSA1B_CONFIG, d_model=128, n_layers=2, d_ffn=256, max_positions=640, max_msa_rows=16)
    rng = np.random.default_rng(8)

    def msa(R, C):
        t = rng.integers(4, 24, (1, R, C))
        t[rng.random((1, R, C)) < 0.1] = 30
        t[..., 0] = 0
        return t.astype(np.int32)
    return cfg, sd, msa


def test_auto_precision_routes_by_the_calls_own_shape_not_by_call_order():
    """an alignment wider than 576 columns used to flip the whole object to bf16, so every LATER narrow
    alignment was sampled in bf16 too -- results depended on call order (and, sharded, on the rank).  Now the wide call alone goes
    to the bf16 engine: wide == a bf16 engine's result, narrow == a fresh fp16 engine's result, in either order, no warning."""
    cfg, sd, msa = _msa_auto_setup()
    wide, narrow, padded = msa(3, 600), msa(3, 40), msa(3, 40)
    padded[0, 2, 30:] = 1                                                      # <pad>: the other shape the fp16 kernels lack
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        a = models.ESM_MSA1(state_dict=sd, config=cfg).model.to("cuda:0")
        b = models.ESM_MSA1(state_dict=sd, config=cfg).model.to("cuda:0")
        w_a, p_a, n_a = a.forward_logits(wide), a.forward_logits(padded), a.forward_logits(narrow)       # wide first
        n_b, w_b = b.forward_logits(narrow), b.forward_logits(wide)                                      # narrow first
    assert a.precision_name == b.precision_name == "fp16" and a.fallbacks == b.fallbacks == 0
    f16 = models.ESM_MSA1(state_dict=sd, config=cfg, precision="fp16").model.to("cuda:0")
    bf = models.ESM_MSA1(state_dict=sd, config=cfg, precision="bf16").model.to("cuda:0")
    for got in (n_a, n_b):
        assert np.array_equal(got.view(np.uint32), f16.forward_logits(narrow).view(np.uint32))
    for got in (w_a, w_b):
        assert np.array_equal(got.view(np.uint32), bf.forward_logits(wide).view(np.uint32))
    assert np.array_equal(p_a.view(np.uint32), bf.forward_logits(padded).view(np.uint32))


def _auto_shard_worker(rank, world, port, q):
    import os
    import random
    import torch
    import torch.distributed as dist
    from protein_gibbs_sampler_amd import esm_msa_sampler
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _auto_shard_job(True, 3 if rank == 0 else 77)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _auto_shard_job(shard, seed):
    import random
    import torch
    from protein_gibbs_sampler_amd import esm_msa_sampler
    cfg, sd, msa = _msa_auto_setup()
    inv = "LAGVSERTIDPKQNFYMHWC"
    def rows(t):
        return ["".join("-" if v == 30 else inv[v - 4] for v in r[1:]) for r in t[0]]
    jobs = [rows(msa(3, 600)), rows(msa(3, 40)), rows(msa(3, 40)), rows(msa(3, 40))]      # rank 0: wide + narrow, rank 1: two narrow
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = esm_msa_sampler.ESM_MSA_sampler(models.ESM_MSA1(state_dict=sd, config=cfg), device="cuda:0")
    s.shard_over_ranks = shard
    random.seed(seed)
    torch.manual_seed(seed)
    out = s.generate_single_batch(jobs, steps=3, passes=2, burn_in=1, target_index=0, k=1, max_batch=2)
    return out, s.model.model.precision_name


def test_sharded_auto_precision_with_a_wide_and_narrow_mix_equals_the_single_gpu_run():
    """The invariant sharding exists to keep, under precision="auto": rank 0 holds the wide alignment (bf16 engine) and a narrow
    one, rank 1 two narrow ones (fp16); both ranks return the single-process result, string for string (two gloo ranks sharing
    this GPU)."""
    import socket
    import torch.multiprocessing as mp
    want, mode = _auto_shard_job(False, 3)
    assert mode == "fp16"
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_auto_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        assert got[rank][0] == want and got[rank][1] == "fp16"


def test_batched_generate_single_overflow_is_repeated_template_by_template(monkeypatch):
    """A batched generate_single call that reports PG_ERR_RANGE is repeated one template at a time: since a template's result is
    that of a call on it alone, the outcome does not depend on how templates were grouped (or sharded).  The overflow is injected
    (PGIBBS_F16_PROBE=0 + fc1 weights that drive GELU(fc1) past 65504 make EVERY template overflow): results == a bf16 engine's."""
    monkeypatch.setenv("PGIBBS_F16_PROBE", "0")
    cfg, sd, msa = _msa_auto_setup()
    sd = {k: v.copy() for k, v in sd.items()}
    for k in sd:
        if k.endswith("fc1.weight"):
            sd[k] *= 1e5 / max(1e-9, float(np.abs(sd[k]).max())) * 0.6
    toks = np.concatenate([msa(3, 40), msa(3, 40)]).astype(np.int32)
    idx = np.tile(np.array([[3, 9, 17]], dtype=np.int32), (2, 2, 1))
    params = [_lib.make_sample_params(True, 32, 1, 0, None, list(range(4, 24)) + [30], rng_seed=5 + b) for b in range(2)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        auto = models.ESM_MSA1(state_dict=sd, config=cfg).model.to("cuda:0")
        bf = models.ESM_MSA1(state_dict=sd, config=cfg, precision="bf16").model.to("cuda:0")
        t_a, t_b = toks.copy(), toks.copy()
        lg_a, st_a = auto.gibbs_single_batch_run(t_a, 2, 0, idx, [1, 0], params, True, True)
        lg_b, st_b = bf.gibbs_single_batch_run(t_b, 2, 0, idx, [1, 0], params, True, True)
    assert auto.fallbacks >= 1 and auto.precision_name == "fp16"
    assert np.array_equal(t_a, t_b) and np.array_equal(st_a, st_b) and np.array_equal(lg_a.view(np.uint32), lg_b.view(np.uint32))


def test_explicit_fp16_reports_the_overflow_instead_of_sampling_from_nan():
    """precision="fp16" has no fallback: forward, Gibbs and log-probability entry points all return PG_ERR_RANGE, the caller's
    token buffer untouched."""
    cfg, sd, tok = _auto_setup(scale_fc1=1e5)
    lm = models.ESM1b(state_dict=sd, config=cfg, precision="fp16").model.to("cuda:0")
    with pytest.raises(_lib.PgError) as ei:
        lm.forward_logits(tok)
    assert ei.value.code == _lib.PG_ERR_RANGE and "fp16 range" in ei.value.msg
    table = np.tile(np.array([3, 9], dtype=np.int32), (2, 3, 1))
    params = _lib.make_sample_params(True, 32, 0, float("inf"), 1.0, list(range(4, 24)), rng_seed=3)
    t = tok.copy()
    with pytest.raises(_lib.PgError) as ei:
        lm.gibbs_run(t, table, params)
    assert ei.value.code == _lib.PG_ERR_RANGE and np.array_equal(t, tok)
    with pytest.raises(_lib.PgError) as ei:
        lm.forward_logprobs(tok, np.arange(3), np.tile(np.array([3, 9], dtype=np.int32), (3, 1)), np.full((3, 2), 5, dtype=np.int32))
    assert ei.value.code == _lib.PG_ERR_RANGE
    ok = models.ESM1b(state_dict=_auto_setup()[1], config=cfg, precision="fp16").model.to("cuda:0")       # ordinary weights: no error
    assert np.isfinite(ok.forward_logits(tok)).all()
