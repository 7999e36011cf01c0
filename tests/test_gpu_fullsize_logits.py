"""ALL emitted logits at the REAL model sizes against the fp32 CPU oracle, in both precision modes.

The call these tests stand for is `self.model.model(batch)["logits"]` (/root/reference/src/pgen/esm_sampler.py:223,
esm_msa_sampler.py:136,236).  north_star's tolerance is 1e-3 on emitted logits: the strict mode (PG_PREC_FP32:
split-bf16 x3 MFMA GEMMs + fp32 attention) is held to it here at 33 layers x d=1280 (ESM-1b) and 12 layers x d=768
(ESM-MSA-1b); the bf16 throughput mode is measured, printed and bounded (it does NOT meet 1e-3).
Weights are synthetic but scaled so the logits have a realistic spread (std 4-8; real ESM-1b logits span about -15..15).
"""
import warnings

import numpy as np
import pytest

from oracle.esm_forward import EsmConfig, esm1b_forward
from oracle.msa_forward import MsaConfig, msa_forward
from protein_gibbs_sampler_amd import models, weights

pytestmark = pytest.mark.gpu

STRICT_TOL = 1e-3      # north_star
# bf16 mode (does NOT meet 1e-3), at logit std ~ 10.  The MAX error and the argmax agreement are noisy
# statistics of a 33-layer bf16 pipeline: two builds whose LayerNorm differs only in the order of two fused multiply-adds
# measured 0.302 / 0.9987 and 0.347 / 0.9948 on the same inputs (MSA-1b 0.242 / 0.9936 and 0.212 / 0.9932; config 1 0.330 and
# 0.269), so they are held to the worse figure plus ~30 %; the MEAN error is stable (ESM-1b 0.0590 / 0.0592, MSA-1b 0.0356 /
# 0.0356) and held to +25 %.  (profiles/r03_gpu_parity_figures.txt)
BF16_MAX_ESM, BF16_AGREE_ESM, BF16_MEAN_ESM = 0.45, 0.990, 0.074
BF16_MAX_MSA, BF16_AGREE_MSA, BF16_MEAN_MSA = 0.32, 0.985, 0.045
BF16_MAX_CFG1 = 0.43
# fp16-operand mode (PG_PREC_F16, round 4): the same kernels, 3 more mantissa bits.  The CPU emulation of the engine's rounding points
# (tests/rounding_ablation.py, profiles/r04_rounding_ablation.txt) predicts max 0.041 / mean 0.0075 for ESM-1b where it predicts
# 0.326 / 0.062 for bf16 (measured 0.30-0.35 / 0.059); bounds = that prediction with the bf16 bounds' headroom.
F16_MAX_ESM, F16_AGREE_ESM, F16_MEAN_ESM = 0.07, 0.997, 0.0095
F16_MAX_MSA, F16_AGREE_MSA, F16_MEAN_MSA = 0.05, 0.996, 0.006
F16_MAX_CFG1 = 0.07


def _kl_valid(got, want, valid):
    """KL(softmax(got[valid]) || softmax(want[valid])) per row: what a bf16 logit error does to the distribution the sampler
    draws from (generate_step restricts to the valid residues before the softmax, esm_sampler.py:28-41)."""
    a = got[..., valid].astype(np.float64)
    b = want[..., valid].astype(np.float64)
    la = a - a.max(-1, keepdims=True)
    la = la - np.log(np.exp(la).sum(-1, keepdims=True))
    lb = b - b.max(-1, keepdims=True)
    lb = lb - np.log(np.exp(lb).sum(-1, keepdims=True))
    return (np.exp(la) * (la - lb)).sum(-1)


@pytest.fixture(scope="module")
def esm_case():
    cfg = dict(weights.ESM1B_CONFIG)
    sd = weights.synthetic_state_dict(cfg, seed=11, std=0.025, embed_std=0.3, ln_jitter=0.1)
    rng = np.random.default_rng(21)
    B, L = 3, 256                                                        # config 2's chain shape (T = 258)
    tok = np.concatenate([np.zeros((B, 1), np.int64), rng.integers(4, 24, (B, L)), np.full((B, 1), 2)], axis=1)
    for b in range(B):
        tok[b, rng.choice(np.arange(1, L + 1), 25, replace=False)] = 32   # 10 % masked, as in a Gibbs iteration
    want = esm1b_forward(sd, EsmConfig(), tok)
    return cfg, sd, tok, want


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_esm1b_full_size_all_logits(esm_case, precision):
    cfg, sd, tok, want = esm_case
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = models.ESM1b(state_dict=sd, config=cfg, precision=precision).model.to("cuda:0")
    got = m.forward_logits(tok)
    assert got.shape == want.shape == (3, 258, 33)
    err = np.abs(got - want)
    agree = (got.argmax(-1) == want.argmax(-1)).mean()
    print("\n[ESM-1b 33 x 1280, %s] max|engine - oracle| = %.3e  mean = %.3e  (logit std %.2f, max|logit| %.1f, argmax agreement %.4f)"
          % (precision, err.max(), err.mean(), want.std(), np.abs(want).max(), agree))
    # the distribution the sampler draws from, at the masked rows (the rows a Gibbs iteration samples)
    masked = tok == 32
    kl = _kl_valid(got[masked], want[masked], list(range(4, 24)))
    print("[ESM-1b 33 x 1280, %s] KL(engine || oracle) over the 20 residues at the %d masked rows: mean %.3e max %.3e"
          % (precision, masked.sum(), kl.mean(), kl.max()))
    if precision == "fp32":
        assert err.max() < STRICT_TOL
        assert agree > 0.999            # flips only between near-tied logits (gap < 2e-3)
        assert kl.max() < 1e-6
    elif precision == "fp16":
        assert err.max() < F16_MAX_ESM and agree >= F16_AGREE_ESM and err.mean() < F16_MEAN_ESM
        assert kl.mean() < 1.5e-5 and kl.max() < 4e-4          # the KL scales with the squared logit error: 64x below bf16's
    else:
        assert err.max() < BF16_MAX_ESM and agree >= BF16_AGREE_ESM and err.mean() < BF16_MEAN_ESM
        assert kl.mean() < 6e-4 and kl.max() < 1.5e-2          # measured 2.9e-4 / 4.7e-3 (ESM-1b), 2.4e-4 / 3.6e-3 (MSA-1b)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_esm1b_full_size_against_huggingface(precision):
    """The engine at the REAL ESM-1b sizes against logits recorded from HuggingFace's EsmForMaskedLM (tests/golden/esm_hf_full.npz,
    generator tests/golden/make_golden.py hf_full: an implementation of the architecture that shares no code and no author with
    oracle/esm_forward.py; the oracle itself agrees with it to 6.3e-5).  Strict mode: north_star's 1e-3 on all 2 x 258 x 33 logits."""
    import json
    from oracle.esm_forward import synthetic_esm_weights
    z = np.load(__file__.rsplit("/", 1)[0] + "/golden/esm_hf_full.npz")
    ck = json.loads(str(z["cfg"]))
    ocfg = EsmConfig(**ck)
    sd = synthetic_esm_weights(ocfg, seed=int(z["seed"]), std=float(z["std"]), embed_std=float(z["embed_std"]), ln_jitter=float(z["ln_jitter"]))
    cfg = dict(weights.ESM1B_CONFIG)
    assert (cfg["d_model"], cfg["n_layers"], cfg["d_ffn"], cfg["max_positions"]) == (ck["d_model"], ck["n_layers"], ck["d_ffn"], ck["max_pos"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = models.ESM1b(state_dict=sd, config=cfg, precision=precision).model.to("cuda:0")
    got = m.forward_logits(z["tokens"])
    err = np.abs(got - z["logits"])
    agree = (got.argmax(-1) == z["logits"].argmax(-1)).mean()
    print("\n[ESM-1b 33 x 1280 vs HuggingFace, %s] max|engine - HF| = %.3e  mean = %.3e  (logit std %.2f, argmax agreement %.4f)"
          % (precision, err.max(), err.mean(), z["logits"].std(), agree))
    if precision == "fp32":
        assert err.max() < STRICT_TOL and agree > 0.999
    elif precision == "fp16":
        assert err.max() < F16_MAX_ESM and agree >= 0.994 and err.mean() < F16_MEAN_ESM
    else:
        # measured 0.310 / mean 0.0604 / agreement 0.9845 (8 of 516 rows: these chains carry 10 % and 20 % masks -- more near-ties)
        assert err.max() < BF16_MAX_ESM and agree >= 0.975 and err.mean() < BF16_MEAN_ESM


@pytest.fixture(scope="module")
def msa_case():
    cfg = dict(weights.MSA1B_CONFIG)
    sd = weights.synthetic_state_dict(cfg, seed=12, std=0.025, embed_std=0.3, ln_jitter=0.1)
    rng = np.random.default_rng(22)
    B, R, C = 1, 32, 257                                                 # one MSA of config 4
    tok = rng.integers(4, 24, (B, R, C))
    tok[rng.random((B, R, C)) < 0.1] = 30                                # gaps
    for r in range(R):
        tok[0, r, rng.choice(np.arange(1, C), 25, replace=False)] = 32
    tok[..., 0] = 0
    want = msa_forward(sd, MsaConfig(), tok)
    return cfg, sd, tok, want


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_msa1b_full_size_all_logits(msa_case, precision):
    cfg, sd, tok, want = msa_case
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = models.ESM_MSA1(state_dict=sd, config=cfg, precision=precision).model.to("cuda:0")
    got = m.forward_logits(tok)
    assert got.shape == want.shape == (1, 32, 257, 33)
    err = np.abs(got - want)
    agree = (got.argmax(-1) == want.argmax(-1)).mean()
    print("\n[MSA-1b 12 x 768, %s] max|engine - oracle| = %.3e  mean = %.3e  (logit std %.2f, max|logit| %.1f, argmax agreement %.4f)"
          % (precision, err.max(), err.mean(), want.std(), np.abs(want).max(), agree))
    masked = tok == 32
    kl = _kl_valid(got[masked], want[masked], list(range(4, 24)) + [30])
    print("[MSA-1b 12 x 768, %s] KL(engine || oracle) over the 21 symbols at the %d masked positions: mean %.3e max %.3e"
          % (precision, masked.sum(), kl.mean(), kl.max()))
    if precision == "fp32":
        assert err.max() < STRICT_TOL
        assert agree > 0.999            # flips only between near-tied logits (gap < 2e-3)
        assert kl.max() < 1e-6
    elif precision == "fp16":
        assert err.max() < F16_MAX_MSA and agree >= F16_AGREE_MSA and err.mean() < F16_MEAN_MSA
        assert kl.mean() < 1.5e-5 and kl.max() < 4e-4
    else:
        assert err.max() < BF16_MAX_MSA and agree >= BF16_AGREE_MSA and err.mean() < BF16_MEAN_MSA
        assert kl.mean() < 6e-4 and kl.max() < 1.5e-2          # measured 2.9e-4 / 4.7e-3 (ESM-1b), 2.4e-4 / 3.6e-3 (MSA-1b)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_config1_full_size_single_chain(esm_case, precision):
    """BASELINE configuration 1 on the REAL ESM-1b shape: one chain of L = 25 (T = 27 tokens), 20 iterations, 10 % of the positions
    (P = 2) per iteration, top_k = 1, burnin = 10 -- the weight-streaming (skinny GEMM) + hipGraph regime of the engine.  Every
    iteration: positions == `random.sample`, the logits the engine sampled from == the fp32 oracle's logits for the engine's own
    masked token buffer, every draw replays bit-exactly from those logits, and the write-back is what the loop says."""
    import random
    from oracle import draw as odraw
    from protein_gibbs_sampler_amd import esm_sampler
    cfg, sd, _, _ = esm_case
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = models.ESM1b(state_dict=sd, config=cfg, precision=precision)
    s = esm_sampler.ESM_sampler(model, device="cuda:0")
    seed = "MEPAATGQEAEECAHSGRGEAWEEV"
    kw = dict(batch_size=1, num_iters=20, burnin=10, mask=True, in_order=False, num_positions_percent=10, top_k=1,
              show_progress_bar=False)
    s.draw_seed, s.record = 21, True
    random.seed(0)
    out = s.generate(1, seed, **kw)
    run = s.last_run[0]
    random.seed(0)
    table = np.asarray([[random.sample(range(1, 26), 2)] for _ in range(20)])
    assert (run["table"] == table).all()
    tok = s.get_init_seq(seed, 25, 1).numpy().astype(np.int32)
    ocfg = EsmConfig()
    worst = 0.0
    for it in range(20):
        tin = tok.copy()
        tin[0, table[it, 0]] = 32
        ref = esm1b_forward(sd, ocfg, tin)[0, table[it, 0]]
        rows = run["sampled_logits"][it].reshape(-1, 33)
        worst = max(worst, float(np.abs(rows - ref).max()))
        want = odraw.draw_rows(rows, s.valid_aa_idx, 1, it < 10, None, [0, 0], it, [0, 1], 0, 21)
        assert (want == run["sampled_tokens"][it].reshape(-1)).all()
        tok[0, table[it, 0]] = want
    print("\n[config 1, ESM-1b 33 x 1280, %s] max|engine - oracle| over 20 iterations = %.3e" % (precision, worst))
    assert worst < {"fp32": STRICT_TOL, "bf16": BF16_MAX_CFG1, "fp16": F16_MAX_CFG1}[precision]
    assert (tok == run["tokens"]).all() and out == s.untokenize_batch(torch_from(tok), True, True)
    # the unrecorded run replays the loop from one captured hipGraph: same strings
    s.record = False
    random.seed(0)
    assert s.generate(1, seed, **kw) == out


def torch_from(a):
    import torch
    return torch.from_numpy(a.astype(np.int64))
