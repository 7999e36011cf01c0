"""ESM-1 architecture (PG_ARCH_ESM1: pgen.models.ESM6 / ESM12 / ESM34, /root/reference/src/pgen/models.py:69-82) on the HIP engine
against the fp32 oracle (oracle/esm1_forward.py): sqrt(d) embedding scale, sinusoidal positions, no embedding LayerNorms, the extra
bias_k / bias_v attention key, LayerNorm eps 1e-12, untied embed_out -- in all three precision modes, at sequence lengths where the
extra key crosses a 16-key block / tile boundary, with right-padded batches; the samplers on the ESM6 holder (tokenisation as
the reference's own tests expect, draws replayed bit-exactly from the emitted logits, log-likelihoods against the oracle); and
the reference's numeric KATs (test/test_esm_sampler.py:269-340), which run as soon as the 43 M checkpoint is present."""
import os
import random
import warnings

import numpy as np
import pytest

from oracle import draw as odraw
from oracle import esm1_forward as E
from protein_gibbs_sampler_amd import esm_sampler, models, weights

pytestmark = pytest.mark.gpu


def _case(n_layers=6, d=768, seed=7):
    cfg = weights.make_config(weights.ESM1_T6_CONFIG, n_layers=n_layers, d_model=d, d_ffn=4 * d)
    sd = weights.synthetic_state_dict(cfg, seed=seed, std=0.03, embed_std=0.05, ln_jitter=0.1)
    ocfg = E.Esm1Config(d_model=d, n_layers=n_layers, n_heads=d // 64, d_ffn=4 * d)
    return cfg, sd, ocfg


def _model(cfg, sd, precision):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return models.ESM6(state_dict=sd, config=cfg, precision=precision)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_esm1_forward_against_the_oracle(precision):
    cfg, sd, ocfg = _case()
    lm = _model(cfg, sd, precision).model.to("cuda:0")
    rng = np.random.default_rng(5)
    worst = 0.0
    # T + 1 keys: 17 (second 16-key block holds only the bias key), 28 (config 1), 65, 161 (the bias key alone in the strict
    # kernel's second 160-key tile), 259, 289 (first length past the 288-key kernel), 600 (long-sequence kernel)
    for T in (16, 27, 64, 160, 258, 288, 600):
        B = 3 if T < 300 else 1
        tok = np.concatenate([np.full((B, 1), 32), rng.integers(4, 24, (B, T - 1))], axis=1)
        tok[:, 2:T:7] = 33
        want = E.esm1_forward(sd, ocfg, tok)
        got = lm.forward_logits(tok)
        assert got.shape == want.shape == (B, T, 35)
        err = np.abs(got - want).max()
        worst = max(worst, err)
        tol = {"fp32": 1e-3, "bf16": 0.25, "fp16": 0.04}[precision]
        assert err < tol, (T, err)
    print("\n[ESM-1 6 x 768, %s] max|engine - oracle| over 7 sequence lengths = %.3e (logit std %.2f)" % (precision, worst, want.std()))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_esm1_right_padded_batch(precision):
    """<pad> keys are masked, positions count non-pad tokens, the bias key stays visible: a padded row equals the row alone."""
    cfg, sd, ocfg = _case(n_layers=3)
    lm = _model(cfg, sd, precision).model.to("cuda:0")
    rng = np.random.default_rng(9)
    tok = np.full((3, 40), 1, dtype=np.int64)
    lens = (40, 23, 9)
    for b, n in enumerate(lens):
        tok[b, 0] = 32
        tok[b, 1:n] = rng.integers(4, 24, n - 1)
    got = lm.forward_logits(tok)
    want = E.esm1_forward(sd, ocfg, tok)
    tol = 1e-3 if precision == "fp32" else 0.2
    for b, n in enumerate(lens):
        assert np.abs(got[b, :n] - want[b, :n]).max() < tol
        alone = lm.forward_logits(tok[b:b + 1, :n])
        assert np.abs(alone[0] - got[b, :n]).max() < (2e-4 if precision == "fp32" else 0.08)


def test_esm6_sampler_tokens_draws_and_likelihoods():
    cfg, sd, ocfg = _case(n_layers=4)
    model = _model(cfg, sd, "fp32")
    s = esm_sampler.ESM_sampler(model, device="cuda:0")
    # the reference's own expectations for the ESM-1 alphabet (test/test_esm_sampler.py:43-60)
    assert s.get_init_seq("", 5, 1).tolist() == [[32, 33, 33, 33, 33, 33]]
    assert s.get_init_seq("AA", 5, 1).tolist() == [[32, 5, 5, 33, 33, 33]]
    assert s.get_init_seq("aa", 5, 1).tolist() == [[32, 5, 5, 33, 33, 33]]
    s.draw_seed, s.record = 11, True
    random.seed(2)
    seed = "MRHGDISSSNDTVGVAVVNYKMPRLHTAAEVLDNAR"
    out = s.generate(4, seed, batch_size=4, num_iters=3, num_positions=4, top_k=2, burnin=1, temperature=1.1, show_progress_bar=False)
    assert len(out) == 4 and all(len(x) == len(seed) for x in out)
    run = s.last_run[0]
    for it in range(3):
        rows = run["sampled_logits"][it].reshape(-1, 35)
        toks = odraw.draw_rows(rows, s.valid_aa_idx, 2, it < 1, 1.1, np.repeat(np.arange(4), 4), it, np.tile(np.arange(4), 4), 0, 11)
        assert (toks == run["sampled_tokens"][it].reshape(-1)).all()
    # log-likelihood (one position masked at a time) against the oracle's log-softmax
    ll, per = s.log_likelihood(seed)
    tok = s.get_init_seq(seed, len(seed), 1).numpy()
    want = []
    for i in range(1, len(seed) + 1):
        t = tok.copy()
        t[0, i] = 33
        lg = E.esm1_forward(sd, ocfg, t)[0, i].astype(np.float64)
        want.append(lg[tok[0, i]] - (lg.max() + np.log(np.exp(lg - lg.max()).sum())))
    assert np.abs(np.asarray(per) - np.asarray(want)).max() < 2e-3 and abs(ll - np.mean(want)) < 1e-3


_CKPT = os.path.expanduser("~/.cache/torch/hub/checkpoints/esm1_t6_43M_UR50S.pt")


@pytest.mark.skipif(not os.path.exists(_CKPT), reason="the reference's KATs need the pretrained esm1_t6_43M_UR50S.pt (no network here)")
def test_reference_kats_with_the_pretrained_checkpoint():
    """/root/reference/test/test_esm_sampler.py:269-291: log-likelihoods of three sequences under esm1_t6_43M_UR50S, with and
    without masking -- the reference's only numeric known-answer tests for this path.  Strict mode (1e-3 on logits)."""
    s = esm_sampler.ESM_sampler(models.ESM6(precision="fp32"), device="cuda:0")
    masked = {"MRHGDISSSNDTVGVAVVNYKMPRLHTAAEVLDNAR": -2.843970775604248, "LTWEEQCKTCKGCRYNFQHE": -3.0787816047668457,
              "ACDEFGHIKLMNPQRSTVWY": -3.290297269821167}
    unmasked = {"MRHGDISSSNDTVGVAVVNYKMPRLHTAAEVLDNAR": -2.1893723011016846, "LTWEEQCKTCKGCRYNFQHE": -2.3772685527801514,
                "ACDEFGHIKLMNPQRSTVWY": -2.412991762161255}
    for seq, v in masked.items():
        ll, per = s.log_likelihood(seq)
        assert ll == pytest.approx(v, abs=2e-3) and ll == pytest.approx(float(np.mean(per)), abs=1e-5)
    for seq, v in unmasked.items():
        assert s.log_likelihood(seq, with_masking=False)[0] == pytest.approx(v, abs=2e-3)
