"""One forward pass at BASELINE configuration 5's shape -- a 128 x 513 alignment, 65 664 tokens, above fair-esm's
`max_tokens_per_msa` = 2^14 (its chunked row / column attention) -- against the fp32 CPU oracle (16.5 TFLOP on the host cores:
about a minute), so that the engine's split-R tied row attention and its > 2^14-token regime are checked against something other
than the engine itself (/root/reference/src/pgen/esm_msa_sampler.py:136 is the call).  Strict mode is held to north_star's 1e-3
on ALL 65 664 x 33 logits; the bf16 mode's error is printed and bounded."""
import time
import warnings

import numpy as np
import pytest

from oracle.msa_forward import MsaConfig, msa_forward
from protein_gibbs_sampler_amd import models, weights

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1800)
def test_config5_shape_forward_against_the_oracle():
    cfg = dict(weights.MSA1B_CONFIG)
    sd = weights.synthetic_state_dict(cfg, seed=12, std=0.025, embed_std=0.3, ln_jitter=0.1)
    rng = np.random.default_rng(55)
    R, C = 128, 513
    tok = rng.integers(4, 24, (1, R, C))
    tok[rng.random((1, R, C)) < 0.1] = 30
    tok[0, R - 1, rng.choice(np.arange(1, C), 52, replace=False)] = 32          # one generate_single step: 52 masks in row -1
    tok[..., 0] = 0
    t0 = time.perf_counter()
    want = msa_forward(sd, MsaConfig(), tok)
    t_cpu = time.perf_counter() - t0
    assert want.shape == (1, R, C, 33) and np.isfinite(want).all()
    res = {}
    for precision in ("fp32", "bf16"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = models.ESM_MSA1(state_dict=sd, config=cfg, precision=precision).model.to("cuda:0")
        got = m.forward_logits(tok)
        err = np.abs(got - want)
        agree = (got.argmax(-1) == want.argmax(-1)).mean()
        res[precision] = (err.max(), err.mean(), agree)
        print("\n[MSA-1b, one 128 x 513 alignment (config 5), %s] max|engine - oracle| = %.3e mean = %.3e argmax agreement %.4f "
              "(logit std %.2f; oracle %.0f s on the host cores)" % (precision, err.max(), err.mean(), agree, want.std(), t_cpu))
        del m
    assert res["fp32"][0] < 1e-3 and res["fp32"][2] > 0.999
    assert res["bf16"][0] < 0.40 and res["bf16"][2] > 0.985
