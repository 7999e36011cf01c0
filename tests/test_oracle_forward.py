"""Pins oracle/esm_forward.py against HuggingFace EsmForMaskedLM logits recorded in
tests/golden/esm_hf_*.npz (independent implementation of the ESM-1b architecture; the
reference's own forward KATs need pretrained checkpoints that are not available offline)."""
import json

import numpy as np
import pytest

from oracle.esm_forward import EsmConfig, synthetic_esm_weights, esm1b_forward
from _standin import GOLDEN


# "full" = the REAL ESM-1b sizes (33 layers, d = 1280, 650 M parameters; ~1 minute of numpy on the CPU container): the fixture
# pins the FULL-SIZE oracle to the independent implementation at a realistic logit scale (std 10), not only the small models
@pytest.mark.parametrize("name", ["tiny", "small", "full"])
def test_esm1b_forward_matches_hf(name):
    z = np.load("%s/esm_hf_%s.npz" % (GOLDEN, name))
    cfg = EsmConfig(**json.loads(str(z["cfg"])))
    w = synthetic_esm_weights(cfg, seed=int(z["seed"]), std=float(z["std"]), embed_std=float(z["embed_std"]),
                              ln_jitter=float(z["ln_jitter"]))
    logits = esm1b_forward(w, cfg, z["tokens"])
    assert np.abs(logits - z["logits"]).max() < 2e-4      # fp32 vs fp32, different summation order
    assert z["logits"].std() > 1.0                         # non-degenerate logits


def test_token_dropout_rescale_matters():
    """every embedding depends on the per-chain mask count."""
    cfg = EsmConfig(d_model=64, n_layers=1, n_heads=1, d_ffn=128, max_pos=40)
    w = synthetic_esm_weights(cfg, seed=1, std=0.1, embed_std=0.5)
    tok = np.array([[0, 5, 6, 7, 8, 9, 10, 2]])
    tok2 = tok.copy()
    tok2[0, 3] = 32
    a, b = esm1b_forward(w, cfg, tok), esm1b_forward(w, cfg, tok2)
    assert np.abs(a[0, 6] - b[0, 6]).max() > 1e-4
