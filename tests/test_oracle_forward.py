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


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_torch_baseline_forward_matches_hf_and_the_numpy_oracle(name):
    """oracle/esm_forward_torch.py (bench.py's CPU-baseline leg, torch CPU ops) against the same recorded HuggingFace logits and
    against the numpy oracle, with <pad> and <mask> in the batch."""
    import torch
    from oracle import esm_forward_torch as eft
    z = np.load("%s/esm_hf_%s.npz" % (GOLDEN, name))
    cfg = EsmConfig(**json.loads(str(z["cfg"])))
    w = synthetic_esm_weights(cfg, seed=int(z["seed"]), std=float(z["std"]), embed_std=float(z["embed_std"]),
                              ln_jitter=float(z["ln_jitter"]))
    wt = eft.torch_state(w)
    got = eft.esm1b_forward(wt, cfg, z["tokens"]).numpy()
    assert np.abs(got - z["logits"]).max() < 2e-4
    tok = np.array(z["tokens"]).copy()
    tok[0, -2:] = cfg.pad_idx
    tok[-1, 2] = cfg.mask_idx
    a, b = eft.esm1b_forward(wt, cfg, tok).numpy(), esm1b_forward(w, cfg, tok)
    real = tok != cfg.pad_idx
    assert np.abs(a - b)[real].max() < 2e-4


def test_torch_baseline_gibbs_loop_is_the_reference_loop():
    """mask -> forward -> one generate_step per (chain, position) in the reference's order; with top_k = 1 and sample = False the
    draw is the argmax over the valid residues of the numpy oracle's logits."""
    import torch
    from oracle import esm_forward_torch as eft
    cfg = EsmConfig(d_model=64, n_layers=2, n_heads=4, d_ffn=128, max_pos=40)
    w = synthetic_esm_weights(cfg, seed=3, std=0.1, embed_std=0.5)
    rng = np.random.default_rng(0)
    tok = np.concatenate([np.zeros((3, 1), np.int64), rng.integers(4, 24, (3, 12)), np.full((3, 1), 2)], axis=1)
    targets = [[[1, 5, 9], [2, 3, 4], [12, 6, 1]]]
    valid = list(range(4, 24))
    out, t_fwd, t_loop = eft.gibbs_iterations(eft.torch_state(w), cfg, tok, targets, valid, top_k=1, temperature=None, sample=False)
    masked = tok.copy()
    for b, kks in enumerate(targets[0]):
        masked[b, kks] = cfg.mask_idx
    logits = esm1b_forward(w, cfg, masked)
    want = tok.copy()
    for b, kks in enumerate(targets[0]):
        for kk in kks:
            want[b, kk] = valid[int(np.argmax(logits[b, kk, valid]))]
    assert (out == want).all() and t_fwd > 0 and t_loop > 0
