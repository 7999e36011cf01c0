#!/usr/bin/env python3
"""Debug aid: the generate() calls of tests/test_gpu_chain_trunk.py, optionally preceded by forward_logits calls."""
import os, random, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_gibbs_sampler_amd import esm_sampler, models, weights
cfg = weights.make_config(weights.ESM1B_CONFIG, n_layers=4)
sd = weights.synthetic_state_dict(cfg, seed=5, std=0.03, embed_std=0.3, ln_jitter=0.1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    s = esm_sampler.ESM_sampler(models.ESM1b(state_dict=sd, config=cfg), device="gpu")
lm = s.model.model
if os.environ.get("PRE", "0") == "1":
    rng = np.random.default_rng(3)
    for (B, T) in [(1, 27), (1, 32), (1, 16), (1, 9), (1, 1), (2, 13), (4, 8), (3, 5), (1, 33)]:
        tok = rng.integers(4, 24, (B, T)); tok[:, 0] = 0
        lm.forward_logits(tok)
random.seed(7)
for it in range(3):
    print(s.generate(1, "MEPAATGQEAEECAHSGRGEAWEEV", batch_size=1, num_iters=12, burnin=6, mask=True, in_order=False,
                     num_positions_percent=10, top_k=1, show_progress_bar=False, rollover_from_start=False))
print(s.generate(2, "MKTAYIAKQR", batch_size=2, num_iters=8, burnin=4, mask=True, in_order=True, num_positions=2, top_k=0, show_progress_bar=False))
