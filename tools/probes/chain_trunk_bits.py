#!/usr/bin/env python3
"""Debug aid: forward_logits of a 4-layer full-width model on several small shapes, repeated; saves the logits so that runs under
different PGIBBS_CHAIN_TRUNK / PGIBBS_LIB_PATH settings can be compared bit for bit."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_gibbs_sampler_amd import models, weights
cfg = weights.make_config(weights.ESM1B_CONFIG, n_layers=int(os.environ.get("LAYERS", "4")))
sd = weights.synthetic_state_dict(cfg, seed=5, std=0.03, embed_std=0.3, ln_jitter=0.1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    lm = models.ESM1b(state_dict=sd, config=cfg).model.to("cuda:0")
rng = np.random.default_rng(3)
out = {}
for rep in range(int(os.environ.get("REPS", "3"))):
    for (B, T) in [(1, 27), (1, 32), (1, 16), (2, 13), (4, 8)]:
        tok = rng.integers(4, 24, (B, T)); tok[:, 0] = 0
        out["r%d_%dx%d" % (rep, B, T)] = lm.forward_logits(tok)
np.savez(sys.argv[1], **out)
