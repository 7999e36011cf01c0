#!/usr/bin/env python3
"""Debug aid: pruned Gibbs iterations (pg_esm_gibbs_run with per-iteration logits) under the current PGIBBS_CHAIN_TRUNK /
PGIBBS_GRAPH settings; saves logits + tokens so two runs can be diffed."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_gibbs_sampler_amd import _lib, models, weights
cfg = weights.make_config(weights.ESM1B_CONFIG, n_layers=int(os.environ.get("LAYERS", "4")))
sd = weights.synthetic_state_dict(cfg, seed=5, std=0.03, embed_std=0.3, ln_jitter=0.1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    lm = models.ESM1b(state_dict=sd, config=cfg).model.to("cuda:0")
rng = np.random.default_rng(3)
out = {}
for (B, T, P, iters) in [(1, 27, 2, 6), (2, 13, 1, 5), (1, 27, 2, 6)]:
    tok = rng.integers(4, 24, (B, T)).astype(np.int32); tok[:, 0] = 0
    idx = rng.integers(1, T, (iters, B, P)).astype(np.int32)
    for want in (True, False):          # with per-iteration logits (eager loop) and without (graph replay)
        t = tok.copy()
        p = _lib.make_sample_params(True, 32, 1, 3, 1.0, list(range(4, 24)), 11)
        lg, st = lm.gibbs_run(t, idx, p, want_logits=want, want_tokens=True)
        key = "%dx%d_%s" % (B, T, "logits" if want else "graph")
        out[key + "_tok"] = t; out[key + "_st"] = st
        if want: out[key + "_lg"] = lg
np.savez(sys.argv[1], **out)
print("saved", sorted(out))
