#!/usr/bin/env python3
"""Per-call wall time of generate() when consecutive calls have different sequence lengths (every call re-captures its graph)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_gibbs_sampler_amd import _cli, esm_sampler, models, weights
cfg = dict(weights.ESM1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    s = esm_sampler.ESM_sampler(models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg), device="gpu")
_cli.seed_everything(11)
seeds = ["MEPAATGQEAEECAHSGRGEAWEEV", "MKTAYIAKQRQISFVKSHFSRQ", "GSHMLEDPVDAFKKLNR", "ACDEFGHIKLMNPQRSTVWYACDEFGHIKL"]
order = [0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 0, 1, 2, 3]
ts = []
for c in order:
    t0 = time.perf_counter()
    s.generate(1, seeds[c], batch_size=1, num_iters=20, burnin=10, mask=True, in_order=False, num_positions_percent=10, top_k=1, show_progress_bar=False)
    ts.append((time.perf_counter() - t0) * 1e3)
lm = s.model.model
print("CT=%s GRAPH=%s per-call ms:" % (os.environ.get("PGIBBS_CHAIN_TRUNK", "1"), os.environ.get("PGIBBS_GRAPH", "1")), " ".join("%d:%.1f" % (o, t) for o, t in zip(order, ts)),
      "captures", lm.get_stat("graph_captures"), "replays", lm.get_stat("graph_replays"))
