#!/usr/bin/env python3
"""cProfile of the Python side of the three sampler entry points at BASELINE-like sizes (what the host does around the native calls)."""
import cProfile, os, pstats, random, sys, warnings, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_gibbs_sampler_amd import _cli, esm_sampler, esm_msa_sampler, models, weights
which = sys.argv[1]
def top(pr, n=14):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(n); print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:n + 12]))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    if which == "esm":
        cfg = weights.make_config(weights.ESM1B_CONFIG, n_layers=2)
        s = esm_sampler.ESM_sampler(models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg), device="gpu")
    else:
        cfg = weights.make_config(weights.MSA1B_CONFIG, n_layers=1)
        s = esm_msa_sampler.ESM_MSA_sampler(models.ESM_MSA1(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg), device="gpu")
_cli.seed_everything(3)
rng = np.random.default_rng(0)
AA = "ACDEFGHIKLMNPQRSTVWY"
if which == "esm":
    seed = "".join(AA[i] for i in rng.integers(0, 20, 256))
    run = lambda: s.generate(256, seed, batch_size=256, num_iters=20, burnin=10, mask=True, in_order=False, num_positions_percent=10, top_k=1, show_progress_bar=False)
elif which == "msa":
    msa = ["".join(AA[i] for i in rng.integers(0, 20, 256)) for _ in range(32)]
    run = lambda: s.generate(64 * 32, msa, batch_size=64, num_iters=10, burnin=5, mask=True, in_order=False, num_positions_percent=10, top_k=1, show_progress_bar=False)
else:
    msas = [["".join(AA[i] for i in rng.integers(0, 20, 512)) for _ in range(128)] for _ in range(4)]
    run = lambda: s.generate_single_batch(msas, steps=10, passes=1, burn_in=1, target_index=0, k=1, max_batch=4)
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
top(pr)
