#!/usr/bin/env python3
"""Soak of the persistent single-chain trunk at full depth (33 layers): N generate() calls of BASELINE config 1's shape; prints a
digest of all generated sequences (compare a PGIBBS_CHAIN_TRUNK=1 run with a =0 run) and the time per iteration.  Two of these at
once on one GPU (`... & ... & wait`) show what happens when two persistent grids share the device."""
import hashlib, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from protein_gibbs_sampler_amd import _cli, esm_sampler, models, weights
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = dict(weights.ESM1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    s = esm_sampler.ESM_sampler(models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg), device="gpu")
_cli.seed_everything(11)
h = hashlib.sha256()
seeds = ["MEPAATGQEAEECAHSGRGEAWEEV", "MKTAYIAKQRQISFVKSHFSRQ", "GSHMLEDPVDAFKKLNR", "ACDEFGHIKLMNPQRSTVWYACDEFGHIKL"]
t0 = time.perf_counter(); iters = 0; errors = 0
for c in range(n_calls):
    try:
        out = s.generate(1, seeds[c % 4], batch_size=1, num_iters=20, burnin=10, mask=True, in_order=False, num_positions_percent=10,
                         top_k=1, show_progress_bar=False)
        h.update(out[0].encode()); iters += 20
    except Exception as e:          # a barrier timeout surfaces as a RuntimeError from the C ABI
        errors += 1
        if errors <= 3: print("ERROR:", str(e)[:200])
dt = time.perf_counter() - t0
print("chain_trunk=%s calls=%d errors=%d digest=%s %.3f ms/iteration" % (os.environ.get("PGIBBS_CHAIN_TRUNK", "1"), n_calls, errors,
      h.hexdigest()[:16], dt * 1e3 / max(iters, 1)))
