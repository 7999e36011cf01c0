"""How the torch-CPU baseline leg of bench.py behaves on the GPU box's host (thread count sweep, few layers); prints as it goes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import esm_forward_torch as eft
from oracle.esm_forward import EsmConfig
from protein_gibbs_sampler_amd import weights

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(weights.ESM1B_CONFIG)
cfg["n_layers"] = layers
sd = weights.synthetic_state_dict(cfg, seed=0, std=0.025, embed_std=0.3, ln_jitter=0.1)
ocfg = EsmConfig(n_layers=layers)
wt = eft.torch_state(sd)
rng = np.random.default_rng(0)
tok = np.concatenate([np.zeros((8, 1), np.int64), rng.integers(4, 24, (8, 256)), np.full((8, 1), 2)], axis=1)
print("cpu_count", os.cpu_count(), "torch threads default", torch.get_num_threads(), flush=True)
for nt in (int(a) for a in (sys.argv[2:] or ["16", "32", "64", "128", "256"])):
    torch.set_num_threads(nt)
    eft.esm1b_forward(wt, ocfg, tok[:1, :20])
    t0 = time.perf_counter()
    out = eft.esm1b_forward(wt, ocfg, tok)
    t1 = time.perf_counter()
    vi = torch.arange(4, 24)
    for i in range(200):
        eft.generate_step(out[0], 1 + i, top_k=0, temperature=1.0, sample=True, valid_idx=vi)
    t2 = time.perf_counter()
    small = tok[:1, :27]
    for _ in range(5):
        eft.esm1b_forward(wt, ocfg, small)
    t3 = time.perf_counter()
    fl = layers * (2.0 * (4 * 1280 * 1280 + 2 * 1280 * 5120) + 4.0 * 258 * 1280) * 8 * 258
    print("threads %3d: forward 8x258 %.3f s (%.0f GFLOP/s), 200 draws %.3f s, 5 forwards of 27 tokens %.3f s" % (nt, t1 - t0, fl / (t1 - t0) / 1e9, t2 - t1, t3 - t2), flush=True)
