"""Attention time of an ESM-1b forward at chain lengths between the rungs of the key-block ladder (run with PGIBBS_ATTN_LADDER=0 / 1)."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, ".")
from protein_gibbs_sampler_amd import models, weights
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cfg = dict(weights.ESM1B_CONFIG)
    lm = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg, precision="bf16").model.to("cuda:0")
rng = np.random.default_rng(0)
outs = []
for L in (80, 150, 200, 256, 300, 400, 500):
    T = L + 2
    B = max(8, 32768 // T)
    tok = np.concatenate([np.zeros((B, 1), np.int64), rng.integers(4, 24, (B, L)), np.full((B, 1), 2)], axis=1)
    lm.forward_logits(tok[:4])
    lm.prof_reset(); lm.prof_enable(True)
    out = lm.forward_logits(tok); lm.synchronize()
    a, g = lm.prof_get("attention")[0], lm.prof_get("gemm")[0]
    outs.append(out[:2].copy())
    print("ladder=%s L=%3d (%2d key blocks) B=%3d: attention %6.2f ms, gemm %6.2f ms" % (os.environ.get("PGIBBS_ATTN_LADDER", "1"), L, (T + 15) // 16, B, a, g))
np.savez(sys.argv[1], *outs)
