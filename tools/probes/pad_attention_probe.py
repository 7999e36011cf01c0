import sys, time, warnings
import numpy as np
sys.path.insert(0, ".")
from protein_gibbs_sampler_amd import models, weights
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cfg = dict(weights.ESM1B_CONFIG)
    lm = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg, precision="bf16").model.to("cuda:0")
rng = np.random.default_rng(0)
B, T = 128, 258
tok = np.concatenate([np.zeros((B, 1), np.int64), rng.integers(4, 24, (B, T - 2)), np.full((B, 1), 2)], axis=1)
rag = tok.copy()
for i in range(B):
    n = int(rng.integers(60, T - 1))
    rag[i, n] = 2; rag[i, n + 1:] = 1
for name, t in (("uniform", tok), ("ragged", rag), ("uniform", tok), ("ragged", rag)):
    lm.prof_reset(); lm.prof_enable(True)
    lm.forward_logits(t[:8])
    t0 = time.perf_counter(); lm.forward_logits(t); lm.synchronize(); dt = time.perf_counter() - t0
    print(name, "forward %.1f ms" % (dt * 1e3), {k: lm.prof_get(k) for k in ("gemm", "attention", "layernorm")})
