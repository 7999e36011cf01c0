#!/usr/bin/env python3
"""Soak test (GPU box): config 2 for many Gibbs iterations, several times -- the final token buffers of every repetition must
be bit-identical (a race in the LDS-DMA ring, the peeled / split-K launches or the graph replay would show up as a diff),
must hold only valid residues, and the chains must have moved."""
import ctypes, hashlib, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from protein_gibbs_sampler_amd import _lib, models, pyrandom, weights

iters, reps = int(os.environ.get("SOAK_ITERS", "60")), int(os.environ.get("SOAK_REPS", "3"))
cfg = dict(weights.ESM1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    lm = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg, precision=os.environ.get("PGIBBS_TOOL_PRECISION", "bf16")).model.to("cuda:0")
L_ = _lib.lib()
dev = torch.device("cuda", 0)
B, L, P = 256, 256, 25
T = L + 2
valid = list(range(4, 24))
rng = np.random.default_rng(1234)
tok0 = np.concatenate([np.zeros((B, 1), np.int64), np.asarray(valid)[rng.integers(0, 20, (B, L))], np.full((B, 1), 2)], axis=1).astype(np.int32)
digests = []
for rep in range(reps):
    tok = torch.from_numpy(tok0.copy()).to(dev)
    pr = pyrandom.NativePyRandom(); pr.seed(0)
    idx = torch.from_numpy(pr.sample_table(list(range(1, L + 1)), P, iters * B).reshape(iters, B, P)).to(dev)
    params = _lib.make_sample_params(True, cfg["mask_idx"], 0, float("inf"), 1.0, valid, rng_seed=11)
    t0 = time.perf_counter()
    _lib.check(L_.pg_esm_gibbs_run_device(lm.handle, ctypes.c_void_p(tok.data_ptr()), B, T, ctypes.c_void_p(idx.data_ptr()), iters, P,
                                          ctypes.byref(params), None, None))
    lm.synchronize()
    dt = time.perf_counter() - t0
    out = tok.cpu().numpy()
    assert np.isin(out[:, 1:-1], valid).all() and (out[:, 0] == 0).all() and (out[:, -1] == 2).all()
    digests.append(hashlib.sha256(out.tobytes()).hexdigest())
    print("rep %d: %d iterations in %.2f s (%.1f ms each), %.1f %% of residues changed, sha256 %s" %
          (rep, iters, dt, 1e3 * dt / iters, 100.0 * (out != tok0).mean(), digests[-1][:16]))
assert len(set(digests)) == 1, "runs differ: " + str(digests)
print("soak OK")

# ---- MSA paths: generate (fused row attention) and generate_single (split-R row attention), twice each
import random
from protein_gibbs_sampler_amd import esm_msa_sampler
mcfg = dict(weights.MSA1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mm = models.ESM_MSA1(state_dict=weights.synthetic_state_dict(mcfg, seed=0), config=mcfg, precision=os.environ.get("PGIBBS_TOOL_PRECISION", "bf16"))
ms = esm_msa_sampler.ESM_MSA_sampler(mm, device="cuda:0")
sym = np.asarray(list("ACDEFGHIKLMNPQRSTVWY"))
r2 = np.random.default_rng(7)
def rand_msa(R, Lm):
    rows = sym[r2.integers(0, 20, (R, Lm))]
    rows[r2.random((R, Lm)) < 0.1] = "-"
    return ["".join(r) for r in rows]
msa_a, msa_b = rand_msa(32, 256), rand_msa(128, 512)
res = []
for rep in range(2):
    ms.draw_seed = 5
    random.seed(3)
    g = ms.generate(16 * 32, msa_a, batch_size=16, num_iters=6, num_positions=25, show_progress_bar=False)
    random.seed(4)
    s1 = ms.generate_single(msa_b, steps=10, passes=2, burn_in=1, k=1)
    res.append((hashlib.sha256("".join(g).encode()).hexdigest(), s1))
    print("msa rep %d: generate sha256 %s, generate_single %s..." % (rep, res[-1][0][:16], s1[:24]))
assert res[0] == res[1], "MSA runs differ"
print("msa soak OK")
