#!/usr/bin/env python3
"""Determinism / shard-invariance check of the config-2 Gibbs job (bf16 mode): whole twice, then 8 and 2 shards."""
import ctypes
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib, models, pyrandom, sharding, weights  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B_ARG = int(sys.argv[2]) if len(sys.argv) > 2 else 256
P_ARG = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cfg = dict(weights.ESM1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    wrapper = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg, precision=os.environ.get("PGIBBS_TOOL_PRECISION", "bf16"))
lm = wrapper.model.to("cuda:0")
valid = sorted(wrapper.alphabet.get_idx(t) for t in "ACDEFGHIKLMNPQRSTVWY")
B, L, P = B_ARG, 256, P_ARG
T = L + 2
rng = np.random.default_rng(1234)
tok_all = np.concatenate([np.zeros((B, 1), np.int64), rng.integers(4, 24, (B, L)), np.full((B, 1), 2)], axis=1).astype(np.int32)
L_ = _lib.lib()


def run(lo, hi, n=iters, want_logits=False):
    r = pyrandom.NativePyRandom()
    r.seed(0)
    lm.set_job_items(B)                      # this call is a shard of the B-item job (pgibbs.h pg_engine_set_job_items)
    table = sharding.local_slice(sharding.global_position_table(r, list(range(1, L + 1)), P, n, B), lo, hi)
    params = _lib.make_sample_params(True, 32, 0, float("inf"), 1.0, valid, rng_seed=0, row_id_base=lo)
    d_tok = torch.from_numpy(tok_all[lo:hi].copy()).cuda()
    d_idx = torch.from_numpy(table).cuda()
    lg = torch.empty((n, hi - lo, P, 33), dtype=torch.float32, device="cuda") if want_logits else None
    _lib.check(L_.pg_esm_gibbs_run_device(lm.handle, ctypes.c_void_p(d_tok.data_ptr()), hi - lo, T, ctypes.c_void_p(d_idx.data_ptr()), n, P,
                                          ctypes.byref(params), ctypes.c_void_p(lg.data_ptr()) if want_logits else None, None))
    lm.synchronize()
    lm.set_job_items(0)
    return d_tok.cpu().numpy(), (lg.cpu().numpy() if want_logits else None)


a, la = run(0, B, 1, True)
b, lb = run(0, B, 1, True)
print("whole vs whole, 1 iteration: tokens differ at", int((a != b).sum()), "logits differ at", int((la != lb).sum()))
parts = [run(*sharding.shard_range(B, 8, g), 1, True) for g in range(8)]
pt = np.concatenate([p[0] for p in parts]); pl = np.concatenate([p[1] for p in parts], axis=1)
d = np.abs(pl - la)
print("8 shards vs whole, 1 iteration: tokens differ at", int((pt != a).sum()), "logits differ at", int((pl != la).sum()), "max |diff| %.3e" % d.max(),
      "chains affected", sorted(set(np.nonzero((pl != la).any(axis=(0, 2, 3)))[0].tolist()))[:20])
w, _ = run(0, B)
for world in (8, 2):
    pt = np.concatenate([run(*sharding.shard_range(B, world, g))[0] for g in range(world)])
    print("world %d, %d iterations: tokens differ at %d" % (world, iters, int((pt != w).sum())))
