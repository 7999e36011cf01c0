#!/usr/bin/env python3
"""Shard invariance of the ESM-MSA-1b Gibbs job: B MSAs (depth R, L columns) whole vs split into `world` contiguous shards."""
import ctypes
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib, models, pyrandom, sharding, weights  # noqa: E402

B, R, L, P, iters = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 32, 256, 25, 2)))
worlds = [int(a) for a in sys.argv[6:]] or [8, 2]
C = L + 1
cfg = dict(weights.MSA1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    wrapper = models.ESM_MSA1(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg, precision=os.environ.get("PGIBBS_TOOL_PRECISION", "bf16"))
lm = wrapper.model.to("cuda:0")
valid = sorted(wrapper.alphabet.get_idx(t) for t in "-ACDEFGHIKLMNPQRSTVWY")
rng = np.random.default_rng(1234)
aa = np.asarray(valid[:20])[rng.integers(0, 20, (B, R, L))]
tok_all = np.concatenate([np.zeros((B, R, 1), np.int64), aa], axis=2).astype(np.int32)
L_ = _lib.lib()


def run(lo, hi):
    r = pyrandom.NativePyRandom()
    r.seed(0)
    lm.set_job_items(B)                      # this call is a shard of the B-item job (pgibbs.h pg_engine_set_job_items)
    table = r.sample_table(list(range(1, L + 1)), P, iters * B * R).reshape(iters, B, R, P)[:, lo:hi].copy()
    params = _lib.make_sample_params(True, cfg["mask_idx"], 0, float("inf"), 1.0, valid, rng_seed=0, row_id_base=lo * R)
    d_tok = torch.from_numpy(tok_all[lo:hi].copy()).cuda()
    d_idx = torch.from_numpy(table).cuda()
    lg = torch.empty((iters, hi - lo, R, P, 33), dtype=torch.float32, device="cuda")
    _lib.check(L_.pg_msa_gibbs_run_device(lm.handle, ctypes.c_void_p(d_tok.data_ptr()), hi - lo, R, C, ctypes.c_void_p(d_idx.data_ptr()), iters, P,
                                          ctypes.byref(params), ctypes.c_void_p(lg.data_ptr()), None))
    lm.synchronize()
    lm.set_job_items(0)
    return d_tok.cpu().numpy(), lg.cpu().numpy()


w, wl = run(0, B)
for world in worlds:
    parts = [run(*sharding.shard_range(B, world, g)) for g in range(world)]
    pt = np.concatenate([p[0] for p in parts]); pl = np.concatenate([p[1] for p in parts], axis=1)
    print("B=%d R=%d L=%d P=%d world %d: tokens differ at %d, logits differ at %d (max |diff| %.3e)"
          % (B, R, L, P, world, int((pt != w).sum()), int((pl != wl).sum()), float(np.abs(pl - wl).max())))
