import ctypes, os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from protein_gibbs_sampler_amd import _lib, models, pyrandom, sharding, weights
cfg = dict(weights.ESM1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    lm = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg, precision="fp32").model.to("cuda:0")
B, L, P, iters = 64, 256, 25, 2
T = L + 2
rng = np.random.default_rng(1)
tok_all = np.concatenate([np.zeros((B, 1), np.int64), rng.integers(4, 24, (B, L)), np.full((B, 1), 2)], axis=1).astype(np.int32)
valid = list(range(4, 24))
L_ = _lib.lib()
def run(lo, hi, job):
    r = pyrandom.NativePyRandom(); r.seed(0)
    table = sharding.local_slice(sharding.global_position_table(r, list(range(1, L + 1)), P, iters, B), lo, hi)
    params = _lib.make_sample_params(True, 32, 0, float("inf"), 1.0, valid, rng_seed=0, row_id_base=lo)
    d_tok = torch.from_numpy(tok_all[lo:hi].copy()).cuda(); d_idx = torch.from_numpy(table).cuda()
    lg = torch.empty((iters, hi - lo, P, 33), dtype=torch.float32, device="cuda")
    lm.set_job_items(job)
    _lib.check(L_.pg_esm_gibbs_run_device(lm.handle, ctypes.c_void_p(d_tok.data_ptr()), hi - lo, T, ctypes.c_void_p(d_idx.data_ptr()), iters, P, ctypes.byref(params), ctypes.c_void_p(lg.data_ptr()), None))
    lm.synchronize(); lm.set_job_items(0)
    return d_tok.cpu().numpy(), lg.cpu().numpy()
a, la = run(0, B, 0); b, lb = run(0, B, 0)
print("strict whole vs whole: logits differ at", int((la != lb).sum()))
for world in (8, 2):
    parts = [run(*sharding.shard_range(B, world, g), B) for g in range(world)]
    pl = np.concatenate([p[1] for p in parts], axis=1)
    print("strict world", world, ": logits differ at", int((pl != la).sum()), "max", float(np.abs(pl - la).max()))
