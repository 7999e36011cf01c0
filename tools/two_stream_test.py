#!/usr/bin/env python3
"""Experiment: the 256 chains of config 2 as S independent sub-batches on S engines / HIP streams of ONE GPU, driven from
S host threads (chains never interact, so this is exact).  Does the hardware overlap one sub-batch's memory-bound phases
(epilogue bursts, LayerNorm, attention loads, ragged last rounds) with the other's MFMA phases?"""
import ctypes, os, sys, threading, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from protein_gibbs_sampler_amd import _lib, models, pyrandom, weights

cfg = dict(weights.ESM1B_CONFIG)
sd = weights.synthetic_state_dict(cfg, seed=0)
dev = torch.device("cuda", 0)
L_ = _lib.lib()
B_total, L, P, K = int(os.environ.get("CHAINS", "256")), 256, 25, int(os.environ.get("STEPS", "6"))
JOB = int(os.environ.get("JOB_ITEMS", "0"))       # > 0: every engine is told it runs a shard of a JOB-chain job (pg_engine_set_job_items)
T = L + 2
valid_idx = list(range(4, 24))
rng = np.random.default_rng(1234)
tok_all = np.concatenate([np.zeros((B_total, 1), np.int64), np.asarray(valid_idx)[rng.integers(0, 20, (B_total, L))], np.full((B_total, 1), 2)], axis=1).astype(np.int32)
for S in tuple(int(x) for x in os.environ.get("SPLITS", "1,2,4").split(",")):
    B = B_total // S
    engines, streams, toks, params, idxs = [], [], [], [], []
    for s_ in range(S):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lm = models.ESM1b(state_dict=sd, config=cfg, precision=os.environ.get("PGIBBS_TOOL_PRECISION", "bf16")).model.to("cuda:0")
        st = torch.cuda.Stream(dev)
        _lib.check(L_.pg_engine_set_stream(lm.handle, ctypes.c_void_p(st.cuda_stream)))
        if JOB:
            lm.set_job_items(JOB)
        engines.append(lm); streams.append(st)
        toks.append(torch.from_numpy(tok_all[s_ * B:(s_ + 1) * B]).to(dev).contiguous())
        params.append(_lib.make_sample_params(True, cfg["mask_idx"], 0, float("inf"), 1.0, valid_idx, rng_seed=0, rng_stream=0, row_id_base=s_ * B))
        pr = pyrandom.NativePyRandom(); pr.seed(s_)
        idxs.append(torch.from_numpy(pr.sample_table(list(range(1, L + 1)), P, (K + 2) * B).reshape(K + 2, B, P)).to(dev))
    torch.cuda.synchronize()

    def work(i, n, base):
        params[i].iter_base = base
        _lib.check(L_.pg_esm_gibbs_run_device(engines[i].handle, ctypes.c_void_p(toks[i].data_ptr()), B, T,
                                              ctypes.c_void_p(idxs[i][base:].data_ptr()), n, P, ctypes.byref(params[i]), None, None))
        engines[i].synchronize()

    def run_all(n, base):
        th = [threading.Thread(target=work, args=(i, n, base)) for i in range(S)]
        [t.start() for t in th]; [t.join() for t in th]

    run_all(2, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_all(K, 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("S=%d sub-batches of %d chains: %.2f ms/iteration, %.0f positions/s" % (S, B, 1e3 * dt / K, B_total * P * K / dt))
    del engines, toks
    torch.cuda.empty_cache()
