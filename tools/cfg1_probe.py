#!/usr/bin/env python3
"""Config 1 latency breakdown: generate() wall time per call and per iteration with the engine on (a) its own non-blocking
stream, (b) torch's current (null) stream, (c) a torch side stream; PGIBBS_GRAPH=0 in the environment gives the eager loop."""
import ctypes, os, sys, time, random, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protein_gibbs_sampler_amd import _lib, esm_sampler, models, weights
cfg = dict(weights.ESM1B_CONFIG)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    s = esm_sampler.ESM_sampler(models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=0), config=cfg, precision=os.environ.get("PGIBBS_TOOL_PRECISION", "bf16")), device="gpu")
lm = s.model.model
L = _lib.lib()
seed = "MEPAATGQEAEECAHSGRGEAWEEV"
kw = dict(batch_size=1, num_iters=20, burnin=10, mask=True, in_order=False, num_positions_percent=10, top_k=1, show_progress_bar=False)
side = torch.cuda.Stream()
for name, ptr in (("own", None), ("null", ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), ("side", ctypes.c_void_p(side.cuda_stream)), ("own", None)):
    _lib.check(L.pg_engine_set_stream(lm.handle, ptr))
    random.seed(0)
    for _ in range(3):
        s.generate(1, seed, **kw)
    for iters in (20, 100):
        kw["num_iters"] = iters
        s.generate(1, seed, **kw)
        t0 = time.perf_counter()
        for _ in range(10):
            s.generate(1, seed, **kw)
        dt = (time.perf_counter() - t0) / 10
        print("stream=%s graph=%s iters=%d: %.2f ms per generate() = %.3f ms/iteration; captures %d replays %d" % (
            name, os.environ.get("PGIBBS_GRAPH", "1"), iters, dt * 1e3, dt * 1e3 / iters, lm.get_stat("graph_captures"), lm.get_stat("graph_replays")))
    kw["num_iters"] = 20
