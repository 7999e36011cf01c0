#!/usr/bin/env python3
"""Few-chain regime sweep (the batch sizes /root/reference/README.md:74-75 documents): 1 ... 64 chains of
L = 256 and L = 100 through the whole engine -- ms per Gibbs iteration, executed TFLOP/s, fraction of the bf16 MFMA peak, the time
per kernel class and the kernel the GEMM dispatch picked for each of the four per-layer projections (pg_prof_get_kernels).
Every row is a job of its own (set_job_items(0): kernels chosen by the local shape).

  python tools/batch_sweep.py [--chains 1,2,4,...] [--lengths 256,100] [--iters 10] > profiles/r06_batch_sweep.txt
"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from protein_gibbs_sampler_amd import _lib, models, pyrandom, weights

ap = argparse.ArgumentParser()
ap.add_argument("--chains", default="1,2,4,8,12,16,24,32,48,64")
ap.add_argument("--lengths", default="256,100")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--precision", default="bf16")
args = ap.parse_args()

cfg = dict(weights.ESM1B_CONFIG)
sd = weights.synthetic_state_dict(cfg, seed=0, std=0.025, embed_std=0.3, ln_jitter=0.1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
lm = models.ESM1b(state_dict=sd, config=cfg, precision=args.precision).model.to(str(dev))
L_ = _lib.lib()
stream = torch.cuda.current_stream(dev)
_lib.check(L_.pg_engine_set_stream(lm.handle, ctypes.c_void_p(stream.cuda_stream)))
lm.set_job_items(0)
valid = list(range(4, 24))
d, f, nl, V = cfg["d_model"], cfg["d_ffn"], cfg["n_layers"], cfg["vocab"]
print("# few-chain sweep, ESM-1b 33 x 1280, %s operands; peak = 2500 TFLOP/s dense bf16; executed FLOPs (last layer pruned)" % args.precision)
print("# %-6s %-4s %-6s %9s %8s %6s | %6s %6s %6s %6s %6s %6s | %6s %6s %6s %6s | kernels: qkv ; out ; fc1 ; fc2" %
      ("chains", "L", "rows", "ms/iter", "TFLOP/s", "frac", "gemm", "attn", "ln", "embed", "head", "sample", "qkv", "out", "fc1", "fc2"))
for L in [int(v) for v in args.lengths.split(",")]:
    T, P = L + 2, max(1, int(L * 10 / 100))
    for B in [int(v) for v in args.chains.split(",")]:
        rng = np.random.default_rng(1234)
        tok = np.concatenate([np.zeros((B, 1), np.int64), np.asarray(valid)[rng.integers(0, 20, (B, L))], np.full((B, 1), 2)], axis=1).astype(np.int32)
        d_tok = torch.from_numpy(tok).to(dev).contiguous()
        pr = pyrandom.NativePyRandom(); pr.seed(0)
        params = _lib.make_sample_params(True, cfg["mask_idx"], 0, float("inf"), 1.0, valid, rng_seed=0)
        done = [0]

        def run(n):
            table = pr.sample_table(list(range(1, L + 1)), P, n * B).reshape(n, B, P)
            d_idx = torch.from_numpy(table).to(dev)
            params.iter_base = done[0]; done[0] += n
            _lib.check(L_.pg_esm_gibbs_run_device(lm.handle, ctypes.c_void_p(d_tok.data_ptr()), B, T, ctypes.c_void_p(d_idx.data_ptr()), n, P,
                                                  ctypes.byref(params), None, None))
            torch.cuda.synchronize(dev)
            return d_idx
        run(3)
        t0 = time.perf_counter(); run(args.iters); ms = 1e3 * (time.perf_counter() - t0) / args.iters
        lm.prof_enable(True); lm.prof_reset(); run(2)
        cls = {c: lm.prof_get(c)[0] / 2 for c in ("gemm", "attention", "layernorm", "embed", "head", "sample")}
        kern = [lm.prof_get_kernels(c) or "-" for c in ("gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2")]
        proj = [lm.prof_get(c)[0] / 2 for c in ("gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2")]
        other = lm.prof_get_kernels("gemm_other")
        lm.prof_enable(False)
        M, S = B * T, B * P
        full = 2.0 * (4 * d * d + 2 * d * f) * M
        last = 2.0 * 3 * d * d * M + 2.0 * (d * d + 2 * d * f) * S
        fl = (nl - 1) * full + last + nl * 4.0 * T * d * M + (2.0 * d * d + 2.0 * V * d) * S
        tf = fl / (ms * 1e-3) / 1e12
        print("  %-6d %-4d %-6d %9.3f %8.1f %6.3f | %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f | %6.3f %6.3f %6.3f %6.3f | %s%s" %
              (B, L, M, ms, tf, tf / 2500.0, cls["gemm"], cls["attention"], cls["layernorm"], cls["embed"], cls["head"], cls["sample"],
               proj[0], proj[1], proj[2], proj[3], " ; ".join(kern), ("   [other: %s]" % other) if other else ""), flush=True)
