#!/usr/bin/env python3
"""Secondary benchmarks (BASELINE.json configs[3], configs[4]): the ESM-MSA-1b Gibbs paths on one MI355X.

  python bench_msa.py [--config 4|5] [--steps K] [--warmup W]

config 4: ESM_MSA_sampler.generate shape -- 64 MSAs x depth 32 x L=256 (C=257), P=25 positions per row per iteration.
config 5: generate_single shape -- depth 128 x L=512 (C=513), steps=10, passes=3, burn_in=2, k=1; templates run
          one after the other with B=1 exactly as the reference does (esm_msa_sampler.py:125).
The headline metric of the repo is bench.py (config 2); this script only adds measured numbers for the MSA rows of
SURVEY.md 8(d).  One JSON line per config.
"""
import argparse
import ctypes
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from protein_gibbs_sampler_amd import _lib, esm_msa_sampler, models, pyrandom, weights  # noqa: E402


def msa_flops_per_forward(cfg, B, R, C):
    d, f, nl, V = cfg["d_model"], cfg["d_ffn"], cfg["n_layers"], cfg["vocab"]
    n_tok = B * R * C
    return n_tok * (nl * (2.0 * (8 * d * d + 2 * d * f) + 4.0 * C * d + 4.0 * R * d))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=0, help="4, 5 or 0 = both")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--templates", type=int, default=2)
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16",
                    help="fp32 = the strict (split-bf16 x3) mode that meets the 1e-3 logit tolerance")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = dict(weights.MSA1B_CONFIG)
    sd = weights.synthetic_state_dict(cfg, seed=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wrapper = models.ESM_MSA1(state_dict=sd, config=cfg, precision=args.precision)
    lm = wrapper.model.to("cuda:0")
    L_ = _lib.lib()
    valid_idx = sorted(wrapper.alphabet.get_idx(t) for t in "-ACDEFGHIKLMNPQRSTVWY")
    rng = np.random.default_rng(1234)

    def random_msa_tokens(B, R, L):
        aa = np.asarray(valid_idx[:20])[rng.integers(0, 20, (B, R, L))]
        aa[rng.random((B, R, L)) < 0.1] = 30                      # 10 % gaps
        return np.concatenate([np.zeros((B, R, 1), np.int64), aa], axis=2).astype(np.int32)

    if args.config in (0, 4):
        B, R, L, P = 64, 32, 256, 25
        C = L + 1
        tok_dev = torch.from_numpy(random_msa_tokens(B, R, L)).to(dev).contiguous()
        pos_rng = pyrandom.NativePyRandom()
        pos_rng.seed(0)
        params = _lib.make_sample_params(True, cfg["mask_idx"], 0, float("inf"), 1.0, valid_idx, rng_seed=0)

        def run(n, base):
            table = pos_rng.sample_table(list(range(1, L + 1)), P, n * B * R).reshape(n, B, R, P)
            d_idx = torch.from_numpy(table).to(dev)
            params.iter_base = base
            _lib.check(L_.pg_msa_gibbs_run_device(lm.handle, ctypes.c_void_p(tok_dev.data_ptr()), B, R, C,
                                                  ctypes.c_void_p(d_idx.data_ptr()), n, P, ctypes.byref(params), None, None))
            lm.synchronize()
            return d_idx

        if args.warmup:
            run(args.warmup, 0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(args.steps, args.warmup)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        lm.prof_enable(True)
        lm.prof_reset()
        run(1, 100)
        split = {c: lm.prof_get(c)[0] for c in ("gemm", "attention", "layernorm", "embed", "head", "sample")}
        lm.prof_enable(False)
        fl = msa_flops_per_forward(cfg, B, R, C)
        print(json.dumps({"metric": "sampled positions/sec, ESM-MSA-1b generate (config 4)", "value": B * R * P * args.steps / el,
                          "unit": "sampled positions/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * el / args.steps, "dtype": "bf16" if args.precision == "bf16" else "bf16x3 (split fp32)", "data": "synthetic",
                          "config": {"workload": "ESM_MSA_sampler.generate: %d MSAs x depth %d x L=%d, P=%d per row" % (B, R, L, P)},
                          "model_tflops": fl * args.steps / el / 1e12, "time_split_ms_per_iter": split}))

    if args.config in (0, 5):
        R, L, steps, passes, burn_in = 128, 512, 10, 3, 2
        C = L + 1
        s = esm_msa_sampler.ESM_MSA_sampler(wrapper, device="cuda:0")
        s.draw_seed = 0
        msa_tok = random_msa_tokens(1, R, L)[0]
        inv = {wrapper.alphabet.get_idx(t): t for t in "-ACDEFGHIKLMNPQRSTVWY"}
        msa = ["".join(inv[int(t)] for t in row[1:]) for row in msa_tok]
        import random
        random.seed(0)
        s.generate_single(msa, steps=steps, passes=1, burn_in=1, target_index=0, k=1)       # warm-up (buffers, clocks)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.templates):
            s.generate_single(msa, steps=steps, passes=passes, burn_in=burn_in, target_index=0, k=1)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        fl = msa_flops_per_forward(cfg, 1, R, C) * steps * passes * args.templates
        print(json.dumps({"metric": "sampled positions/sec, ESM-MSA-1b generate_single (config 5)",
                          "value": L * passes * args.templates / el, "unit": "sampled positions/s", "n_gpus": 1,
                          "templates": args.templates, "ms_per_forward": 1e3 * el / (steps * passes * args.templates),
                          "dtype": "bf16" if args.precision == "bf16" else "bf16x3 (split fp32)", "data": "synthetic",
                          "config": {"workload": "generate_single: depth %d x L=%d, steps=%d passes=%d burn_in=%d k=1, B=1 per call "
                                                 "(includes tokenisation, host shuffle/partition and PCIe of the host-buffer entry)"
                                                 % (R, L, steps, passes, burn_in)},
                          "model_tflops": fl / el / 1e12}))


if __name__ == "__main__":
    main()
