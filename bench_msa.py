#!/usr/bin/env python3
"""Secondary benchmarks (BASELINE.json configs[3], configs[4]): the ESM-MSA-1b Gibbs paths on one MI355X.

  python bench_msa.py [--config 4|5] [--steps K] [--warmup W] [--templates N]

config 4: ESM_MSA_sampler.generate shape -- 64 MSAs x depth 32 x L=256 (C=257), P=25 positions per row per iteration.
config 5: generate_single shape -- depth 128 x L=512 (C=513), steps=10, passes=3, burn_in=2, k=1.  BASELINE config 5 is
          "batch=32 templates sharded over 8 x MI355X" = 4 templates per GPU: the templates of one GPU go through
          ESM_MSA_sampler.generate_single_batch as ONE native call (pg_msa_gibbs_single_batch_run, 4 MSAs per forward);
          `--templates 1` is the reference's own shape, one template per call (esm_msa_sampler.py:125).
The headline metric of the repo is bench.py (config 2), which embeds both results under "msa" at N = 1 (run_config4 /
run_config5 below are what it calls); stand-alone this script prints one JSON line per config.
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

MFMA_BF16_PEAK_TFLOPS = 2500.0


def msa_flops_per_forward(cfg, B, R, C):
    """GEMMs (4 projections x 2 attention blocks + FFN) + tied-row and column attention contractions, 2 FLOP/MAC."""
    d, f, nl = cfg["d_model"], cfg["d_ffn"], cfg["n_layers"]
    n_tok = B * R * C
    return n_tok * (nl * (2.0 * (8 * d * d + 2 * d * f) + 4.0 * C * d + 4.0 * R * d))


_SD_CACHE = {}


def state_dict(realistic=True):
    """The seeded synthetic ESM-MSA-1b weights every leg of this file runs with (one copy per process)."""
    if realistic not in _SD_CACHE:
        kw = dict(std=0.025, embed_std=0.3, ln_jitter=0.1) if realistic else {}
        _SD_CACHE[realistic] = weights.synthetic_state_dict(dict(weights.MSA1B_CONFIG), seed=0, **kw)
    return _SD_CACHE[realistic]


def build(precision="bf16", device="cuda:0", realistic=True):
    """(wrapper, engine, cfg): ESM-MSA-1b with seeded synthetic weights (realistic = logit std ~ 10, as the parity tests use)."""
    cfg = dict(weights.MSA1B_CONFIG)
    sd = state_dict(realistic)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wrapper = models.ESM_MSA1(state_dict=sd, config=cfg, precision=precision)
    return wrapper, wrapper.model.to(device), cfg


def _valid_idx(wrapper):
    return sorted(wrapper.alphabet.get_idx(t) for t in "-ACDEFGHIKLMNPQRSTVWY")


def random_msa_tokens(rng, valid_idx, B, R, L):
    aa = np.asarray(valid_idx[:20])[rng.integers(0, 20, (B, R, L))]
    aa[rng.random((B, R, L)) < 0.1] = 30                      # 10 % gaps
    return np.concatenate([np.zeros((B, R, 1), np.int64), aa], axis=2).astype(np.int32)


def _split(lm):
    return {c: lm.prof_get(c)[0] for c in ("gemm", "attention", "layernorm", "embed", "head", "sample")}


def run_config4(wrapper, lm, cfg, steps=3, warmup=1, precision="bf16", dev=None):
    dev = dev if dev is not None else torch.device("cuda", 0)
    L_ = _lib.lib()
    valid_idx = _valid_idx(wrapper)
    rng = np.random.default_rng(1234)
    B, R, L, P = 64, 32, 256, 25
    C = L + 1
    tok_dev = torch.from_numpy(random_msa_tokens(rng, valid_idx, B, R, L)).to(dev).contiguous()
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

    if warmup:
        run(warmup, 0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(steps, warmup)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    lm.prof_enable(True)
    lm.prof_reset()
    run(1, 100)
    split = _split(lm)
    lm.prof_enable(False)
    final = tok_dev.cpu().numpy()
    assert np.isin(final[:, :, 1:], valid_idx).all(), "MSA rows left the 21-symbol alphabet"
    fl = msa_flops_per_forward(cfg, B, R, C)
    tf = fl * steps / el / 1e12 * (3 if precision == "fp32" else 1)
    return {"metric": "sampled positions/sec, ESM-MSA-1b generate (config 4)", "value": B * R * P * steps / el,
            "unit": "sampled positions/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * el / steps, "dtype": "bf16" if precision == "bf16" else "bf16x3 (split fp32)", "data": "synthetic",
            "config": {"workload": "ESM_MSA_sampler.generate: %d MSAs x depth %d x L=%d, P=%d per row" % (B, R, L, P)},
            "model_tflops": fl * steps / el / 1e12, "frac_of_bf16_mfma_peak": tf / MFMA_BF16_PEAK_TFLOPS,
            "time_split_ms_per_iter": split}


def run_config5(wrapper, lm, cfg, templates=4, precision="bf16", dev=None, max_batch=4):
    """`templates` distinct template MSAs (depth 128 x L=512) resampled by generate_single_batch: max_batch of them per native
    call (4 = one GPU's share of BASELINE config 5's 32 templates over 8 GPUs; 1 = the reference's one call per template)."""
    import random
    dev = dev if dev is not None else torch.device("cuda", 0)
    valid_idx = _valid_idx(wrapper)
    rng = np.random.default_rng(4321)
    R, L, steps, passes, burn_in = 128, 512, 10, 3, 2
    C = L + 1
    s = esm_msa_sampler.ESM_MSA_sampler(wrapper, device=str(dev))
    s.draw_seed = 0
    inv = {wrapper.alphabet.get_idx(t): t for t in "-ACDEFGHIKLMNPQRSTVWY"}
    toks = random_msa_tokens(rng, valid_idx, templates, R, L)
    msas = [["".join(inv[int(t)] for t in row[1:]) for row in toks[b]] for b in range(templates)]
    random.seed(0)
    s.generate_single_batch(msas[:max_batch], steps=steps, passes=1, burn_in=1, target_index=0, k=1, max_batch=max_batch)   # warm-up
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = s.generate_single_batch(msas, steps=steps, passes=passes, burn_in=burn_in, target_index=0, k=1, max_batch=max_batch)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    assert len(out) == templates and all(len(o) == L for o in out)
    # time split of one batched forward pass sequence (one pass over the templates of one call)
    lm.prof_enable(True)
    lm.prof_reset()
    s.generate_single_batch(msas[:max_batch], steps=steps, passes=1, burn_in=1, target_index=0, k=1, max_batch=max_batch)
    split = {c: v / steps for c, v in _split(lm).items()}
    lm.prof_enable(False)
    n_calls = (templates + max_batch - 1) // max_batch
    fl = msa_flops_per_forward(cfg, 1, R, C) * steps * passes * templates
    tf = fl / el / 1e12 * (3 if precision == "fp32" else 1)
    return {"metric": "sampled positions/sec, ESM-MSA-1b generate_single (config 5)",
            "value": L * passes * templates / el, "unit": "sampled positions/s", "n_gpus": 1,
            "templates": templates, "templates_per_call": min(max_batch, templates),
            "ms_per_forward": 1e3 * el / (steps * passes * n_calls), "ms_per_template_forward": 1e3 * el / (steps * passes * templates),
            "dtype": "bf16" if precision == "bf16" else "bf16x3 (split fp32)", "data": "synthetic",
            "config": {"workload": "generate_single_batch: %d templates of depth %d x L=%d, steps=%d passes=%d burn_in=%d k=1, %d per "
                                   "native call (includes tokenisation, host shuffle/partition and PCIe of the host-buffer entry)"
                                   % (templates, R, L, steps, passes, burn_in, min(max_batch, templates))},
            "model_tflops": fl / el / 1e12, "frac_of_bf16_mfma_peak": tf / MFMA_BF16_PEAK_TFLOPS,
            "time_split_ms_per_forward": split}


def logit_check(engines, R=32, L=256, P=25, realistic=True):
    """max |logit error| of each engine in `engines` ({mode name: NativeMaskedLM}) against the numpy fp32 oracle
    (oracle/msa_forward.py -- the checker, never the product) on ONE config-4 alignment (depth 32 x 257 columns, P = 25 <mask> per
    row) of THIS file's weights, at the masked rows: the figure north_star's 1e-3 tolerance is held against.  ~10-20 s of host time."""
    from oracle.msa_forward import MsaConfig, msa_forward
    sd = state_dict(realistic)
    rng = np.random.default_rng(77)
    tok = random_msa_tokens(rng, list(range(4, 24)) + [30], 1, R, L).astype(np.int64)
    for r in range(R):
        tok[0, r, rng.choice(np.arange(1, L + 1), P, replace=False)] = 32
    t0 = time.perf_counter()
    want = msa_forward(sd, MsaConfig(), tok)
    out = {"alignment": "1 x %d x %d (config 4's MSA shape), %d <mask> per row" % (R, L + 1, P), "rows_checked": int((tok == 32).sum()),
           "logit_std": float(want.std()), "checker": "oracle/msa_forward.py (numpy fp32)", "oracle_seconds": time.perf_counter() - t0}
    m = tok == 32
    for name, lm in engines.items():
        got = lm.forward_logits(tok.astype(np.int32))
        err = np.abs(got[m] - want[m])
        out[name + "_max_abs_logit_err"] = float(err.max())
        out[name + "_mean_abs_logit_err"] = float(err.mean())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=0, help="4, 5 or 0 = both")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--templates", type=int, default=4)
    ap.add_argument("--max-batch", type=int, default=4, help="templates per native call in config 5")
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16",
                    help="fp32 = the strict (split-bf16 x3) mode that meets the 1e-3 logit tolerance")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wrapper, lm, cfg = build(args.precision)
    if args.config in (0, 4):
        print(json.dumps(run_config4(wrapper, lm, cfg, args.steps, args.warmup, args.precision, dev)))
    if args.config in (0, 5):
        print(json.dumps(run_config5(wrapper, lm, cfg, args.templates, args.precision, dev, args.max_batch)))


if __name__ == "__main__":
    main()
