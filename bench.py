#!/usr/bin/env python3
"""Headline benchmark: sampled positions / second of the ESM-1b Gibbs hot path (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one Gibbs iteration over the batch resident on one GPU: draw P = 25 positions for each of
B = 256 chains of L = 256 residues (CPython-exact random.sample stream, native), scatter <mask>, run
the 33-layer ESM-1b forward (T = 258 tokens per chain), evaluate the LM head at the sampled rows, draw
tokens (top_k = 0, temperature = 1, burnin = inf) and write them back -- all inside one C-ABI call
(pg_esm_gibbs_run_device) with the token buffer resident in HBM.  Weights are seeded synthetic weights
of the ESM-1b architecture (no checkpoints offline); throughput does not depend on their values.

Multi-GPU: chains are independent, so each rank owns a contiguous block of chains (weak scaling:
256 chains per GPU) and the only collective is one RCCL all-gather of the final token buffers.

Output: ONE JSON line (rank 0) with the driver's contract plus
  roofline     -- the GEMM kernel family (96.5 % of the FLOPs): algorithmic FLOPs / HIP-event time
  cpu_baseline -- the fp32 CPU oracle (numpy/OpenBLAS port of the same path) on a bounded sample
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

from protein_gibbs_sampler_amd import _lib, models, pyrandom, sharding, weights  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def gemm_flops_per_iter(cfg, n_tokens, n_sampled):
    d, f, nl = cfg["d_model"], cfg["d_ffn"], cfg["n_layers"]
    return nl * 2.0 * (4 * d * d + 2 * d * f) * n_tokens + 2.0 * d * d * n_sampled


def total_flops_per_iter(cfg, n_tokens, T, n_sampled):
    """GEMM + attention contractions, 2 FLOP/MAC, LM head at the sampled rows only."""
    d, V = cfg["d_model"], cfg["vocab"]
    return gemm_flops_per_iter(cfg, n_tokens, n_sampled) + cfg["n_layers"] * 4.0 * T * d * n_tokens + 2.0 * V * d * n_sampled


def measured_gemm_traffic():
    """Fabric (L2 -> Infinity Cache/HBM) bytes per GEMM launch from the committed PMC passes
    (tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs, calibrated on the LayerNorm
    kernel whose traffic is known exactly).  None when the profile file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")
    if not os.path.exists(path):
        return None
    k = json.load(open(path))["kernels"]
    gem = [(v["launches"], v["read_MB_per_launch"] + v["write_MB_per_launch"]) for n, v in k.items()
           if "gemm_bf16_pp_kernel<0" in n or "gemm_bf16_pp_kernel<1" in n or "gemm_bf16_pp_kernel<2" in n]
    if not gem:
        return None
    return 1e6 * sum(n * mb for n, mb in gem) / sum(n for n, _ in gem)


def cpu_baseline(cfg, sd, B, L, P, valid_idx, target_seconds=15.0):
    """Times the CPU oracle (checker, never the product) on a bounded sample of the same workload:
    b chains x one full Gibbs iteration (mask, fp32 forward, draw).  Scales linearly in chains."""
    from threadpoolctl import threadpool_info
    from oracle import draw as odraw
    from oracle.esm_forward import EsmConfig, esm1b_trunk, lm_head
    ocfg = EsmConfig(vocab=cfg["vocab"], d_model=cfg["d_model"], n_layers=cfg["n_layers"], n_heads=cfg["n_heads"],
                     d_ffn=cfg["d_ffn"], max_pos=cfg["max_positions"])
    rng = np.random.default_rng(1234)

    def one(b):
        tok = np.concatenate([np.zeros((b, 1), np.int64), rng.integers(4, 24, (b, L)), np.full((b, 1), 2)], axis=1)
        idx = np.stack([rng.choice(np.arange(1, L + 1), P, replace=False) for _ in range(b)])
        t0 = time.perf_counter()
        for i in range(b):
            tok[i, idx[i]] = cfg["mask_idx"]
        x = esm1b_trunk(sd, ocfg, tok)
        rows = np.stack([x[i, idx[i]] for i in range(b)]).reshape(b * P, -1)
        logits = lm_head(sd, rows)
        toks = odraw.draw_rows(logits, valid_idx, 0, True, 1.0, np.repeat(np.arange(b), P), 0, np.tile(np.arange(P), b), 0, 0)
        for i in range(b):
            tok[i, idx[i]] = toks[i * P:(i + 1) * P]
        return time.perf_counter() - t0

    t1 = one(1)
    b = int(max(1, min(B, round(target_seconds / max(t1, 1e-3)))))
    t = one(b) if b > 1 else t1
    threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    return {"value": b * P / t, "unit": "sampled positions/s", "cores": int(threads), "kind": "port",
            "sample": "%d of %d chains x 1 Gibbs iteration (L=%d, P=%d), fp32 numpy/OpenBLAS oracle, %.1f s" % (b, B, L, P, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains-per-gpu", type=int, default=256)
    ap.add_argument("--length", type=int, default=256)
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (the JSON then says so)")
    ap.add_argument("--greedy-after-burnin", action="store_true",
                    help="SURVEY 8d variant: top_k=1, burnin=25 (argmax after 25 sampled iterations) instead of all-sampling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    cfg = dict(weights.ESM1B_CONFIG)
    if args.layers:
        cfg["n_layers"] = args.layers
    B, L = args.chains_per_gpu, args.length
    T = L + 2
    P = int(L * 10 / 100)
    K, W = args.steps, args.warmup
    B_total = B * world

    sd = weights.synthetic_state_dict(cfg, seed=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wrapper = models.ESM1b(state_dict=sd, config=cfg)
    lm = wrapper.model.to("cuda:%d" % local_rank)
    L_ = _lib.lib()
    valid_idx = sorted(wrapper.alphabet.get_idx(t) for t in "ACDEFGHIKLMNPQRSTVWY")

    # seeds: B_total chains, L residues i.i.d. uniform over the 20 amino acids, numpy default_rng(1234)
    rng = np.random.default_rng(1234)
    aa = rng.integers(0, 20, (B_total, L))
    tok_all = np.concatenate([np.zeros((B_total, 1), np.int64), np.asarray(valid_idx)[aa], np.full((B_total, 1), 2)], axis=1)
    tok_dev = torch.from_numpy(tok_all[rank * B:(rank + 1) * B].astype(np.int32)).to(dev).contiguous()
    gathered = None

    pos_rng = pyrandom.NativePyRandom()
    pos_rng.seed(0)
    population = list(range(1, L + 1))
    top_k, burnin = (1, 25.0) if args.greedy_after_burnin else (0, float("inf"))
    params = _lib.make_sample_params(True, cfg["mask_idx"], top_k, burnin, 1.0, valid_idx, rng_seed=0, rng_stream=0,
                                     row_id_base=rank * B)
    stream = torch.cuda.current_stream(dev)
    _lib.check(L_.pg_engine_set_stream(lm.handle, ctypes.c_void_p(stream.cuda_stream)))

    def run(n_iters, iter_base):
        """n_iters Gibbs iterations: native position table for ALL chains (same stream on every rank), slice, upload, run."""
        table = sharding.global_position_table(pos_rng, population, P, n_iters, B_total)
        d_idx = torch.from_numpy(sharding.local_slice(table, rank * B, (rank + 1) * B)).to(dev, non_blocking=True)
        params.iter_base = iter_base
        _lib.check(L_.pg_esm_gibbs_run_device(lm.handle, ctypes.c_void_p(tok_dev.data_ptr()), B, T,
                                              ctypes.c_void_p(d_idx.data_ptr()), n_iters, P, ctypes.byref(params), None, None))
        return d_idx

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if W > 0:
        keep = run(W, 0)
    barrier()
    t0 = time.perf_counter()
    keep = run(K, W)                                                   # noqa: F841 (keeps the index table alive)
    if dist is not None:
        lm.synchronize()                                               # engine stream -> before the collective reads tokens
        gathered = sharding.gather_tokens(dist, tok_dev)               # the one collective: final token buffers
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final = (gathered if gathered is not None else tok_dev).cpu().numpy()
    assert ((final[:, 1:-1] >= 4) & (final[:, 1:-1] <= 23)).all(), "chains left the 20-amino-acid alphabet"

    value = B_total * P * K / elapsed
    out = {"metric": "sampled positions/sec (whole node), ESM-1b L=256 B=256 Gibbs", "value": value,
           "unit": "sampled positions/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "ESM_sampler ESM-1b (33 layers, d=1280) Gibbs: %d chains/GPU x L=%d (T=%d), P=%d positions "
                                  "per chain per iteration, mask=True top_k=%d temperature=1.0 burnin=%s; bf16 MFMA "
                                  "operands, fp32 accumulate + fp32 residual stream; synthetic N(0,0.02) weights"
                                  % (B, L, T, P, top_k, "inf" if burnin == float("inf") else int(burnin)),
                      "global_batch": B_total, "seq_len": L, "parallelism": "chains sharded %d-way, 1 RCCL all-gather at end" % world,
                      "n_layers": cfg["n_layers"]}}

    if rank == 0:
        n_tok, n_samp = B * T, B * P
        flops_iter = total_flops_per_iter(cfg, n_tok, T, n_samp)
        out["model_tflops_per_gpu"] = flops_iter * K / elapsed / 1e12
        out["frac_of_bf16_mfma_peak"] = out["model_tflops_per_gpu"] / MFMA_BF16_PEAK_TFLOPS
    if not args.no_roofline:
        # dominant kernel family = the bf16 MFMA GEMM; HIP events on the engine's stream around every launch
        lm.prof_enable(True)
        lm.prof_reset()
        n_prof = min(K, 3)
        run(n_prof, W + K)
        torch.cuda.synchronize(dev)
        ms, launches = lm.prof_get("gemm")
        parts = {c: lm.prof_get(c) for c in ("gemm", "attention", "layernorm", "embed", "head", "sample")}
        lm.prof_enable(False)
        if rank == 0 and launches:
            # FLOPs the GEMM launches actually executed: every layer on all B*T rows, except that the last layer's
            # out-proj / fc1 / fc2 run on the B*P sampled rows only (exact pruning); head GEMM timed under "head"
            d_, f_, nl_ = cfg["d_model"], cfg["d_ffn"], cfg["n_layers"]
            full = 2.0 * (4 * d_ * d_ + 2 * d_ * f_) * (B * T)
            last = 2.0 * (3 * d_ * d_) * (B * T) + 2.0 * (d_ * d_ + 2 * d_ * f_) * (B * P)
            gf = ((nl_ - 1) * full + last) * n_prof
            achieved = gf / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_bf16_pp_kernel (all %d launches/iteration)" % (launches // n_prof),
                               "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": measured_gemm_traffic(),
                               "traffic_note": "bytes/launch L2<->fabric (FETCH_SIZE+WRITE_SIZE, calibrated), avg over the 4 per-layer GEMMs; "
                                               "algorithmic minimum 0.63 GB/launch -- the excess is X/W panel re-reads served by the 256 MB Infinity Cache",
                               "avg_launch_ms": ms / launches, "flops_per_launch": gf / launches}
            out["time_split_ms_per_iter"] = {c: v[0] / n_prof for c, v in parts.items()}
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd, B, L, P, valid_idx)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
