#!/usr/bin/env python3
"""Experiment: config 2 as TWO 128-chain sub-batches on two engines whose HIP streams are restricted to disjoint halves of the CUs
(hipExtStreamCreateWithCUMask), driven from two host threads: one half's memory-bound phases (LayerNorm, attention, epilogue bursts)
against the other half's MFMA main loops, systematically instead of by chance (tools/two_stream_test.py has no masks).
MASK patterns: 'evenodd' = CU bits 0,2,4.. / 1,3,5..; 'halves' = low 128 bits / high 128 bits; 'pairs' = bit pairs 0-1,4-5.. / 2-3,6-7..;
'none' = two unmasked streams; 'single' = the ordinary one-engine run (baseline)."""
import ctypes, os, sys, threading, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from protein_gibbs_sampler_amd import _lib, models, pyrandom, weights

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
cfg = dict(weights.ESM1B_CONFIG)
sd = weights.synthetic_state_dict(cfg, seed=0, std=0.025, embed_std=0.3, ln_jitter=0.1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
L_ = _lib.lib()
B_total, L, P, K = 256, 256, 25, int(os.environ.get("STEPS", "8"))
T = L + 2
valid_idx = list(range(4, 24))
rng = np.random.default_rng(1234)
tok_all = np.concatenate([np.zeros((B_total, 1), np.int64), np.asarray(valid_idx)[rng.integers(0, 20, (B_total, L))], np.full((B_total, 1), 2)], axis=1).astype(np.int32)
NCU = torch.cuda.get_device_properties(dev).multi_processor_count


def masks(kind):
    bits = [[0] * NCU, [0] * NCU]
    for i in range(NCU):
        if kind == "evenodd":
            h = i & 1
        elif kind == "halves":
            h = 0 if i < NCU // 2 else 1
        elif kind == "pairs":
            h = (i >> 1) & 1
        elif kind == "x8":           # bit i -> XCD i % 8 (if CUs are enumerated round-robin over the XCDs): CUs 0-15 / 16-31 of every XCD
            h = ((i >> 3) & 1)
        else:
            raise ValueError(kind)
        bits[h][i] = 1
    out = []
    for b in bits:
        words = (ctypes.c_uint32 * ((NCU + 31) // 32))()
        for i, v in enumerate(b):
            if v:
                words[i >> 5] |= 1 << (i & 31)
        out.append(words)
    return out


def run(kind):
    S = 1 if kind == "single" else 2
    B = B_total // S
    engines, streams, toks, params, idxs = [], [], [], [], []
    mk = masks(kind) if kind not in ("single", "none") else None
    for s_ in range(S):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lm = models.ESM1b(state_dict=sd, config=cfg, precision=os.environ.get("PGIBBS_TOOL_PRECISION", "bf16")).model.to("cuda:0")
        st = ctypes.c_void_p()
        if mk is not None:
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(mk[s_]), mk[s_])
            assert rc == 0, rc
        else:
            rc = hip.hipStreamCreateWithFlags(ctypes.byref(st), 1)
            assert rc == 0, rc
        _lib.check(L_.pg_engine_set_stream(lm.handle, st))
        lm.set_job_items(B_total)
        engines.append(lm); streams.append(st)
        toks.append(torch.from_numpy(tok_all[s_ * B:(s_ + 1) * B]).to(dev).contiguous())
        params.append(_lib.make_sample_params(True, cfg["mask_idx"], 0, float("inf"), 1.0, valid_idx, rng_seed=0, rng_stream=0, row_id_base=s_ * B))
        pr = pyrandom.NativePyRandom(); pr.seed(0)
        table = pr.sample_table(list(range(1, L + 1)), P, (K + 2) * B_total).reshape(K + 2, B_total, P)
        idxs.append(torch.from_numpy(np.ascontiguousarray(table[:, s_ * B:(s_ + 1) * B])).to(dev))
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
    final = np.concatenate([t.cpu().numpy() for t in toks])
    print("%-8s: %.2f ms/iteration, %.0f positions/s, token checksum %d" % (kind, 1e3 * dt / K, B_total * P * K / dt, int(final.astype(np.int64).sum())))
    del engines, toks
    torch.cuda.empty_cache()


for kind in os.environ.get("KINDS", "single,none,evenodd,halves,pairs,x8").split(","):
    run(kind)
